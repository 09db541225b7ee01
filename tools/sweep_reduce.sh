#!/bin/bash
# ON THE GPU BOX: slab-reduction workgroup shape (MUCON_REDUCE_LANES) x batched weight-gradient slab count
# (MUCON_TN_BATCH_TARGET, MUCON_TN_MC_CAP): hot-path step + end-to-end step
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() {
  env "$@" python3 bench.py --steps 200 --warmup 30 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$*', 'ms/step %.4f'%d['ms_per_step'], 'wg %.4f' % d['roofline']['all_weight_gradients_launch_ms'], 'e2e', d.get('end_to_end',{}).get('ms_per_video'))"
}
run MUCON_TN_MC_CAP=2048
run MUCON_TN_MC_CAP=4096
run MUCON_TN_MC_CAP=2048 MUCON_TN_BATCH_TARGET=96
run MUCON_TN_MC_CAP=4096 MUCON_TN_BATCH_TARGET=96
run MUCON_TN_MC_CAP=2048
