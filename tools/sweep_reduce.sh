#!/bin/bash
# ON THE GPU BOX: A/B of a tuning hook on the hot-path step
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() {
  env "$@" python3 bench.py --steps 200 --warmup 30 --no-viterbi --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$*', 'ms/step %.4f'%d['ms_per_step'], 'wg %.4f' % d['roofline']['all_weight_gradients_launch_ms'])"
}
for i in 1 2; do
run MUCON_NT_SPLIT=0
run MUCON_NT_SPLIT=1
done
