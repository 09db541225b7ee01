#!/bin/bash
# ON THE GPU BOX: A/B of a tuning hook on a chosen problem size (hot-path step only)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() {
  env "$@" python3 bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-viterbi --batch $BATCH --frames $FRAMES 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('B=$BATCH T=$FRAMES $*', 'ms/step %.4f'%d['ms_per_step'], 'first conv %.4f' % d['roofline_first_conv_fwd']['avg_launch_ms'])"
}
for sz in "1 6000" "2 4000" "1 5000"; do set -- $sz; BATCH=$1; FRAMES=$2
run MUCON_FIRST_CONV_KSPLIT_ROWS=4096
run MUCON_FIRST_CONV_KSPLIT_ROWS=8192
done
