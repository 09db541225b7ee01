#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for e in 1 8 1 8 1000; do
  timeout 300 python3 bench.py --steps 100 --warmup 10 --no-viterbi --no-cpu-baseline --time-every $e 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('time-every $e', d['ms_per_step'], d['ms_per_step_repeats'], 'ts', d['roofline']['avg_launch_ms'], 'n', d['roofline']['launches_timed'])"
done
