#!/bin/bash
# ON THE GPU BOX: the hot-path bench under alternative ENVIRONMENTS, alternated on one box:  bash tools/env_ab.sh "" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0"
# (each argument: a space-separated list of VAR=value, "" = the plain environment); WITH_E2E=1 adds the end-to-end / evaluation legs.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
extra="--no-viterbi"
[ "${WITH_E2E:-0}" = "1" ] && extra=""
for rnd in 1 2; do
  for e in "$@"; do
    env $e python3 bench.py --steps 100 --warmup 10 --repeats 3 --no-cpu-baseline --no-traffic $extra ${BENCH_ARGS:-} 2>/dev/null | tail -1 | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline())
e = d.get('end_to_end', {}).get('ms_per_video'), d.get('evaluation', {}).get('ms_per_video')
print('env [$e]', 'ms/step', d['ms_per_step'], d['ms_per_step_repeats'], 'weight-gradient launch', d['roofline']['avg_launch_ms'],
      'first_conv fwd', d['roofline_first_conv_fwd']['avg_launch_ms'], 'e2e / eval ms per video', e)"
  done
done
