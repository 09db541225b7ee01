#!/bin/bash
# ON THE GPU BOX: the hot-path bench under two builds of the library, alternated (A B A B), same box:  bash tools/flag_ab.sh "<flags A>" "<flags B>" [bench args]
# (MUCON_HIPCC_FLAGS of each build; prints ms_per_step and the two event-timed launches per run)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
fa=$1; fb=$2; shift 2
for rnd in 1 2; do
  for f in "$fa" "$fb"; do
    MUCON_HIPCC_FLAGS="$f" python3 -m mucon_amd.build --force > /dev/null 2>&1 || { echo "build failed: $f"; exit 1; }
    python3 bench.py --no-viterbi --no-cpu-baseline --no-calibration "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('flags [%s]: ms_per_step %.4f  repeats %s  wgrad %.4f ms  first_conv %.4f ms' % ('$f', d['ms_per_step'], d['ms_per_step_repeats'], d['roofline']['avg_launch_ms'], d['roofline_first_conv_fwd']['avg_launch_ms']))"
  done
done
