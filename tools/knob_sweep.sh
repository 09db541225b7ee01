#!/bin/bash
# ON THE GPU BOX: A/B of one tuning knob ("Tuning / regression knobs") on the hot-path step, same box, back to back:
#   bash tools/knob_sweep.sh MUCON_TS_MC_CAP 512 1024 2048
#   BENCH_ARGS="--batch 1 --frames 6000" bash tools/knob_sweep.sh MUCON_FIRST_CONV_KSPLIT_ROWS 4096 8192
# Prints ms per step (median of 3 regions of 100 steps), the dominant launch and, with WITH_E2E=1, the end-to-end / evaluation legs.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
knob=$1
shift
extra="--no-viterbi"
[ "${WITH_E2E:-0}" = "1" ] && extra=""
for v in "$@"; do
    env "$knob=$v" python3 bench.py --steps 100 --warmup 10 --repeats 3 --no-cpu-baseline $extra ${BENCH_ARGS:-} 2>/dev/null | tail -1 | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline())
e = d.get('end_to_end', {}).get('ms_per_video'), d.get('evaluation', {}).get('ms_per_video')
print('$knob=$v', 'ms/step', d['ms_per_step'], d['ms_per_step_repeats'], 'weight-gradient launch', d['roofline']['avg_launch_ms'],
      'first_conv fwd', d['roofline_first_conv_fwd']['avg_launch_ms'], 'e2e / eval ms per video', e)"
done
