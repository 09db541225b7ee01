# ON THE GPU BOX: the driver's command (--steps 20 --warmup 5) from a cold start and behind 200 / 400 / 800 / 1600 untimed steps
for w in 0 800 0 400 1600 200 0; do
    python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-viterbi --no-cpu-baseline --prewarm-steps $w 2>/dev/null | tail -1 | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('prewarm-steps $w', d['ms_per_step'], d['ms_per_step_repeats'], 'ts', d['roofline']['avg_launch_ms'], 'first_conv', d['roofline_first_conv_fwd']['avg_launch_ms'])"
done
