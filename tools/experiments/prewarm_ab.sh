# ON THE GPU BOX: is the hot-path step's speed a function of the weights?  The driver's command (--steps 20 --warmup 5) with the learning rate at 0 / at its
# default, with the init weights restored (the default) and left to drift behind 800 untimed SGD steps; then the default command.
for a in "MUCON_BENCH_LR=0 --prewarm-steps 100" "MUCON_BENCH_LR=0 --prewarm-steps 800 --keep-drift" "MUCON_BENCH_LR=0.01 --prewarm-steps 0" "MUCON_BENCH_LR=0.01 --prewarm-steps 100" "MUCON_BENCH_LR=0.01 --prewarm-steps 800 --keep-drift"; do
    set -- $a
    env $1 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-viterbi --no-cpu-baseline $2 $3 $4 2>/dev/null | tail -1 | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('$a:', d['ms_per_step'], d['ms_per_step_repeats'], 'ts', d['roofline']['avg_launch_ms'], 'first_conv', d['roofline_first_conv_fwd']['avg_launch_ms'])"
done
python3 bench.py --no-viterbi --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('default command:', d['ms_per_step'], d['ms_per_step_repeats'], 'ts', d['roofline']['avg_launch_ms'])"
