# ON THE GPU BOX: the hot-path bench's timed regions with and without a stream drain every n steps (same box, back to back)
for d in 0 32 0 64 0 16; do
    python3 bench.py --steps 200 --warmup 20 --repeats 7 --no-viterbi --no-cpu-baseline --drain-every $d 2>/dev/null | tail -1 | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('drain-every $d', d['ms_per_step'], d['ms_per_step_repeats'])"
done
