mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_dense.py -m gpu -x -q -k "overlap_hook or static_runs" 2>&1 | tail -5
MUCON_BENCH_FORCE_DIST=1 python3 bench.py --steps 50 --warmup 10 --repeats 3 --no-viterbi --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 > gpurun_out/r06/bench_force_dist_overlap.json
MUCON_BENCH_FORCE_DIST=1 MUCON_BENCH_OVERLAP=0 python3 bench.py --steps 50 --warmup 10 --repeats 3 --no-viterbi --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 > gpurun_out/r06/bench_force_dist_plain.json
python3 bench.py --steps 50 --warmup 10 --repeats 3 --no-viterbi --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 > gpurun_out/r06/bench_n1.json
python3 -c "
import json
for f in ('overlap','plain'):
    d=json.loads(open('gpurun_out/r06/bench_force_dist_%s.json'%f).read()); print(f, d['ms_per_step'], d['ms_per_step_repeats'], d.get('rccl'))
d=json.loads(open('gpurun_out/r06/bench_n1.json').read()); print('n1', d['ms_per_step'])"
