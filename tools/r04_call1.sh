#!/bin/bash
# Round 4, first GPU call: the new Viterbi parity tests (throughput schedule, label formats), a rocprofv3 kernel list of them,
# the batched-evaluation tests, and a bench line.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_c1
rm -rf $OUT && mkdir -p $OUT
cd $R
timeout 900 python3 -m pytest tests/test_gpu_viterbi_batch.py tests/test_gpu_viterbi.py tests/test_gpu_eval_batched.py tests/test_gpu_model.py -x -q > $OUT/tests.log 2>&1
echo "tests rc=$?" >> $OUT/tests.log
tail -15 $OUT/tests.log
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/vit_trace -- python3 -m pytest $R/tests/test_gpu_viterbi_batch.py $R/tests/test_gpu_viterbi.py -q -p no:cacheprovider > $OUT/vit_trace.log 2>&1
find $OUT/vit_trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/viterbi_tests_kernel_stats.csv
rm -rf $OUT/vit_trace
cd $R
timeout 1200 python3 bench.py --steps 30 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"
python3 - <<'PY'
import json,sys
try:
    d=json.loads(open("gpurun_out/r04_c1/bench.json").read().strip().splitlines()[-1])
    print("ms_per_step", d["ms_per_step"], "value", d["value"])
    print("viterbi", {k:v for k,v in d["viterbi"].items() if k.startswith("ms_")})
    print("e2e", d["end_to_end"]["ms_per_video"], "eval", d["evaluation"]["ms_per_video"])
except Exception as e:
    print("bench parse failed", e)
PY
