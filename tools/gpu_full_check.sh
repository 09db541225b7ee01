#!/bin/bash
# ON THE GPU BOX (one gpurun call): the whole -m gpu suite, the kernel list of the Viterbi tests under rocprofv3 (every instantiation of
# the throughput schedule must appear: profiles/r04_viterbi_tests_kernel_list.csv), the Viterbi host breakdown, one default bench line.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/full_check
rm -rf $OUT && mkdir -p $OUT
cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1
echo "tests rc=$?" >> $OUT/tests.log
tail -5 $OUT/tests.log
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/vit_trace -- python3 -m pytest $R/tests/test_gpu_viterbi_batch.py $R/tests/test_gpu_viterbi.py -q -p no:cacheprovider > $OUT/vit_trace.log 2>&1
find $OUT/vit_trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/viterbi_tests_kernel_list.csv
rm -rf $OUT/vit_trace
cd $R
timeout 600 python3 tools/vit_host_breakdown.py > $OUT/vit_host_breakdown.txt 2>&1
grep -v "uint8\|int32" $OUT/vit_host_breakdown.txt
timeout 1200 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"
python3 - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/full_check/bench.json").read().strip().splitlines()[-1])
    print("ms_per_step", d["ms_per_step"], "value", d["value"], "roofline.frac", d["roofline"]["frac"], "ts ms", d["roofline"]["avg_launch_ms"])
    print("viterbi", {k:v for k,v in d["viterbi"].items() if k.startswith("ms_")})
    print("e2e", d["end_to_end"]["ms_per_video"], "eval", d["evaluation"]["ms_per_video"], "cpu", d["cpu_baseline"]["value"])
except Exception as e:
    print("bench parse failed", e)
PY
