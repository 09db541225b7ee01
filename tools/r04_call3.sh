#!/bin/bash
# full GPU suite + bench + viterbi breakdown
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_c3
rm -rf $OUT && mkdir -p $OUT
cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1
echo "tests rc=$?" >> $OUT/tests.log
tail -5 $OUT/tests.log
timeout 600 python3 tools/vit_host_breakdown.py > $OUT/vit_host_breakdown.txt 2>&1
grep -v "uint8\|int32" $OUT/vit_host_breakdown.txt
timeout 1200 python3 bench.py --steps 30 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"
python3 - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r04_c3/bench.json").read().strip().splitlines()[-1])
    print("ms_per_step", d["ms_per_step"], "value", d["value"])
    print("viterbi", {k:v for k,v in d["viterbi"].items() if k.startswith("ms_")})
    print("e2e", d["end_to_end"]["ms_per_video"], "eval", d["evaluation"]["ms_per_video"])
except Exception as e:
    print("bench parse failed", e)
PY
