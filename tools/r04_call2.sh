#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_c2
rm -rf $OUT && mkdir -p $OUT
cd $R
timeout 900 python3 -m pytest tests/test_gpu_viterbi_batch.py tests/test_gpu_viterbi.py -x -q > $OUT/tests.log 2>&1
echo "tests rc=$?" >> $OUT/tests.log
tail -5 $OUT/tests.log
timeout 600 python3 tools/vit_host_breakdown.py > $OUT/vit_host_breakdown.txt 2>&1
cat $OUT/vit_host_breakdown.txt
timeout 600 python3 tools/two_stream_probe.py > $OUT/two_stream_probe.txt 2>&1
cat $OUT/two_stream_probe.txt
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/vit_trace -- python3 $R/tools/vit_profile_run.py > $OUT/vit_trace.log 2>&1
find $OUT/vit_trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/viterbi_kernel_stats.csv
rm -rf $OUT/vit_trace
python3 - <<'PY'
import csv,os
p=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r04_c2/viterbi_kernel_stats.csv")
for r in csv.reader(open(p)):
    print(r[0][:90], r[1:4], r[5:7])
PY
