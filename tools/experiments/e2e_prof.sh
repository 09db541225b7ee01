#!/bin/bash
# ON THE GPU BOX: the end-to-end (batch-1) step plain, then under rocprofv3 --kernel-trace with the median step's timeline
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/e2e_prof
rm -rf $OUT && mkdir -p $OUT
python3 $R/tools/experiments/e2e_loop.py 40 2>&1 | tail -1 | cut -c1-120
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/tools/experiments/e2e_loop.py 30 > $OUT/log.txt 2>&1
cd $R
python3 tools/e2e_trace_summary.py $(find $OUT/trace -name "*kernel_trace.csv" | head -1) v > $OUT/summary.txt
head -${1:-14} $OUT/summary.txt
rm -rf $OUT/trace
