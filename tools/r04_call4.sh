#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_c4
rm -rf $OUT && mkdir -p $OUT
cd $R
echo "== coarse protocol probe"
timeout 300 tools/build/coarse_protocol_probe 300 > $OUT/coarse_protocols.txt 2>&1; cat $OUT/coarse_protocols.txt
timeout 300 tools/build/coarse_protocol_probe 300 >> $OUT/coarse_protocols.txt 2>&1; tail -8 $OUT/coarse_protocols.txt
cd /tmp && export TMPDIR=/tmp
echo "== viterbi in-flight sweep"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/sweep -- python3 $R/tools/vit_inflight_sweep.py > $OUT/sweep.log 2>&1
f=$(find $OUT/sweep -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_kernels.py "$f" 'viterbi' > $OUT/vit_inflight_sweep.txt 2>&1; cat $OUT/vit_inflight_sweep.txt
rm -rf $OUT/sweep
echo "== step kernels"
cd $R
bash tools/kstat.sh 'pack|reduce|sgd|gn_|head' 2>&1 | tail -12
