#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_c9
rm -rf $OUT && mkdir -p $OUT
cd $R
timeout 600 python3 -m pytest tests/test_gpu_shead.py -m gpu -x -q -k "eight or golden" > $OUT/tests.log 2>&1
echo "tests rc=$?" >> $OUT/tests.log; tail -3 $OUT/tests.log
timeout 300 python3 tools/dec_ab.py 2>&1 | grep "DEC_MW=1" | tee $OUT/dec_ab.txt
echo "== no sleep in the polls"
MUCON_HIPCC_FLAGS=-DMW_NO_SLEEP timeout 600 python3 -m mucon_amd.build > $OUT/build.log 2>&1
timeout 300 python3 tools/dec_ab.py 2>&1 | grep "DEC_MW=1" | tee $OUT/dec_ab_nosleep.txt
