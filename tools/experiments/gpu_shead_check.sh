#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/shead_check
rm -rf $OUT && mkdir -p $OUT
cd $R
timeout 900 python3 -m pytest tests/test_gpu_shead.py tests/test_gpu_model.py tests/test_gpu_trajectory.py tests/test_gpu_fused_step.py -m gpu -x -q > $OUT/tests.log 2>&1
echo "tests rc=$?" >> $OUT/tests.log; tail -25 $OUT/tests.log
timeout 300 python3 tools/dec_ab.py 2>&1 | tee $OUT/dec_ab.txt
