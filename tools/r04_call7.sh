#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_c7
rm -rf $OUT && mkdir -p $OUT
cd $R
timeout 900 python3 -m pytest tests/test_gpu_dense.py tests/test_encoder_variants.py tests/test_gpu_model.py -m gpu -x -q > $OUT/tests.log 2>&1
echo "tests rc=$?" >> $OUT/tests.log; tail -3 $OUT/tests.log
bash tools/kstat.sh 'pack|reduce|sgd' 2>&1 | tail -6
for v in 0 1; do
  timeout 300 python3 bench.py --steps 100 --warmup 10 --no-viterbi --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step', d['ms_per_step'], d['ms_per_step_repeats'])"
done
