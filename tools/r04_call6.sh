#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_c6
rm -rf $OUT && mkdir -p $OUT
cd $R
timeout 600 python3 -m pytest tests/test_gpu_dense.py -m gpu -x -q -k "packers or golden or oracle_f64 or full_size_batch or trajectory" > $OUT/tests.log 2>&1
echo "tests rc=$?" >> $OUT/tests.log; tail -3 $OUT/tests.log
echo "== clock probe"; timeout 120 tools/build/clock_probe > $OUT/clock_probe.txt 2>&1; cat $OUT/clock_probe.txt
for v in 0 1 0 1; do
  MUCON_PACK_GATHER=$v timeout 300 python3 bench.py --steps 100 --warmup 10 --no-viterbi --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PACK_GATHER=$v', d['ms_per_step'], d['ms_per_step_repeats'])"
done
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
MUCON_PACK_GATHER=$v rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t$v -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-viterbi > $OUT/log$v.txt 2>&1
f=$(find $OUT/t$v -name "*kernel_stats.csv" | head -1)
python3 - "$f" $v <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'pack_all' in r['Name']: print("PACK_GATHER=%s pack_all_kernel avg %.1f us" % (sys.argv[2], float(r['AverageNs'])/1e3))
PY
rm -rf $OUT/t$v
done
