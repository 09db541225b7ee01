#!/bin/bash
# ON THE GPU BOX: rocprofv3 per-kernel averages of a short default bench run, filtered by a pattern:  bash tools/kstat.sh 'sgd|head'
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/kstat; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/log.txt 2>&1
cd $R
f=$(find $OUT/t -name "*kernel_stats.csv" | head -1)
python3 - "$f" "${1:-.}" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name'].replace('void ', '').split('(')[0]
    if re.search(sys.argv[2], n):
        print(f"  {n[:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}  max {float(r['MaxNs'])/1e3:8.1f}")
PY
rm -rf $OUT/t
