#!/bin/bash
# ON THE GPU BOX: per-kernel averages of the hot-path step under rocprofv3 (TRAINING=0/1)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/stepk; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $R/tools/hot_loop.py > $OUT/log.txt 2>&1
cd $R
f=$(find $OUT/t -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
tot = 0
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name'].replace('void ', '').split('(')[0]
    if any(k in n for k in ('gemm', 'fused', 'batched', 'head', 'reduce', 'gn_', 'pack', 'unpool', 'sgd_', 'split')):
        print(f"  {n[:66]:66s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us total/step {float(r['TotalDurationNs'])/12e3:8.1f}")
        tot += float(r['TotalDurationNs']) / 12e3
print("  kernel time per step: %.1f us" % tot)
PY
rm -rf $OUT
