#!/bin/bash
# ON THE GPU BOX: per-kernel averages of the s-head kernels (rocprofv3 --kernel-trace --stats of tools/shead_breakdown.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/shead_prof
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/shead_breakdown.py > $OUT/log.txt 2>&1
cd $R
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if any(k in n for k in ('decoder_', 'lstm_', 'dec_', 'loss_', 'sgd_')):
        print(f"{n.split('(')[0][:40]:40s} calls {r['Calls']:>6s}  avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:9.1f} us")
PY
rm -rf $OUT/trace   # tens of MB of per-launch rows; the summary above is what is kept
