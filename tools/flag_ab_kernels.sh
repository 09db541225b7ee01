#!/bin/bash
# ON THE GPU BOX: per-kernel averages of the hot-path step under two builds of the library (rocprofv3 --kernel-trace --stats, A then B then A then B):
#   bash tools/flag_ab_kernels.sh "<flags A>" "<flags B>"
# Prints every kernel's average duration (us) per run side by side and the sum over one step -- which kernels a flag moves, and by how much the
# others move with it (code placement: a flag that touches one kernel of mucon_hip.hip shifts every kernel behind it in the code object).
R=${GRAFT_REPO_ROOT:-$(pwd)}
fa=$1; fb=$2
OUT=$R/gpurun_out/flag_ab_kernels
rm -rf $OUT && mkdir -p $OUT
i=0
for rnd in 1 2; do
  for f in "$fa" "$fb"; do
    i=$((i+1))
    cd $R && MUCON_HIPCC_FLAGS="$f" python3 -m mucon_amd.build --force > /dev/null 2>&1 || { echo "build failed: $f"; exit 1; }
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t$i -- python3 $R/bench.py --steps 100 --warmup 10 --repeats 1 --no-viterbi --no-cpu-baseline --no-calibration > $OUT/b$i.log 2>&1
    cp $(ls $OUT/t$i/*/*kernel_stats.csv | head -1) $OUT/stats$i.csv
    rm -rf $OUT/t$i
  done
done
cd $R
python3 - "$fa" "$fb" <<'PY'
import csv, sys, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out", "flag_ab_kernels")
runs = []
for i in (1, 2, 3, 4):
    d = {}
    for r in csv.DictReader(open(os.path.join(out, f"stats{i}.csv"))):
        d[r["Name"]] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
    runs.append(d)
names = sorted(runs[0], key=lambda k: -runs[0][k][0] * runs[0][k][1])
steps = max(c for _, c in runs[0].values() if True)
print(f"A = [{sys.argv[1]}]   B = [{sys.argv[2]}]   (average us per launch: A B A B | launches per step)")
tot = [0.0] * 4
for n in names:
    calls = runs[0][n][1]
    per_step = calls / min(c for k, (a, c) in runs[0].items() if "ts_" in k or "tn_batched" in k)
    row = [runs[j].get(n, (0.0, 0))[0] for j in range(4)]
    if per_step < 0.5:
        continue
    for j in range(4):
        tot[j] += row[j] * per_step
    print(f"  {row[0]:8.2f} {row[1]:8.2f} {row[2]:8.2f} {row[3]:8.2f} | {per_step:4.1f}  {n[:90]}")
print(f"  {tot[0]:8.1f} {tot[1]:8.1f} {tot[2]:8.1f} {tot[3]:8.1f} |       sum over one step")
PY
