#!/bin/bash
# ON THE GPU BOX: per-kernel averages of the hot-path step of TWO TREES (rocprofv3 --kernel-trace --stats; other, this, other, this):
#   bash tools/tree_ab_kernels.sh .r05_tree        (the frozen tree: tools/tree_ab.sh)
# Prints every kernel's average duration (us) per run side by side (kernels that exist in one tree only show 0 in the other) and the sum over one step.
R=${GRAFT_REPO_ROOT:-$(pwd)}
other=$1; shift     # further arguments go to bench.py (e.g. --batch 1 --frames 2000: the one-video step's kernels)
OUT=$R/gpurun_out/tree_ab_kernels
rm -rf $OUT && mkdir -p $OUT
i=0
for rnd in 1 2; do
  for t in "$R/$other" "$R"; do
    i=$((i+1))
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t$i -- python3 $t/bench.py --steps 100 --warmup 10 --repeats 1 --no-viterbi --no-cpu-baseline --no-calibration --no-traffic "$@" > $OUT/b$i.log 2>&1
    cp $(ls $OUT/t$i/*/*kernel_stats.csv | head -1) $OUT/stats$i.csv
    rm -rf $OUT/t$i
  done
done
cd $R
python3 - "$other" <<'PY'
import csv, sys, os, re
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out", "tree_ab_kernels")
runs = []
for i in (1, 2, 3, 4):
    d = {}
    for r in csv.DictReader(open(os.path.join(out, f"stats{i}.csv"))):
        d[r["Name"]] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
    runs.append(d)
def steps(d):
    return min(c for k, (a, c) in d.items() if re.search(r"ts_(persist|runs|batched)_kernel|tn_batched", k))
names = sorted(set(runs[0]) | set(runs[1]), key=lambda k: -max(runs[0].get(k, (0, 0))[0] * runs[0].get(k, (0, 0))[1], runs[1].get(k, (0, 0))[0] * runs[1].get(k, (0, 0))[1]))
print(f"A = [{sys.argv[1]}]   B = [this tree]   (average us per launch: A B A B | launches per step)")
tot = [0.0] * 4
for n in names:
    ps = [runs[j].get(n, (0.0, 0))[1] / steps(runs[j]) for j in range(4)]
    if max(ps) < 0.5:
        continue
    row = [runs[j].get(n, (0.0, 0))[0] for j in range(4)]
    for j in range(4):
        tot[j] += row[j] * ps[j]
    print(f"  {row[0]:8.2f} {row[1]:8.2f} {row[2]:8.2f} {row[3]:8.2f} | {max(ps):4.1f}  {n[:90]}")
print(f"  {tot[0]:8.1f} {tot[1]:8.1f} {tot[2]:8.1f} {tot[3]:8.1f} |       sum over one step")
PY
