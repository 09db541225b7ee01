#!/bin/bash
# ON THE GPU BOX: HBM bytes per launch of the weight-gradient launch and of first_conv forward (two separate rocprofv3 --pmc passes, as MI355X_MICROARCH.md
# prescribes: FETCH_SIZE x 2 for wide coalesced reads on gfx950, both x 1024 B) -- what bench.py measures for roofline.traffic, alone:  bash tools/traffic_quick.sh [ENV=VALUE ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/traffic_quick
rm -rf $OUT && mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/$c -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-viterbi --no-calibration --no-traffic > $OUT/$c.log 2>&1
done
python3 - "$OUT" "$*" <<'PY'
import csv, glob, sys
out = sys.argv[1]
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/{c}/*/*counter_collection.csv")[0]
    acc = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c:
            continue
        k = r["Kernel_Name"].split("(")[0]
        a = acc.setdefault(k, [0.0, 0])
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    for k, (v, n) in acc.items():
        tot.setdefault(k, {})[c] = v / n * 1024.0 * (2.0 if c == "FETCH_SIZE" else 1.0)
print(f"HBM bytes per launch ({sys.argv[2]}): fetch (x 2 corrected) + write = total [MB]")
for k, d in sorted(tot.items(), key=lambda kv: -sum(kv[1].values()))[:8]:
    print(f"  {d.get('FETCH_SIZE', 0) / 1e6:8.1f} + {d.get('WRITE_SIZE', 0) / 1e6:7.1f} = {sum(d.values()) / 1e6:8.1f}   {k[:80]}")
PY
