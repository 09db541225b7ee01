#!/bin/bash
# ON THE GPU BOX: MFMA-pipe utilisation and wait breakdown per kernel of the hot-path step (one PMC pass, SQ + GRBM only).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_mfma
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-viterbi > $OUT/log.txt 2>&1
cd $R
f=$(find $OUT/pmc -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY' | tee $OUT/summary.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
per = collections.OrderedDict()
for r in rows:
    k = (r['Kernel_Name'].replace('void ', '').split('(')[0][:58], r.get('Grid_Size', ''))
    d = per.setdefault(k, collections.defaultdict(list))
    d[r['Counter_Name']].append(float(r['Counter_Value']))
print(f"{'kernel':58s} {'grid':>8s} {'n':>4s} {'cycles':>6s} {'mfma%':>6s} {'wait_any%':>9s} {'wait_inst%':>10s} {'active%':>8s} {'lds_stall%':>10s}")
for (k, g), d in per.items():
    if 'GRBM_GUI_ACTIVE' not in d or not any(s in k for s in ('gemm', 'fused', 'batched', 'persist', 'ts_runs', 'head', 'reduce', 'gn_', 'split', 'fs_kernel', 'cs_kernel', 'ct_kernel')):
        continue
    avg = {c: sum(v) / len(v) for c, v in d.items()}
    gui = avg['GRBM_GUI_ACTIVE'] / 8.0            # summed over 8 XCDs
    wc = avg['SQ_WAVE_CYCLES']                     # quad-cycles summed over waves
    mf = avg['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui * 1024.0) * 100 if gui else 0
    print(f"{k:58s} {g:>8s} {len(d['GRBM_GUI_ACTIVE']):4d} {int(gui):6d} {mf:6.1f} {avg['SQ_WAIT_ANY']/wc*100:9.1f} {avg['SQ_WAIT_INST_ANY']/wc*100:10.1f} {avg['SQ_ACTIVE_INST_ANY']/wc*100:8.1f} {avg['SQ_WAIT_INST_LDS']/wc*100:10.1f}")
PY
rm -rf $OUT/pmc
