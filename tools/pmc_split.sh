#!/bin/bash
# ON THE GPU BOX: SQ counters of the split first_conv kernel (two PMC passes)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_split
rm -rf $OUT && mkdir -p $OUT
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -- python3 $R/tools/split_bench.py > $OUT/log1.txt 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -- python3 $R/tools/split_bench.py > $OUT/log2.txt 2>&1
cd $R
for d in p1 p2; do
f=$(find $OUT/$d -name "*counter_collection.csv" | head -1)
[ -z "$f" ] && { echo "no counters for $d"; tail -5 $OUT/log${d#p}.txt; continue; }
python3 - "$f" <<'PY'
import csv, sys, collections
per = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].replace('void ', '').split('(')[0][:50]
    per.setdefault(k, collections.defaultdict(list))[r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in per.items():
    if 'split' not in k and 'nt_gemm' not in k: continue
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
done | tee $OUT/summary.txt
rm -rf $OUT/p1 $OUT/p2
