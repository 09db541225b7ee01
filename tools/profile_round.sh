#!/bin/bash
# Run ON THE GPU BOX (through gpurun):  bash tools/profile_round.sh r03
# Produces the rocprofv3 summaries that profiles/ keeps for this round: kernel-trace stats of the default
# bench command, and two PMC passes (FETCH_SIZE, WRITE_SIZE) of a short bench run.
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profile_$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 30 --warmup 5 > $OUT/bench_trace.log 2>&1
# the hot-path leg alone (no Viterbi / end-to-end / evaluation legs): per-kernel averages that are not mixed with the small
# launches of the batch-1 end-to-end leg (same kernels, 10x smaller grids)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_hot -- python3 $R/bench.py --steps 30 --warmup 5 --no-viterbi --no-cpu-baseline > $OUT/bench_trace_hot.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-viterbi > $OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-viterbi > $OUT/bench_pmc_write.log 2>&1
cd $R
python3 bench.py --steps 30 --warmup 5 > $OUT/bench_plain.log 2>&1
python3 tools/summarize_profiles.py $OUT $TAG > $OUT/summary.txt 2>&1
cat $OUT/summary.txt | head -60
rm -rf $OUT/trace $OUT/trace_hot $OUT/pmc_fetch $OUT/pmc_write   # raw rocprofv3 output: large, already condensed above
