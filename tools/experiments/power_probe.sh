#!/bin/bash
# ON THE GPU BOX: board power and shader clock while a timing harness loops:  bash tools/power_probe.sh "<hipcc flags>" [harness args]
# (builds tools/ts_ablate.hip with the flags, runs it for a few seconds in the background, samples rocm-smi beside it)
R=${GRAFT_REPO_ROOT:-$(pwd)}
flags=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $flags $R/tools/ts_ablate.hip -o /tmp/pp_bin 2>/dev/null || { echo "build failed"; exit 1; }
echo "== $flags $*"
/tmp/pp_bin "$@" &
pid=$!
sleep 1.5
for i in 1 2 3; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr -s ' ' | tr '\n' ';'
    echo
    sleep 0.7
done
wait $pid
