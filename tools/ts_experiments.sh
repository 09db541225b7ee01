#!/bin/bash
# ON THE GPU BOX: build tools/ts_ablate.hip in several variants (flags after --) and run each:  bash tools/ts_experiments.sh "-DTS_STAMP=1" "-DTS_ABL=6" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
i=0
for flags in "$@"; do
    out=/tmp/ts_exp_$i
    if /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $flags $R/tools/ts_ablate.hip -o $out 2>/tmp/ts_exp_$i.err; then
        echo "== $flags"; $out ${TS_ARGS:-}
    else
        echo "== $flags: BUILD FAILED"; tail -5 /tmp/ts_exp_$i.err
    fi
    i=$((i + 1))
done
