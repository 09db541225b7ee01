# ON THE GPU BOX: timing ablations of the first_conv forward kernel (gemm_split.hpp S2_ABL_*: the results are wrong by construction).
# Leaves the library built WITHOUT ablations.
run() { PYTHONPATH=. timeout 300 python tools/split_bench.py 2>&1 | tail -1; }
for f in S2_ABL_BASE S2_ABL_NOW S2_ABL_NOSPLIT S2_ABL_NOA S2_ABL_NOMFMA "S2_ABL_NOA -DS2_ABL_NOMFMA" "S2_ABL_NOA -DS2_ABL_NOW -DS2_ABL_NOSPLIT"; do
    MUCON_HIPCC_FLAGS="-D$f" python -m mucon_amd.build --force > /dev/null 2>&1; echo $f; run
done
MUCON_HIPCC_FLAGS= python -m mucon_amd.build --force > /dev/null 2>&1
