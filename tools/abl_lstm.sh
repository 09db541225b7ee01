# ON THE GPU BOX: timing ablations of the LSTM forward recurrence (lstm.hpp LSTM_ABL: the results are wrong by construction).
# 0 as shipped | 1 no mat-vec FMAs | 3 plain v_fma_f32 instead of v_pk_fma_f32.
# r4 (Tz = 125, whole ops.lstm_forward incl. the 5.5 us input projection): 83 | 58 | 89 us -- the mat-vec is 25 us of the recurrence's 76, the rest is the
# gate tail + barrier + LDS round trip: ~55 instructions per wave and step beside the 64 packed FMAs, two waves per SIMD, 4 cycles each = the step's 1,460 cycles.  Leaves the library built WITHOUT ablations.
for f in 0 1 3; do
    MUCON_HIPCC_FLAGS="-DLSTM_ABL=$f" python -m mucon_amd.build --force > /dev/null 2>&1; echo "LSTM_ABL=$f"
    PYTHONPATH=. timeout 300 python tools/shead_breakdown.py 2>&1 | grep "biLSTM HIP"
done
MUCON_HIPCC_FLAGS= python -m mucon_amd.build --force > /dev/null 2>&1
