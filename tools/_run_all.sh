mkdir -p gpurun_out/r06
bash tools/knob_sweep.sh MUCON_FS_RB2 1 0 1 0 > gpurun_out/r06/fs_rb2_ab.txt 2>&1
python -m pytest tests/test_gpu_dense.py -m gpu -q -x -k "golden or oracle_f64 or full_size_batch or full_size_training or coarse_row" 2>&1 | tail -4 > gpurun_out/r06/t.txt
cat gpurun_out/r06/fs_rb2_ab.txt gpurun_out/r06/t.txt
