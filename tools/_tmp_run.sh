mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_tn_split.py tests/test_gpu_dense.py tests/test_encoder_variants.py tests/test_gpu_fused_step.py tests/test_gpu_trajectory.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r06/t.txt
bash tools/tree_ab_kernels.sh .r05_tree > gpurun_out/r06/tree_ab_kernels2.txt 2>&1
bash tools/tree_ab.sh .r05_tree > gpurun_out/r06/tree_ab2.txt 2>&1
grep -E "ts_|reduce_batch|sum over" gpurun_out/r06/tree_ab_kernels2.txt; cat gpurun_out/r06/t.txt gpurun_out/r06/tree_ab2.txt
