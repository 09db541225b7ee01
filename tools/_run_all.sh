mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_dense.py tests/test_gpu_fuzz.py tests/test_gpu_model.py tests/test_gpu_trajectory.py -m gpu -q -x 2>&1 | tail -6 > gpurun_out/r06/t.txt
bash tools/tree_ab_kernels.sh .r05_tree > gpurun_out/r06/tree_ab_kernels3.txt 2>&1
cat gpurun_out/r06/t.txt; grep -E "cs_kernel|ct_kernel|pack_all|sum over" gpurun_out/r06/tree_ab_kernels3.txt
