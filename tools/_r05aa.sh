mkdir -p gpurun_out/r05aa
python -m pytest tests/test_gpu_tn_split.py -x -q 2>&1 | tail -2 > gpurun_out/r05aa/out.txt
bash tools/flag_ab.sh "-DTS_BRANCHFREE=0" "" --repeats 3 >> gpurun_out/r05aa/out.txt 2>&1
export TS_ARGS="65536 2048 2048 30"
TS_ST=0 bash tools/ts_experiments.sh "-fno-slp-vectorize -DTS_BRANCHFREE=0" "-fno-slp-vectorize -DTS_BRANCHFREE=1" >> gpurun_out/r05aa/out.txt 2>&1
cat gpurun_out/r05aa/out.txt
