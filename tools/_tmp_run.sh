mkdir -p gpurun_out/r06
python3 tools/kernel_cycles.py > gpurun_out/r06/kernel_cycles.txt 2>&1; tail -3 gpurun_out/r06/kernel_cycles.txt
python3 bench.py > gpurun_out/r06/bench_mid2.json 2> gpurun_out/r06/bench_mid2.err; tail -2 gpurun_out/r06/bench_mid2.err
