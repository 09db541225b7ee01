mkdir -p gpurun_out/r06
bash tools/profile_round.sh r06 > gpurun_out/r06/profile_round.log 2>&1
bash tools/pmc_mfma.sh > gpurun_out/r06/pmc_mfma.log 2>&1
cp gpurun_out/pmc_mfma/summary.txt gpurun_out/r06/mfma_utilisation.txt
ls gpurun_out/profile_r06 | head -30
