# Round 4, call 15: what the f32 matrix pipe sustains in the kernels' instruction mixes (scripts/microbench/mfma_peak.hip)
cd $GRAFT_REPO_ROOT
./scripts/microbench/mfma_peak 2>&1 | tee gpurun_out/mfma_peak.txt
