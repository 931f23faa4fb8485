# Round 4, call 16: matrix-pipe cycles lost per interleaved ds_read / global load / VALU op (scripts/microbench/mfma_mix.hip)
cd $GRAFT_REPO_ROOT
./scripts/microbench/mfma_mix 2>&1 | tee gpurun_out/mfma_mix.txt
