# Round 4, call 39: phase stamps of the fused MRF stage kernels at batch 64 / 16 / 1 (scripts/stamps_mrf.py, tuning build)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for b in 64 16 1; do timeout 200 python scripts/stamps_mrf.py medium $b 128; done 2>&1 | grep -v Warning | tee gpurun_out/r04_stamps_mrf.txt
