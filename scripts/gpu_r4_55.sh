# Round 4, call 55: extreme call shapes (scripts/exp_limits.py): huge batches of short texts, very long texts, a long stream
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python scripts/exp_limits.py medium 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r04_limits.txt
