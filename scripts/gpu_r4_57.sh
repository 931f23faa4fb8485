# Round 4, call 57: very long single utterances against the oracle (scripts/exp_long.py)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python scripts/exp_long.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r04_long.txt
