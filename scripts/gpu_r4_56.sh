# Round 4, call 56: extreme call shapes again with the workspace budget, then the whole GPU suite
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python scripts/exp_limits.py medium 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r04_limits.txt
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
