# Round 4, call 34: phase stamps of attno_kernel (tuning build), then the whole batched GPU module on the current build
# (profile row names as rocprofv3 prints them; the half-group gate conv's parity cases).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python scripts/stamps.py medium 128 2>&1 | grep -v amdgpu.ids | head -3
timeout 1500 python -m pytest tests/test_gpu_batched.py -m gpu -x -q 2>&1 | tail -4
