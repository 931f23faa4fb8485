cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python scripts/exp_pair.py medium 64 2>&1 | grep -v amdgpu.ids
timeout 600 python scripts/exp_pair.py medium 32 2>&1 | grep -v amdgpu.ids
timeout 600 python scripts/exp_pair.py high 64 2>&1 | grep -v amdgpu.ids
