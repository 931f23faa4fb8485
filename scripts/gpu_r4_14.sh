# Round 4, call 14: a 64-utterance call as N concurrent sub-batches (N engines on one GPU) against one engine call.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python scripts/exp_split.py medium 64 2>&1 | grep -v amdgpu.ids
timeout 600 python scripts/exp_split.py high 64 2>&1 | grep -v amdgpu.ids
timeout 300 python scripts/exp_split.py medium 16 2>&1 | grep -v amdgpu.ids
