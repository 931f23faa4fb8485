# Round 4, call 52: the soak with 64 / 256 / 1024 cached graphs per engine
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for g in 64 256 1024; do echo "== PIPER_HIP_GRAPHS=$g"; PIPER_HIP_GRAPHS=$g timeout 600 python scripts/soak.py 6000 medium 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -4; done | tee gpurun_out/r04_soak_graphs.txt
