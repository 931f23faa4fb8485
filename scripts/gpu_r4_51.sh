# Round 4, call 51: soak of the small-call path (scripts/soak.py): 6000 calls, random texts / batch sizes / length scales
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python scripts/soak.py 6000 medium 2>&1 | grep -v Warning | tee gpurun_out/r04_soak.txt
