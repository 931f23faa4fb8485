cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python scripts/stress_parity.py 24 > gpurun_out/stress.log 2>&1
tail -3 gpurun_out/stress.log
