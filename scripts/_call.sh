cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r05/gpu_pytest.log; tail -4 gpurun_out/r05/gpu_pytest.log
timeout 2400 bash scripts/collect_r05.sh 2>&1 | tail -45
