cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/c10; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -30 > $O/pytest_gpu.log
tail -8 $O/pytest_gpu.log
bash scripts/collect_r05.sh 2>&1 | tail -45
