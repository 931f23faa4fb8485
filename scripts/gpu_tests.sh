cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${1:-tests}
mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -q -s 2>&1 | tail -120 > $O/pytest_gpu.log
tail -40 $O/pytest_gpu.log
