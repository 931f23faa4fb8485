cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/c14; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -q --deselect tests/test_gpu_batched.py::test_every_profiled_instantiation_is_parity_tested 2>&1 | tail -30 > $O/pytest_gpu.log
tail -8 $O/pytest_gpu.log
bash scripts/collect_r05.sh 2>&1 | tail -12
