cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/call; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/gpu_pytest.log; cat $O/gpu_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
