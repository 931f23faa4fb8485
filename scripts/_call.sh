cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r6c12; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/gpu_pytest.log; cat $O/gpu_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python bench.py > $O/bench_default.stdout 2> $O/bench_default.err
cp bench_full.json $O/bench_default_full.json
echo "last line bytes: $(tail -n 1 $O/bench_default.stdout | wc -c)"; tail -n 1 $O/bench_default.stdout | cut -c1-1200
