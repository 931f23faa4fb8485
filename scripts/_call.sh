# scratch script of the last `gpurun -- 'bash scripts/_call.sh'` call (rewritten per call; see profiles/r06_notes.md)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/last; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.stdout 2> $O/bench_driver.err; tail -n 1 $O/bench_driver.stdout | cut -c1-300
