# Round 4 (second session), call 12: state check on a fresh box -- the full -m gpu suite, smoke, the driver's command.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4l
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q --durations=15 > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -25 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > $O/bench_default.stdout 2> $O/bench_default.err
cp bench_full.json $O/bench_default_full.json
echo "last line bytes: $(tail -n 1 $O/bench_default.stdout | wc -c)"; tail -n 1 $O/bench_default.stdout
