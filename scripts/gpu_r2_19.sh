cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2s
mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "group_two or persist or golden or speculative" 2>&1 | tail -15 > $O/pytest_gpu.log
timeout 300 python bench.py --no-cpu-baseline --steps 300 > $O/bench_b1_1.json 2>> $O/err.log
cat $O/pytest_gpu.log
python scripts/_show.py $O/bench_*.json | grep -v "^    "
tail -3 $O/err.log
