# round 2, GPU call 1: launch-floor microbenchmark, full GPU test suite (incl. the new batched-kernel parity tests),
# default bench line with the new methodology, kernel trace with per-launch gaps
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2a
mkdir -p $O
timeout 120 ./scripts/microbench/launch_floor > $O/launch_floor.txt 2>&1
timeout 2400 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -60 > $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 600 python bench.py > $O/bench_b1.json 2> $O/bench_b1.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/st_b1 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --steps 100 > /dev/null 2>&1)
python scripts/trace_gaps.py $O/st_b1 > $O/trace_gaps_b1.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
find $O -name "*agent_info.csv" -delete
cat $O/launch_floor.txt; tail -5 $O/pytest_gpu.log; cat $O/smoke.log; head -c 600 $O/bench_b1.json; echo; head -30 $O/trace_gaps_b1.txt
