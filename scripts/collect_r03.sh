# round-3 collection (one build, one box): GPU suite, smoke, the ONE-invocation bench line (headline + extra_configs),
# rocprofv3 kernel stats of the headline / configs[3]-share / high commands, PMC traffic passes (FETCH_SIZE / WRITE_SIZE
# in separate runs, counters only with --kernel-trace), RCCL at world size 1
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
PIPER_BENCH_DIST=1 timeout 300 python bench.py --no-extra --no-cpu-baseline --no-roofline --steps 50 > $O/bench_nccl_ws1.json 2>> $O/err.log
run_stats() {  # name, bench args
  n=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/st_$n -- python $GRAFT_REPO_ROOT/bench.py --no-extra --no-cpu-baseline --no-roofline "$@" > /dev/null 2>&1)
}
run_pmc() {
  n=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_fetch_$n -- python $GRAFT_REPO_ROOT/bench.py --no-extra --no-cpu-baseline --no-roofline --min-seconds 0 "$@" > /dev/null 2>&1)
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_write_$n -- python $GRAFT_REPO_ROOT/bench.py --no-extra --no-cpu-baseline --no-roofline --min-seconds 0 "$@" > /dev/null 2>&1)
}
run_stats b1 --steps 50
run_stats b64 --config 4 --steps 5 --warmup 2
run_stats high_b64 --config 3 --steps 2 --warmup 1 --min-seconds 0
run_pmc b1 --steps 50
run_pmc b64 --config 4 --steps 3 --warmup 1
python scripts/pmc_traffic.py medium/b1/t128 $O/pmc_fetch_b1 $O/pmc_write_b1 $O/r03_pmc_traffic.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, WRITE_SIZE) -- python bench.py --no-extra --no-cpu-baseline --no-roofline --steps 50" > $O/traffic.log 2>&1
python scripts/pmc_traffic.py medium/b64/t128 $O/pmc_fetch_b64 $O/pmc_write_b64 $O/r03_pmc_traffic.json "same with --config 4 --steps 3" >> $O/traffic.log 2>&1
for n in b1 b64 high_b64; do f=$(find $O/st_$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r03_${n}_kernel_stats.csv; done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -delete
cat $O/pytest_gpu.log $O/smoke.log; tail -3 $O/traffic.log; tail -3 $O/err.log $O/bench_default.err
python scripts/_show_r03.py
