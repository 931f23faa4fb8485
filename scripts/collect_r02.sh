# round-2 collection: tests, smoke, bench lines of every BASELINE config, rocprof kernel stats of the same commands,
# PMC traffic passes (FETCH_SIZE / WRITE_SIZE in separate runs), 2-rank self-launch smoke (gloo, one GPU)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -5 > $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 600 python bench.py > $O/bench_b1.json 2> $O/bench_b1.err
timeout 300 python bench.py --no-cpu-baseline --batch 16 --steps 20 > $O/bench_b16.json 2>> $O/err.log
timeout 300 python bench.py --no-cpu-baseline --config 4 --steps 10 > $O/bench_b64.json 2>> $O/err.log
timeout 300 python bench.py --no-cpu-baseline --config 3 --steps 4 --warmup 2 > $O/bench_high_b64.json 2>> $O/err.log
timeout 300 python bench.py --no-cpu-baseline --preset high > $O/bench_high_b1.json 2>> $O/err.log
timeout 300 python bench.py --stream-latency > $O/stream_medium.json 2>> $O/err.log
timeout 300 python bench.py --config 5 > $O/stream_high.json 2>> $O/err.log
PIPER_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 50 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_2rank_gloo_1gpu.json 2>> $O/err.log
run_prof() {  # name, bench args
  n=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/st_$n -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline "$@" > /dev/null 2>&1)
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_fetch_$n -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --min-seconds 0 "$@" > /dev/null 2>&1)
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_write_$n -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --min-seconds 0 "$@" > /dev/null 2>&1)
}
run_prof b1 --steps 50
python scripts/trace_gaps.py $O/st_b1 > $O/trace_gaps_b1.txt 2>&1
# in-kernel phase stamps + per-launch device trace of one replayed step (tuning build of the same sources)
[ -f piper_amd/libpiper_hip_stamps.so ] && timeout 300 python scripts/stamps.py medium 128 > $O/stamps_b1.txt 2>> $O/err.log
run_prof b16 --batch 16 --steps 5 --warmup 2
run_prof high_b8 --preset high --batch 8 --steps 3 --warmup 1
python scripts/pmc_traffic.py medium/b1/t128 $O/pmc_fetch_b1 $O/pmc_write_b1 $O/r02_pmc_traffic.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, WRITE_SIZE) -- python bench.py --no-cpu-baseline --no-roofline --steps 50" > $O/traffic.log 2>&1
python scripts/pmc_traffic.py medium/b16/t128 $O/pmc_fetch_b16 $O/pmc_write_b16 $O/r02_pmc_traffic.json "same with --batch 16 --steps 5" >> $O/traffic.log 2>&1
python scripts/pmc_traffic.py high/b8/t128 $O/pmc_fetch_high_b8 $O/pmc_write_high_b8 $O/r02_pmc_traffic.json "same with --preset high --batch 8 --steps 3" >> $O/traffic.log 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -delete
cat $O/pytest_gpu.log $O/smoke.log; cat $O/traffic.log; tail -3 $O/err.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02/bench_*.json")):
    try:
        d=json.load(open(f)); r=d.get("roofline")
        line="%s ms %.3f val %.1fM x%.0f launches %s" % (f.split("/")[-1], d["ms_per_step"], d["value"]/1e6, d["x_realtime"], d["config"]["kernel_launches_per_step"])
        if r: line += " stages %s hifiTF %.1f top %s %.2f step %.3f" % ({k:round(v,3) for k,v in r["stage_ms"].items()}, r["stage_tflops"]["hifigan"], r["kernel"], r["frac"], r["step"]["frac"])
        print(line)
    except Exception as e: print(f, "ERR", e)
PY
