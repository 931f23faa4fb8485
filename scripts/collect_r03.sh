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
# shader clock / power under sustained matrix load (configs[2], ~140 ms steps): sampled beside the run, ~2 samples / s
( for i in $(seq 1 40); do rocm-smi -c -P --json 2>/dev/null | head -c 2000; echo; sleep 0.2; done > $O/clocks_high_b64.jsonl ) &
SMI=$!
timeout 300 python bench.py --no-extra --no-cpu-baseline --no-roofline --config 3 --steps 60 --warmup 2 --min-seconds 0 > $O/bench_high_b64_clk.json 2>> $O/err.log
kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
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
python - <<'PY'
import json
sc=[];pw=[]
for ln in open("gpurun_out/r03/clocks_high_b64.jsonl"):
    try: d=json.loads(ln)
    except Exception: continue
    for card,v in d.items():
        for k,x in v.items():
            if "sclk" in k.lower():
                try: sc.append(float(str(x).strip("()Mhz ")))
                except Exception: pass
            if "power" in k.lower():
                try: pw.append(float(x))
                except Exception: pass
print("sclk samples", len(sc), "min/median/max", (min(sc), sorted(sc)[len(sc)//2], max(sc)) if sc else None, "power max", max(pw) if pw else None)
PY
python scripts/_show_r03.py
