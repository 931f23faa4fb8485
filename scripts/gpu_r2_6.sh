cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2f
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/pytest_gpu.log
for v in "" "PIPER_HIP_PERSIST_DP=0" "PIPER_HIP_SPEC=0" "PIPER_HIP_PERSIST_DP=0 PIPER_HIP_SPEC=0"; do
  n=$(echo "$v" | tr -d ' =' | tr -c 'A-Za-z0-9_' '_'); n=${n:-default}
  env $v timeout 300 python bench.py --no-cpu-baseline --steps 200 > $O/bench_b1_$n.json 2>> $O/err.log
done
timeout 300 python bench.py --no-cpu-baseline --batch 16 --steps 20 > $O/bench_b16.json 2>> $O/err.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/st_b1 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --steps 100 > /dev/null 2>&1)
python scripts/trace_gaps.py $O/st_b1 > $O/trace_gaps_b1.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
cat $O/pytest_gpu.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2f/bench_*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]
        print(f.split("/")[-1], "ms %.3f"%d["ms_per_step"], "dev-only %.3f"%d["device_pipeline_only_ms_per_step"], "api %.3f"%d["api_inclusive"]["ms_per_call"], "launches", d["config"]["kernel_launches_per_step"], "stages", {k:round(v,3) for k,v in r["stage_ms"].items()}, "top", r["kernel"], "%.2f"%r["frac"], "step frac %.3f"%r["step"]["frac"])
    except Exception as e: print(f, "ERR", e)
PY
head -30 $O/trace_gaps_b1.txt
