cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2h
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "golden or full_size or intermediate or fused_dp or FUSE_DP or product_path or stress" 2>&1 | tail -8 > $O/pytest_gpu.log
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --steps 300 > $O/bench_b1_$i.json 2>> $O/err.log; done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/st_b1 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --steps 100 > /dev/null 2>&1)
python scripts/trace_gaps.py $O/st_b1 > $O/trace_gaps_b1.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
cat $O/pytest_gpu.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2h/bench_*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]
        print(f.split("/")[-1], "ms %.3f"%d["ms_per_step"], "dev-only %.3f"%d["device_pipeline_only_ms_per_step"], "launches", d["config"]["kernel_launches_per_step"], "stages", {k:round(v,3) for k,v in r["stage_ms"].items()})
    except Exception as e: print(f, "ERR", e)
PY
head -24 $O/trace_gaps_b1.txt
