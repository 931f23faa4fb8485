cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2i
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "COLCHAIN or golden or full_size or intermediate or product_path or stress or medium-B16" 2>&1 | tail -8 > $O/pytest_gpu.log
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --steps 300 > $O/bench_b1_$i.json 2>> $O/err.log
  PIPER_HIP_COLCHAIN=0 timeout 300 python bench.py --no-cpu-baseline --steps 300 > $O/bench_b1_nochain_$i.json 2>> $O/err.log
done
timeout 300 python bench.py --no-cpu-baseline --batch 16 --steps 50 > $O/bench_b16.json 2>> $O/err.log
timeout 300 python scripts/stamps.py medium 128 > $O/stamps_b1.txt 2>> $O/err.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/st_b1 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --steps 100 > /dev/null 2>&1)
python scripts/trace_gaps.py $O/st_b1 > $O/trace_gaps_b1.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
cat $O/pytest_gpu.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2i/bench_*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]
        print(f.split("/")[-1], "ms %.3f"%d["ms_per_step"], "dev-only %.3f"%d["device_pipeline_only_ms_per_step"], "launches", d["config"]["kernel_launches_per_step"], "stages", {k:round(v,3) for k,v in r["stage_ms"].items()})
    except Exception as e: print(f, "ERR", e)
PY
cat $O/stamps_b1.txt
head -26 $O/trace_gaps_b1.txt
tail -5 $O/err.log
