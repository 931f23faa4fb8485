cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2e
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest_gpu.log
timeout 300 python bench.py --no-cpu-baseline --steps 200 > $O/bench_b1.json 2>> $O/err.log
PIPER_HIP_FOLD_LN=0 timeout 300 python bench.py --no-cpu-baseline --steps 200 > $O/bench_b1_foldln0.json 2>> $O/err.log
PIPER_HIP_FUSE_DP=0 timeout 300 python bench.py --no-cpu-baseline --steps 200 > $O/bench_b1_fusedp0.json 2>> $O/err.log
timeout 300 python bench.py --no-cpu-baseline --batch 2 --steps 100 > $O/bench_b2.json 2>> $O/err.log
PIPER_HIP_MRF2=0 timeout 300 python bench.py --no-cpu-baseline --batch 2 --steps 100 > $O/bench_b2_mrf2_0.json 2>> $O/err.log
timeout 300 python bench.py --no-cpu-baseline --batch 16 --steps 20 > $O/bench_b16.json 2>> $O/err.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/st_b1 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --steps 100 > /dev/null 2>&1)
python scripts/trace_gaps.py $O/st_b1 > $O/trace_gaps_b1.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
cat $O/pytest_gpu.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2e/bench_*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]
        print(f.split("/")[-1], "ms %.3f"%d["ms_per_step"], "dev-only %.3f"%d["device_pipeline_only_ms_per_step"], "val %.1fM"%(d["value"]/1e6), "launches", d["config"]["kernel_launches_per_step"], "stages", {k:round(v,3) for k,v in r["stage_ms"].items()}, "hifigan TF %.1f"%(r["stage_tflops"]["hifigan"]), "top", r["kernel"], "%.2f"%r["frac"], "step frac %.3f"%r["step"]["frac"])
    except Exception as e: print(f, "ERR", e)
PY
head -30 $O/trace_gaps_b1.txt
