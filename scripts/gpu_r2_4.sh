# round 2, GPU call 4: full GPU suite with mrf2 + fused DP; A/B benches
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2d
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest_gpu.log
for m in 0 1; do
  PIPER_HIP_MRF2=$m timeout 300 python bench.py --no-cpu-baseline --steps 100 > $O/bench_b1_mrf2_$m.json 2>> $O/err.log
  PIPER_HIP_MRF2=$m timeout 300 python bench.py --no-cpu-baseline --batch 16 --steps 20 > $O/bench_b16_mrf2_$m.json 2>> $O/err.log
done
PIPER_HIP_MRF2=1 timeout 300 python bench.py --no-cpu-baseline --batch 4 --steps 40 > $O/bench_b4_mrf2_1.json 2>> $O/err.log
PIPER_HIP_MRF2=0 timeout 300 python bench.py --no-cpu-baseline --batch 4 --steps 40 > $O/bench_b4_mrf2_0.json 2>> $O/err.log
PIPER_HIP_FUSE_DP=0 timeout 300 python bench.py --no-cpu-baseline --steps 100 > $O/bench_b1_fusedp_0.json 2>> $O/err.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/st_b1 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --steps 100 > /dev/null 2>&1)
python scripts/trace_gaps.py $O/st_b1 > $O/trace_gaps_b1.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
cat $O/pytest_gpu.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2d/bench_*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]
        print(f.split("/")[-1], "ms %.3f"%d["ms_per_step"], "val %.1fM"%(d["value"]/1e6), "launches", d["config"]["kernel_launches_per_step"], "stages", {k:round(v,3) for k,v in r["stage_ms"].items()}, "hifigan TF %.1f"%(r["stage_tflops"]["hifigan"]), "top", r["kernel"], "%.2f"%r["frac"], "step frac %.3f"%r["step"]["frac"])
        for k,v in r["kernels"].items():
            if k.startswith("mrf2"): print("    ",k,{a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items()})
    except Exception as e: print(f, "ERR", e)
PY
head -40 $O/trace_gaps_b1.txt
