# round 2, GPU call 3: mrf2 parity + A/B benches (fused MRF stage kernel on/off) at B=1 and B=16, kernel stats
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2c
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batched.py -m gpu -x -q -k "mrf2 or intermediate or medium_b64" 2>&1 | tail -15 > $O/pytest_mrf2.log
for m in 0 1; do
  PIPER_HIP_MRF2=$m timeout 300 python bench.py --no-cpu-baseline --steps 100 > $O/bench_b1_mrf2_$m.json 2>> $O/err.log
  PIPER_HIP_MRF2=$m timeout 300 python bench.py --no-cpu-baseline --batch 16 --steps 20 > $O/bench_b16_mrf2_$m.json 2>> $O/err.log
  PIPER_HIP_MRF2=$m timeout 300 python bench.py --no-cpu-baseline --config 4 --steps 10 > $O/bench_b64_mrf2_$m.json 2>> $O/err.log
done
PIPER_HIP_MRF2=1 timeout 300 python bench.py --no-cpu-baseline --config 3 --steps 4 --warmup 2 > $O/bench_high_b64_mrf2_1.json 2>> $O/err.log
PIPER_HIP_MRF2=0 timeout 300 python bench.py --no-cpu-baseline --config 3 --steps 4 --warmup 2 > $O/bench_high_b64_mrf2_0.json 2>> $O/err.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/st_b1 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --steps 100 > /dev/null 2>&1)
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
cat $O/pytest_mrf2.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2c/bench_*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]
        print(f.split("/")[-1], "ms %.3f"%d["ms_per_step"], "val %.1fM"%(d["value"]/1e6), "launches", d["config"]["kernel_launches_per_step"], "hifigan ms %.3f TF %.1f"%(r["stage_ms"]["hifigan"], r["stage_tflops"]["hifigan"]), "top", r["kernel"], "%.2f"%r["frac"], "step frac %.3f"%r["step"]["frac"])
        for k,v in r["kernels"].items():
            if k.startswith("mrf2"): print("    ",k,{a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items()})
    except Exception as e: print(f, "ERR", e)
PY
