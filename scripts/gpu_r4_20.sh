# Round 4, call 20: L2 warm-up of a launch's weights at kernel entry (PIPER_HIP_L2WARM bits: 1 fused stage kernels, 2 split-K
# convs, 4 tiled convs) against off, one box, each setting twice; parity of the new build first.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or medium_t128 or intermediate" 2>&1 | tail -3
BQ="--no-extra --no-cpu-baseline --min-seconds 0.4 --steps 300 --warmup 10"
for r in a b; do for w in 0 1 2 4 7; do
  PIPER_HIP_L2WARM=$w PIPER_BENCH_FULL=$O/w${w}_$r.json timeout 300 python bench.py $BQ > /dev/null 2>> $O/err.log
done; done
grep -v amdgpu.ids $O/err.log | tail -3
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r4o/w*.json")):
    d=json.load(open(f)); r=d.get("roofline") or {}
    print("%-10s ms %8.4f stages %s hifiTF %.1f" % (os.path.basename(f)[:-5], d["ms_per_step"], {k[:4]:round(v,4) for k,v in r.get("stage_ms",{}).items()}, r.get("stage_tflops",{}).get("hifigan",0)))
    row=[]
    for k,v in r.get("kernels",{}).items():
        if any(x in k for x in ("mrf_kernel","splitk_sum","splitk_group","conv_mfma","splitk16_kernel<false","splitk_kernel<1")): row.append("%s %.1f" % (k.replace("conv_","").replace("_kernel","")[:22], v["avg_launch_us"]))
    print("     "+" | ".join(row))
PY
