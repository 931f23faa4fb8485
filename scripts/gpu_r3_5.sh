# round 3, call 5: matrix mode bf16x3 (conv_bf3_kernel) -- parity tests on the GPU, A/B against f32 at B=1/16/64 medium and
# high B=8/64, MRF policy (fused f32 stage kernel vs conv by conv on the bf16 pipe)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3e
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batched.py -m gpu -q -s -k "bf16x3" 2>&1 | tail -25 > $O/pytest_bf3.log
BQ="--no-extra --no-cpu-baseline --min-seconds 0.5"
for m in f32 bf16x3; do
  timeout 300 python bench.py $BQ --matrix $m --steps 200 > $O/b1_$m.json 2>> $O/err.log
  timeout 300 python bench.py $BQ --matrix $m --batch 16 --steps 20 --warmup 3 > $O/b16_$m.json 2>> $O/err.log
  timeout 300 python bench.py $BQ --matrix $m --config 4 --steps 10 --warmup 3 > $O/b64_$m.json 2>> $O/err.log
  timeout 300 python bench.py $BQ --matrix $m --preset high --batch 8 --steps 5 --warmup 2 > $O/h8_$m.json 2>> $O/err.log
  timeout 300 python bench.py $BQ --matrix $m --config 3 --steps 4 --warmup 2 > $O/h64_$m.json 2>> $O/err.log
done
PIPER_HIP_MRF=2 timeout 300 python bench.py $BQ --matrix bf16x3 --batch 16 --steps 20 --warmup 3 > $O/b16_bf16x3_fusedmrf.json 2>> $O/err.log
PIPER_HIP_MRF=2 timeout 300 python bench.py $BQ --matrix bf16x3 --config 4 --steps 10 --warmup 3 > $O/b64_bf16x3_fusedmrf.json 2>> $O/err.log
PIPER_HIP_BF3_MINF=0 timeout 300 python bench.py $BQ --matrix bf16x3 --steps 200 > $O/b1_bf16x3_minf0.json 2>> $O/err.log
PIPER_HIP_BF3_MINF=0 timeout 300 python bench.py $BQ --matrix bf16x3 --batch 4 --steps 50 > $O/b4_bf16x3_minf0.json 2>> $O/err.log
timeout 300 python bench.py $BQ --matrix bf16x3 --batch 4 --steps 50 > $O/b4_bf16x3.json 2>> $O/err.log
cat $O/pytest_bf3.log; grep -v amdgpu.ids $O/err.log | tail -5
python - <<'PY'
import json,glob,os
O="gpurun_out/r3e/"
for f in sorted(glob.glob(O+"*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f),"ERR",e); continue
    r=d.get("roofline") or {}
    print("%-28s ms %8.3f val %7.1fM stages %s step %.3f" % (os.path.basename(f), d["ms_per_step"], d["value"]/1e6,
          {k[:4]:round(v,3) for k,v in r.get("stage_ms",{}).items()}, r.get("step",{}).get("frac",0)))
    ks=sorted(r.get("kernels",{}).items(), key=lambda kv:-kv[1]["ms_per_step"])[:6]
    for k,v in ks:
        print("      %-46s n %5.1f us %8.2f TF %6.1f GB/s %s" % (k[:46], v["launches_per_step"], v["avg_launch_us"], v["tflops"], round(v.get("algorithmic_gb_per_s") or 0)))
PY
