# Round 4, call 6: where the tiled conv kernel's time goes at batch -- one level-2 profile row per conv SHAPE
# (PIPER_HIP_PROF_SITES=1) for configs[3]'s per-GPU share (medium, 64 x 128) and for B=1.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4f
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_batched.py -m gpu -q -k "engine_group_weight" 2>&1 | tail -3
BQ="--no-extra --no-cpu-baseline --min-seconds 0.4"
PIPER_HIP_PROF_SITES=1 PIPER_BENCH_FULL=$O/b64_sites.json timeout 300 python bench.py $BQ --config 4 --steps 8 --warmup 2 > $O/b64.line 2>> $O/err.log
PIPER_HIP_PROF_SITES=1 PIPER_BENCH_FULL=$O/b1_sites.json timeout 300 python bench.py $BQ --steps 200 --warmup 10 > $O/b1.line 2>> $O/err.log
PIPER_HIP_PROF_SITES=1 PIPER_BENCH_FULL=$O/high_b64_sites.json timeout 300 python bench.py $BQ --config 3 --steps 3 --warmup 1 > $O/high.line 2>> $O/err.log
grep -v amdgpu.ids $O/err.log | tail -3
python - <<'PY'
import json
for f in ("b64_sites","b1_sites","high_b64_sites"):
    d=json.load(open("gpurun_out/r4f/%s.json"%f))
    ks=d["roofline"]["kernels"]
    print(f, "ms/step %.4f" % d["ms_per_step"])
    for k,v in sorted(ks.items(), key=lambda kv:-kv[1]["ms_per_step"])[:28]:
        print("  %-66s %5.1f x %9.2f us = %8.1f us %6.1f TF  GB/s %s" % (k[:66], v["launches_per_step"], v["avg_launch_us"], v["ms_per_step"]*1e3, v["tflops"], ("%.0f" % v["algorithmic_gb_per_s"]) if v.get("algorithmic_gb_per_s") else "-"))
PY
