# Round 4, call 43: per-conv-shape rows of the tiled kernel at medium 64 x 128 (PIPER_HIP_PROF_SITES=1)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4y; mkdir -p $O
PIPER_HIP_PROF_SITES=1 PIPER_BENCH_FULL=$O/b64_sites.json timeout 600 python bench.py --no-extra --no-cpu-baseline --config 4 --steps 10 --warmup 3 > /dev/null 2>> $O/err.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4y/b64_sites.json")); r=d["roofline"]
print("ms_per_step", d["ms_per_step"])
rows=sorted(r["kernels"].items(), key=lambda kv: -kv[1]["avg_launch_us"]*kv[1]["launches_per_step"])
for k,v in rows:
    print("%-70s x%-3.0f %9.1f us  %7.1f TF  %5.3f  tot %7.1f us" % (k[:70], v["launches_per_step"], v["avg_launch_us"], v["tflops"], v["frac_of_mfma_peak"], v["avg_launch_us"]*v["launches_per_step"]))
PY
