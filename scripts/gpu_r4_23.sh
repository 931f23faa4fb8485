# Round 4, call 23: conv1x1_kernel (one-tap convs, B operand straight from global memory) against the tiled kernel,
# PIPER_HIP_CONV1X1 = 0 | 1 (64 x 64 tiles) | 2 (128 x 64), one box, each setting twice; parity of the batched family first.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4q; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_batched.py -m gpu -x -q -k "b64 or b16 or intermediate or forced" 2>&1 | tail -3
BQ="--no-extra --no-cpu-baseline --min-seconds 0.3 --warmup 2"
for r in a b; do for m in 0 1 2; do
  PIPER_HIP_PROF_SITES=1 PIPER_HIP_CONV1X1=$m PIPER_BENCH_FULL=$O/b64_m${m}_$r.json timeout 300 python bench.py $BQ --steps 6 --config 4 > /dev/null 2>> $O/err.log
done; done
for m in 0 1 2; do
  PIPER_HIP_CONV1X1=$m PIPER_BENCH_FULL=$O/b16_m${m}.json timeout 300 python bench.py $BQ --steps 10 --batch 16 > /dev/null 2>> $O/err.log
  PIPER_HIP_CONV1X1=$m PIPER_BENCH_FULL=$O/b4_m${m}.json timeout 300 python bench.py $BQ --steps 30 --batch 4 > /dev/null 2>> $O/err.log
  PIPER_HIP_CONV1X1=$m PIPER_BENCH_FULL=$O/high_m${m}.json timeout 300 python bench.py $BQ --steps 3 --config 3 > /dev/null 2>> $O/err.log
done
grep -v amdgpu.ids $O/err.log | tail -3
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r4q/*.json")):
    d=json.load(open(f)); r=d.get("roofline") or {}
    print("%-12s ms %9.4f" % (os.path.basename(f)[:-5], d["ms_per_step"]))
    if "b64" in f and f.endswith("_a.json"):
        for k,v in r.get("kernels",{}).items():
            if ("x1 " in k or "x1|" in k or "conv1x1" in k) and "|" in k: print("     %-64s %4.1f x %8.2f us %6.1f TF" % (k[:64], v["launches_per_step"], v["avg_launch_us"], v.get("tflops",0)))
PY
