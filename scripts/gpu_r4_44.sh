# Round 4, call 44: conv1x1_kernel with 1 | 2 | 4 consecutive columns per lane (PIPER_HIP_CONV1X1), parity first, then
# medium 64 x 128 (each setting twice), medium 16 x 128, high 64 x 128.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4z; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_batched.py -m gpu -x -q -k "forced or b64 or ragged" 2>&1 | tail -3
BQ="--no-extra --no-cpu-baseline --min-seconds 0"
for r in a b; do for h in 1 2 4; do
  PIPER_HIP_CONV1X1=$h PIPER_BENCH_FULL=$O/m64_c${h}_$r.json timeout 300 python bench.py $BQ --config 4 --steps 10 --warmup 3 > /dev/null 2>> $O/err.log
done; done
for h in 1 2 4; do
  PIPER_HIP_CONV1X1=$h PIPER_BENCH_FULL=$O/m16_c${h}.json timeout 300 python bench.py $BQ --steps 30 --warmup 5 --batch 16 > /dev/null 2>> $O/err.log
  PIPER_HIP_CONV1X1=$h PIPER_BENCH_FULL=$O/h64_c${h}.json timeout 300 python bench.py $BQ --config 3 --steps 4 --warmup 2 > /dev/null 2>> $O/err.log
done
grep -v amdgpu.ids $O/err.log | tail -3
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r4z/*.json")):
    d=json.load(open(f)); r=d.get("roofline") or {}
    row=["%s %.2f x%.0f %.1fTF" % (k[:22], v["avg_launch_us"], v["launches_per_step"], v["tflops"]) for k,v in r.get("kernels",{}).items() if k.startswith("conv1x1")]
    print("%-12s ms %8.4f  %s" % (os.path.basename(f)[:-5], d["ms_per_step"], " | ".join(row)))
PY
