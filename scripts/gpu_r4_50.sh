# Round 4, call 50: with the spills gone, is 4 output units per wave (mrf_kernel<32,4,1>, N = 512) still slower than 3 at batch?
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4w2; mkdir -p $O
BQ="--no-extra --no-cpu-baseline --min-seconds 0.3"
for r in a b; do for u in 0 4; do
  PIPER_HIP_MRF_OU=$u PIPER_BENCH_FULL=$O/m64_u${u}_$r.json timeout 300 python bench.py $BQ --config 4 --steps 10 --warmup 3 > /dev/null 2>> $O/err.log
  PIPER_HIP_MRF_OU=$u PIPER_BENCH_FULL=$O/m16_u${u}_$r.json timeout 300 python bench.py $BQ --steps 30 --warmup 5 --batch 16 > /dev/null 2>> $O/err.log
  PIPER_HIP_MRF_OU=$u PIPER_BENCH_FULL=$O/m04_u${u}_$r.json timeout 300 python bench.py $BQ --steps 50 --warmup 5 --batch 4 > /dev/null 2>> $O/err.log
done; done
grep -v amdgpu.ids $O/err.log | tail -3
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r4w2/*.json")):
    d=json.load(open(f)); r=d.get("roofline") or {}
    row=["%s %.1f" % (k[11:], v["avg_launch_us"]) for k,v in r.get("kernels",{}).items() if k.startswith("mrf_kernel")]
    print("%-12s ms %9.4f  %s" % (os.path.basename(f)[:-5], d["ms_per_step"], " | ".join(row)))
PY
