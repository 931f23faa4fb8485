# Round 4, call 21: mrf_kernel<32,1,1> with two workgroups resident per CU (row stride 240: 61 KB of LDS, 128 registers)
# against the one-per-CU geometries, medium voice at batch.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batched.py -m gpu -x -q -k "fused_mrf or generator_tail" 2>&1 | tail -3
BQ="--no-extra --no-cpu-baseline --min-seconds 0.3 --warmup 2 --steps 6"
for r in a b; do for ou in 0 1 2; do
  PIPER_HIP_MRF_OU=$ou PIPER_BENCH_FULL=$O/b64_ou${ou}_$r.json timeout 300 python bench.py $BQ --config 4 > /dev/null 2>> $O/err.log
done; done
PIPER_HIP_MRF_OU=1 PIPER_BENCH_FULL=$O/b16_ou1.json timeout 300 python bench.py $BQ --batch 16 > /dev/null 2>> $O/err.log
PIPER_HIP_MRF_OU=0 PIPER_BENCH_FULL=$O/b16_ou0.json timeout 300 python bench.py $BQ --batch 16 > /dev/null 2>> $O/err.log
grep -v amdgpu.ids $O/err.log | tail -3
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r4p/*.json")):
    d=json.load(open(f)); r=d.get("roofline") or {}
    row=["%s %.1f us %.1f TF" % (k[:18], v["avg_launch_us"], v.get("tflops",0)) for k,v in r.get("kernels",{}).items() if "mrf_kernel" in k]
    print("%-14s ms %8.3f  %s" % (os.path.basename(f)[:-5], d["ms_per_step"], " | ".join(row)))
PY
