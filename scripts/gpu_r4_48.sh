# Round 4, call 48: mrf_kernel's B operand buffered in half steps (40 registers fewer for the five-unit instantiations: no
# spilled VGPRs in <64,3,2> / <32,4,1>): library variants h0 = none (before), h1 = five-unit instantiations only, h2 = all.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4v; mkdir -p $O
BQ="--no-extra --no-cpu-baseline --min-seconds 0.3"
for r in a b; do for h in 0 1 2; do
  cp piper_amd/libpiper_hip_h$h.so piper_amd/libpiper_hip.so
  PIPER_BENCH_FULL=$O/b1_h${h}_$r.json timeout 300 python bench.py $BQ --steps 300 --warmup 10 > /dev/null 2>> $O/err.log
  PIPER_BENCH_FULL=$O/m64_h${h}_$r.json timeout 300 python bench.py $BQ --config 4 --steps 10 --warmup 3 > /dev/null 2>> $O/err.log
done; done
for h in 0 1 2; do
  cp piper_amd/libpiper_hip_h$h.so piper_amd/libpiper_hip.so
  PIPER_BENCH_FULL=$O/m16_h${h}.json timeout 300 python bench.py $BQ --steps 30 --warmup 5 --batch 16 > /dev/null 2>> $O/err.log
  PIPER_BENCH_FULL=$O/h64_h${h}.json timeout 300 python bench.py $BQ --config 3 --steps 4 --warmup 2 > /dev/null 2>> $O/err.log
done
cp piper_amd/libpiper_hip_h1.so piper_amd/libpiper_hip.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batched.py -m gpu -x -q -k "golden or medium_t128 or b64 or forced or ragged" 2>&1 | tail -3
grep -v amdgpu.ids $O/err.log | tail -3
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r4v/*.json")):
    d=json.load(open(f)); r=d.get("roofline") or {}
    row=["%s %.1f" % (k[11:], v["avg_launch_us"]) for k,v in r.get("kernels",{}).items() if k.startswith("mrf_kernel")]
    print("%-12s ms %9.4f  %s" % (os.path.basename(f)[:-5], d["ms_per_step"], " | ".join(row)))
PY
