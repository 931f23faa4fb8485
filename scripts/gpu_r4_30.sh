# Round 4, call 30: mrf_kernel<64, OU, 3> (halo units of the 64-channel stage as (unit, row tile) singles, three per wave)
# against the previous commit's whole-unit form: parity of the stage-kernel family, then B=1 / 16 / 64 medium, alternating twice.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4t; mkdir -p $O
cp piper_amd/libpiper_hip.so /tmp/new.so
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batched.py -m gpu -x -q -k "golden or medium_t128 or b64 or b16 or fused_mrf or generator_tail or stress or sentences" 2>&1 | tail -3
BQ="--no-extra --no-cpu-baseline --min-seconds 0.3"
for r in 1 2; do for w in base new; do
  if [ $w = base ]; then cp build/ab/libpiper_hip_base.so piper_amd/libpiper_hip.so; else cp /tmp/new.so piper_amd/libpiper_hip.so; fi
  PIPER_BENCH_FULL=$O/b1_${w}_$r.json timeout 300 python bench.py $BQ --steps 300 --warmup 10 > /dev/null 2>> $O/err.log
  PIPER_BENCH_FULL=$O/b64_${w}_$r.json timeout 300 python bench.py $BQ --steps 6 --warmup 2 --config 4 > /dev/null 2>> $O/err.log
  [ $r = 1 ] && PIPER_BENCH_FULL=$O/b16_${w}_$r.json timeout 300 python bench.py $BQ --steps 10 --warmup 2 --batch 16 > /dev/null 2>> $O/err.log
done; done
cp /tmp/new.so piper_amd/libpiper_hip.so
grep -v amdgpu.ids $O/err.log | tail -3
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r4t/*.json")):
    d=json.load(open(f)); r=d.get("roofline") or {}
    row=["%s %.1f us %.1f TF" % (k[:18], v["avg_launch_us"], v.get("tflops",0)) for k,v in r.get("kernels",{}).items() if "mrf_kernel<64" in k]
    print("%-14s ms %9.4f  %s" % (os.path.basename(f)[:-5], d["ms_per_step"], " | ".join(row)))
PY
