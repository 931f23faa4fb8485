# Round 4, calls 24 ...: A/B of the tree's build against build/ab/libpiper_hip_base.so (the previous commit) on B=1, 64 x 128 medium and high, one box, alternating twice
# three elsewhere) in the tiled and split-K kernels + conv1x1_kernel<1> as the default, against the previous commit's build:
# B=1, 64 x 128 medium, 64 x 128 high, one box, alternating twice.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4r; mkdir -p $O
cp piper_amd/libpiper_hip.so /tmp/new.so
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batched.py -m gpu -x -q -k "golden or medium_t128 or b64 or b16 or forced or intermediate or sentences" 2>&1 | tail -3
BQ="--no-extra --no-cpu-baseline --min-seconds 0.3"
for r in 1 2; do for w in base new; do
  if [ $w = base ]; then cp build/ab/libpiper_hip_base.so piper_amd/libpiper_hip.so; else cp /tmp/new.so piper_amd/libpiper_hip.so; fi
  PIPER_BENCH_FULL=$O/b1_${w}_$r.json timeout 300 python bench.py $BQ --steps 300 --warmup 10 > /dev/null 2>> $O/err.log
  PIPER_BENCH_FULL=$O/b64_${w}_$r.json timeout 300 python bench.py $BQ --steps 6 --warmup 2 --config 4 > /dev/null 2>> $O/err.log
  [ $r = 1 ] && PIPER_BENCH_FULL=$O/high_${w}_$r.json timeout 300 python bench.py $BQ --steps 3 --warmup 1 --config 3 > /dev/null 2>> $O/err.log
done; done
cp /tmp/new.so piper_amd/libpiper_hip.so
grep -v amdgpu.ids $O/err.log | tail -3
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r4r/*.json")):
    d=json.load(open(f)); r=d.get("roofline") or {}
    row=["%s %.1f" % (k.replace("conv_","").replace("_kernel","")[:26], v["avg_launch_us"]) for k,v in r.get("kernels",{}).items() if any(x in k for x in ("true","splitk16","conv1x1"))]
    print("%-14s ms %9.4f  %s" % (os.path.basename(f)[:-5], d["ms_per_step"], " | ".join(row)))
PY
