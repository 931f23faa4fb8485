# FIRST call of the next round (prepared at the end of round 3, when the GPU-minutes were spent): the opt-in fused WN
# layers (PIPER_HIP_WN=1, kernels/wn.h) on the hardware for the first time -- parity (tests/test_gpu_zz_optin.py), then
# on / off per batch size on one box, with the per-kernel table and the box's XCD dispatch pattern (xcd_dispatch).
# If it wins: make it the default (engine.h wn_ = 1; pack unconditionally), move its parity case into FORCED, re-collect.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4a
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_zz_optin.py -m gpu -q 2>&1 | tail -4
BQ="--no-extra --no-cpu-baseline --min-seconds 0.4"
for b in 1 2 4; do
  for w in 0 1 0 1; do
    PIPER_HIP_WN=$w timeout 300 python bench.py $BQ --batch $b --steps 300 --warmup 10 > $O/b${b}_wn${w}_$RANDOM.json 2>> $O/err.log
  done
done
# the preloaded small-K up-convs (PIPER_HIP_UPPRE=1, kernels/conv_small.h), B=1
for u in 0 1 0 1; do
  PIPER_HIP_UPPRE=$u timeout 300 python bench.py $BQ --steps 300 --warmup 10 > $O/b1_upre${u}_$RANDOM.json 2>> $O/err.log
done
# the 16-deep weight ring of the K-concatenated stage-1 launch (PIPER_HIP_SUMD=16), B=1
for d in 2 16 2 16; do
  PIPER_HIP_SUMD=$d timeout 300 python bench.py $BQ --steps 300 --warmup 10 > $O/b1_sumd${d}_$RANDOM.json 2>> $O/err.log
done
grep -v amdgpu.ids $O/err.log | tail -3
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r4a/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f),"ERR",e); continue
    r=d.get("roofline") or {}
    print("%-22s ms %8.4f launches %s stages %s" % (os.path.basename(f), d["ms_per_step"], d["config"].get("kernel_launches_per_step"), {k[:4]:round(v,3) for k,v in r.get("stage_ms",{}).items()}))
    for k,v in r.get("kernels",{}).items():
        if any(x in k for x in ("wn_kernel","splitk16_kernel<true","colchain4","conv_small","conv_mfma","conv_splitk_sum")): print("     %-40s %5.1f x %7.2f us" % (k, v["launches_per_step"], v["avg_launch_us"]))
print(d.get("xcd_dispatch"))
PY
