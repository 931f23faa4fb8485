# Round 4, call 41: the last WN layer's res/skip conv in front of the post + pre chain launch (PIPER_HIP_CHAIN_RS 0 | 1),
# parity first, then B=1 (T = 128, 64), B = 2, each setting twice.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4x; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batched.py -m gpu -x -q -k "golden or medium_t128 or intermediate or sentences or multi_speaker or ragged or forced" 2>&1 | tail -3
BQ="--no-extra --no-cpu-baseline --min-seconds 0.4"
for r in a b; do for h in 0 1; do
  PIPER_HIP_CHAIN_RS=$h PIPER_BENCH_FULL=$O/b1_s${h}_$r.json timeout 300 python bench.py $BQ --steps 300 --warmup 10 > /dev/null 2>> $O/err.log
done; done
for h in 0 1; do
  PIPER_HIP_CHAIN_RS=$h PIPER_BENCH_FULL=$O/t64_s${h}.json timeout 300 python bench.py $BQ --steps 200 --warmup 5 --ids 64 > /dev/null 2>> $O/err.log
  PIPER_HIP_CHAIN_RS=$h PIPER_BENCH_FULL=$O/b2_s${h}.json timeout 300 python bench.py $BQ --steps 100 --warmup 5 --batch 2 > /dev/null 2>> $O/err.log
done
grep -v amdgpu.ids $O/err.log | tail -3
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r4x/*.json")):
    d=json.load(open(f)); r=d.get("roofline") or {}
    row=["%s %.2f x%.0f" % (k[:22], v["avg_launch_us"], v["launches_per_step"]) for k,v in r.get("kernels",{}).items() if k.startswith("colchain4_kernel")]
    print("%-12s ms %8.4f launches %s flow %.4f  %s" % (os.path.basename(f)[:-5], d["ms_per_step"], d["config"]["kernel_launches_per_step"], r.get("stage_ms",{}).get("regulate+flow",0), " | ".join(row)))
PY
