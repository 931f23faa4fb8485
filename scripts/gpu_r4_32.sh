# Round 4, call 32: the one-utterance WN gate conv on half channel groups (conv_splitk16_kernel<true,6,5,2>: 324 workgroups of
# six waves, every weight step in flight at entry) against whole groups on twelve waves (<true,12,2>: 162 workgroups),
# PIPER_HIP_GATE_HALF = 0 | 1, one box, each setting twice; B = 1, 2, 4 and T = 64 / 256.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4u; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batched.py -m gpu -x -q -k "golden or medium_t128 or intermediate or sentences or ragged or forced" 2>&1 | tail -3
BQ="--no-extra --no-cpu-baseline --min-seconds 0.4"
for r in a b; do for h in 0 1; do
  PIPER_HIP_GATE_HALF=$h PIPER_BENCH_FULL=$O/b1_h${h}_$r.json timeout 300 python bench.py $BQ --steps 300 --warmup 10 > /dev/null 2>> $O/err.log
done; done
for h in 0 1; do
  PIPER_HIP_GATE_HALF=$h PIPER_BENCH_FULL=$O/b2_h${h}.json timeout 300 python bench.py $BQ --steps 100 --warmup 5 --batch 2 > /dev/null 2>> $O/err.log
  PIPER_HIP_GATE_HALF=$h PIPER_BENCH_FULL=$O/b4_h${h}.json timeout 300 python bench.py $BQ --steps 100 --warmup 5 --batch 4 > /dev/null 2>> $O/err.log
  PIPER_HIP_GATE_HALF=$h PIPER_BENCH_FULL=$O/t64_h${h}.json timeout 300 python bench.py $BQ --steps 200 --warmup 5 --ids 64 > /dev/null 2>> $O/err.log
  PIPER_HIP_GATE_HALF=$h PIPER_BENCH_FULL=$O/t256_h${h}.json timeout 300 python bench.py $BQ --steps 200 --warmup 5 --ids 256 > /dev/null 2>> $O/err.log
done
grep -v amdgpu.ids $O/err.log | tail -3
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r4u/*.json")):
    d=json.load(open(f)); r=d.get("roofline") or {}
    row=["%s %.2f x%.0f" % (k.replace("conv_","").replace("_kernel","")[:24], v["avg_launch_us"], v["launches_per_step"]) for k,v in r.get("kernels",{}).items() if "true" in k]
    print("%-12s ms %8.4f flow %.4f  %s" % (os.path.basename(f)[:-5], d["ms_per_step"], r.get("stage_ms",{}).get("regulate+flow",0), " | ".join(row)))
PY
