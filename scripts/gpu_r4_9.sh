# Round 4, call 9: A/B on one box of (1) XCD-aware tile orders -- split-K convs / fused FFN row part-major (PIPER_HIP_XCD_ROWS),
# tiled conv column tile-major (PIPER_HIP_XCD_TILE) -- and (2) speculative graphs planned for the expected frame counts
# instead of the bucket capacity (PIPER_HIP_SPEC_EXPECT). Parity of the new orders first.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4i
mkdir -p $O
timeout 700 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
BQ="--no-extra --no-cpu-baseline --min-seconds 0.4"
run() { PIPER_BENCH_FULL=$O/$1.json env $2 timeout 300 python bench.py $BQ $3 > $O/$1.line 2>> $O/err.log; }
OFF="PIPER_HIP_XCD_ROWS=0 PIPER_HIP_XCD_TILE=0"
B1="--steps 300 --warmup 10"
for r in a b; do
  run b1_base_$r "$OFF PIPER_HIP_SPEC_EXPECT=0" "$B1"
  run b1_expect_$r "$OFF" "$B1"
  run b1_rows_$r "PIPER_HIP_XCD_TILE=0" "$B1"
  run b1_tile_$r "PIPER_HIP_XCD_ROWS=0" "$B1"
  run b1_all_$r "" "$B1"
  run b64_tile0_$r "PIPER_HIP_XCD_TILE=0" "--config 4 --steps 8 --warmup 2"
  run b64_tile1_$r "" "--config 4 --steps 8 --warmup 2"
done
run high_tile0 "PIPER_HIP_XCD_TILE=0" "--config 3 --steps 3 --warmup 1"
run high_tile1 "" "--config 3 --steps 3 --warmup 1"
for b in 2 4 8; do
  run b${b}_off "$OFF" "--batch $b --steps 100 --warmup 5"
  run b${b}_on "" "--batch $b --steps 100 --warmup 5"
done
grep -v amdgpu.ids $O/err.log | tail -3
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r4i/*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(os.path.basename(f),"ERR",e); continue
    r=d.get("roofline") or {}
    sm=r.get("stage_ms",{})
    print("%-14s ms %9.4f dev %8.4f | %s" % (os.path.basename(f)[:-5], d["ms_per_step"], d.get("device_pipeline_only_ms_per_step") or 0,
          " ".join("%s %.3f" % (k[:4], v) for k,v in sm.items())))
    ks=r.get("kernels",{})
    tot=sum(v["ms_per_step"] for v in ks.values()) or 1
    for k,v in sorted(ks.items(), key=lambda kv:-kv[1]["ms_per_step"]):
        if v["ms_per_step"]/tot > 0.025: print("     %-44s %5.1f x %9.2f us %6.1f TF" % (k[:44], v["launches_per_step"], v["avg_launch_us"], v["tflops"]))
PY
