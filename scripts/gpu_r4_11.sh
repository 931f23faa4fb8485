# Round 4, call 11: the up-conv tile stored as 16-byte pieces straight from the accumulators (PIPER_HIP_CONVT_VEC=1, new)
# against the previous default (through LDS for stride 8, element-wise otherwise), one box, each setting twice.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4k
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batched.py tests/test_gpu_parity.py -m gpu -x -q -k "CONVT or golden or high_b64 or medium_b64 or b16 or intermediate" 2>&1 | tail -3
BQ="--no-extra --no-cpu-baseline --min-seconds 0.4"
run() { PIPER_HIP_PROF_SITES=1 PIPER_BENCH_FULL=$O/$1.json env $2 timeout 300 python bench.py $BQ $3 > $O/$1.line 2>> $O/err.log; }
for r in a b; do for v in 0 1; do
  run b1_vec${v}_$r PIPER_HIP_CONVT_VEC=$v "--steps 300 --warmup 10"
  run b64_vec${v}_$r PIPER_HIP_CONVT_VEC=$v "--config 4 --steps 8 --warmup 2"
done; done
for v in 0 1; do run high_vec${v} PIPER_HIP_CONVT_VEC=$v "--config 3 --steps 3 --warmup 1"; run b8_vec${v} PIPER_HIP_CONVT_VEC=$v "--batch 8 --steps 50 --warmup 5"; done
grep -v amdgpu.ids $O/err.log | tail -3
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r4k/*_vec*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(os.path.basename(f),"ERR",e); continue
    r=d.get("roofline") or {}
    print("%-14s ms %9.4f hifi %.4f ms %.1f TF" % (os.path.basename(f)[:-5], d["ms_per_step"], r.get("stage_ms",{}).get("hifigan",0), r.get("stage_tflops",{}).get("hifigan",0)))
    for k,v in r.get("kernels",{}).items():
        if " e6 " in k: print("     %-62s %4.1f x %9.2f us %6.1f TF" % (k[:62], v["launches_per_step"], v["avg_launch_us"], v["tflops"]))
PY
