# Round 4, call 1: (a) the opt-in kernels of round 3 on the hardware: parity (tests/test_gpu_zz_optin.py), then on / off
# per batch size on ONE box (wn_kernel, conv_small_kernel, conv_splitk_sum_kernel<4,16>); (b) the default `python bench.py`
# exactly as the driver runs it: last stdout line must be the compact object (< 4 KB), full tables in bench_full.json.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4a
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_zz_optin.py -m gpu -q 2>&1 | tail -4
BQ="--no-extra --no-cpu-baseline --min-seconds 0.4"
run() { # name, env assignment, extra args
  PIPER_BENCH_FULL=$O/$1.json env $2 timeout 300 python bench.py $BQ $3 > $O/$1.line 2>> $O/err.log
}
for b in 1 2 4; do
  for r in a b; do for w in 0 1; do run b${b}_wn${w}_$r PIPER_HIP_WN=$w "--batch $b --steps 300 --warmup 10"; done; done
done
for r in a b; do for u in 0 1; do run b1_upre${u}_$r PIPER_HIP_UPPRE=$u "--steps 300 --warmup 10"; done; done
for r in a b; do for d in 2 16; do run b1_sumd${d}_$r PIPER_HIP_SUMD=$d "--steps 300 --warmup 10"; done; done
run b1_all_a "PIPER_HIP_WN=1 PIPER_HIP_UPPRE=1 PIPER_HIP_SUMD=16" "--steps 300 --warmup 10"
grep -v amdgpu.ids $O/err.log | tail -3
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r4a/*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(os.path.basename(f),"ERR",e); continue
    r=d.get("roofline") or {}
    print("%-18s ms %8.4f launches %s stages %s hifiTF %.1f" % (os.path.basename(f)[:-5], d["ms_per_step"], d["config"].get("kernel_launches_per_step"), {k[:4]:round(v,3) for k,v in r.get("stage_ms",{}).items()}, r.get("stage_tflops",{}).get("hifigan",0)))
    for k,v in r.get("kernels",{}).items():
        if any(x in k for x in ("wn_kernel","splitk16_kernel<true","colchain4","conv_small","conv_mfma","conv_splitk_sum")): print("     %-40s %5.1f x %7.2f us" % (k, v["launches_per_step"], v["avg_launch_us"]))
print(d.get("xcd_dispatch"))
PY
# (b) the driver's command
( time timeout 900 python bench.py > $O/default.stdout 2> $O/default.stderr ) 2>&1 | tail -3
cp bench_full.json $O/default_full.json
tail -c 300 $O/default.stderr
echo "last line bytes: $(tail -n 1 $O/default.stdout | wc -c)"
tail -n 1 $O/default.stdout
