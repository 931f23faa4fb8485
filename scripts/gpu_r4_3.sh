# Round 4, call 3: (a) the GPU tests call 2 did not reach (it stopped at the first failure, a test that read a key the
# compact bench line renamed) plus the new ones: fused up-conv (mrf_kernel<..., true>), graph cache, group broadcast, 2-rank
# bench; (b) A/B of the fused up-conv on ONE box: PIPER_HIP_UPF=0 / 1 at B = 1, 2, 4 and 0 / 2 at B = 16, 64.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4c
mkdir -p $O
( time timeout 2400 python -m pytest tests/test_gpu_batched.py -m gpu -q -k "fused_upconv or forced_kernel or rccl or engine_group or graph_cache or bench_two or fused_mrf or generator_tail or speculative or every_profiled or b64" 2>&1 | tail -25 ) > $O/pytest_a.log 2>&1
cat $O/pytest_a.log
( time timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -8 ) > $O/pytest_b.log 2>&1
cat $O/pytest_b.log
BQ="--no-extra --no-cpu-baseline --min-seconds 0.4"
run() { PIPER_BENCH_FULL=$O/$1.json env $2 timeout 300 python bench.py $BQ $3 > $O/$1.line 2>> $O/err.log; }
for b in 1 2 4; do
  for r in a b; do for u in 0 1; do run b${b}_upf${u}_$r PIPER_HIP_UPF=$u "--batch $b --steps 300 --warmup 10"; done; done
done
for b in 16 64; do
  for r in a b; do for u in 0 2; do run b${b}_upf${u}_$r PIPER_HIP_UPF=$u "--batch $b --steps 8 --warmup 2"; done; done
done
grep -v amdgpu.ids $O/err.log | tail -3
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r4c/b*_upf*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(os.path.basename(f),"ERR",e); continue
    r=d.get("roofline") or {}
    print("%-16s ms %8.4f launches %s hifi %.4f ms %.1f TF" % (os.path.basename(f)[:-5], d["ms_per_step"], d["config"].get("kernel_launches_per_step"), r.get("stage_ms",{}).get("hifigan",0), r.get("stage_tflops",{}).get("hifigan",0)))
    for k,v in r.get("kernels",{}).items():
        if k.startswith("mrf_kernel") or k.startswith("conv_mfma_kernel<2,2,1,1,16,false,64"): print("     %-42s %5.1f x %8.2f us" % (k, v["launches_per_step"], v["avg_launch_us"]))
PY
