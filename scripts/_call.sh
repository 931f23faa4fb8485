cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r6c5; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_batched.py -m gpu -q -x -s -k "split_fused or split_matrix_modes_at_baseline" 2>&1 | grep -E "worst|passed|failed|Error|error|assert" | cut -c1-300 | tee $O/split_tests.txt
for m in f16x3 bf16x3; do
  for ms in 1 0; do
  PIPER_HIP_MRF_SPLIT=$ms PIPER_BENCH_FULL=$O/full_b64_${m}_ms$ms.json timeout 300 python bench.py --no-extra --no-cpu-baseline --config 4 --steps 10 --warmup 3 --min-seconds 0 --matrix $m > /dev/null 2>> $O/err.log
  PIPER_HIP_MRF_SPLIT=$ms PIPER_BENCH_FULL=$O/full_high_b64_${m}_ms$ms.json timeout 300 python bench.py --no-extra --no-cpu-baseline --config 3 --steps 5 --warmup 2 --min-seconds 0 --matrix $m > /dev/null 2>> $O/err.log
  PIPER_HIP_MRF_SPLIT=$ms PIPER_BENCH_FULL=$O/full_b1_${m}_ms$ms.json timeout 300 python bench.py --no-extra --no-cpu-baseline --steps 100 --min-seconds 0 --matrix $m > /dev/null 2>> $O/err.log
  done
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r6c5/full_*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(os.path.basename(f),"ERR",e); continue
    r=d.get("roofline") or {}
    print("%-32s ms %8.4f val %7.2fM stages %s" % (os.path.basename(f), d["ms_per_step"], d["value"]/1e6, {k[:4]:round(v,3) for k,v in r.get("stage_ms",{}).items()}))
    ks=r.get("kernels",{})
    for n,k in sorted(ks.items(), key=lambda kv:-kv[1]["ms_per_step"])[:6]:
        print("     %-50s %5.1f x %8.1f us = %7.3f ms %6.1f TF" % (n,k["launches_per_step"],k["avg_launch_us"],k["ms_per_step"],k["tflops"]))
PY
grep -v amdgpu.ids $O/err.log | tail -5
