# Round 4, call 13: tile-quantisation probe. Per-kernel launch durations of the medium voice at batch sizes chosen so that
# the tiled gate conv's workgroup count sits just under / just over a whole number of residency generations
# (21 workgroups per utterance, 768 resident): B = 36 (0.98), 40 (1.09), 55 (1.50), 64 (1.75), 73 (2.00), 80 (2.19).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4m
mkdir -p $O
BQ="--no-extra --no-cpu-baseline --min-seconds 0.3 --warmup 2 --steps 6"
for B in 36 40 55 64 73 80; do
  PIPER_HIP_PROF_SITES=1 PIPER_BENCH_FULL=$O/b$B.json timeout 300 python bench.py $BQ --batch $B > $O/b$B.line 2>> $O/err.log
done
grep -v amdgpu.ids $O/err.log | tail -3
python - <<'PY'
import json,glob,os,re
for B in (36,40,55,64,73,80):
    f="gpurun_out/r4m/b%d.json"%B
    try: d=json.load(open(f))
    except Exception as e: print(B,"ERR",e); continue
    r=d.get("roofline") or {}
    print("B=%d ms %.3f  ms/utt %.4f"%(B,d["ms_per_step"],d["ms_per_step"]/B))
    for k,v in sorted(r.get("kernels",{}).items(), key=lambda kv:-kv[1]["ms_per_step"])[:14]:
        print("   %-70s %5.1f x %9.2f us  %6.1f TF  us/utt %7.3f" % (k[:70], v["launches_per_step"], v["avg_launch_us"], v.get("tflops",0), v["avg_launch_us"]/B))
PY
