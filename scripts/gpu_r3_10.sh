# round 3, call 10: per-kernel table with the polyphase up-convs forced to the split-K form at B=1
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3j
mkdir -p $O
BQ="--no-extra --no-cpu-baseline --min-seconds 0.5 --steps 300"
PIPER_HIP_SPLITK_MAX=450 timeout 300 python bench.py $BQ > $O/b1_splitk450.json 2>> $O/err.log
timeout 300 python bench.py $BQ > $O/b1_default.json 2>> $O/err.log
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r3j/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
    print(os.path.basename(f), "ms %.4f" % d["ms_per_step"], {k[:4]:round(v,3) for k,v in r["stage_ms"].items()})
    for k,v in r["kernels"].items():
        print("   %-48s %5.1f x %7.2f us = %7.1f us %6.1f TF" % (k, v["launches_per_step"], v["avg_launch_us"], v["ms_per_step"]*1e3, v["tflops"]))
PY
