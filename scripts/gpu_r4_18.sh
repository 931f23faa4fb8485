# Round 4, call 18 (scratch build, not committed): every layer reads layer 0's weights (PIPER_DBG_SAMEW) -- the step time
# with the weights L2-hot bounds what a prefetch of the next launch's weights can give.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4n; mkdir -p $O
BQ="--no-extra --no-cpu-baseline --min-seconds 0.4 --steps 300 --warmup 10"
for r in a b; do
  PIPER_BENCH_FULL=$O/base_$r.json timeout 300 python bench.py $BQ > /dev/null 2>> $O/err.log
  PIPER_DBG_SAMEW=1 PIPER_BENCH_FULL=$O/same_$r.json timeout 300 python bench.py $BQ > /dev/null 2>> $O/err.log
done
grep -v amdgpu.ids $O/err.log | tail -3
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r4n/*.json")):
    d=json.load(open(f)); r=d.get("roofline") or {}
    print("%-12s ms %8.4f stages %s" % (os.path.basename(f), d["ms_per_step"], {k[:4]:round(v,4) for k,v in r.get("stage_ms",{}).items()}))
    for k,v in r.get("kernels",{}).items():
        if v["launches_per_step"]>=6: print("     %-46s %5.1f x %7.2f us" % (k[:46], v["launches_per_step"], v["avg_launch_us"]))
PY
