cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4v; mkdir -p $O
cp piper_amd/libpiper_hip.so /tmp/new.so
BQ="--no-extra --no-cpu-baseline --min-seconds 0.4 --steps 300 --warmup 10"
for r in a b; do
  cp build/ab/libpiper_hip_base.so piper_amd/libpiper_hip.so
  PIPER_BENCH_FULL=$O/base_$r.json timeout 300 python bench.py $BQ > /dev/null 2>> $O/err.log
  cp /tmp/new.so piper_amd/libpiper_hip.so
  PIPER_BENCH_FULL=$O/new128_$r.json timeout 300 python bench.py $BQ > /dev/null 2>> $O/err.log
  PIPER_DBG_KCH64=1 PIPER_BENCH_FULL=$O/new64_$r.json timeout 300 python bench.py $BQ > /dev/null 2>> $O/err.log
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r4v/*.json")):
    d=json.load(open(f)); r=d.get("roofline") or {}
    row=["%s %.2f" % (k[:22], v["avg_launch_us"]) for k,v in r.get("kernels",{}).items() if "attno" in k]
    print("%-12s ms %8.4f text %.4f %s" % (os.path.basename(f)[:-5], d["ms_per_step"], r.get("stage_ms",{}).get("text_encoder",0), " | ".join(row)))
PY
