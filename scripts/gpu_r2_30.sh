cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2ad
mkdir -p $O
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --steps 300 > $O/bench_b1_$i.json 2>> $O/err.log; done
timeout 300 python bench.py --no-cpu-baseline --batch 16 --steps 30 > $O/bench_b16.json 2>> $O/err.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2ad/bench_*.json")):
    d=json.load(open(f)); print(f.split("/")[-1], "ms %.4f"%d["ms_per_step"], "%.1fM"%(d["value"]/1e6), "dev-only %.4f"%d["device_pipeline_only_ms_per_step"], "api %.4f"%d["api_inclusive"]["ms_per_call"], d["config"].get("samples_per_step"))
PY
tail -2 $O/err.log
