cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2z
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "golden or specul or product_path or smoke or en_us or stream or cpp or parity" 2>&1 | tail -5 > $O/pytest_gpu.log
for i in 1 2 3; do
  timeout 300 python bench.py --no-cpu-baseline --steps 300 > $O/bench_b1_$i.json 2>> $O/err.log
  PIPER_HIP_PCM_ZC=0 timeout 300 python bench.py --no-cpu-baseline --steps 300 > $O/bench_b1_nozc_$i.json 2>> $O/err.log
done
cat $O/pytest_gpu.log
python scripts/_show.py $O/bench_*.json | grep -v "^    "
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2z/bench_*.json")):
    d=json.load(open(f)); print(f.split("/")[-1], "ms %.4f"%d["ms_per_step"], "dev-only %.4f"%d["device_pipeline_only_ms_per_step"], "api %.4f"%d["api_inclusive"]["ms_per_call"])
PY
tail -3 $O/err.log
