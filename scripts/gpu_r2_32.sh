cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2af
mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "ragged or B64 or group_two or golden or specul or multi_speaker" 2>&1 | tail -4 > $O/pytest_gpu.log
timeout 300 python bench.py --no-cpu-baseline --no-roofline --batch 16 --steps 20 > $O/bench_b16.json 2>> $O/err.log
PIPER_HIP_PCM_ZC=0 timeout 300 python bench.py --no-cpu-baseline --no-roofline --batch 16 --steps 20 > $O/bench_b16_nozc.json 2>> $O/err.log
timeout 300 python bench.py --no-cpu-baseline --no-roofline --config 4 --steps 10 > $O/bench_b64.json 2>> $O/err.log
PIPER_HIP_PCM_ZC=0 timeout 300 python bench.py --no-cpu-baseline --no-roofline --config 4 --steps 10 > $O/bench_b64_nozc.json 2>> $O/err.log
cat $O/pytest_gpu.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2af/bench_*.json")):
    d=json.load(open(f)); print(f.split("/")[-1], "ms %.4f"%d["ms_per_step"], "%.1fM"%(d["value"]/1e6), "dev-only %.4f"%d["device_pipeline_only_ms_per_step"], "api %.4f"%d["api_inclusive"]["ms_per_call"])
PY
tail -2 $O/err.log
