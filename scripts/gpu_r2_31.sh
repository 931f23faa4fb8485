# bench lines only (final bench.py harness), for profiles/r02_bench_*.json
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2ae
mkdir -p $O
timeout 600 python bench.py > $O/bench_b1.json 2> $O/bench_b1.err
timeout 300 python bench.py --no-cpu-baseline --batch 16 --steps 20 > $O/bench_b16.json 2>> $O/err.log
timeout 300 python bench.py --no-cpu-baseline --config 4 --steps 10 > $O/bench_b64.json 2>> $O/err.log
timeout 300 python bench.py --no-cpu-baseline --config 3 --steps 4 --warmup 2 > $O/bench_high_b64.json 2>> $O/err.log
timeout 300 python bench.py --no-cpu-baseline --preset high > $O/bench_high_b1.json 2>> $O/err.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2ae/bench_*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]
        print(f.split("/")[-1], "ms %.4f"%d["ms_per_step"], "%.1fM"%(d["value"]/1e6), "x%.0f"%d["x_realtime"], "launches", d["config"]["kernel_launches_per_step"], "step %.3f"%r["step"]["frac"], "hifiTF %.1f"%r["stage_tflops"]["hifigan"], "api %.3f"%d["api_inclusive"]["ms_per_call"])
    except Exception as e: print(f, "ERR", e)
PY
