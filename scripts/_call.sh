cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/c2; mkdir -p $O
BQ="--no-extra --no-cpu-baseline --min-seconds 0.5 --steps 300"
for r in 1 2; do
  for a in 0 1; do
    PIPER_HIP_ATTN4=$a timeout 300 python bench.py $BQ > $O/attn4_${a}_$r.json 2>> $O/err.log
  done
done
for T in 64 256; do for a in 0 2; do PIPER_HIP_ATTN4=$a timeout 300 python bench.py $BQ --ids $T > $O/attn4_T${T}_${a}.json 2>> $O/err.log; done; done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/c2/attn4_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f),"ERR",e); continue
    full=json.load(open("bench_full.json")) if False else None
    r=d.get("roofline") or {}
    print("%-22s ms %8.4f launches %s stages %s" % (os.path.basename(f), d["ms_per_step"], d["config"].get("kernel_launches_per_step"), r.get("stage_ms")))
PY
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $O/pytest_gpu.log
tail -25 $O/pytest_gpu.log
