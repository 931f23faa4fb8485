cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2ac
mkdir -p $O
for B in 32 64; do
  timeout 200 python bench.py --no-cpu-baseline --no-roofline --batch $B --steps 20 > $O/b${B}_default.json 2>> $O/err.log
  PIPER_HIP_COLCHAIN=2 timeout 200 python bench.py --no-cpu-baseline --no-roofline --batch $B --steps 20 > $O/b${B}_chainall.json 2>> $O/err.log
done
timeout 200 python bench.py --no-cpu-baseline --no-roofline --preset high --batch 2 --steps 50 > $O/high_b2_default.json 2>> $O/err.log
PIPER_HIP_GROUP_MRF=0 timeout 200 python bench.py --no-cpu-baseline --no-roofline --preset high --batch 2 --steps 50 > $O/high_b2_nogroup.json 2>> $O/err.log
PIPER_HIP_COLCHAIN=2 timeout 200 python bench.py --no-cpu-baseline --no-roofline --preset high --batch 8 --steps 20 > $O/high_b8_chainall.json 2>> $O/err.log
timeout 200 python bench.py --no-cpu-baseline --no-roofline --preset high --batch 8 --steps 20 > $O/high_b8_default.json 2>> $O/err.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2ac/*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], "ms %.4f"%d["ms_per_step"], "%.1fM"%(d["value"]/1e6))
    except Exception as e: print(f, "ERR", e)
PY
