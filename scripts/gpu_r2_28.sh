cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2ab
mkdir -p $O
for B in 2 4; do
  PIPER_HIP_GROUP_MRF=2 timeout 200 python bench.py --no-cpu-baseline --no-roofline --batch $B --steps 100 > $O/b${B}_nosum.json 2>> $O/err.log
  PIPER_HIP_GROUP_MRF=0 timeout 200 python bench.py --no-cpu-baseline --no-roofline --batch $B --steps 100 > $O/b${B}_nogroup.json 2>> $O/err.log
  timeout 200 python bench.py --no-cpu-baseline --no-roofline --batch $B --steps 100 > $O/b${B}_default.json 2>> $O/err.log
done
for B in 12 16; do
  timeout 200 python bench.py --no-cpu-baseline --no-roofline --batch $B --steps 50 > $O/b${B}_default.json 2>> $O/err.log
  PIPER_HIP_COLCHAIN=2 timeout 200 python bench.py --no-cpu-baseline --no-roofline --batch $B --steps 50 > $O/b${B}_chainall.json 2>> $O/err.log
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2ab/b*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], "ms %.4f"%d["ms_per_step"], "%.1fM"%(d["value"]/1e6))
    except Exception as e: print(f, "ERR", e)
PY
