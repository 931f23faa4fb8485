# round 3, call 12: batch-size sweep of the 4-column small-call kernels (PIPER_HIP_COL4=0 off / 2 always) for the column
# limit of the default policy
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3l
mkdir -p $O
BQ="--no-extra --no-cpu-baseline --no-roofline --min-seconds 0.4"
for b in 1 2 4 8 16 32; do
  for c in 0 2; do
    PIPER_HIP_COL4=$c PIPER_HIP_COLCHAIN=2 timeout 300 python bench.py $BQ --batch $b --steps 100 --warmup 5 > $O/b${b}_c$c.json 2>> $O/err.log
  done
done
python - <<'PY'
import json,glob,os
for b in (1,2,4,8,16,32):
    r=[]
    for c in (0,2):
        try: r.append(json.loads(open(f"gpurun_out/r3l/b{b}_c{c}.json").read().strip().splitlines()[-1])["ms_per_step"])
        except Exception as e: r.append(float("nan"))
    print("B=%-3d col4 off %.4f ms   always %.4f ms   (%+.1f %%)" % (b, r[0], r[1], (r[1]/r[0]-1)*100))
PY
