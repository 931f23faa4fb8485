# round 3, call 13: batch-size sweep of the fused FFN (PIPER_HIP_FFN=0 / 1) and of the 4-column chains with it, for the
# column limits of the default policy (PIPER_HIP_COL4_MAXC raised so that the knob alone decides)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3m
mkdir -p $O
BQ="--no-extra --no-cpu-baseline --no-roofline --min-seconds 0.4"
for b in 1 2 4 8 12 16; do
  for c in "0 0" "2 0" "2 1"; do
    set -- $c
    PIPER_HIP_COL4=$1 PIPER_HIP_FFN=$2 PIPER_HIP_COLCHAIN=2 timeout 300 python bench.py $BQ --batch $b --steps 100 --warmup 5 > $O/b${b}_c$1_f$2.json 2>> $O/err.log
  done
done
python - <<'PY'
import json
for b in (1,2,4,8,12,16):
    r=[]
    for c in ("c0_f0","c2_f0","c2_f1"):
        try: r.append(json.loads(open(f"gpurun_out/r3m/b{b}_{c}.json").read().strip().splitlines()[-1])["ms_per_step"])
        except Exception as e: r.append(float("nan"))
    print("B=%-3d 16-col chains %.4f ms   4-col chains %.4f (%+.1f %%)   + fused FFN %.4f (%+.1f %%)" % (b, r[0], r[1], (r[1]/r[0]-1)*100, r[2], (r[2]/r[0]-1)*100))
PY
