# Round 4, call 46: speculative sizing of stage B beyond 4 utterances (PIPER_HIP_SPEC_MAXB 4 | 16): medium, 128 ids per utterance
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4t; mkdir -p $O
BQ="--no-extra --no-cpu-baseline --no-roofline --min-seconds 0.3"
for b in 5 6 8 12 16; do for m in 4 16; do
  PIPER_HIP_SPEC_MAXB=$m PIPER_BENCH_FULL=$O/b${b}_m$m.json timeout 300 python bench.py $BQ --steps 20 --warmup 5 --batch $b > /dev/null 2>> $O/err.log
done; done
grep -v amdgpu.ids $O/err.log | tail -3
python - <<'PY'
import json,glob,os
rows=[]
for f in glob.glob("gpurun_out/r4t/b*.json"):
    d=json.load(open(f)); n=os.path.basename(f)[1:-5]; b,m=n.split("_m")
    rows.append((int(b),int(m),d["ms_per_step"],d["value"],d["config"]["kernel_launches_per_step"],d.get("speculation")))
for b,m,ms,v,l,sp in sorted(rows):
    print("B=%-3d maxb %-2d %9.4f ms/step  %7.1f M samples/s  %d launches  %s" % (b,m,ms,v/1e6,l,sp))
PY
