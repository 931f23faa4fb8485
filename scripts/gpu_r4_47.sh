# Round 4, call 47: is the id limit of the 16-column chains (4096 ids per call) still right? medium, 128 ids per utterance
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4u; mkdir -p $O
BQ="--no-extra --no-cpu-baseline --no-roofline --min-seconds 0.3"
for b in 24 32 40 48 64; do for m in 0 1 2; do
  PIPER_HIP_COLCHAIN=$m PIPER_BENCH_FULL=$O/b${b}_m$m.json timeout 300 python bench.py $BQ --steps 10 --warmup 3 --batch $b > /dev/null 2>> $O/err.log
done; done
grep -v amdgpu.ids $O/err.log | tail -3
python - <<'PY'
import json,glob,os
rows=[]
for f in glob.glob("gpurun_out/r4u/b*.json"):
    d=json.load(open(f)); n=os.path.basename(f)[1:-5]; b,m=n.split("_m")
    rows.append((int(b),int(m),d["ms_per_step"],d["value"],d["config"]["kernel_launches_per_step"]))
for b,m,ms,v,l in sorted(rows):
    print("B=%-3d colchain %d %9.4f ms/step  %7.1f M samples/s  %d launches" % (b,m,ms,v/1e6,l))
PY
