# Round 4, call 45: batch sweep of the medium voice (128 ids per utterance): ms per step and samples/s per batch size
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s; mkdir -p $O
BQ="--no-extra --no-cpu-baseline --no-roofline --min-seconds 0.3"
for b in 1 2 3 4 5 6 8 12 16 24 32 48 64; do
  PIPER_BENCH_FULL=$O/b$b.json timeout 300 python bench.py $BQ --steps 20 --warmup 5 --batch $b > /dev/null 2>> $O/err.log
done
grep -v amdgpu.ids $O/err.log | tail -3
python - <<'PY'
import json,glob,os
rows=[]
for f in glob.glob("gpurun_out/r4s/b*.json"):
    d=json.load(open(f)); b=int(os.path.basename(f)[1:-5])
    rows.append((b,d["ms_per_step"],d["value"],d["config"]["kernel_launches_per_step"]))
for b,ms,v,l in sorted(rows):
    print("B=%-3d %9.4f ms/step  %8.4f ms/utterance  %7.1f M samples/s  %d launches" % (b,ms,ms/b,v/1e6,l))
PY
