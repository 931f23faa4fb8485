# Round 4, call 53: single-utterance sweep over the text length (medium voice): ms per step, samples/s, launches
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4x2; mkdir -p $O
BQ="--no-extra --no-cpu-baseline --no-roofline --min-seconds 0.3"
for t in 16 32 64 96 128 160 192 256 320 384 512 768 1024; do
  PIPER_BENCH_FULL=$O/t$t.json timeout 300 python bench.py $BQ --steps 50 --warmup 5 --ids $t > /dev/null 2>> $O/err.log
done
grep -v amdgpu.ids $O/err.log | tail -3
python - <<'PY'
import json,glob,os
rows=[]
for f in glob.glob("gpurun_out/r4x2/t*.json"):
    d=json.load(open(f)); t=int(os.path.basename(f)[1:-5])
    rows.append((t,d["ms_per_step"],d["value"],d["config"]["kernel_launches_per_step"],d["config"]["frames_per_step"]))
for t,ms,v,l,fr in sorted(rows):
    print("T=%-5d frames %5d %9.4f ms/step  %7.3f us/frame  %7.1f M samples/s  %d launches" % (t,fr,ms,ms*1e3/fr,v/1e6,l))
PY
