# Round 4, call 54: utterances beyond ~830 ids (attention score slabs in global memory): parity, then a length sweep
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4y2; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batched.py -m gpu -x -q -k "extreme or long_utterance or forced or golden" 2>&1 | tail -4
BQ="--no-extra --no-cpu-baseline --no-roofline --min-seconds 0.3"
for t in 768 832 896 1024 1536 2048 4096; do
  PIPER_BENCH_FULL=$O/t$t.json timeout 300 python bench.py $BQ --steps 20 --warmup 3 --ids $t > /dev/null 2>> $O/err.log
done
grep -v amdgpu.ids $O/err.log | tail -5
python - <<'PY'
import json,glob,os
rows=[]
for f in glob.glob("gpurun_out/r4y2/t*.json"):
    d=json.load(open(f)); t=int(os.path.basename(f)[1:-5])
    rows.append((t,d["ms_per_step"],d["value"],d["config"]["kernel_launches_per_step"],d["config"]["frames_per_step"]))
for t,ms,v,l,fr in sorted(rows):
    print("T=%-5d frames %5d %9.4f ms/step  %7.3f us/frame  %7.1f M samples/s  %d launches" % (t,fr,ms,ms*1e3/fr,v/1e6,l))
PY
