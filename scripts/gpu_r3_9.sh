# round 3, call 9: N single-utterance PROCESSES sharing the one GPU (bench.py's multi-process path with gloo, every rank on
# device 0) against N engines in one process (call 8): how much of a latency-bound B=1 pipeline's idle chip independent
# requests can use
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3i
mkdir -p $O
for n in 2 4 8; do
  PIPER_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus $n --config 2 --no-extra --no-cpu-baseline --no-roofline --steps 400 --min-seconds 0.5 > $O/ranks$n.json 2>> $O/err.log
done
grep -v amdgpu.ids $O/err.log | tail -5
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r3i/ranks*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f),"ERR",e); continue
    print(os.path.basename(f), "n", d["n_gpus"], "ms/step %.4f" % d["ms_per_step"], "aggregate %.1fM" % (d["value"]/1e6), [round(x/1e6,1) for x in d.get("per_rank_samples_per_s",[])])
PY
