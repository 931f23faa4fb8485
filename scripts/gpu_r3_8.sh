# round 3, call 8: what is left at batch 1 outside the kernels -- kernel-argument placement (HIP_FORCE_DEV_KERNARG),
# which convs take the split-K form (PIPER_HIP_SPLITK_MAX), and N independent single-utterance engines sharing the GPU
# (pe_group_*: the "per-GPU independent streams" of north_star at the B=1 latency point)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3h
mkdir -p $O
BQ="--no-extra --no-cpu-baseline --no-roofline --min-seconds 0.5"
run() { # name, env..., -- bench args
  n=$1; shift
  envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs timeout 300 python bench.py $BQ "$@" > $O/$n.json 2>> $O/err.log
}
run b1_a -- --steps 500
run b1_kernarg0 HIP_FORCE_DEV_KERNARG=0 -- --steps 500
run b1_kernarg1 HIP_FORCE_DEV_KERNARG=1 -- --steps 500
run b1_splitk450 PIPER_HIP_SPLITK_MAX=450 -- --steps 500
run b1_splitk64 PIPER_HIP_SPLITK_MAX=64 -- --steps 500
run b1_b -- --steps 500
timeout 600 python - > $O/concurrent.json 2>> $O/err.log <<'PY'
import json, bench
from piper_amd import weights as W
class Ctx: dev_index = 0
print(json.dumps(bench.concurrent_streams(Ctx(), W.preset("medium"), counts=(1, 2, 3, 4, 6, 8))))
PY
grep -v amdgpu.ids $O/err.log | tail -5
python - <<'PY'
import json,glob,os
O="gpurun_out/r3h/"
for f in sorted(glob.glob(O+"b1_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f),"ERR",e); continue
    print("%-20s ms %8.4f val %7.2fM launches %s" % (os.path.basename(f), d["ms_per_step"], d["value"]/1e6, d["config"].get("kernel_launches_per_step")))
try:
    d=json.load(open(O+"concurrent.json"))
    for n,e in d["by_engines"].items(): print("engines", n, "%.1fM samples/s" % (e["value"]/1e6), "p50 ms/call %.3f" % e["ms_per_call_p50"])
except Exception as e: print("concurrent ERR", e)
PY
