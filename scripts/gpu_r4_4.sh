# Round 4, call 4: attno_kernel (attention + conv_o + LayerNorm in one launch) on the hardware: parity (the forced-variant
# matrix), then on / off per batch size on ONE box; and why ncclCommInitAll failed inside pe_group_create in call 3.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4d
mkdir -p $O
( time timeout 1800 python -m pytest tests/test_gpu_batched.py -m gpu -q -k "forced_kernel or intermediate or reference_test_sentences or ragged" 2>&1 | tail -12 ) > $O/pytest_a.log 2>&1
cat $O/pytest_a.log
BQ="--no-extra --no-cpu-baseline --min-seconds 0.4"
run() { PIPER_BENCH_FULL=$O/$1.json env $2 timeout 300 python bench.py $BQ $3 > $O/$1.line 2>> $O/err.log; }
for b in 1 2 4 8; do
  for r in a b; do for u in 0 1; do run b${b}_ao${u}_$r PIPER_HIP_ATTNO=$u "--batch $b --steps 300 --warmup 10"; done; done
done
run t64_ao0 PIPER_HIP_ATTNO=0 "--ids 64 --steps 300 --warmup 10"
run t64_ao1 PIPER_HIP_ATTNO=1 "--ids 64 --steps 300 --warmup 10"
run t256_ao0 PIPER_HIP_ATTNO=0 "--ids 256 --steps 200 --warmup 10"
run t256_ao1 PIPER_HIP_ATTNO=1 "--ids 256 --steps 200 --warmup 10"
grep -v amdgpu.ids $O/err.log | tail -3
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r4d/*_ao*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(os.path.basename(f),"ERR",e); continue
    r=d.get("roofline") or {}
    print("%-14s ms %8.4f launches %s text %.4f ms" % (os.path.basename(f)[:-5], d["ms_per_step"], d["config"].get("kernel_launches_per_step"), r.get("stage_ms",{}).get("text_encoder",0)))
    for k,v in r.get("kernels",{}).items():
        if k.startswith("attn") or k.startswith("colchain4") or k.startswith("lngemm4") or k.startswith("ffn"): print("     %-24s %5.1f x %8.2f us" % (k, v["launches_per_step"], v["avg_launch_us"]))
PY
# ---- RCCL inside pe_group_create: the library's own diagnostics
NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,ENV PIPER_HIP_GROUP_BCAST=rccl timeout 300 python - > $O/rccl_debug.log 2>&1 <<'PY'
import sys
sys.path.insert(0, ".")
from piper_amd import weights as W
from piper_amd.group import EngineGroup
cfg = W.preset("tiny")
blob = W.pack_blob(cfg, W.synthetic_weights(cfg, 1234))
g = EngineGroup(blob, [0, 0])
print("broadcast_path:", g.broadcast_path)
g.close()
PY
grep -v amdgpu.ids $O/rccl_debug.log | tail -40
