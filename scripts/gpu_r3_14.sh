# round 3, call 14: the XCD dispatch pattern of the box (pe_xcc_pattern) and the tile order of the 4-column kernels with /
# without it (PIPER_HIP_XCD=0: tiles in workgroup order)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3n
mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or full_size" 2>&1 | tail -2
BQ="--no-extra --no-cpu-baseline --min-seconds 0.4 --steps 300"
timeout 100 python bench.py $BQ > $O/b1_probe.json 2>> $O/err.log
PIPER_HIP_XCD=0 timeout 100 python bench.py $BQ > $O/b1_xcd0.json 2>> $O/err.log
timeout 100 python bench.py $BQ > $O/b1_probe2.json 2>> $O/err.log
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r3n/b1_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]["kernels"]
    print(os.path.basename(f), "ms %.4f" % d["ms_per_step"], d.get("xcd_dispatch",{}).get("round_robin_period"), {k[:12]:round(v["avg_launch_us"],2) for k,v in r.items() if any(x in k for x in ("attn","lngemm4","colchain4","dds_layer4","ffn"))})
print(json.loads(open("gpurun_out/r3n/b1_probe.json").read().strip().splitlines()[-1])["xcd_dispatch"])
PY
