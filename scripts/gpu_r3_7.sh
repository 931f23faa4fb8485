# round 3, call 7: mrf_kernel with 4 output units per wave on 32 channels (N = 512: one utterance's last stage in one round)
# and the generator tail (conv_post + tanh + peak) fused into the last stage's kernel -- parity tests, A/B per batch size
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3g
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batched.py -m gpu -q -x -k "fused_mrf or generator_tail" 2>&1 | tail -12 > $O/pytest.log
BQ="--no-extra --no-cpu-baseline --min-seconds 0.5"
run() { # name, env..., -- bench args
  n=$1; shift
  envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs timeout 300 python bench.py $BQ "$@" > $O/$n.json 2>> $O/err.log
}
for t in 0 1; do
  run b1_ou2_tail$t PIPER_HIP_MRF_OU=2 PIPER_HIP_MRF_TAIL=$t -- --steps 300
  run b1_ou4_tail$t PIPER_HIP_MRF_OU=4 PIPER_HIP_MRF_TAIL=$t -- --steps 300
  run b16_ou3_tail$t PIPER_HIP_MRF_OU=3 PIPER_HIP_MRF_TAIL=$t -- --batch 16 --steps 30 --warmup 3
  run b16_ou4_tail$t PIPER_HIP_MRF_OU=4 PIPER_HIP_MRF_TAIL=$t -- --batch 16 --steps 30 --warmup 3
  run b64_ou3_tail$t PIPER_HIP_MRF_OU=3 PIPER_HIP_MRF_TAIL=$t -- --config 4 --steps 10 --warmup 3
  run b64_ou4_tail$t PIPER_HIP_MRF_OU=4 PIPER_HIP_MRF_TAIL=$t -- --config 4 --steps 10 --warmup 3
done
run b1_default -- --steps 300
run b4_default -- --batch 4 --steps 100
run b16_default -- --batch 16 --steps 30 --warmup 3
run b64_default -- --config 4 --steps 10 --warmup 3
run h1_default -- --preset high --steps 50
run h1_tail0 PIPER_HIP_MRF_TAIL=0 -- --preset high --steps 50
cat $O/pytest.log; grep -v amdgpu.ids $O/err.log | tail -5
python - <<'PY'
import json,glob,os
O="gpurun_out/r3g/"
for f in sorted(glob.glob(O+"*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f),"ERR",e); continue
    r=d.get("roofline") or {}
    print("%-20s ms %8.3f val %7.1fM launches %s stages %s hifiTF %.1f" % (os.path.basename(f), d["ms_per_step"], d["value"]/1e6, d["config"].get("kernel_launches_per_step"),
          {k[:4]:round(v,3) for k,v in r.get("stage_ms",{}).items()}, r.get("stage_tflops",{}).get("hifigan",0)))
    for k,v in r.get("kernels",{}).items():
        if k.startswith("mrf") or k.startswith("conv_post"):
            print("      %-30s n %5.1f us %8.2f TF %.1f" % (k, v["launches_per_step"], v["avg_launch_us"], v["tflops"]))
PY
