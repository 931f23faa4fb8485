# Round 4, call 29: column tiles walked per workgroup by the tiled kernel (PIPER_HIP_TPB) at ONE utterance, where only the two
# late up-convs take that kernel (424 / 418 workgroups of 64 x 64 on 256 CUs): 1 (default) against 2 / 3 / 4, each twice.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s; mkdir -p $O
BQ="--no-extra --no-cpu-baseline --min-seconds 0.4 --steps 300 --warmup 10"
for r in a b; do for t in 0 2 3 4; do
  PIPER_HIP_TPB=$t PIPER_BENCH_FULL=$O/tpb${t}_$r.json timeout 300 python bench.py $BQ > /dev/null 2>> $O/err.log
done; done
grep -v amdgpu.ids $O/err.log | tail -3
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r4s/*.json")):
    d=json.load(open(f)); r=d.get("roofline") or {}
    row=["%s %.1f" % (k.replace("conv_","").replace("_kernel","")[:24], v["avg_launch_us"]) for k,v in r.get("kernels",{}).items() if "mfma" in k]
    print("%-10s ms %8.4f hifi %.4f %s" % (os.path.basename(f)[:-5], d["ms_per_step"], r.get("stage_ms",{}).get("hifigan",0), " | ".join(row)))
PY
