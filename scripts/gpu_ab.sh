# A/B of two builds of the library on ONE box: piper_amd/libab_base.so (the previous commit's build) against the
# tree's piper_amd/libpiper_hip.so. Usage: gpu_ab.sh "<pytest -k expression or empty>" "<bench args>" [kernel name filter]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/ab
mkdir -p $O
KEXPR="$1"; BARGS="$2"; KFILT="$3"
cp piper_amd/libpiper_hip.so /tmp/new.so
if [ -n "$KEXPR" ]; then timeout 900 python -m pytest tests -m gpu -q -x -k "$KEXPR" 2>&1 | tail -4; fi
BQ="--no-extra --no-cpu-baseline --min-seconds 0.5"
for r in 1 2; do
  cp piper_amd/libab_base.so piper_amd/libpiper_hip.so
  timeout 300 python bench.py $BQ $BARGS > $O/base_$r.json 2>> $O/err.log
  cp /tmp/new.so piper_amd/libpiper_hip.so
  timeout 300 python bench.py $BQ $BARGS > $O/new_$r.json 2>> $O/err.log
done
grep -v amdgpu.ids $O/err.log | tail -3
KFILT="$KFILT" python - <<'PY'
import json,glob,os
filt=os.environ.get("KFILT","")
for f in sorted(glob.glob("gpurun_out/ab/*_[12].json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f),"ERR",e); continue
    r=d.get("roofline") or {}
    print("%-12s ms %8.4f val %7.2fM launches %s stages %s hifiTF %.1f" % (os.path.basename(f), d["ms_per_step"], d["value"]/1e6, d["config"].get("kernel_launches_per_step"),
          {k[:4]:round(v,3) for k,v in r.get("stage_ms",{}).items()}, r.get("stage_tflops",{}).get("hifigan",0)))
    for k,v in r.get("kernels",{}).items():
        if filt and any(x in k for x in filt.split(",")):
            print("     %-46s %5.1f x %7.2f us = %7.1f us" % (k, v["launches_per_step"], v["avg_launch_us"], v["ms_per_step"]*1e3))
PY
