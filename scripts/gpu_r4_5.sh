# Round 4, call 5: (a) the tests that failed in calls 3 / 4 for reasons on the test side; (b) workspace memory type at B=1
# (PIPER_HIP_WS_MEM = 0 default / 1 uncached / 2 fine-grained): does a dependent launch get cheaper when the L2s hold
# nothing dirty at the kernel boundary?; (c) the driver's command again (changing-input leg after the wider warm-up).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4e
mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_batched.py -m gpu -q -k "forced_kernel or engine_group or every_profiled or graph_cache" 2>&1 | tail -12 ) > $O/pytest_a.log 2>&1
cat $O/pytest_a.log
BQ="--no-extra --no-cpu-baseline --min-seconds 0.4"
run() { PIPER_BENCH_FULL=$O/$1.json env $2 timeout 300 python bench.py $BQ $3 > $O/$1.line 2>> $O/err.log; }
for r in a b; do for m in 0 1 2; do run b1_mem${m}_$r PIPER_HIP_WS_MEM=$m "--steps 300 --warmup 10"; done; done
for m in 0 1; do run b4_mem${m} PIPER_HIP_WS_MEM=$m "--batch 4 --steps 200 --warmup 10"; done
grep -v amdgpu.ids $O/err.log | tail -3
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r4e/b*_mem*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(os.path.basename(f),"ERR",e); continue
    r=d.get("roofline") or {}
    print("%-12s ms %8.4f stages %s" % (os.path.basename(f)[:-5], d["ms_per_step"], {k[:4]:round(v,4) for k,v in r.get("stage_ms",{}).items()}))
    ks=r.get("kernels",{})
    print("     ", {k.split('<')[0][:14]: round(v["avg_launch_us"],2) for k,v in list(ks.items())[:12]})
PY
( time timeout 900 python bench.py > $O/default.stdout 2> $O/default.stderr ) 2>&1 | tail -3
cp bench_full.json $O/default_full.json
tail -n 1 $O/default.stdout
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4e/default_full.json"))
for e in d.get("extra_configs",[]):
    if "changing" in e.get("leg",""): print(json.dumps({k:v for k,v in e.items() if k!="config"}))
PY
