# Round 4, call 2: the whole -m gpu suite on the build with the reworked graph cache (shape buckets, LRU, pe_warmup,
# pinned input block), then the driver's command (compact line; the changing-input leg after one pe_warmup).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4b
mkdir -p $O
python -c "import os; print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count()); print(open('/sys/fs/cgroup/cpu.max').read() if os.path.exists('/sys/fs/cgroup/cpu.max') else 'no cpu.max')" 
( time timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 ) > $O/pytest_gpu.log 2>&1
cat $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout 900 python bench.py > $O/default.stdout 2> $O/default.stderr ) 2>&1 | tail -3
cp bench_full.json $O/default_full.json
tail -c 400 $O/default.stderr
echo "last line bytes: $(tail -n 1 $O/default.stdout | wc -c)"
tail -n 1 $O/default.stdout
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4b/default_full.json"))
print("cpu_baseline", json.dumps(d.get("cpu_baseline"))[:1500])
for e in d.get("extra_configs",[]):
    if "changing" in e.get("leg",""): print(json.dumps({k:v for k,v in e.items() if k!="config"}))
print("api", d["api_inclusive"])
PY
