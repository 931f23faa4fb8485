cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r6c14; mkdir -p $O
for v in 1 2; do PIPER_HIP_GATE4_XCD=$v timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "medium or golden" 2>&1 | tail -1; done
for r in 1 2 3; do for v in 0 1 2; do
  PIPER_HIP_GATE4_XCD=$v PIPER_BENCH_FULL=$O/full_g${v}_$r.json timeout 300 python bench.py --no-extra --no-cpu-baseline --min-seconds 0.5 > /dev/null 2>> $O/err.log
done; done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r6c14/full_g*_*.json")):
    d=json.load(open(f)); k=d["roofline"]["kernels"].get("gate4_kernel",{})
    print("%-18s ms %8.4f resident %.4f gate4 %5.2f us x %s" % (os.path.basename(f), d["ms_per_step"], d.get("device_resident_ms") or 0, k.get("avg_launch_us",0), k.get("launches_per_step")))
PY
BA="--no-extra --no-cpu-baseline --no-roofline --steps 50 --min-seconds 0"
prof() { d=$1; shift; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/$d "$@" > /dev/null 2>&1); }
for v in 0 1 2; do
  export PIPER_HIP_GATE4_XCD=$v
  prof fetch_$v --pmc FETCH_SIZE -- python $GRAFT_REPO_ROOT/bench.py $BA
  prof write_$v --pmc WRITE_SIZE -- python $GRAFT_REPO_ROOT/bench.py $BA
  prof st_$v --stats -- python $GRAFT_REPO_ROOT/bench.py $BA
  python scripts/pmc_traffic.py medium/b1/t128 $O/fetch_$v $O/write_$v $O/traffic_$v.json "gate4 xcd=$v" > /dev/null 2>&1
  python -c "
import json,csv,glob
d=json.load(open('$O/traffic_$v.json'))['medium/b1/t128']['kernels']['gate4_kernel']
f=glob.glob('$O/st_$v/**/*kernel_stats.csv',recursive=True)[0]
us=[float(r['AverageNs'])/1e3 for r in csv.DictReader(open(f)) if 'gate4' in r['Name']][0]
print('GATE4_XCD=$v: hbm bytes per launch', d['hbm_bytes_per_launch'], ' rocprof avg us %.2f' % us)"
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -delete
