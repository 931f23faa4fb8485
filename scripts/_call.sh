cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r6c1; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -4
# A/B: ids read from pinned host memory by embed_kernel (1) against one H2D copy in front of the graph (0); headline only
for r in 1 2; do for z in 0 1; do
  PIPER_HIP_IDS_ZC=$z PIPER_BENCH_FULL=$O/full_zc${z}_$r.json timeout 300 python bench.py --no-extra --no-cpu-baseline --min-seconds 0.5 > $O/zc${z}_$r.json 2>> $O/err.log
done; done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r6c1/zc*_[12].json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f),"ERR",e); continue
    print("%-12s ms %8.4f resident %s devonly %s val %7.2fM launches %s api %s" % (os.path.basename(f), d["ms_per_step"], d.get("device_resident_ms"), d.get("device_pipeline_only_ms_per_step"), d["value"]/1e6, d["config"].get("kernel_launches_per_step"), (d.get("api_inclusive") or {}).get("ms_per_call")))
PY
timeout 1200 python bench.py > $O/bench_default.stdout 2> $O/bench_default.err
cp bench_full.json $O/bench_default_full.json
grep -v amdgpu.ids $O/err.log | tail -3; tail -2 $O/bench_default.err
echo "last line bytes: $(tail -n 1 $O/bench_default.stdout | wc -c)"; tail -n 1 $O/bench_default.stdout
