# Round 4, call 65: colchain4_kernel / lngemm4_kernel with their weight fragments requested BEHIND the input loads of the
# phase (libab_base.so = the build of call 64): parity cases, B=1 at 128 / 64 / 256 ids, each build three times, alternating.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4_65
mkdir -p $O
cp piper_amd/libpiper_hip.so /tmp/new.so
timeout 600 python -m pytest tests -m gpu -q -x -k "golden or full_size or intermediate or ragged or reference_test_sentences or forced_kernel_variants" 2>&1 | tail -2
BQ="--no-extra --no-cpu-baseline --min-seconds 0.5"
for ids in 128 64 256; do
  for r in 1 2 3; do
    for w in base new; do
      case $w in base) cp piper_amd/libab_base.so piper_amd/libpiper_hip.so;; new) cp /tmp/new.so piper_amd/libpiper_hip.so;; esac
      PIPER_BENCH_FULL=$O/${w}_${ids}_$r.json timeout 300 python bench.py $BQ --ids $ids > /dev/null 2>> $O/err.log
    done
  done
done
cp /tmp/new.so piper_amd/libpiper_hip.so
grep -v amdgpu.ids $O/err.log | tail -3
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r4_65/*_[123].json")):
    d=json.load(open(f)); r=d.get("roofline") or {}
    ks=r.get("kernels",{})
    a=[(k,v) for k,v in ks.items() if "colchain4" in k or "lngemm4" in k]
    print("%-16s ms %8.4f  %s" % (os.path.basename(f), d["ms_per_step"], " ".join("%s %.2f" % (k.replace("_kernel",""), v["avg_launch_us"]) for k,v in a)))
PY
