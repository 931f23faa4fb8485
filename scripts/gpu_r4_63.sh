# Round 4, call 63: attno_kernel -- previous build (V chunks through LDS) against V fragments straight from global memory
# with Q / relative tables requested first and both key tiles preloaded: parity cases, B=1 at 128 / 64 / 256 / 384 ids, each
# build twice, alternating, on one box; then the phase stamps of both forms (tuning builds).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4_63
mkdir -p $O
cp piper_amd/libpiper_hip.so /tmp/new.so
timeout 600 python -m pytest tests -m gpu -q -x -k "golden or full_size or intermediate or ragged or reference_test_sentences" 2>&1 | tail -2
BQ="--no-extra --no-cpu-baseline --min-seconds 0.5"
for ids in 128 64 256 384; do
  for r in 1 2; do
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
for f in sorted(glob.glob("gpurun_out/r4_63/*_[12].json")):
    d=json.load(open(f)); r=d.get("roofline") or {}
    ks=r.get("kernels",{})
    a=[(k,v) for k,v in ks.items() if "attno" in k]
    print("%-16s ms %8.4f  %s" % (os.path.basename(f), d["ms_per_step"], " ".join("%s %.2f us" % (k, v["avg_launch_us"]) for k,v in a)))
PY
for w in base new; do for T in 128 64; do echo "== stamps $w $T"; PIPER_STAMPS_LIB=$GRAFT_REPO_ROOT/piper_amd/libab_${w}_st.so timeout 200 python scripts/stamps.py medium $T 2>&1 | grep "attn" | head -2; done; done
