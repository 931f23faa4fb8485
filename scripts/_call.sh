cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/c16; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -x -k "medium_t128 or intermediate or ragged or sentences or forced and FFN" 2>&1 | tail -3
BQ="--no-extra --no-cpu-baseline --min-seconds 0.3"
cp piper_amd/libpiper_hip.so /tmp/new.so
for r in 1 2 3; do
  cp piper_amd/libab_base.so piper_amd/libpiper_hip.so
  PIPER_BENCH_FULL=$O/base_$r.json timeout 300 python bench.py $BQ --steps 200 > /dev/null 2>> $O/err.log
  cp /tmp/new.so piper_amd/libpiper_hip.so
  PIPER_BENCH_FULL=$O/new_$r.json timeout 300 python bench.py $BQ --steps 200 > /dev/null 2>> $O/err.log
done
python scripts/_show_kernels.py lngemm4,ffn $O/base_*.json $O/new_*.json
grep -v amdgpu.ids $O/err.log | tail -5
