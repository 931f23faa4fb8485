cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -8 > gpurun_out/t1.log
for v in old new old new; do
  cp scripts/ab/lib_$v.so piper_amd/libpiper_hip.so
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/li_${v} -- python bench.py --no-cpu-baseline > gpurun_out/li_${v}.json 2> gpurun_out/at.err
  python bench.py --no-cpu-baseline --batch 16 --steps 20 > gpurun_out/li16_${v}.json 2> gpurun_out/at.err
  python bench.py --no-cpu-baseline --preset high --batch 8 --steps 10 > gpurun_out/lih_${v}.json 2> gpurun_out/at.err
done
cp scripts/ab/lib_new.so piper_amd/libpiper_hip.so
find gpurun_out -name "*kernel_trace.csv" -delete
cat gpurun_out/t1.log
