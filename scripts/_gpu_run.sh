cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -8 > gpurun_out/t1.log
for v in 0 1 2 0 1 2; do
  PIPER_HIP_SPLITK16=$v rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/s16_${v} -- python bench.py --no-cpu-baseline > gpurun_out/s16_${v}.json 2> gpurun_out/at.err
  PIPER_HIP_SPLITK16=$v python bench.py --no-cpu-baseline --batch 4 > gpurun_out/s16b4_${v}.json 2> gpurun_out/at.err
done
find gpurun_out -name "*kernel_trace.csv" -delete
cat gpurun_out/t1.log
