cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in 0 1 0 1; do
  PIPER_HIP_WIDE_SPLITK=$v rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ws_${v} -- python bench.py --no-cpu-baseline > gpurun_out/ws_${v}.json 2> gpurun_out/at.err
done
find gpurun_out -name "*kernel_trace.csv" -delete
