cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -8 > gpurun_out/t1.log
timeout 1200 python scripts/stress_parity.py 15 > gpurun_out/stress.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/an_b1 -- python bench.py --no-cpu-baseline > gpurun_out/an_b1.json 2> gpurun_out/at.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/an_b16 -- python bench.py --no-cpu-baseline --batch 16 --steps 20 > gpurun_out/an_b16.json 2> gpurun_out/at.err
find gpurun_out -name "*kernel_trace.csv" -delete
cat gpurun_out/t1.log; tail -2 gpurun_out/stress.log
