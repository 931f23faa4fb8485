# final round-1 collection: tests, smoke, bench lines, rocprof kernel stats of the same commands, PMC traffic passes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
export TMPDIR=/tmp
O=gpurun_out/final
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_b1 -- python bench.py --no-cpu-baseline --steps 5 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_b1 -- python bench.py --no-cpu-baseline --steps 5 --warmup 1 > /dev/null 2>&1
python scripts/pmc_traffic.py medium/b1/t128 $O/pmc_fetch_b1 $O/pmc_write_b1 profiles/r01_pmc_traffic.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, WRITE_SIZE) -- python bench.py --no-cpu-baseline --steps 5 --warmup 1" > $O/traffic.log 2>&1
cp profiles/r01_pmc_traffic.json $O/r01_pmc_traffic.json
python bench.py > $O/bench_b1.json 2> $O/bench_b1.err
python bench.py --no-cpu-baseline --batch 16 --steps 20 > $O/bench_b16.json 2>> $O/err.log
python bench.py --no-cpu-baseline --batch 64 --steps 10 > $O/bench_b64.json 2>> $O/err.log
python bench.py --no-cpu-baseline --preset high --batch 8 --steps 10 > $O/bench_high_b8.json 2>> $O/err.log
python bench.py --no-cpu-baseline --preset high > $O/bench_high_b1.json 2>> $O/err.log
python bench.py --no-cpu-baseline --preset high --batch 64 --steps 3 --warmup 1 > $O/bench_high_b64.json 2>> $O/err.log
python bench.py --stream-latency > $O/stream_medium.json 2>> $O/err.log
python bench.py --stream-latency --preset high > $O/stream_high.json 2>> $O/err.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_b1 -- python bench.py --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_b16 -- python bench.py --no-cpu-baseline --batch 16 --steps 20 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_high -- python bench.py --no-cpu-baseline --preset high --batch 8 --steps 10 > /dev/null 2>&1
find $O -name "*kernel_trace.csv" -delete
PIPER_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_2rank_gloo_1gpu.json 2>> $O/err.log
cat $O/pytest_gpu.log $O/smoke.log
