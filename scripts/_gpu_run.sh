cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32"
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d gpurun_out/pmc_sq_b16 -- python bench.py --no-cpu-baseline --batch 16 --steps 3 --warmup 1 > gpurun_out/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d gpurun_out/pmc_sq_b1 -- python bench.py --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch_b1 -- python bench.py --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write_b1 -- python bench.py --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/pmc4.log 2>&1
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d gpurun_out/pmc_sq_h8 -- python bench.py --no-cpu-baseline --preset high --batch 8 --steps 2 --warmup 1 > gpurun_out/pmc5.log 2>&1
find gpurun_out -name "*kernel_trace.csv" -delete
du -sh gpurun_out/pmc_*
