# SQ counter pass of the final round-2 build (same counters as profiles/r01_final_pmc_summary.txt)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2x
mkdir -p $O
C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32"
run() { n=$1; shift; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $GRAFT_REPO_ROOT/$O/$n -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --min-seconds 0 "$@" > /dev/null 2>&1); }
run b1 --steps 20 --warmup 2
run b16 --batch 16 --steps 4 --warmup 1
run high_b8 --preset high --batch 8 --steps 2 --warmup 1
{
echo "# round-2 final build. rocprofv3 --kernel-trace --pmc $C -- python bench.py --no-cpu-baseline --no-roofline --min-seconds 0 [--steps 20 | --batch 16 --steps 4 | --preset high --batch 8 --steps 2]"
echo "# wait = wave parked on s_waitcnt/barrier, winst = issue stall (MFMA dependency / pipe), active = issuing; percentages of SQ_WAVE_CYCLES"
echo "# mfma_busy/busy is SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES (summed over shader engines: ~35 per 1 % of f32 MFMA peak); MFMA GFLOP/call = executed matrix FLOPs"
python scripts/pmc_summary.py $O/b1 $O/b16 $O/high_b8
} > $O/r02_pmc_summary.txt 2>&1
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
grep -n "mrf2\|group\|sum_kernel\|conv_post\|#" $O/r02_pmc_summary.txt | cut -c1-210
