# round-6 collection (ONE build, ONE box): smoke, the driver's command (compact line + bench_full.json), rocprofv3 kernel
# stats of the headline / configs[3]-share / configs[2] / high-voice-B=1 commands in f32 and of the two throughput configs
# in matrix mode f16x3, PMC traffic passes (FETCH_SIZE and WRITE_SIZE in separate runs), SQ counter passes for the same
# workloads, the randomised parity sweep (f32, and the same sweep in mode f16x3).
# Counters are collected with --kernel-trace only (no other trace domains).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 1500 python bench.py > $O/bench_default.stdout 2> $O/bench_default.err
cp bench_full.json $O/bench_default_full.json
BA="--no-extra --no-cpu-baseline --no-roofline"
W_b1="--steps 50"
W_b64="--config 4 --steps 4 --warmup 2 --min-seconds 0"
W_high="--config 3 --steps 2 --warmup 1 --min-seconds 0"
W_hb1="--preset high --steps 30 --min-seconds 0"
W_b64h="--config 4 --steps 4 --warmup 2 --min-seconds 0 --matrix f16x3"
W_highh="--config 3 --steps 2 --warmup 1 --min-seconds 0 --matrix f16x3"
W_b1h="--steps 50 --matrix f16x3"
# per-workload full result objects (algorithmic flops / bytes per kernel from the engine's level-2 profile rows)
FA="--no-extra --no-cpu-baseline"
PIPER_BENCH_FULL=$O/full_b1.json timeout 300 python bench.py $FA --steps 100 > /dev/null 2>> $O/err.log
PIPER_BENCH_FULL=$O/full_b64.json timeout 300 python bench.py $FA --config 4 --steps 5 --warmup 2 > /dev/null 2>> $O/err.log
PIPER_BENCH_FULL=$O/full_high_b64.json timeout 300 python bench.py $FA --config 3 --steps 3 --warmup 1 > /dev/null 2>> $O/err.log
PIPER_BENCH_FULL=$O/full_high_b1.json timeout 300 python bench.py $FA --preset high --steps 30 > /dev/null 2>> $O/err.log
PIPER_BENCH_FULL=$O/full_b64_f16x3.json timeout 300 python bench.py $FA --config 4 --steps 5 --warmup 2 --matrix f16x3 > /dev/null 2>> $O/err.log
PIPER_BENCH_FULL=$O/full_high_b64_f16x3.json timeout 300 python bench.py $FA --config 3 --steps 3 --warmup 1 --matrix f16x3 > /dev/null 2>> $O/err.log
PIPER_BENCH_FULL=$O/full_b1_f16x3.json timeout 300 python bench.py $FA --steps 100 --matrix f16x3 > /dev/null 2>> $O/err.log
prof() {  # dir, rocprofv3 args..., -- bench args
  d=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/$d "$@" > /dev/null 2>&1)
}
B="python $GRAFT_REPO_ROOT/bench.py $BA"
prof st_b1 --stats -- $B $W_b1
prof st_b64 --stats -- $B $W_b64
prof st_high_b64 --stats -- $B $W_high
prof st_high_b1 --stats -- $B $W_hb1
prof st_b64_f16x3 --stats -- $B $W_b64h
prof st_high_b64_f16x3 --stats -- $B $W_highh
prof st_b1_f16x3 --stats -- $B $W_b1h
for n in b1 b64 high_b64 b64_f16x3; do
  case $n in b1) W="$W_b1 --min-seconds 0";; b64) W="$W_b64";; high_b64) W="$W_high";; *) W="$W_b64h";; esac
  prof pmc_fetch_$n --pmc FETCH_SIZE -- $B $W
  prof pmc_write_$n --pmc WRITE_SIZE -- $B $W
  prof pmc_sq_$n --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 -- $B $W
done
CMD="rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, WRITE_SIZE) -- python bench.py $BA"
python scripts/pmc_traffic.py medium/b1/t128 $O/pmc_fetch_b1 $O/pmc_write_b1 $O/r06_pmc_traffic.json "$CMD $W_b1" $O/full_b1.json > $O/traffic.log 2>&1
python scripts/pmc_traffic.py medium/b64/t128 $O/pmc_fetch_b64 $O/pmc_write_b64 $O/r06_pmc_traffic.json "$CMD $W_b64" $O/full_b64.json >> $O/traffic.log 2>&1
python scripts/pmc_traffic.py high/b64/t128 $O/pmc_fetch_high_b64 $O/pmc_write_high_b64 $O/r06_pmc_traffic.json "$CMD $W_high" $O/full_high_b64.json >> $O/traffic.log 2>&1
python scripts/pmc_traffic.py medium-f16x3/b64/t128 $O/pmc_fetch_b64_f16x3 $O/pmc_write_b64_f16x3 $O/r06_pmc_traffic.json "$CMD $W_b64h" $O/full_b64_f16x3.json >> $O/traffic.log 2>&1
{
  echo "# round-6 build. rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 -- python bench.py $BA [$W_b1 | $W_b64 | $W_high | $W_b64h]"
  echo "# wait = wave parked on s_waitcnt / barrier, winst = issue stall (MFMA dependency / pipe), active = issuing: percentages of SQ_WAVE_CYCLES"
  echo "# mfma_busy/busy = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES (summed over shader engines); MFMA GFLOP/call = executed matrix FLOPs (SQ_INSTS_VALU_MFMA_MOPS_F32 x 512: f32 matrix ops only -- the split kernels' 16-bit ops are not in this counter)"
  python scripts/pmc_summary.py $O/pmc_sq_b1 $O/pmc_sq_b64 $O/pmc_sq_high_b64 $O/pmc_sq_b64_f16x3
} > $O/r06_pmc_summary.txt 2>> $O/err.log
for n in b1 b64 high_b64 high_b1 b64_f16x3 high_b64_f16x3 b1_f16x3; do f=$(find $O/st_$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r06_${n}_kernel_stats.csv; done
{ echo; python scripts/pmc_util.py $O/r06_pmc_summary.txt pmc_sq_b1=$O/r06_b1_kernel_stats.csv pmc_sq_b64=$O/r06_b64_kernel_stats.csv pmc_sq_high_b64=$O/r06_high_b64_kernel_stats.csv; } >> $O/r06_pmc_summary.txt 2>> $O/err.log
# randomised parity sweeps against the oracle (random lengths, ragged batches, scales, speakers; four voices): f32, then f16x3
timeout 600 python scripts/stress_parity.py 24 > $O/r06_stress_parity.log 2>&1; tail -1 $O/r06_stress_parity.log
PIPER_HIP_MATRIX=f16x3 timeout 600 python scripts/stress_parity.py 24 7 > $O/r06_stress_parity_f16x3.log 2>&1; tail -1 $O/r06_stress_parity_f16x3.log
PIPER_HIP_MATRIX=bf16x6 timeout 600 python scripts/stress_parity.py 12 9 > $O/r06_stress_parity_bf16x6.log 2>&1; tail -1 $O/r06_stress_parity_bf16x6.log
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -delete
cat $O/smoke.log; tail -3 $O/traffic.log; grep -v amdgpu.ids $O/err.log | tail -3; tail -2 $O/bench_default.err
echo "last line bytes: $(tail -n 1 $O/bench_default.stdout | wc -c)"; tail -n 1 $O/bench_default.stdout | cut -c1-600
head -40 $O/r06_pmc_summary.txt
