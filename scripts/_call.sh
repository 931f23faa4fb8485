cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/c8; mkdir -p $O
BQ="--no-extra --no-cpu-baseline --min-seconds 0.3"
run() { name=$1; shift; PIPER_BENCH_FULL=$O/$name.json timeout 300 python bench.py $BQ "$@" > /dev/null 2>> $O/err.log; }
PIPER_STAMPS_LIB=$GRAFT_REPO_ROOT/piper_amd/libab_stamps.so PIPER_HIP_ATTN4=1 timeout 300 python scripts/stamps.py medium 128 2>&1 | head -1
for r in 1 2 3; do
  PIPER_HIP_ATTN4=0 run a4off_$r --steps 200
  PIPER_HIP_ATTN4=1 run a4on_$r --steps 200
done
python scripts/_show_kernels.py attn $O/a4*.json
for T in 64 160 192 256; do PIPER_HIP_ATTN4=0 run a4off_T$T --steps 100 --ids $T; PIPER_HIP_ATTN4=2 run a4on_T$T --steps 100 --ids $T; done
python scripts/_show_kernels.py attn $O/a4*_T*.json
timeout 600 python -m pytest tests -m gpu -q -x -k "medium_t128 or intermediate or ragged or sentences or no_kernel_reads" 2>&1 | tail -4
grep -v amdgpu.ids $O/err.log | tail -5
