cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/c11; mkdir -p $O
BQ="--no-extra --no-cpu-baseline --min-seconds 0.3"
run() { name=$1; shift; PIPER_BENCH_FULL=$O/$name.json timeout 300 python bench.py $BQ "$@" > /dev/null 2>> $O/err.log; }
for r in 1 2; do PIPER_HIP_ATTN4=0 run a4off_$r --steps 200; PIPER_HIP_ATTN4=1 run a4on_$r --steps 200; done
python scripts/_show_kernels.py attn $O/a4*.json
for T in 64 192 256 320 384 512; do PIPER_HIP_ATTN4=0 run a4off_T$T --steps 60 --ids $T; PIPER_HIP_ATTN4=2 run a4on_T$T --steps 60 --ids $T; done
python scripts/_show_kernels.py attn $O/a4*_T*.json
timeout 1200 python -m pytest tests -m gpu -q -x -k "forced or medium_t128 or intermediate or ragged or sentences or no_kernel_reads or long" 2>&1 | tail -4
grep -v amdgpu.ids $O/err.log | tail -5
