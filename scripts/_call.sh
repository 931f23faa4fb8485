cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/c3; mkdir -p $O
BQ="--no-extra --no-cpu-baseline --min-seconds 0.3"
run() { name=$1; shift; PIPER_BENCH_FULL=$O/$name.json timeout 300 python bench.py $BQ "$@" > /dev/null 2>> $O/err.log; }
for r in 1 2; do
  PIPER_HIP_ATTN4=0 run a4off_$r --steps 200
  PIPER_HIP_ATTN4=1 run a4on_$r --steps 200
done
python scripts/_show_kernels.py attn,ffn,lngemm $O/a4*.json
for r in 1 2; do
  PIPER_HIP_GROUP_TILED=0 run hi1_g0_$r --preset high --steps 40
  PIPER_HIP_GROUP_TILED=1 run hi1_g1_$r --preset high --steps 40
done
PIPER_HIP_GROUP_TILED=1 PIPER_HIP_MRF=2 run hi1_g1_mrf2 --preset high --steps 40
python scripts/_show_kernels.py conv_mfma,group,mrf,sum $O/hi1_*.json
for B in 2 4 8 16; do for g in 0 1; do PIPER_HIP_GROUP_TILED=$g run med_b${B}_g$g --batch $B --steps 30; done; done
python scripts/_show_kernels.py conv_mfma,group,mrf_sum $O/med_b*.json
for B in 2 4; do for g in 0 1; do PIPER_HIP_GROUP_TILED=$g run hi_b${B}_g$g --preset high --batch $B --steps 10; done; done
python scripts/_show_kernels.py zzz $O/hi_b*.json
timeout 900 python -m pytest tests -m gpu -q -x -k "forced or stage_boundar or medium_b16 or high_b64 or poison or no_kernel_reads" 2>&1 | tail -5
grep -v amdgpu.ids $O/err.log | tail -5
