cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for r in 1 2 3; do
for v in 0 1; do
  PIPER_HIP_DDS16=$v python bench.py --no-cpu-baseline > gpurun_out/dr${r}_${v}.json 2> gpurun_out/at.err
  PIPER_HIP_DDS16=$v python bench.py --no-cpu-baseline --batch 16 --steps 20 > gpurun_out/dr16_${r}_${v}.json 2> gpurun_out/at.err
done
done
