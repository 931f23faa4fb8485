cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for m in 0 512 4096; do
  PIPER_HIP_S2_MIN=$m python bench.py --no-cpu-baseline --batch 16 --steps 20 > gpurun_out/s2_${m}_b16.json 2> gpurun_out/f.err
  PIPER_HIP_S2_MIN=$m python bench.py --no-cpu-baseline --batch 64 --steps 10 > gpurun_out/s2_${m}_b64.json 2>> gpurun_out/f.err
  PIPER_HIP_S2_MIN=$m python bench.py --no-cpu-baseline --preset high --batch 8 --steps 10 > gpurun_out/s2_${m}_h8.json 2>> gpurun_out/f.err
done
