cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for m in 160 96 48 0; do
  PIPER_HIP_SPLITK_MAX=$m python bench.py --no-cpu-baseline > gpurun_out/sk${m}_b1.json 2> gpurun_out/f.err
  PIPER_HIP_SPLITK_MAX=$m python bench.py --no-cpu-baseline --batch 4 > gpurun_out/sk${m}_b4.json 2>> gpurun_out/f.err
done
