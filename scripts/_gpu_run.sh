cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for m in 96 200 400 800; do
  for b in 4 8 16 32; do
    PIPER_HIP_SPLITK_MAX=$m python bench.py --no-cpu-baseline --batch $b --steps 20 > gpurun_out/skm${m}_b$b.json 2> gpurun_out/f.err
  done
done
