set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -5 > gpurun_out/t1.log
for f in 0 1; do
  PIPER_HIP_FUSE_MRF=$f timeout 300 python bench.py --no-cpu-baseline > gpurun_out/b1_f$f.json 2> gpurun_out/b1_f$f.err
  PIPER_HIP_FUSE_MRF=$f timeout 300 python bench.py --no-cpu-baseline --batch 16 --steps 20 > gpurun_out/b16_f$f.json 2>> gpurun_out/b1_f$f.err
  PIPER_HIP_FUSE_MRF=$f timeout 300 python bench.py --no-cpu-baseline --preset high --batch 8 --steps 10 > gpurun_out/h8_f$f.json 2>> gpurun_out/b1_f$f.err
done
cat gpurun_out/t1.log
