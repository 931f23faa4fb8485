cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -8 > gpurun_out/t1.log
for r in 0 1 0 1; do
  PIPER_HIP_RES_X=$r python bench.py --no-cpu-baseline > gpurun_out/rx${r}_b1.json 2> gpurun_out/f.err
  PIPER_HIP_RES_X=$r python bench.py --no-cpu-baseline --batch 16 --steps 20 > gpurun_out/rx${r}_b16.json 2>> gpurun_out/f.err
  PIPER_HIP_RES_X=$r python bench.py --no-cpu-baseline --batch 64 --steps 10 > gpurun_out/rx${r}_b64.json 2>> gpurun_out/f.err
done
cat gpurun_out/t1.log
