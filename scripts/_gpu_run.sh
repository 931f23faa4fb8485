cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -8 > gpurun_out/t1.log
python bench.py --no-cpu-baseline > gpurun_out/n_b1.json 2> gpurun_out/f.err
python bench.py --no-cpu-baseline --batch 16 --steps 20 > gpurun_out/n_b16.json 2>> gpurun_out/f.err
python bench.py --no-cpu-baseline --batch 64 --steps 10 > gpurun_out/n_b64.json 2>> gpurun_out/f.err
python bench.py --no-cpu-baseline --preset high --batch 8 --steps 10 > gpurun_out/n_h8.json 2>> gpurun_out/f.err
cat gpurun_out/t1.log
