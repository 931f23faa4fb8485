cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2j
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "COLCHAIN or golden or full_size or intermediate or product_path or stress or medium-B16" 2>&1 | tail -15 > $O/pytest_gpu.log
timeout 300 python scripts/stamps.py medium 128 > $O/stamps_b1.txt 2>> $O/err.log
PIPER_HIP_COLCHAIN=0 timeout 300 python scripts/stamps.py medium 128 > $O/stamps_b1_nochain.txt 2>> $O/err.log
timeout 300 python bench.py --no-cpu-baseline --steps 300 > $O/bench_b1.json 2>> $O/err.log
cat $O/pytest_gpu.log
python scripts/_show.py $O/bench_b1.json | head -3
cat $O/stamps_b1.txt
echo ------ nochain
tail -22 $O/stamps_b1_nochain.txt
tail -3 $O/err.log
