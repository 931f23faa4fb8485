cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2k
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "COLCHAIN or golden or full_size or intermediate or product_path or stress or medium-B16 or x-low" 2>&1 | tail -15 > $O/pytest_gpu.log
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --steps 300 > $O/bench_b1_$i.json 2>> $O/err.log
  PIPER_HIP_COLCHAIN=0 timeout 300 python bench.py --no-cpu-baseline --steps 300 > $O/bench_b1_nochain_$i.json 2>> $O/err.log
  PIPER_HIP_PAR_MRF=1 timeout 300 python bench.py --no-cpu-baseline --steps 300 > $O/bench_b1_parmrf_$i.json 2>> $O/err.log
done
timeout 300 python scripts/stamps.py medium 128 > $O/stamps_b1.txt 2>> $O/err.log
cat $O/pytest_gpu.log
python scripts/_show.py $O/bench_b1_*.json | grep -v "^    "
python scripts/_show.py $O/bench_b1_1.json | grep "^    "
head -12 $O/stamps_b1.txt
tail -18 $O/stamps_b1.txt
tail -3 $O/err.log
