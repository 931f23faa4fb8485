cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2v
mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "forced and default" 2>&1 | tail -5 > $O/pytest_gpu.log
for i in 1 2 3; do
  timeout 300 python bench.py --no-cpu-baseline --steps 300 > $O/bench_b1_$i.json 2>> $O/err.log
  PIPER_HIP_SUM_D4=1 timeout 300 python bench.py --no-cpu-baseline --steps 300 > $O/bench_b1_d4_$i.json 2>> $O/err.log
done
timeout 300 python scripts/stamps.py medium 128 > $O/stamps_b1.txt 2>> $O/err.log
PIPER_HIP_SUM_D4=1 timeout 300 python scripts/stamps.py medium 128 > $O/stamps_b1_d4.txt 2>> $O/err.log
cat $O/pytest_gpu.log
python scripts/_show.py $O/bench_*.json | grep -v "^    "
grep -n "conv_splitk_group \|conv_splitk_sum \|conv_mfma  " $O/stamps_b1.txt | head -8
grep -n "conv_splitk_group \|conv_splitk_sum \|conv_mfma  " $O/stamps_b1_d4.txt | head -8
tail -3 $O/err.log
