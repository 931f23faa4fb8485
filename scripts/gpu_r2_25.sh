cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2y
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "forced or golden" 2>&1 | tail -5 > $O/pytest_gpu.log
for i in 1 2 3; do
  timeout 300 python bench.py --no-cpu-baseline --steps 300 > $O/bench_b1_$i.json 2>> $O/err.log
done
timeout 300 python scripts/stamps.py medium 128 > $O/stamps_b1.txt 2>> $O/err.log
cat $O/pytest_gpu.log
python scripts/_show.py $O/bench_*.json | grep -v "^    "
head -3 $O/stamps_b1.txt
grep "conv_splitk_sum\|conv_splitk_group" $O/stamps_b1.txt | head -4
tail -3 $O/err.log
