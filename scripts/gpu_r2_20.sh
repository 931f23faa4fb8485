cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2t
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "forced or golden or full_size or stress" 2>&1 | tail -15 > $O/pytest_gpu.log
for i in 1 2 3; do
  timeout 300 python bench.py --no-cpu-baseline --steps 300 > $O/bench_b1_$i.json 2>> $O/err.log
  PIPER_HIP_GROUP_MRF=2 timeout 300 python bench.py --no-cpu-baseline --steps 300 > $O/bench_b1_nosum_$i.json 2>> $O/err.log
done
timeout 300 python bench.py --no-cpu-baseline --config 3 --batch 1 --steps 100 > $O/bench_high_b1.json 2>> $O/err.log
PIPER_HIP_GROUP_MRF=2 timeout 300 python bench.py --no-cpu-baseline --config 3 --batch 1 --steps 100 > $O/bench_high_b1_nosum.json 2>> $O/err.log
timeout 300 python scripts/stamps.py medium 128 > $O/stamps_b1.txt 2>> $O/err.log
cat $O/pytest_gpu.log
python scripts/_show.py $O/bench_*.json | grep -v "^    "
tail -26 $O/stamps_b1.txt | head -12
tail -3 $O/err.log
