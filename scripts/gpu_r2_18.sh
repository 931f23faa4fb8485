cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2r
mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "persist or group_two" 2>&1 | tail -15 > $O/pytest_gpu.log
for i in 1 2 3; do
  timeout 300 python bench.py --no-cpu-baseline --steps 300 > $O/bench_b1_$i.json 2>> $O/err.log
  PIPER_HIP_PERSIST_DP=1 timeout 300 python bench.py --no-cpu-baseline --steps 300 > $O/bench_b1_persist_$i.json 2>> $O/err.log
done
PIPER_HIP_PERSIST_DP=1 timeout 300 python scripts/stamps.py medium 128 > $O/stamps_b1_persist.txt 2>> $O/err.log
cat $O/pytest_gpu.log
python scripts/_show.py $O/bench_*.json | grep -v "^    "
grep -n "dp_persist\|duration\|randn\|sum in-WG" $O/stamps_b1_persist.txt | head
tail -3 $O/err.log
