cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r6c27; mkdir -p $O
S=$SECONDS; python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.stdout 2> $O/bench_driver.err; echo "rc $? wall $((SECONDS-S)) s"
echo "bytes $(tail -n 1 $O/bench_driver.stdout | wc -c)"; tail -n 1 $O/bench_driver.stdout | cut -c1-200
