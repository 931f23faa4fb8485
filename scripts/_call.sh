cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
bash scripts/collect_r06.sh > gpurun_out/r06_collect.log 2>&1
tail -60 gpurun_out/r06_collect.log
