# Round 4, call 38: a persistent kernel on ONE XCD exchanging data through that XCD's L2 (scripts/microbench/xcd_persist.hip)
cd $GRAFT_REPO_ROOT
timeout 120 ./scripts/microbench/xcd_persist 2>&1 | tee gpurun_out/xcd_persist.txt
