# Round 4, call 17: does an XCD's L2 content survive a kernel boundary in a hipGraph; does prefetching the next kernel's weights pay
cd $GRAFT_REPO_ROOT
./scripts/microbench/l2_boundary 2>&1 | tee gpurun_out/l2_boundary.txt
