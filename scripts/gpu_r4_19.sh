# Round 4, call 19: first-use cost of a kernel's code inside a replayed graph; code prefetch by the previous kernel
cd $GRAFT_REPO_ROOT
./scripts/microbench/icache_boundary 2>&1 | tee gpurun_out/icache_boundary.txt
