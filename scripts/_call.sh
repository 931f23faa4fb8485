cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r6c4; mkdir -p $O
./scripts/microbench/mfma16x16x32 2>&1 | tee $O/mfma16x16x32.txt
