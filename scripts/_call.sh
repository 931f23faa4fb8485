cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
echo "base = EXPERIMENT build without x loads in conv_split_kernel (wrong results, timing only); new = the tree"
bash scripts/gpu_ab.sh "" "--config 3 --steps 5 --warmup 2 --matrix f16x3" "conv_split" 2>&1 | grep -v "^$" | cut -c1-200
bash scripts/gpu_ab.sh "" "--config 4 --steps 10 --warmup 3 --matrix f16x3" "conv_split" 2>&1 | grep -v "^$" | cut -c1-200
