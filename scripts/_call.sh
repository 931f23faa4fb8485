cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_batched.py -m gpu -q -x -s -k "split" 2>&1 | grep -E "B=64|heavy|passed|failed|rror" | cut -c1-200
bash scripts/gpu_ab.sh "" "--config 3 --steps 5 --warmup 2 --matrix f16x3" "conv_split,mrf_split" 2>&1 | grep -v "^$" | cut -c1-200
bash scripts/gpu_ab.sh "" "--config 4 --steps 10 --warmup 3 --matrix f16x3" "conv_split,mrf_split" 2>&1 | grep -v "^$" | cut -c1-200
bash scripts/gpu_ab.sh "" "--steps 100 --matrix f16x3" "mrf_split" 2>&1 | grep -v "^$" | cut -c1-200
