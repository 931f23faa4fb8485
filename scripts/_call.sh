cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/c13; mkdir -p $O
export PIPER_STAMPS_LIB=$GRAFT_REPO_ROOT/piper_amd/libab_stamps.so
timeout 300 python scripts/stamps.py medium 128 2>&1 | tee $O/stamps.txt | head -8
grep -E "conv_splitk16|gate|colchain   " $O/stamps.txt | tail -12
PIPER_HIP_GATE4=0 timeout 300 python scripts/stamps.py medium 128 2>&1 | grep -E "conv_splitk16  " | tail -4
