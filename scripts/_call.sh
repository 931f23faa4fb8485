cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/c4; mkdir -p $O
export PIPER_STAMPS_LIB=$GRAFT_REPO_ROOT/piper_amd/libab_stamps.so
for a in 0 1; do echo "== ATTN4=$a"; PIPER_HIP_ATTN4=$a timeout 300 python scripts/stamps.py medium 128 2>&1 | tee $O/stamps_a$a.txt | head -60; done
