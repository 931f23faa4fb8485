# round 3, call 11: device-side trace of one replayed B=1 step with the round-3 build (tuning build: make stamps)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3k
mkdir -p $O
timeout 300 python scripts/stamps.py medium 128 > $O/stamps_b1.txt 2>> $O/err.log
tail -5 $O/err.log | grep -v amdgpu.ids
cat $O/stamps_b1.txt | tail -150
