# Round 4, call 40: host-side breakdown of a whole single-utterance call (scripts/exp_api.py), with and without interrupt-driven waits
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 200 python scripts/exp_api.py medium 128
  HSA_ENABLE_INTERRUPT=0 timeout 200 python scripts/exp_api.py medium 128
  timeout 200 python scripts/exp_api.py medium 64 ) 2>&1 | grep -v Warning | tee gpurun_out/r04_exp_api.txt
