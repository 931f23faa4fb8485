# Round 4, call 31: device-side trace of one replayed B=1 step of the current build (tuning build with PE_STAMPS):
# per launch the time inside workgroup (0,0,0) and the gap to the previous launch's exit.
cd $GRAFT_REPO_ROOT
python scripts/stamps.py medium 128 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_stamps_b1.txt | tail -130
