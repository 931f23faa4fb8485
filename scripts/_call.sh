cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r6c28; mkdir -p $O
timeout 900 python scripts/stress_parity.py 48 4242 > $O/stress_f32_s4242.log 2>&1; tail -1 $O/stress_f32_s4242.log
PIPER_HIP_MATRIX=f16x3 timeout 900 python scripts/stress_parity.py 48 4243 > $O/stress_f16x3_s4243.log 2>&1; tail -1 $O/stress_f16x3_s4243.log
PIPER_HIP_MATRIX=bf16x6 timeout 900 python scripts/stress_parity.py 32 4244 > $O/stress_bf16x6_s4244.log 2>&1; tail -1 $O/stress_bf16x6_s4244.log
timeout 900 python scripts/stress_parity.py 24 4245 300 150 12 > $O/stress_f32_s4245_b12.log 2>&1; tail -1 $O/stress_f32_s4245_b12.log
