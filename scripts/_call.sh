cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r6c29; mkdir -p $O
cp piper_amd/libpiper_hip.so /tmp/new.so
BQ="--no-extra --no-cpu-baseline --no-roofline --min-seconds 1"
run() { l=$1; c=$2; m=$3
  timeout 300 python bench.py $BQ --config $c --matrix $m > $O/$l.json 2>> $O/err.log
  python -c "
import json;d=json.loads(open('$O/$l.json').read().strip().splitlines()[-1]);print('%-22s cfg $c %-7s ms %.3f'%('$l',d['ms_per_step']))" "$m"
}
for r in 1 2; do
cp piper_amd/libab_base.so piper_amd/libpiper_hip.so
run base_c4_f32_$r 4 f32; run base_c4_h_$r 4 f16x3; run base_c4_b6_$r 4 bf16x6; run base_c3_f32_$r 3 f32; run base_c3_h_$r 3 f16x3
cp /tmp/new.so piper_amd/libpiper_hip.so
run new_c4_f32_$r 4 f32; run new_c4_h_$r 4 f16x3; run new_c4_b6_$r 4 bf16x6; run new_c3_f32_$r 3 f32; run new_c3_h_$r 3 f16x3
done
grep -v amdgpu.ids $O/err.log | tail -3
timeout 900 python -m pytest tests/test_gpu_batched.py tests/test_gpu_parity.py -m gpu -x -q -k "not baseline_sizes" 2>&1 | tail -3
