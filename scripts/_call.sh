cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r6c23; mkdir -p $O
cp piper_amd/libpiper_hip.so /tmp/new.so
BQ="--no-extra --no-cpu-baseline --no-roofline --min-seconds 1"
run() { l=$1; c=$2; m=$3
  timeout 300 python bench.py $BQ --config $c --matrix $m > $O/$l.json 2>> $O/err.log
  python -c "
import json;d=json.loads(open('$O/$l.json').read().strip().splitlines()[-1]);print('%-28s cfg $c $m ms %.3f'%('$l',d['ms_per_step']))"
}
for r in 1 2 3; do
cp piper_amd/libab_base.so piper_amd/libpiper_hip.so
run base_c3_$r 3 f16x3
cp /tmp/new.so piper_amd/libpiper_hip.so
run new_c3_$r 3 f16x3
done
for r in 1 2; do
cp piper_amd/libab_base.so piper_amd/libpiper_hip.so
run base_c3b6_$r 3 bf16x6; run base_c4b6_$r 4 bf16x6
cp /tmp/new.so piper_amd/libpiper_hip.so
run new_c3b6_$r 3 bf16x6; run new_c4b6_$r 4 bf16x6
done
grep -v amdgpu.ids $O/err.log | tail -3
