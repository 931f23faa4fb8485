cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r6c25; mkdir -p $O
BQ="--no-extra --no-cpu-baseline --no-roofline --min-seconds 1"
run() { l=$1; c=$2; m=$3; shift 3
  env "$@" timeout 300 python bench.py $BQ --config $c --matrix $m > $O/$l.json 2>> $O/err.log
  python -c "
import json;d=json.loads(open('$O/$l.json').read().strip().splitlines()[-1]);print('%-28s cfg $c $m ms %.3f'%('$l',d['ms_per_step']))"
}
for r in 1 2; do
for u in 0 4 6 12 21; do run u${u}_c4_$r 4 f16x3 PIPER_HIP_SPLIT_BM64_MAXU=$u; done
done
for u in 0 4 12 0 4 12; do run u${u}_c3 3 f16x3 PIPER_HIP_SPLIT_BM64_MAXU=$u; done
for u in 0 4 12 0 4 12; do run b6_u${u}_c4 4 bf16x6 PIPER_HIP_SPLIT_BM64_MAXU=$u; done
grep -v amdgpu.ids $O/err.log | tail -3
PIPER_HIP_SPLIT_BM64_MAXU=21 timeout 600 python -m pytest tests/test_gpu_batched.py -m gpu -x -q -k "split and not baseline" 2>&1 | tail -3
