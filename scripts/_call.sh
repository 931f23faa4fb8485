cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r6c16; mkdir -p $O
S=$(date +%s.%N)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd.stdout 2> $O/driver_cmd.err
E=$(date +%s.%N)
echo "driver command wall seconds: $(python3 -c "print(round($E-$S,1))")"
echo "last line bytes: $(tail -n 1 $O/driver_cmd.stdout | wc -c)"
tail -n 1 $O/driver_cmd.stdout | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step','steps','warmup','device_resident_ms')}, d['config']['frames_per_id'], d['roofline']['kernel_frac'], d['roofline']['kernel_clock'])
for e in d['extra_configs']: print(e)
"
