cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r6c20; mkdir -p $O
cp piper_amd/libpiper_hip.so /tmp/new.so
B="python $GRAFT_REPO_ROOT/bench.py --no-extra --no-cpu-baseline --no-roofline --min-seconds 1"
for v in base new base2 new2; do
  case $v in base*) cp piper_amd/libab_base.so piper_amd/libpiper_hip.so;; *) cp /tmp/new.so piper_amd/libpiper_hip.so;; esac
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/$v -- $B > $GRAFT_REPO_ROOT/$O/$v.json 2>/dev/null)
done
cp /tmp/new.so piper_amd/libpiper_hip.so
python - <<'PY'
import csv,glob
for v in ('base','new','base2','new2'):
    f=glob.glob('gpurun_out/r6c20/%s/**/*kernel_stats.csv'%v,recursive=True)[0]
    rows=list(csv.DictReader(open(f)))
    n=[int(r['Calls']) for r in rows if 'embed' in r['Name']][0]
    tot=sum(float(r['TotalDurationNs']) for r in rows)/n/1e3
    s=' '.join('%s %.2fx%.2f'%(r['Name'].split('(')[0].replace('void pe::','').replace('pe::','')[:22], int(r['Calls'])/n, float(r['AverageNs'])/1e3) for r in rows if any(k in r['Name'] for k in ('dds_layer4','randn','embed','lngemm4')))
    print(v, 'sum/step %.1f us'%tot, s)
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
timeout 600 python -m pytest tests/test_gpu_batched.py tests/test_gpu_parity.py -m gpu -x -q -k "noise or rng or seed or replay or drawn or zero_copy or upload" 2>&1 | tail -3
