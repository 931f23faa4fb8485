cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r6c17; mkdir -p $O
BA="--no-extra --no-cpu-baseline --no-roofline"
W="--config 4 --steps 3 --warmup 2 --min-seconds 0 --matrix f16x3"
prof() { d=$1; shift; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/$d "$@" > /dev/null 2>&1); }
B="python $GRAFT_REPO_ROOT/bench.py $BA"
prof sq2 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -- $B $W
python - <<'PY'
import csv,glob,collections,re
acc=collections.defaultdict(lambda: collections.defaultdict(float)); calls=collections.Counter()
for f in glob.glob('gpurun_out/r6c17/sq2/**/*counter_collection.csv',recursive=True):
    seen=set()
    for r in csv.DictReader(open(f)):
        k=re.sub(r'\(.*','',r['Kernel_Name']).replace('void ','').replace('pe::','')
        acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
        if r['Dispatch_Id'] not in seen: seen.add(r['Dispatch_Id']); calls[k]+=1
for k,n in calls.most_common(40):
    if 'split' not in k and 'conv_mfma' not in k: continue
    a=acc[k]
    print('%-46s calls %4d VALU/MFMA %.2f SALU/MFMA %.2f VMEM/MFMA %.2f LDS/MFMA %.2f  mfma_busy/busy %.0f%%' % (k[:46],n,a['SQ_INSTS_VALU']/max(a['SQ_INSTS_MFMA'],1),a['SQ_INSTS_SALU']/max(a['SQ_INSTS_MFMA'],1),a['SQ_INSTS_VMEM_RD']/max(a['SQ_INSTS_MFMA'],1),a['SQ_INSTS_LDS']/max(a['SQ_INSTS_MFMA'],1), 100*a['SQ_VALU_MFMA_BUSY_CYCLES']/max(a['SQ_BUSY_CYCLES'],1)))
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -delete
