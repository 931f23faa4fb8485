"""Aggregate rocprofv3 --pmc counter_collection.csv files per kernel (sum over dispatches / calls).
usage: pmc_summary.py <dir-with-*_counter_collection.csv> [...]"""
import csv, glob, sys, collections, re
for d in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '').replace('pe::', '')
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
            key = (r['Dispatch_Id'])
            if key not in seen:
                seen.add(key); calls[k] += 1
    names = sorted({c for k in acc for c in acc[k]})
    print('#', d, names)
    for k, n in calls.most_common(30):
        a = acc[k]
        line = '%-48s calls %5d' % (k[:48], n)
        if 'SQ_WAVE_CYCLES' in a:
            wc = a['SQ_WAVE_CYCLES'] or 1
            line += '  wait %4.1f%% winst %4.1f%% active %4.1f%%' % (100*a['SQ_WAIT_ANY']/wc, 100*a['SQ_WAIT_INST_ANY']/wc, 100*a['SQ_ACTIVE_INST_ANY']/wc)
            if 'SQ_WAIT_INST_LDS' in a: line += ' wlds %4.1f%%' % (100*a['SQ_WAIT_INST_LDS']/wc)
            if 'SQ_VALU_MFMA_BUSY_CYCLES' in a and a.get('SQ_BUSY_CYCLES'):
                line += '  mfma_busy/busy %5.1f%%' % (100*a['SQ_VALU_MFMA_BUSY_CYCLES']/a['SQ_BUSY_CYCLES'])
            if 'SQ_INSTS_VALU_MFMA_MOPS_F32' in a: line += '  MFMA GFLOP/call %.3f' % (a['SQ_INSTS_VALU_MFMA_MOPS_F32']*512/n/1e9)
        for c in ('FETCH_SIZE', 'WRITE_SIZE'):
            if c in a: line += '  %s KB/call raw %.1f' % (c, a[c]/n)
        print(line)
