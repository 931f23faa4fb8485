import csv,sys
for f in sys.argv[1:]:
    rows=list(csv.DictReader(open(f)))
    tot=sum(float(r['TotalDurationNs']) for r in rows)
    print(f, 'total ms', tot/1e6)
    for r in rows[:22]:
        print('  ', r['Name'][:72].ljust(72), r['Calls'].rjust(6), '%8.1f'%(float(r['AverageNs'])/1e3), '%5.1f%%'%(100*float(r['TotalDurationNs'])/tot))
