import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['roofline']
        print(f, 'ms=%.3f'%d['ms_per_step'], {k:round(v,3) for k,v in r['stage_ms'].items()}, 'hifiTF=%.1f'%r['stage_tflops']['hifigan'])
        print('   ', r['kernel'], '%.1f'%r['achieved'], {k:(round(v['ms_per_step'],3),v['launches_per_step'],round(v['avg_launch_us'],1),round(v['tflops'],1)) for k,v in r['kernels'].items()})
    except Exception as e: print(f,'ERR',e)
