"""MFMA pipe utilisation per kernel = executed matrix FLOPs per call (SQ_INSTS_VALU_MFMA_MOPS_F32 x 512, from a
pmc_summary.py section) / (average launch duration from the rocprofv3 kernel_stats.csv of the same command) / the f32
matrix peak (157.3 TFLOP/s). SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES is NOT a utilisation on gfx950: the two counters
are summed over different numbers of instances (values > 100 %), so the summary's ratio column is only comparable
between kernels, and this table is the figure to quote.
usage: pmc_util.py <pmc_summary.txt> <section-substring>=<kernel_stats.csv> [...]"""
import csv, re, sys
PEAK = 157.3e12
def norm(k): return re.sub(r'\s+', '', re.sub(r'\(.*', '', k).replace('void ', '').replace('pe::', ''))
txt = open(sys.argv[1]).read().splitlines()
for spec in sys.argv[2:]:
    sec, path = spec.split('=', 1)
    avg = {}
    for r in csv.DictReader(open(path)):
        avg[norm(r['Name'])] = (float(r['AverageNs']), int(r['Calls']))
    on = False
    print('# MFMA pipe utilisation, section %s x %s' % (sec, path.split('/')[-1]))
    print('# %-46s %9s %12s %12s %8s' % ('kernel', 'avg us', 'GFLOP exec', 'TFLOP/s exec', 'of peak'))
    for line in txt:
        if line.startswith('# ') and 'pmc_sq_' in line:
            on = (sec + ' ') in (line + ' ') or line.split()[1].endswith(sec)
            continue
        if not on or line.startswith('#'): continue
        m = re.match(r'(.{48}) calls\s+(\d+).*MFMA GFLOP/call ([0-9.]+)', line)
        if not m: continue
        k = norm(m.group(1)); gf = float(m.group(3))
        if gf <= 0 or k not in avg: continue
        us = avg[k][0] / 1e3
        tf = gf * 1e9 / (us * 1e-6)
        print('  %-46s %9.2f %12.3f %12.1f %7.1f%%' % (k[:46], us, gf, tf / 1e12, 100 * tf / PEAK))
