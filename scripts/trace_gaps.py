"""Per-kernel duration AND gap-to-predecessor from a rocprofv3 --kernel-trace CSV (steady state of a bench run):
separates what a launch costs inside the kernel from what the boundary before it costs.
usage: trace_gaps.py <dir with *kernel_trace.csv> [skip_fraction]"""
import collections, csv, glob, os, re, sys

d = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
rows = rows[int(len(rows) * skip):]
dur, gap, cnt = collections.defaultdict(float), collections.defaultdict(float), collections.Counter()
prev_end = None
tot_gap = tot_dur = 0
for s, e, n in rows:
    n = re.sub(r"\(.*", "", n).replace("void ", "").replace("pe::", "").replace(" ", "")
    dur[n] += e - s
    cnt[n] += 1
    if prev_end is not None and 0 <= s - prev_end < 50000:      # ignore host-side pauses (> 50 us)
        gap[n] += s - prev_end
        tot_gap += s - prev_end
    tot_dur += e - s
    prev_end = e
print(f"{len(rows)} dispatches, kernel time {tot_dur/1e3:.0f} us, gaps (<50us) {tot_gap/1e3:.0f} us, "
      f"avg duration {tot_dur/len(rows)/1e3:.2f} us, avg gap {tot_gap/len(rows)/1e3:.2f} us")
print(f"{'kernel':70s} {'calls':>7s} {'avg_us':>8s} {'gap_before_us':>14s}")
for n in sorted(dur, key=lambda k: -dur[k]):
    print(f"{n[:70]:70s} {cnt[n]:7d} {dur[n]/cnt[n]/1e3:8.2f} {gap[n]/cnt[n]/1e3:14.2f}")
