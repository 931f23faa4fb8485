"""Per-kernel averages of every counter in a rocprofv3 --pmc pass (+ kernel durations when the pass also traced kernels).
usage: pmc_dump.py <dir> [name-filter]"""
import collections
import csv
import glob
import re
import sys

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("pe::", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            calls[k] += 1
dur = collections.defaultdict(list)
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("pe::", "")
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, n in calls.most_common(40):
    if flt and flt not in k:
        continue
    a = acc[k]
    us = sum(dur[k]) / len(dur[k]) if dur.get(k) else 0.0
    print("%-52s calls %5d avg_us %9.2f  " % (k[:52], n, us) + "  ".join("%s=%.4g" % (c, a[c] / n) for c in sorted(a)))
