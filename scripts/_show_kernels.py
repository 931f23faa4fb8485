"""Prints the headline and the per-kernel table (launches x average us) of bench_full.json files: `_show_kernels.py [filter,filter] file...`"""
import json, os, sys
filt = [x for x in sys.argv[1].split(",") if x] if len(sys.argv) > 1 else []
for f in sys.argv[2:]:
    try:
        d = json.load(open(f))
    except Exception as e:
        print(os.path.basename(f), "ERR", e); continue
    r = d.get("roofline") or {}
    print("%-28s ms %8.4f launches %s stages %s" % (os.path.basename(f), d["ms_per_step"], d["config"].get("kernel_launches_per_step"),
          {k[:4]: round(v, 4) for k, v in (r.get("stage_ms") or {}).items()}))
    for k, v in sorted((r.get("kernels") or {}).items(), key=lambda kv: -kv[1]["ms_per_step"]):
        if not filt or any(x in k for x in filt):
            print("     %-46s %5.1f x %8.2f us = %8.1f us  %6.1f TF" % (k, v["launches_per_step"], v["avg_launch_us"], v["ms_per_step"] * 1e3, v["tflops"]))
