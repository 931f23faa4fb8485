import json, glob, os
O = "gpurun_out/r03/"
def show(f):
    try:
        d = json.loads(open(O + f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "ERR", e); return None
    r = d.get("roofline") or {}
    print(f, "ms %.3f val %.1fM launches %s" % (d["ms_per_step"], d["value"] / 1e6, d["config"].get("kernel_launches_per_step")),
          "top", r.get("kernel"), "frac %.3f" % r.get("frac", 0), "step %.3f" % r.get("step", {}).get("frac", 0),
          "hifigan TF %.1f" % r.get("stage_tflops", {}).get("hifigan", 0), "spec", d.get("speculation"))
    return d
d = show("bench_default.json")
if d:
    for e in d.get("extra_configs", []):
        r = e.get("roofline") or {}
        print("  leg", e.get("leg"), "%.1fs" % e.get("leg_seconds", 0), e.get("error") or ("val %.4g %s ms %.3f" % (e.get("value", 0), e.get("unit"), e.get("ms_per_step", 0) or e.get("ms_per_call_mean", 0))),
              "top", r.get("kernel"), "frac %.3f" % r.get("frac", 0), "step %.3f" % r.get("step", {}).get("frac", 0), "hifigan %.1f" % r.get("stage_tflops", {}).get("hifigan", 0), e.get("speculation", ""))
    print("  cpu", d.get("cpu_baseline", {}).get("value"), "stage_ms", d["roofline"]["stage_ms"])
    for k, v in sorted(d["roofline"]["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"])[:14]:
        print("   %-56s n %5.1f us %7.2f TF %6.1f" % (k[:56], v["launches_per_step"], v["avg_launch_us"], v["tflops"]))
show("bench_nccl_ws1.json")
