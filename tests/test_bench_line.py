"""The driver parses the LAST stdout line of bench.py from a size-capped tail: it must be one small JSON object
(round 3 lost its measurement to a 47 KB line). Canned input: the full result object of a real run."""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CANNED = os.path.join(ROOT, "profiles", "r03_bench_default.json")


def _full():
    return json.load(open(CANNED))


def _check(line, n_gpus):
    assert "\n" not in line
    assert len(line) < bench.COMPACT_LIMIT, len(line)
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["n_gpus"] == n_gpus and d["unit"] == "samples/s" and d["dtype"] == "f32"
    assert "model" not in d["config"] and d["config"]["workload"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    return d


def test_single_gpu_line_is_compact_and_complete():
    full = _full()
    assert len(json.dumps(full)) > 40000                 # the canned object really is the oversized one
    d = _check(bench.compact_line(full, "bench_full.json"), 1)
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert abs(d["value"] - full["value"]) / full["value"] < 1e-5
    assert abs(d["ms_per_step"] - full["ms_per_step"]) / full["ms_per_step"] < 1e-4
    legs = d["extra_configs"]
    assert len(legs) == len(full["extra_configs"])
    assert all(len(json.dumps(l, separators=(",", ":"))) <= 250 for l in legs)
    assert d["full"] == "bench_full.json"


def test_honesty_fields_are_in_the_compact_line():
    """the sustained leg's ms_per_step and the frames-per-id of the synthetic utterance are in the
    line the driver reads; a bf16x3 leg names the peak its fraction is divided by (and that fraction is <= 1); the high
    voice's B=1 leg and configs[4]'s streaming rate are there."""
    full = _full()
    full["sustained"] = {"steps": 2300, "seconds": 2.01, "value": 1.22e8, "ms_per_step": 0.8741}
    full["config"]["frames_per_id"] = 417 / 128
    legs = full["extra_configs"]
    bf = [e for e in legs if str(e.get("dtype", "")).startswith("bf16x3") and e.get("roofline")]
    assert bf, "the canned run has bf16x3 legs"
    legs.append({"leg": "high voice, B=1", "dtype": "f32", "value": 3.1e7, "unit": "samples/s", "ms_per_step": 3.4, "steps": 50,
                 "frames_per_id": 3.3, "roofline": {"kernel": "conv_mfma_kernel<2,2,1,1,16,false,64>", "step": {"achieved": 80.0, "frac": 0.51}}})
    legs.append({"leg": "configs[4]", "dtype": "f32", "value": 1.9, "unit": "ms", "steps": 100, "streaming_samples_per_s": 4.2e7})
    d = _check(bench.compact_line(full, "bench_full.json"), 1)
    assert d["sustained"]["ms_per_step"] == 0.8741 and abs(d["config"]["frames_per_id"] - 3.258) < 1e-3
    by = {l["leg"]: l for l in d["extra_configs"]}
    assert by["high voice, B=1"]["frac"] == 0.51 and by["high voice, B=1"]["frames_per_id"] == 3.3
    assert by["configs[4]"]["streaming_samples_per_s"] == 4.2e7
    for l in d["extra_configs"]:
        if l.get("dtype") == "bf16x3" and "frac" in l:
            assert l["peak_tflops"] == 833 and 0 < l["frac"] <= 1.0, l


def test_multi_gpu_line_is_compact():
    full = _full()
    full.pop("extra_configs")
    full.pop("cpu_baseline")
    full["n_gpus"] = 8
    full["ranks"] = {"world_size": 8, "backend": "nccl", "rccl": "2.26.6",
                     "pci_bus_ids": ["0000:%02x:00.0" % (5 + 16 * i) for i in range(8)], "distinct_devices": 8}
    full["headline_note"] = "n" * 300
    full["per_rank_samples_per_s"] = [3.4e8 + i for i in range(8)]
    full["single_gpu_reference"] = {"value": 3.45e8, "ms_per_step": 18.8, "steps": 10, "what": "x" * 300}
    full["batched_per_gpu"] = {"config": {"workload": "y" * 300}, "value": 2.8e9, "unit": "samples/s", "x_realtime": 1.3e5,
                               "ms_per_step": 18.6, "steps": 10, "per_rank_samples_per_s": [3.5e8] * 8,
                               "single_gpu_value": 3.6e8, "single_gpu_ms_per_step": 18.1, "speedup_over_single_gpu": 7.78}
    full["weight_broadcast_bytes"] = 139000000
    full["weight_broadcast_s"] = 0.0021
    d = _check(bench.compact_line(full, None), 8)
    assert len(d["per_rank_samples_per_s"]) == 8 and d["batched_per_gpu"]["value"] == 2.8e9
    assert d["batched_per_gpu"]["speedup_over_single_gpu"] == 7.78 and len(d["batched_per_gpu"]["workload"]) <= 120
    assert d["weight_broadcast"]["bytes"] == 139000000
    # auditable: who took part
    assert d["ranks"]["world_size"] == 8 and d["ranks"]["distinct_devices"] == 8 and len(d["ranks"]["pci_bus_ids"]) == 8
    assert d["ranks"]["backend"] == "nccl" and d["ranks"]["rccl"]
    assert d["batched_per_gpu"]["single_gpu_value"] == 3.6e8


def test_batched_scaling_and_rank_list_are_shed_last():
    full = _full()
    full["n_gpus"] = 8
    full["ranks"] = {"world_size": 8, "backend": "nccl", "rccl": "2.26.6", "pci_bus_ids": ["0000:%02x:00.0" % i for i in range(8)],
                     "distinct_devices": 8}
    full["batched_per_gpu"] = {"config": {"workload": "y" * 300}, "value": 2.8e9, "ms_per_step": 18.6, "steps": 10,
                               "single_gpu_value": 3.6e8, "speedup_over_single_gpu": 7.78}
    leg = copy.deepcopy(full["extra_configs"][0])
    full["extra_configs"] = [dict(leg, leg=f"leg {i} " + "z" * 40) for i in range(60)]
    d = json.loads(bench.compact_line(full, "bench_full.json"))
    assert "extra_configs" not in d and d["batched_per_gpu"]["speedup_over_single_gpu"] == 7.78 and d["ranks"]["distinct_devices"] == 8


def test_line_sheds_optional_parts_rather_than_overflow():
    full = _full()
    leg = copy.deepcopy(full["extra_configs"][0])
    full["extra_configs"] = [dict(leg, leg=f"leg {i} " + "z" * 40) for i in range(40)]
    line = bench.compact_line(full, "bench_full.json")
    d = _check(line, 1)
    assert "extra_configs" not in d and "cpu_baseline" in d and "roofline" in d


def test_failed_leg_is_reported_short():
    e = bench.compact_leg({"leg": "configs[2]", "error": "RuntimeError: " + "q" * 1000})
    assert len(json.dumps(e)) < 250 and e["error"].startswith("RuntimeError")


def test_headline_is_the_infer_seconds_span_and_names_both_clocks():
    """the driver-timed step is what the reference's inferSeconds spans (piper.cpp:385-395) -- host ids
    in, device pipeline with BOTH noise sites drawn by the engine, int16 PCM on the host -- and says so in
    config.workload; the resident-input figure sits beside it; the dominant kernel's fraction names the clock it was
    computed on and carries both durations."""
    from piper_amd import weights as W
    txt = bench.workload_text(2, "medium", W.preset("medium"), 1, 128)
    assert "host ids in (pe_upload)" in txt and "both noise sites" in txt and "int16 PCM to host (pe_fetch)" in txt
    full = _full()
    full["config"]["workload"] = txt
    full["device_resident_ms"] = 0.8512
    full["roofline"].update({"clock": "rocprofv3", "avg_launch_us": 10.02, "avg_launch_us_event_pairs": 8.82,
                             "avg_launch_us_rocprof": 10.02, "rocprof_source": "r06_b1_kernel_stats.csv"})
    d = _check(bench.compact_line(full, "bench_full.json"), 1)
    assert d["config"]["workload"] == txt                              # not truncated
    assert d["device_resident_ms"] == 0.8512
    r = d["roofline"]
    assert r["kernel_clock"] == "rocprofv3" and r["kernel_us_rocprof"] == 10.02 and r["kernel_us_event_pairs"] == 8.82
    assert r["kernel_avg_launch_us"] == 10.02 and r["rocprof_source"].endswith("kernel_stats.csv")


def test_rocprof_clock_lookup_reads_the_committed_summaries():
    k = bench.rocprof_kernel_us("medium", 1, "gate4_kernel")
    assert k and 5.0 < k["avg_us"] < 20.0 and k["source"].endswith("_b1_kernel_stats.csv")
    assert bench.rocprof_kernel_us("medium", 7, "gate4_kernel") is None
    assert bench.rocprof_kernel_us("medium", 1, "no_such_kernel") is None
