#!/usr/bin/env python3
"""Throughput of the synthesis hot path on MI355X (BASELINE.json metric: audio samples/s and x real-time).

A "step" is one pass of the hot path over one batch of synthetic input, timed over the span the reference's
`inferSeconds` covers (src/cpp/piper.cpp:385-395: host tensors in, output tensor out): `pe_upload` of the phoneme ids
from host memory, `pe_run` (the whole device pipeline; the ENGINE draws both noise sites -- the graph's two
RandomNormalLike nodes, vits/models.py:111,718 -- fresh every step) and `pe_fetch` of the int16 PCM into host memory.
The same step with the ids left resident in HBM (upload once, then `pe_run` + `pe_fetch`) is reported beside it as
`device_resident_ms`.

ONE invocation covers every BASELINE.json configuration (the driver only ever runs `python bench.py --gpus N`):

    python bench.py                  # N=1. Headline line = configs[1]: en_US-lessac-medium architecture, ONE utterance
                                     # of 128 ids (what one piper::synthesize() call does). `extra_configs` carries
                                     # time-boxed legs for configs[2] (high, 64 x 128), configs[3]'s per-GPU share
                                     # (medium, 64 x 128), configs[4] (streaming p50 first chunk) and a B=1 leg with
                                     # changing text + noise every call (speculation misses), each with its own roofline
    python bench.py --gpus N         # N>1: launches N ranks itself (torch.distributed.run). Headline = the SAME per-GPU
                                     # workload as N=1 (configs[1]: one utterance per GPU per step, RCCL broadcast of the
                                     # packed voice at load), so the per-N values are one weak-scaling curve;
                                     # `batched_per_gpu` = configs[3] (medium, 64 utterances per GPU, 512 over 8) with
                                     # rank 0's single-GPU rate on that share and the speed-up over it
    python bench.py --config 3       # any single configuration as the headline (2..5, 1-based like SURVEY.md 8d)

The rate of the one-call form `pe_synthesize_batch` (the same span plus the float waveform copied to the host as well) is
reported beside the headline as `api_inclusive`. Rank 0 prints ONE compact JSON
line (< 4 KB: headline, roofline, cpu_baseline, one short entry per leg) as the last line of stdout and writes the full
per-kernel tables of every leg to bench_full.json.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MATRIX_PEAK_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32
# opt-in split-operand matrix modes (kernels/conv_bf3.h: conv_split_kernel<SM,...>): 3 / 3 / 6 sixteen-bit MFMAs
# (v_mfma_f32_32x32x16_{bf16,f16}) per f32-equivalent product, dense 16-bit peak ~2500 TFLOP/s (same guide) -> 833 / 833 /
# 417 TFLOP/s of algorithmic (f32-equivalent) FLOPs for the kernels that use them
SPLIT_MODES = {"bf16x3": (0, 3), "f16x3": (1, 3), "bf16x6": (2, 6)}          # PIPER_HIP_MATRIX -> (kernel SM, MFMAs per product)
BF16X3_PEAK_TFLOPS = 2500.0 / 3.0
HBM_PEAK_GBPS = 8000.0              # same guide: HBM3E ~8 TB/s
DTYPE_SPLIT = {
    "bf16x3": "bf16x3 (flow + generator conv GEMMs: f32 operands split into two bf16 terms = 16 significand bits, 3 bf16 MFMAs, "
              "f32 accumulate; text encoder, duration predictor and all small-batch split-K launches f32)",
    "f16x3": "f16x3 (flow + generator conv GEMMs: f32 operands split into two f16 terms = 22 significand bits, 3 f16 MFMAs, f32 "
             "accumulate; everything else f32)",
    "bf16x6": "bf16x6 (flow + generator conv GEMMs: f32 operands split into three bf16 terms = all 24 significand bits, 6 bf16 "
              "MFMAs, f32 accumulate; everything else f32)",
}
DTYPE_BF3 = DTYPE_SPLIT["bf16x3"]


def split_peak(mode):
    return 2500.0 / SPLIT_MODES[mode][1]


def short_dtype(dt):
    dt = str(dt)
    for m in SPLIT_MODES:
        if dt.startswith(m):
            return m
    return dt


def kernel_peak(name):
    for m, (sm, n) in SPLIT_MODES.items():
        if name.startswith(f"conv_split_kernel<{sm},"):
            return 2500.0 / n
        if name.startswith(f"mrf_split_kernel<{sm},"):
            return 2500.0 / n
    return BF16X3_PEAK_TFLOPS if name.startswith("conv_bf3_kernel") else FP32_MATRIX_PEAK_TFLOPS
SCALES = (0.667, 1.0, 0.8)

# BASELINE.json configs (1-based like SURVEY.md section 8d): preset, utterances per GPU, ids per utterance
CONFIGS = {2: ("medium", 1, 128), 3: ("high", 64, 128), 4: ("medium", 64, 128), 5: ("high", 1, 128)}


# ---------------------------------------------------------------------------------------------------------------------
# The result line. The driver keeps only the tail of stdout (~8 KB) and parses its LAST line, so that line is a compact
# object (< 4 KB, like the reference harness's one small JSON object, src/benchmark/benchmark_onnx.py:73-81); the full
# per-kernel tables of every leg go to a file (bench_full.json beside this script, or $PIPER_BENCH_FULL).
COMPACT_LIMIT = 4096


def _r(x, sig=5):
    """Floats to `sig` significant digits (ints and None unchanged): the line is read by people and a size-capped parser."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    try:
        v = float(f"{float(x):.{sig}g}")
        return int(v) if abs(v) >= 1e5 and v == int(v) else v          # (123400000, not 123400000.0: the line is size-capped)
    except (TypeError, ValueError):
        return None


def _short(s, n):
    s = str(s)
    return s if len(s) <= n else s[:n - 1] + "~"


def compact_roofline(roof):
    """`frac` is the STEP-level fraction (all algorithmic FLOPs of a step over the timed step, against the f32 matrix
    peak): the per-kernel winner changes from box to box at B=1 (17 % of the time at most), the step figure does not.
    The kernel with the largest share of device time and the one that carries the most FLOPs are named beside it."""
    if not roof:
        return None
    ks = roof.get("kernels") or {}
    st = roof.get("step") or {}

    def kview(name):
        k = ks.get(name)
        if not k:
            return None
        return {"name": _short(name, 60), "launches": _r(k["launches_per_step"], 4), "avg_us": _r(k["avg_launch_us"], 4),
                "tflops": _r(k["tflops"], 4), "frac": _r(k["frac_of_mfma_peak"], 3)}

    top_flop = max(ks, key=lambda n: ks[n]["algorithmic_gflop_per_launch"] * ks[n]["launches_per_step"]) if ks else None
    tr = roof.get("traffic")
    out = {"bound": roof.get("bound"), "scope": "step", "achieved": _r(st.get("achieved")),
           "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": _r(st.get("frac"), 4),
           "kernel": _short(roof.get("kernel"), 60), "kernel_frac": _r(roof.get("frac"), 3),
           "kernel_achieved": _r(roof.get("achieved"), 4), "kernel_share_of_time": _r(roof.get("share_of_profiled_kernel_time"), 3),
           "kernel_avg_launch_us": _r(roof.get("avg_launch_us"), 4), "kernel_clock": roof.get("clock"),
           "kernel_us_event_pairs": _r(roof.get("avg_launch_us_event_pairs"), 4),
           "kernel_us_rocprof": _r(roof.get("avg_launch_us_rocprof"), 4), "rocprof_source": roof.get("rocprof_source"),
           "traffic": ({"hbm_bytes_per_launch": tr.get("hbm_bytes_per_launch"),
                        "algorithmic_bytes_per_launch": _r(tr.get("algorithmic_bytes_per_launch"), 6),
                        "source": tr.get("source")} if tr else None),
           "kernel_launches": _r((ks.get(roof.get("kernel")) or {}).get("launches_per_step"), 4),
           "top_flop_kernel": kview(top_flop),
           "step": {"algorithmic_gflop": _r(st.get("algorithmic_gflop"))},
           "stage_ms": {k[:4]: _r(v, 4) for k, v in (roof.get("stage_ms") or {}).items()},
           "hifigan": {"tflops": _r((roof.get("stage_tflops") or {}).get("hifigan"), 4),
                       "frac": _r(((roof.get("stage_tflops") or {}).get("hifigan") or 0.0) / FP32_MATRIX_PEAK_TFLOPS, 3)}}
    return out


def compact_leg(e):
    """One extra_configs leg in <= 250 bytes."""
    roof = e.get("roofline") or {}
    c = {"leg": _short(e.get("leg", ""), 48)}
    if "error" in e:
        c["error"] = _short(e["error"], 120)
        return c
    dt = short_dtype(e.get("dtype", "f32"))
    if dt != "f32":                          # (f32 and samples/s are the defaults of a leg: said once, in the headline)
        c["dtype"] = dt
    c["value"] = _r(e.get("value"))
    if e.get("unit") != "samples/s":
        c["unit"] = e.get("unit")
    for k in ("ms_per_step", "ms_per_call_p50", "ms_per_call_mean", "steps", "calls", "engines"):
        if e.get(k) is not None:
            c[k] = _r(e[k], 4)
    if e.get("graphs"):
        c["captures"] = e["graphs"].get("captures_in_timed_calls")
    for k in ("threads_value", "racing_value", "batch_value", "batch_ms_p50"):
        if e.get(k) is not None:
            c[k] = _r(e[k], 4)
    if e.get("row_ms_p50"):
        c["row_ms_p50"] = [_r(v, 3) for v in e["row_ms_p50"]]
        c["row_ms_p95"] = [_r(v, 3) for v in e["row_ms_p95"]]
    if e.get("by_requests"):         # coalesced concurrent requests: samples/s at 2 / 4 / 8 requests in flight (target 250 M at 8)
        c["coalesced"] = {n: _r(v["group"]["value"], 4) for n, v in e["by_requests"].items() if v.get("group")}
    if e.get("streaming_samples_per_s") is not None:
        c["streaming_samples_per_s"] = _r(e["streaming_samples_per_s"])
    if e.get("frames_per_id") is not None:
        c["frames_per_id"] = _r(e["frames_per_id"], 3)
    if e.get("clock") and dt == "f32" and (e.get("ms_per_step") or 0) > 2.0:          # (the batched f32 legs: where the fraction is a throughput figure)
        c["sclk_mhz"] = _r(e["clock"]["sclk_mhz_median"], 4)
        c["clock_adjusted_frac"] = _r(e.get("clock_adjusted_frac"), 3)
    if roof:
        c["kernel"] = _short(roof.get("kernel"), 44)
        st = roof.get("step") or {}
        if dt in SPLIT_MODES:
            # priced against the split-operand peak of the instruction these legs run (2500 / 3 or 2500 / 6 TFLOP/s of
            # f32-equivalent FLOPs), and said so: a fraction of the f32 matrix peak would exceed 1 here
            c["frac"] = _r((st.get("achieved") or 0.0) / split_peak(dt), 3)
            c["peak_tflops"] = round(split_peak(dt))          # (the denominator of `frac`: 2500 / MFMAs per product)
        else:
            c["frac"] = _r(st.get("frac"), 3)
    return c


def compact_line(full, full_path=None):
    """The driver-facing line from the full result object; asserts nothing, never raises on missing optional parts."""
    cfgf = full.get("config") or {}
    out = {k: full.get(k) for k in ("metric", "unit", "n_gpus", "steps", "warmup", "higher_is_better", "scaling",
                                     "vs_baseline")}
    out["value"] = _r(full.get("value"), 7)
    out["x_realtime"] = _r(full.get("x_realtime"))
    out["ms_per_step"] = _r(full.get("ms_per_step"), 6)
    out["dtype"] = short_dtype(full.get("dtype", "f32"))
    out["data"] = "synthetic"
    out["config"] = {"workload": _short(cfgf.get("workload", ""), 280), "frames_per_step": cfgf.get("frames_per_step"),
                     "samples_per_step": cfgf.get("samples_per_step"),
                     "frames_per_id": _r(cfgf.get("frames_per_id"), 4),
                     "kernel_launches_per_step": cfgf.get("kernel_launches_per_step"),
                     "parallelism": _short(cfgf.get("parallelism", ""), 80)}
    su = full.get("sustained")
    if su:
        out["sustained"] = {"ms_per_step": _r(su.get("ms_per_step"), 5), "steps": su.get("steps"), "seconds": _r(su.get("seconds"), 3)}
    if full.get("ranks"):
        out["ranks"] = full["ranks"]
    if full.get("headline_note"):
        out["headline_note"] = full["headline_note"]
    if full.get("roofline"):
        out["roofline"] = compact_roofline(full["roofline"])
    cb = full.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {"value": _r(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"),
                               "kind": cb.get("kind"), "x_realtime": _r(cb.get("x_realtime"), 4),
                               "sample": _short(cb.get("sample", ""), 150)}
        for k in ("all_cores",):
            if cb.get(k):
                out["cpu_baseline"][k] = cb[k]
    api = full.get("api_inclusive")
    if api:
        out["api_inclusive"] = {"value": _r(api.get("value")), "ms_per_call": _r(api.get("ms_per_call"), 4)}
    if full.get("device_resident_ms") is not None:
        out["device_resident_ms"] = _r(full["device_resident_ms"], 5)
    if full.get("device_pipeline_only_ms_per_step") is not None:
        out["device_pipeline_only_ms_per_step"] = _r(full["device_pipeline_only_ms_per_step"], 5)
    if full.get("per_rank_samples_per_s") and (full.get("n_gpus") or 1) > 1:
        out["per_rank_samples_per_s"] = [_r(v, 4) for v in full["per_rank_samples_per_s"]]
    for k in ("single_gpu_reference", "batched_per_gpu"):
        v = full.get(k)
        if v:
            out[k] = {"value": _r(v.get("value")), "ms_per_step": _r(v.get("ms_per_step"), 5), "steps": v.get("steps")}
            if v.get("config"):
                out[k]["workload"] = _short(v["config"].get("workload", "").split(";")[0], 120)
            for kk in ("single_gpu_value", "speedup_over_single_gpu"):
                if v.get(kk) is not None:
                    out[k][kk] = _r(v[kk])
    if full.get("weight_broadcast_bytes"):
        out["weight_broadcast"] = {"bytes": full["weight_broadcast_bytes"], "s": _r(full.get("weight_broadcast_s"), 3)}
    if full.get("speculation"):
        out["speculation"] = full["speculation"]
    if full.get("extra_configs"):
        out["extra_configs"] = [compact_leg(e) for e in full["extra_configs"]]
    if full_path:
        out["full"] = full_path
    line = json.dumps(out, separators=(",", ":"))
    # belt and braces: slim the legs first (kernel names, then units / step counts), then shed the optional parts, until
    # the line fits
    for drop in (("steps", "calls", "captures"), ("kernel",), ("frames_per_id", "row_ms_p95", "threads_value", "racing_value", "sclk_mhz")):
        if len(line) <= COMPACT_LIMIT or not out.get("extra_configs"):
            break
        out["extra_configs"] = [{k: v for k, v in e.items() if k not in drop} for e in out["extra_configs"]]
        line = json.dumps(out, separators=(",", ":"))
    # (the per-rank device list and, last of all, batched_per_gpu -- the 1 -> N scaling of batched throughput north_star asks for)
    for k in ("extra_configs", "api_inclusive", "speculation", "sustained", "single_gpu_reference", "headline_note",
              "per_rank_samples_per_s", "ranks", "batched_per_gpu"):
        if len(line) <= COMPACT_LIMIT:
            break
        out.pop(k, None)
        line = json.dumps(out, separators=(",", ":"))
    return line


def emit(full):
    """Full object -> file; compact object -> the LAST line of stdout."""
    path = os.environ.get("PIPER_BENCH_FULL") or os.path.join(ROOT, "bench_full.json")
    try:
        with open(path, "w") as f:
            json.dump(full, f)
        shown = os.path.relpath(path, ROOT) if path.startswith(ROOT) else path
    except OSError:
        shown = None
    print(compact_line(full, shown), flush=True)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=None, choices=sorted(CONFIGS),
                    help="BASELINE.json configs[] entry (1-based) used as the headline: 2 medium B=1 (default at "
                         "--gpus 1), 3 high B=64, 4 medium 64 utterances per GPU (default at --gpus N>1), 5 streaming")
    ap.add_argument("--preset", default=None)
    ap.add_argument("--ids", type=int, default=None, help="phoneme ids per utterance")
    ap.add_argument("--batch", type=int, default=None, help="utterances per step (per GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the per-kernel event passes (profiling runs)")
    ap.add_argument("--no-extra", action="store_true", help="headline only: skip the extra_configs legs")
    ap.add_argument("--matrix", default="f32", choices=["f32"] + sorted(SPLIT_MODES),
                    help="matrix mode of the engine (PIPER_HIP_MATRIX): f32 = the reference's arithmetic (default, the "
                         "headline); bf16x3 / f16x3 / bf16x6 = opt-in split-operand conv GEMMs for flow + generator")
    ap.add_argument("--stream-latency", action="store_true",
                    help="BASELINE configs[4]: p50 time to the first chunk of a chunked (45-frame) decode, then exit")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--min-seconds", type=float, default=2.0,
                    help="after the K timed steps keep stepping (reported separately as `sustained`) until the GPU "
                         "has been busy this long, so that external samplers see the run")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def dbg(msg):
    """Progress markers on stderr (PIPER_BENCH_DEBUG=1): where a run that ends without its JSON line stopped."""
    if os.environ.get("PIPER_BENCH_DEBUG") == "1":
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


class Ctx:
    """Process-wide state of one bench run: rank / world, the torch.distributed handle (or None), device."""
    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        self.cdev = None
        self.dev_index = 0
        self.backend = os.environ.get("PIPER_BENCH_BACKEND", "nccl")   # gloo: single-GPU smoke test of the N-rank path

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def reduce(self, x, op):
        import torch
        if self.dist is None:
            return float(x)
        t = torch.tensor([x], dtype=torch.float64, device=self.cdev)
        self.dist.all_reduce(t, op=getattr(self.dist.ReduceOp, op))
        return float(t.item())

    def gather(self, x):
        import torch
        if self.dist is None:
            return [float(x)]
        pr = [torch.zeros(1, dtype=torch.float64, device=self.cdev) for _ in range(self.world)]
        self.dist.all_gather(pr, torch.tensor([x], dtype=torch.float64, device=self.cdev))
        return [float(v.item()) for v in pr]


class ClockSampler:
    """Shader clock of the device while a leg runs ("if the bound is the clock, log sclk from the same
    run"): a thread reads the amdgpu hwmon `freq1_input` of the device's PCI function every 10 ms (sysfs, no subprocess);
    `stats()` = median / min / max MHz over the samples taken under load, or None where sysfs does not show it. The f32
    matrix peak (157.3 TFLOP/s) is 256 CUs x 4 SIMDs x 64 FLOP/clk at 2.4 GHz: `clock_adjusted_frac` = frac x 2400 / sclk."""
    PEAK_MHZ = 2400.0

    def __init__(self, bdf):
        import glob
        self.paths = sorted(glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*/freq1_input")) if bdf else []
        self.samples, self._stop, self._th = [], False, None

    def __enter__(self):
        if self.paths:
            import threading

            def loop():
                while not self._stop:
                    try:
                        self.samples.append(int(open(self.paths[0]).read().strip()) / 1e6)
                    except (OSError, ValueError):
                        return
                    time.sleep(0.01)
            self._th = threading.Thread(target=loop, daemon=True)
            self._th.start()
        return self

    def __exit__(self, *a):
        self._stop = True
        if self._th:
            self._th.join(timeout=1.0)

    def stats(self):
        v = sorted(x for x in self.samples if x > 500.0)          # (idle samples before / after the loop do not count)
        if len(v) < 3:
            return None
        return {"sclk_mhz_median": v[len(v) // 2], "sclk_mhz_min": v[0], "sclk_mhz_max": v[-1], "samples": len(v)}


def rank_audit(ctx):
    """What makes an N > 1 record auditable: the process group's world size and backend, the RCCL version torch.distributed
    runs on, and every rank's device as its PCI bus id (through the C ABI: pe_device_pci_bus_id) -- N distinct ids = N
    distinct GPUs took part. Gathered on every rank (a collective), returned on all."""
    from piper_amd import _lib as L
    try:
        mine = L.device_pci_bus_id(ctx.dev_index)
    except Exception as ex:          # noqa: BLE001 -- a diagnostic must not take the run down
        mine = f"unknown ({type(ex).__name__})"
    if ctx.dist is None:
        return {"world_size": 1, "backend": None, "pci_bus_ids": [mine], "distinct_devices": 1}
    ids = [None] * ctx.world
    ctx.dist.all_gather_object(ids, mine)
    rccl = None
    try:
        import torch
        v = torch.cuda.nccl.version()
        rccl = ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
    except Exception:          # noqa: BLE001
        pass
    return {"world_size": int(ctx.dist.get_world_size()), "backend": str(ctx.dist.get_backend()), "rccl": rccl,
            "pci_bus_ids": ids, "distinct_devices": len(set(ids))}


def make_inputs(cfg, B, T, rank):
    """SURVEY.md section 8d synthetic inputs: fixed-length id sequences shaped like phonemizer output; the duration noise
    is fixed (resident), the prior noise is drawn on the device every step."""
    from piper_amd import weights as W
    id_max = min(cfg.n_vocab - 1, 129)
    id_lists = [W.synthetic_phoneme_ids(T, rank * B + i, id_max=id_max) for i in range(B)]
    rng = np.random.default_rng(1234 + rank)
    noise_w = rng.standard_normal((B, 2, T)).astype(np.float32)
    return id_lists, noise_w


def timed_leg(ctx, eng, cfg, preset, B, T, steps, warmup, sync_ranks=True, scales=SCALES, resident_steps=0):
    """W untimed + exactly K timed steps bracketed by barrier + device synchronisation; max over ranks. A step spans what
    the reference's inferSeconds spans (piper.cpp:385-395): ids from host memory (pe_upload), the device pipeline with
    both noise sites drawn by the engine (pe_run), int16 PCM in host memory (pe_fetch)."""
    import torch
    id_lists, noise_w = make_inputs(cfg, B, T, ctx.rank)
    eng.set_seed(1234 + ctx.rank)
    host_in = eng.pack_host(id_lists, scales)     # int64 ids + offsets in host memory, as the caller of piper::synthesize holds them

    def step():
        # the result views are used as the C ABI hands them out -- no Python-side copy of the samples inside the timed region.
        # Returns the samples this step produced: the engine draws fresh duration noise every step (models.py:111), so the
        # frame counts differ from step to step
        eng.upload_host(host_in)
        eng.run()
        return eng.fetch_views(False, True).sample_offsets[B]

    for _ in range(max(1, warmup)):          # (at least one untimed step: graph capture, and the frame counts below)
        step()
    res = eng.fetch(False, True)             # the last warm-up step's result as numpy copies, for the bookkeeping
    torch.cuda.synchronize()
    frames = res.frames
    samples_per_step = int(frames.sum()) * eng.hop
    launches = eng.run_launches
    if sync_ranks:
        ctx.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    total_samples = 0
    for _ in range(steps):
        total_samples += step()
    torch.cuda.synchronize()
    if sync_ranks:
        ctx.barrier()
    elapsed_local = time.perf_counter() - t0
    total_local = float(total_samples)
    if sync_ranks and ctx.dist is not None:
        elapsed = ctx.reduce(elapsed_local, "MAX")
        total = ctx.reduce(total_local, "SUM")
        per_rank = ctx.gather(total_local / elapsed_local)
    else:
        elapsed, total, per_rank = elapsed_local, total_local, [total_local / elapsed_local]
    # the same step with the ids left resident in HBM (what rounds 1-5 reported as the headline): upload once, then
    # pe_run + pe_fetch per step
    resident_ms = None
    if resident_steps > 0:
        eng.upload_host(host_in)
        eng.run(); eng.fetch_views(False, True)
        t1 = time.perf_counter()
        for _ in range(resident_steps):
            eng.run()
            eng.fetch_views(False, True)
        resident_ms = (time.perf_counter() - t1) / resident_steps * 1e3
    return {"value": total / elapsed, "ms_per_step": elapsed / steps * 1e3, "elapsed_local": elapsed_local,
            "frames": frames, "samples_per_step": samples_per_step, "launches": launches, "per_rank": per_rank,
            "id_lists": id_lists, "noise_w": noise_w, "step": step, "steps": steps, "warmup": warmup,
            "device_resident_ms": resident_ms,
            # frames per step averaged over the timed steps (every step draws its own duration noise)
            "avg_frames_per_step": total_local / max(steps, 1) / eng.hop}


def device_only_ms(eng, id_lists, noise_w, n, scales=SCALES):
    """Device pipeline only (no PCM delivery to the host), graphs replayed: the sum of the kernels' durations."""
    eng.upload(id_lists, scales)
    eng.run(); eng.fetch(False, False)
    t1 = time.perf_counter()
    for _ in range(n):
        eng.run()
    eng.fetch(False, False)
    return (time.perf_counter() - t1) / n * 1e3


def workload_text(cfgno, preset, cfg, B, T):
    return (f"BASELINE configs[{cfgno - 1}]: {preset} VITS voice ({cfg.sample_rate} Hz), {B} utterance(s) x {T} phoneme "
            f"ids per step per GPU, scales 0.667/1.0/0.8; step = inferSeconds span: host ids in (pe_upload) + pe_run, engine "
            f"draws both noise sites + int16 PCM to host (pe_fetch)")


def main():
    args = parse_args()
    ctx = Ctx()
    if args.gpus > 1 and "RANK" not in os.environ:
        self_launch(args)
    if args.gpus != ctx.world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={ctx.world}")
    # headline at EVERY N: configs[1] -- the configuration BASELINE.json's metric is quoted on -- one utterance per GPU
    # per step, so the per-N `value`s of the driver's N = 1, 2, 4, 8 runs are one weak-scaling curve over a fixed
    # per-GPU workload. The batched split north_star names (configs[3]: 64 utterances per GPU, 512 over 8) is timed in
    # the same invocation and reported beside it (`batched_per_gpu`, with rank 0's single-GPU rate on the same share).
    cfgno = args.config or 2
    preset, B, T = CONFIGS[cfgno]
    preset = args.preset or preset
    B = args.batch or B
    T = args.ids or T
    if cfgno == 5:
        args.stream_latency = True

    import torch
    from piper_amd import weights as W
    from piper_amd.engine import Engine

    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if ctx.world > ndev and ctx.backend == "nccl":
        raise SystemExit(f"--gpus {ctx.world} but only {ndev} GPU(s) visible")
    ctx.dev_index = ctx.local_rank % ndev
    torch.cuda.set_device(ctx.dev_index)
    ctx.cdev = torch.device("cuda", ctx.dev_index) if ctx.backend == "nccl" else torch.device("cpu")
    # a world of one still goes through the process group + RCCL when asked to (PIPER_BENCH_DIST=1): the only way to
    # exercise the nccl load path on a single-GPU box
    if ctx.world > 1 or os.environ.get("PIPER_BENCH_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if ctx.backend == "nccl":
            dist.init_process_group("nccl", rank=ctx.rank, world_size=ctx.world, device_id=ctx.cdev)
        else:
            dist.init_process_group(ctx.backend, rank=ctx.rank, world_size=ctx.world)
        assert dist.get_world_size() == args.gpus
        ctx.dist = dist
        dbg(f"process group up: backend {ctx.backend}, world {ctx.world}")

    ranks_info = rank_audit(ctx)
    # A scaling point is only a scaling point when every rank drove its OWN GPU: under RCCL a run whose ranks share
    # devices (a mis-set HIP_VISIBLE_DEVICES / LOCAL_RANK) fails here, on every rank, instead of printing a curve point.
    # (PIPER_BENCH_BACKEND=gloo is the single-GPU smoke test of the N-rank path and says so in `ranks.backend`.)
    if ctx.dist is not None and ctx.backend == "nccl" and ranks_info["distinct_devices"] != ranks_info["world_size"]:
        if ctx.rank == 0:
            print(f"bench.py: {ranks_info['world_size']} ranks on {ranks_info['distinct_devices']} distinct device(s) "
                  f"{ranks_info['pci_bus_ids']}: not a multi-GPU measurement", file=sys.stderr, flush=True)
        finish(ctx)
        raise SystemExit(3)

    os.environ["PIPER_HIP_MATRIX"] = args.matrix          # read once, at engine creation
    cfg = W.preset(preset)
    # ---- voice: rank 0 builds / parses / packs it; the others lay out an identical weight arena from the blob header and
    # receive the PACKED weights by one device-to-device broadcast into that arena (RCCL over xGMI)
    wts = W.synthetic_weights(cfg, 1234) if ctx.rank == 0 else None
    t_bcast, bcast_bytes = 0.0, 0
    if ctx.dist is not None:
        from piper_amd.dist import load_sharded
        eng, t_bcast, bcast_bytes = load_sharded(W.pack_blob(cfg, wts) if ctx.rank == 0 else None, 0, ctx.dev_index)
    else:
        eng = Engine(blob=W.pack_blob(cfg, wts), device=ctx.dev_index)
    dbg(f"engine ready (broadcast {bcast_bytes} bytes in {t_bcast:.3f} s)")

    if args.stream_latency:
        out = stream_latency(eng, cfg, preset, T, args.steps, args.warmup, ctx.rank)
        if ctx.rank == 0:
            print(json.dumps(out), flush=True)
        finish(ctx)
        return

    # ---- N>1: rank 0's rate on the SAME workload with the other GPUs idle, so that the scaling of `value` can be read
    # against a single-GPU number from the same invocation (the driver's N=1 run has the B=1 headline)
    single_ref = None
    if ctx.world > 1:
        if ctx.rank == 0:
            r1 = timed_leg(ctx, eng, cfg, preset, B, T, max(3, min(args.steps, 10)), 2, sync_ranks=False)
            single_ref = {"value": r1["value"], "ms_per_step": r1["ms_per_step"], "steps": r1["steps"],
                          "what": "rank 0 alone (other ranks waiting at a barrier), same workload, same build"}
        ctx.barrier()

    leg = timed_leg(ctx, eng, cfg, preset, B, T, args.steps, args.warmup, resident_steps=max(10, args.steps))
    dbg(f"timed leg done: {leg['ms_per_step']:.3f} ms/step")
    frames, id_lists, noise_w = leg["frames"], leg["id_lists"], leg["noise_w"]

    # ---- sustained: keep the GPU busy for >= --min-seconds in total (same step), reported separately
    sustained = None
    if leg["elapsed_local"] < args.min_seconds:
        per = leg["elapsed_local"] / args.steps
        n_more = int(min(200000, max(1, (args.min_seconds - leg["elapsed_local"]) / per)))
        t1 = time.perf_counter()
        for _ in range(n_more):
            leg["step"]()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        sustained = {"steps": n_more, "seconds": dt, "value": leg["samples_per_step"] * n_more / dt,
                     "ms_per_step": dt / n_more * 1e3}          # (value: at the warm-up step's frame count; ms is exact)

    # ---- the one-call form of the C ABI: the same span plus the float waveform on the host (Python list handling and
    # numpy copies of both outputs included)
    n_api = max(3, min(50, args.steps))
    eng.synthesize_batch(id_lists, SCALES)
    t1 = time.perf_counter()
    for _ in range(n_api):
        r_api = eng.synthesize_batch(id_lists, SCALES)
    dt_api = time.perf_counter() - t1
    api = {"value": sum(p.size for p in r_api.pcm) * n_api / dt_api, "unit": "samples/s", "calls": n_api,
           "ms_per_call": dt_api / n_api * 1e3,
           "what": "pe_synthesize_batch through the Python binding (ids H2D, device pipeline with engine-drawn noise, float + "
                   "int16 D2H, numpy copies of both)"}
    dev_ms = device_only_ms(eng, id_lists, noise_w, max(3, min(50, args.steps)))

    roof = None
    if ctx.rank == 0 and not args.no_roofline:
        roof = roofline(eng, preset, B, T, id_lists, noise_w, args.steps, leg["ms_per_step"], dev_ms)

    # ---- N>1: configs[3]'s per-GPU share (64 utterances x 128 ids on every GPU at once) beside the headline, after rank
    # 0's rate on that share with the other GPUs idle: the 1 -> N scaling of BATCHED throughput from one invocation
    batched_line = None
    if ctx.world > 1 and cfgno == 2 and not args.no_extra and not (args.preset or args.batch or args.ids):
        _, B4, T4 = CONFIGS[4]
        n4 = max(3, min(args.steps, 10))
        alone = None
        if ctx.rank == 0:
            alone = timed_leg(ctx, eng, cfg, preset, B4, T4, n4, 2, sync_ranks=False)
        ctx.barrier()
        l4 = timed_leg(ctx, eng, cfg, preset, B4, T4, n4, 2)
        batched_line = {"config": {"workload": workload_text(4, preset, cfg, B4, T4)}, "value": l4["value"],
                        "unit": "samples/s", "x_realtime": l4["value"] / cfg.sample_rate, "ms_per_step": l4["ms_per_step"],
                        "steps": l4["steps"], "per_rank_samples_per_s": l4["per_rank"]}
        if alone is not None:
            batched_line["single_gpu_value"] = alone["value"]
            batched_line["single_gpu_ms_per_step"] = alone["ms_per_step"]
            batched_line["speedup_over_single_gpu"] = l4["value"] / alone["value"]

    out = None
    if ctx.rank == 0:
        value = leg["value"]
        out = {
            "metric": "audio samples/sec",
            "value": value,
            "unit": "samples/s",
            "x_realtime": value / cfg.sample_rate,
            "n_gpus": ctx.world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": leg["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.matrix == "f32" else DTYPE_SPLIT[args.matrix],
            "data": "synthetic (seeded random-weight voice of the named architecture, synthetic phoneme ids)",
            "config": {"workload": workload_text(cfgno, preset, cfg, B, T),
                       "frames_per_step": int(frames.sum()), "samples_per_step": leg["samples_per_step"],
                       "frames_per_id": float(leg["avg_frames_per_step"]) / float(B * T),
                       "kernel_launches_per_step": leg["launches"],
                       "parallelism": f"utterance-parallel x{ctx.world}, one process per GPU, RCCL weight broadcast"},
            "ranks": ranks_info,
            "api_inclusive": api,
            "device_resident_ms": leg["device_resident_ms"],
            "device_pipeline_only_ms_per_step": dev_ms,
            "sustained": sustained,
            "per_rank_samples_per_s": leg["per_rank"],
            "weight_broadcast_s": t_bcast,
            "weight_broadcast_bytes": bcast_bytes,
            "speculation": dict(zip(("runs", "misses"), eng.speculation_stats)),
            "xcd_dispatch": dict(zip(("xcc_of_workgroups_0_63", "round_robin_period"), eng.xcc_pattern)),
        }
        if single_ref is not None:
            out["single_gpu_reference"] = single_ref
        if batched_line is not None:
            out["batched_per_gpu"] = batched_line
        if ctx.world > 1 and cfgno == 2:
            out["headline_note"] = ("N>1 `value` = configs[1] per GPU (1 utterance x 128 ids per rank per step: one weak-scaling "
                                    "curve with the N=1 line; rounds 1-3 used configs[3] here). The batched 1->N scaling "
                                    "north_star asks for is `batched_per_gpu` (64 utterances per GPU): value / single_gpu_value")
        if roof is not None:
            out["roofline"] = roof
        if ctx.world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, wts, id_lists[0], SCALES, noise_w[0], args.cpu_seconds, preset)

    # ---- the other BASELINE configurations, time-boxed, in the same invocation (single GPU, default headline only)
    if ctx.rank == 0 and ctx.world == 1 and cfgno == 2 and args.config is None and not args.no_extra and \
            not (args.preset or args.batch or args.ids):
        out["extra_configs"] = extra_configs(ctx, eng, cfg, args)
    if ctx.rank == 0:
        dbg("printing the result line")
        emit(out)
    finish(ctx)
    dbg("done")


def finish(ctx):
    if ctx.dist is not None:
        ctx.dist.barrier()
        ctx.dist.destroy_process_group()


def extra_configs(ctx, eng_medium, cfg_medium, args):
    """configs[2], configs[3]'s per-GPU share, configs[4] and a changing-input B=1 leg, each a short timed region
    (warm-up, then K steps between device synchronisations) with its own per-kernel roofline. Failures of one leg are
    reported in its entry, never raised: the headline line must come out."""
    from piper_amd import weights as W
    from piper_amd.engine import Engine
    from piper_amd import _lib as L
    legs = []
    try:
        bdf = L.device_pci_bus_id(ctx.dev_index).lower()
    except Exception:          # noqa: BLE001
        bdf = None

    def batched(cfgno, eng, cfg, preset, B, T, steps, warmup, dtype="f32", scales=SCALES, what=None):
        with ClockSampler(bdf) as clk:
            l = timed_leg(ctx, eng, cfg, preset, B, T, steps, warmup, sync_ranks=False, scales=scales)
        dev_ms = device_only_ms(eng, l["id_lists"], l["noise_w"], max(2, min(5, steps)), scales=scales)
        e = {"config": {"workload": what or workload_text(cfgno, preset, cfg, B, T), "frames_per_step": int(l["frames"].sum()),
                        "samples_per_step": l["samples_per_step"], "kernel_launches_per_step": l["launches"]},
             "metric": "audio samples/sec", "value": l["value"], "unit": "samples/s", "dtype": dtype,
             "x_realtime": l["value"] / cfg.sample_rate, "ms_per_step": l["ms_per_step"], "steps": steps,
             "warmup": warmup, "device_pipeline_only_ms_per_step": dev_ms,
             "frames_per_id": float(l["avg_frames_per_step"]) / float(B * T)}
        if not args.no_roofline:
            e["roofline"] = roofline(eng, preset, B, T, l["id_lists"], l["noise_w"], 3, l["ms_per_step"], dev_ms, scales=scales)
        ck = clk.stats()
        if ck:
            e["clock"] = ck
            fr = ((e.get("roofline") or {}).get("step") or {}).get("frac")
            if fr:
                e["clock_adjusted_frac"] = fr * ClockSampler.PEAK_MHZ / ck["sclk_mhz_median"]
        return e

    def survey_shape():
        """configs[1] at SURVEY.md section 8(d)'s stated shape -- about 2.7 frames per id (F ~ 330-350 for 128 ids): the
        synthetic voice's duration predictor gives 3.26 at length_scale 1, which dilutes the per-id front end with ~20 %
        more samples than the stated shape; here length_scale is set so that the utterance lands at 2.7 +- 0.2."""
        id_lists, noise_w = make_inputs(cfg_medium, 1, 128, ctx.rank)
        ls, fpi = 1.0, None
        for _ in range(5):         # (the engine draws the duration noise: the mean of 8 calls per trial, like the timed steps will)
            fr = [float(eng_medium.synthesize_batch(id_lists, (SCALES[0], ls, SCALES[2])).frames.sum()) for _ in range(8)]
            fpi = sum(fr) / len(fr) / 128.0
            if abs(fpi - 2.7) <= 0.05:
                break
            ls *= 2.7 / fpi
        sc = (SCALES[0], float(f"{ls:.3f}"), SCALES[2])
        e = batched(2, eng_medium, cfg_medium, "medium", 1, 128, 200, 10, scales=sc,
                    what=f"configs[1] at SURVEY 8(d)'s shape: medium voice, 1 utterance x 128 ids, length_scale {sc[1]} "
                         f"(-> ~2.7 frames per id on average; the headline runs the scales 0.667/1.0/0.8 = ~3.1 frames per id)")
        e["length_scale"] = sc[1]
        return e

    def guarded(name, fn):
        t0 = time.perf_counter()
        try:
            e = fn()
        except Exception as ex:          # noqa: BLE001 -- one leg must not take the headline down
            e = {"error": f"{type(ex).__name__}: {ex}"}
        e["leg"] = name
        e["leg_seconds"] = time.perf_counter() - t0
        legs.append(e)

    # configs[3] per-GPU share: the medium engine is already there
    guarded("configs[1] at 2.7 frames/id", survey_shape)
    guarded("configs[3] per-GPU share", lambda: batched(4, eng_medium, cfg_medium, "medium", 64, 128, 10, 3))
    guarded("changing inputs, B=1", lambda: varied_inputs(eng_medium, cfg_medium))
    guarded("en-us rows, sequential B=1", lambda: survey_rows(eng_medium, cfg_medium))
    guarded("concurrent requests, coalesced", lambda: concurrent_streams(ctx, cfg_medium))
    # configs[2] + configs[4]: the high-quality architecture (ResBlock1, four upsampling stages)
    hi = {}

    def high_single():
        # the high-quality architecture's single utterance (en_US-lessac-high / -ryan-high shape: 264 GFLOP per step, the one
        # B=1 workload that is matrix-bound) -- on a fresh engine, before the batched leg grows its workspaces
        hi["cfg"] = W.preset("high")
        hi["eng"] = Engine(blob=W.pack_blob(hi["cfg"], W.synthetic_weights(hi["cfg"], 1234)), device=ctx.dev_index)
        return batched(5, hi["eng"], hi["cfg"], "high", 1, 128, 50, 5,
                       what="high VITS voice (configs[2] / configs[4] architecture), 1 utterance x 128 ids per step, scales "
                            "0.667/1.0/0.8; step = pe_run + int16 PCM to host")

    def high_batched():
        if "eng" not in hi:
            hi["cfg"] = W.preset("high")
            hi["eng"] = Engine(blob=W.pack_blob(hi["cfg"], W.synthetic_weights(hi["cfg"], 1234)), device=ctx.dev_index)
        return batched(3, hi["eng"], hi["cfg"], "high", 64, 128, 5, 2)

    guarded("high voice, B=1", high_single)
    guarded("configs[2]", high_batched)
    if "eng" in hi:
        guarded("configs[4]", lambda: stream_latency(hi["eng"], hi["cfg"], "high", 128, 100, 5, 0))
        hi["eng"].close()

    # ---- the same two throughput configurations in the opt-in split-operand matrix modes (separately labelled dtype and
    # peak; the headline and the legs above stay f32). Parity gate of the modes = the f32 path's own (durations equal, max
    # |d audio| < 2e-4 on all 64 utterances of both configurations): tests/test_gpu_batched.py
    # test_split_matrix_modes_at_baseline_sizes.
    def split_leg(mode, cfgno, preset, steps, warmup, B=64):
        os.environ["PIPER_HIP_MATRIX"] = mode
        try:
            c = W.preset(preset)
            e = Engine(blob=W.pack_blob(c, W.synthetic_weights(c, 1234)), device=ctx.dev_index)
        finally:
            os.environ["PIPER_HIP_MATRIX"] = "f32"
        try:
            return batched(cfgno, e, c, preset, B, 128, steps, warmup, dtype=DTYPE_SPLIT[mode])
        finally:
            e.close()

    # the headline's own workload (configs[1]: one utterance) in the near-exact mode f16x3: reported as a leg, never as `value`
    guarded("configs[1], f16x3", lambda: split_leg("f16x3", 2, "medium", 200, 10, B=1))

    # (bf16x3 -- rounds 3-5's mode, 16 significand bits -- costs what f16x3 costs and is 8x less accurate: `--matrix bf16x3`
    # still times it, the default line carries the two near-exact modes)
    for mode in ("f16x3", "bf16x6"):
        guarded(f"configs[3] share, {mode}", lambda m=mode: split_leg(m, 4, "medium", 10, 3))
        guarded(f"configs[2], {mode}", lambda m=mode: split_leg(m, 3, "high", 5, 2))
    return legs


def concurrent_streams(ctx, cfg, counts=(2, 4, 8), T=128, calls=60):
    """Several single-utterance requests pending at once on ONE GPU (a server's situation; north_star's "per-GPU
    independent streams"). A B=1 pipeline leaves most of the chip idle, but N engines racing for the HIP runtime's launch
    path reach only ~1.5x of one (`racing`: PIPER_HIP_GROUP_COALESCE=0, N engines x one utterance each -- what round 4
    reported). The product COALESCES instead: pe_group_* gives a device's small share to ONE of its engines as a
    batched call (`group`), and pe_coalescer_* does the same for independent caller threads (`threads`: N Python
    threads, each with its own utterance, one engine). Whole C-ABI calls with host inputs and outputs, engine-drawn noise;
    p50 = time from a request to its PCM."""
    import threading
    from piper_amd import weights as W
    from piper_amd.engine import Engine
    from piper_amd.group import Coalescer, EngineGroup
    blob = W.pack_blob(cfg, W.synthetic_weights(cfg, 1234))
    id_max = min(cfg.n_vocab - 1, 129)
    out = {"config": {"workload": f"medium VITS voice, N concurrent single-utterance requests x {T} ids on ONE GPU, coalesced "
                                  "into batched engine calls (pe_group_* / pe_coalescer_*); host inputs and outputs"},
           "metric": "audio samples/sec", "unit": "samples/s", "dtype": "f32", "by_requests": {}}

    def group_leg(n, coalesce):
        os.environ["PIPER_HIP_GROUP_COALESCE"] = "1" if coalesce else "0"
        try:
            grp = EngineGroup(blob, [ctx.dev_index] * n)
        finally:
            os.environ.pop("PIPER_HIP_GROUP_COALESCE", None)
        try:
            grp.set_seed(1234)
            texts = [W.synthetic_phoneme_ids(T, 1234 + i, id_max=id_max) for i in range(n)]
            for _ in range(5):
                grp.synthesize_batch(texts, SCALES)
            samples, ms = 0, []
            t0 = time.perf_counter()
            for _ in range(calls):
                t1 = time.perf_counter()
                r = grp.synthesize_batch(texts, SCALES)
                ms.append((time.perf_counter() - t1) * 1e3)
                samples += sum(p.size for p in r.pcm)
            tot = time.perf_counter() - t0
            used = len(set(grp.assignment(n)))
        finally:
            grp.close()
        ms.sort()
        return {"value": samples / tot, "ms_per_call_p50": ms[len(ms) // 2], "engines_used": used}

    def thread_leg(n):
        eng = Engine(blob=blob, device=ctx.dev_index)
        co = Coalescer(eng, max_batch=8, max_wait_us=150)
        try:
            eng.set_seed(77)
            texts = [W.synthetic_phoneme_ids(T, 1234 + i, id_max=id_max) for i in range(n)]
            lat, samples = [[] for _ in range(n)], [0] * n
            go = threading.Barrier(n + 1)

            def worker(i):
                for _ in range(5):
                    co.synthesize(texts[i], SCALES)
                go.wait()
                for _ in range(calls):
                    t1 = time.perf_counter()
                    pcm, _, _, _ = co.synthesize(texts[i], SCALES)
                    lat[i].append((time.perf_counter() - t1) * 1e3)
                    samples[i] += pcm.size

            th = [threading.Thread(target=worker, args=(i,)) for i in range(n)]
            [t.start() for t in th]
            go.wait()
            t0 = time.perf_counter()
            [t.join() for t in th]
            tot = time.perf_counter() - t0
            c, q = co.stats
        finally:
            co.close()
            eng.close()
        allms = sorted(x for l in lat for x in l)
        return {"value": sum(samples) / tot, "ms_per_call_p50": allms[len(allms) // 2], "requests_per_engine_call": q / max(c, 1)}

    best = None
    for n in counts:
        e = {"group": group_leg(n, True), "threads": thread_leg(n)}
        if n == counts[-1]:
            e["racing"] = group_leg(n, False)
        out["by_requests"][str(n)] = e
        if best is None or e["group"]["value"] > best[1]["group"]["value"]:
            best = (n, e)
    out["engines"] = best[0]
    out["value"] = best[1]["group"]["value"]
    out["x_realtime"] = out["value"] / cfg.sample_rate
    out["ms_per_call_p50"] = best[1]["group"]["ms_per_call_p50"]
    out["calls"] = calls
    out["threads_value"] = best[1]["threads"]["value"]
    out["racing_value"] = (out["by_requests"][str(counts[-1])].get("racing") or {}).get("value")
    return out


def survey_rows(eng, cfg, reps=20):
    """SURVEY.md section 8(d) Config 2's real inputs: the 7 rows of the reference's etc/test_sentences/test_en-us.jsonl
    (113-381 phoneme ids; tests/golden/phoneme_ids_en-us.json holds their `phoneme_ids`) on the medium architecture.
    `sequential` = one call per row, one after the other, the way piper.cpp:549-582 walks the phrases of a text (host ids
    in, engine-drawn noise, int16 PCM out; per-row p50 / p95 over `reps` passes); `batch` = the 7 rows as ONE call (what
    textToWavFile does with a whole text here)."""
    with open(os.path.join(ROOT, "tests", "golden", "phoneme_ids_en-us.json")) as f:
        rows = [r["phoneme_ids"] for r in json.load(f)["rows"]]
    eng.set_seed(4242)
    eng.warmup(max_batch=len(rows), max_ids=max(len(r) for r in rows), frames_per_id=0.0, scales=SCALES, sample_ids=rows[0])
    packed = [eng.pack_host([r], SCALES) for r in rows]
    allp = eng.pack_host(rows, SCALES)

    def call(pk, B):
        eng.upload_host(pk)
        eng.run()
        return eng.fetch_views(False, True).sample_offsets[B]

    for pk in packed:                       # graphs of every row's bucket, speculation ratio settled
        for _ in range(3):
            call(pk, 1)
    per_row = [[] for _ in rows]
    samples, t_all = 0, 0.0
    for _ in range(reps):
        for i, pk in enumerate(packed):
            t0 = time.perf_counter()
            samples += call(pk, 1)
            dt = time.perf_counter() - t0
            per_row[i].append(dt * 1e3)
            t_all += dt
    for _ in range(3):
        call(allp, len(rows))
    bms, bsamples = [], 0
    for _ in range(reps):
        t0 = time.perf_counter()
        bsamples += call(allp, len(rows))
        bms.append((time.perf_counter() - t0) * 1e3)

    def pct(v, q):
        v = sorted(v)
        return v[min(len(v) - 1, int(len(v) * q))]

    table = [{"ids": len(r), "ms_p50": pct(m, 0.5), "ms_p95": pct(m, 0.95)} for r, m in zip(rows, per_row)]
    return {"config": {"workload": "medium VITS voice, the 7 rows of the reference's test_en-us.jsonl (113-381 ids): sequential B=1 calls "
                                   "like piper.cpp:549-582, and the 7 rows as one batched call; host ids in, engine-drawn noise, int16 out"},
            "metric": "audio samples/sec", "unit": "samples/s", "dtype": "f32", "value": samples / t_all,
            "x_realtime": samples / t_all / cfg.sample_rate, "calls": reps * len(rows), "rows": table,
            "ms_per_call_p50": pct([x for m in per_row for x in m], 0.5),
            "row_ms_p50": [round(t["ms_p50"], 3) for t in table], "row_ms_p95": [round(t["ms_p95"], 3) for t in table],
            "batch_value": bsamples / (sum(bms) * 1e-3), "batch_ms_p50": pct(bms, 0.5)}


def varied_inputs(eng, cfg, n=64):
    """What real use looks like at batch 1: another text and fresh duration noise on every call, so the frame
    count changes from call to call and the speculative sizing of stage B can miss. Whole calls with host inputs and
    outputs as `piper::synthesize` makes them: pe_upload, pe_run, pe_fetch of the int16 PCM."""
    from piper_amd import weights as W
    id_max = min(cfg.n_vocab - 1, 129)
    rng = np.random.default_rng(4321)
    texts = [W.synthetic_phoneme_ids(int(rng.integers(60, 200)), 1000 + i, id_max=id_max) for i in range(n)]
    eng.set_seed(99)
    # what a server does once after loading the voice (include/piper_hip.h: pe_warmup): workspaces sized for its longest
    # text, the single-utterance graphs of every id bucket captured from one representative utterance
    t_w = time.perf_counter()
    eng.warmup(max_batch=1, max_ids=256, frames_per_id=0.0, scales=SCALES, sample_ids=texts[0])
    warm_s = time.perf_counter() - t_w
    def call(t):                                          # the C-ABI calls piper::synthesize makes (piper_shim.cpp:355-358):
        eng.upload([t], SCALES)                           # ids H2D, the engine's own noise: durations differ every call
        eng.run()
        return eng.fetch(False, True)                     # the int16 PCM the reference's synthesize() returns, on the host
    for t in texts[:4]:
        call(t)
    r0, m0 = eng.speculation_stats
    g0 = eng.graph_stats
    ms, samples = [], 0
    for t in texts:
        t0 = time.perf_counter()
        r = call(t)
        ms.append((time.perf_counter() - t0) * 1e3)
        samples += r.pcm[0].size
    r1, m1 = eng.speculation_stats
    g1 = eng.graph_stats
    tot = sum(ms) * 1e-3
    ms.sort()
    return {"config": {"workload": f"medium VITS voice, {n} calls of ONE utterance each (pe_upload + pe_run + pe_fetch of the int16 PCM, "
                                   "as piper::synthesize makes them), 60..200 ids, another text and fresh duration + prior noise every call, after one pe_warmup"},
            "metric": "audio samples/sec", "value": samples / tot, "unit": "samples/s", "dtype": "f32",
            "x_realtime": samples / tot / cfg.sample_rate, "ms_per_call_p50": ms[len(ms) // 2],
            "ms_per_call_mean": tot / n * 1e3, "ms_per_call_max": ms[-1], "calls": n,
            "graphs": {"warmup_s": warm_s, "captures_in_warmup": g0[1], "captures_in_timed_calls": g1[1] - g0[1],
                       "cached": g1[0]},
            "speculation": {"runs": r1 - r0, "misses": m1 - m0,
                            "what": "calls whose vocoder half was enqueued for a guessed frame bucket / guesses that "
                                    "were too small and cost a second pass (include/piper_hip.h: pe_speculation_stats)"}}


def roofline(eng, preset, B, T, id_lists, noise_w, steps, ms_per_step, dev_ms, scales=SCALES):
    """HIP events on the engine's stream (pe_profile_enable): one pass with a pair per pipeline stage, one pass with
    a pair around every conv / attention / layer-norm / fused-stage launch. `kernel` is the kernel with the largest
    share of device time, whatever its bound; `achieved` its algorithmic FLOPs (2 * rows * Cin * taps per output
    column) over its summed launch durations. `step` prices the whole step the same way.

    An event pair brackets a little more than the kernel (the two timestamp writes and the dispatch between them): the
    excess is calibrated per run as (sum of all event-pair durations - the replayed pipeline's own duration) / launches --
    the replayed graph runs the same kernels back to back with ~0 gaps (profiles/r02_trace_gaps_b1.txt) -- and
    subtracted, so that `avg_launch_us` agrees with rocprofv3's kernel durations (profiles/*_kernel_stats.csv); the raw
    figure stays beside it."""
    eng.upload(id_lists, scales)                  # both noise sites drawn by the engine: the product path's launch count
    nprof = max(3, min(10, steps))
    eng.profile_enable(1)
    eng.profile_reset()
    for _ in range(nprof):
        eng.run()
    eng.fetch(False, False)
    rows = eng.profile()[:5]
    stage_ms = {r["name"]: r["ms"] / nprof for r in rows}
    stage_tf = {r["name"]: (r["flops"] / (r["ms"] * 1e-3) / 1e12 if r["ms"] > 0 else 0.0) for r in rows}
    step_flops = sum(r["flops"] for r in rows) / nprof
    eng.profile_enable(2)
    eng.profile_reset()
    for _ in range(nprof):
        eng.run()
    eng.fetch(False, False)
    allrows = eng.profile()
    krows = [r for r in allrows[5:] if r["launches"]]
    eng.profile_enable(0)
    n_launch = sum(r["launches"] for r in krows) / nprof
    ev_sum_ms = sum(r["ms"] for r in krows) / nprof
    # launches without an event pair (element-wise glue: embed, randn, regulate, durations, post, pcm) are inside
    # dev_ms but not in ev_sum_ms: price them at the profiled stage-level remainder so the calibration is not biased
    ov_us = max(0.0, (ev_sum_ms - dev_ms) / max(n_launch, 1.0) * 1e3)
    ov_us = min(ov_us, 4.0)                       # an event pair cannot cost more than a few microseconds
    kernels = {}
    for r in krows:
        raw_us = r["ms"] / r["launches"] * 1e3
        us = max(raw_us - ov_us, 0.25 * raw_us)
        tf = r["flops"] / r["launches"] / (us * 1e-6) / 1e12
        kernels[r["name"]] = {"ms_per_step": us * 1e-3 * r["launches"] / nprof, "launches_per_step": r["launches"] / nprof,
                              "avg_launch_us": us, "avg_launch_us_event_pair": raw_us, "tflops": tf,
                              "frac_of_mfma_peak": tf / kernel_peak(r["name"]), "peak": kernel_peak(r["name"]),
                              "algorithmic_gflop_per_launch": r["flops"] / r["launches"] / 1e9,
                              "algorithmic_bytes_per_launch": (r["bytes"] / r["launches"] if r.get("bytes") else None),
                              "algorithmic_gb_per_s": (r["bytes"] / r["launches"] / (us * 1e-6) / 1e9 if r.get("bytes") else None)}
    ksum = sum(k["ms_per_step"] for k in kernels.values())
    top = max(kernels, key=lambda n: kernels[n]["ms_per_step"])
    k = kernels[top]
    traffic = pmc_traffic(preset, B, T, top)
    if traffic:
        traffic["algorithmic_bytes_per_launch"] = k["algorithmic_bytes_per_launch"]
    # the dominant kernel on rocprofv3's clock (committed summary of the same command): `frac` is computed on the slower
    # of the two clocks, both durations are printed
    rp = rocprof_kernel_us(preset, B, top)
    us_ev = k["avg_launch_us"]
    us_frac = max(us_ev, rp["avg_us"]) if rp else us_ev
    clock = "rocprofv3" if (rp and rp["avg_us"] >= us_ev) else "hip-event-pairs"
    k_tf = k["algorithmic_gflop_per_launch"] * 1e9 / (us_frac * 1e-6) / 1e12
    k_gbs = (k["algorithmic_bytes_per_launch"] / (us_frac * 1e-6) / 1e9) if k["algorithmic_bytes_per_launch"] else 0.0
    # the family view: all split-K launches / all tiled-GEMM launches together
    fam = {}
    for name, kk in kernels.items():
        d = fam.setdefault(name.split("<")[0], {"ms": 0.0, "flops": 0.0, "launches": 0.0})
        d["ms"] += kk["ms_per_step"]
        d["flops"] += kk["algorithmic_gflop_per_launch"] * 1e9 * kk["launches_per_step"]
        d["launches"] += kk["launches_per_step"]
    families = {f: {"share_of_profiled_kernel_time": d["ms"] / ksum, "launches_per_step": d["launches"],
                    "tflops": d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0} for f, d in fam.items()}
    step_tf = step_flops / (ms_per_step * 1e-3) / 1e12
    # the bound of the dominant kernel = the roof it sits closer to (algorithmic bytes over HBM peak vs algorithmic FLOPs
    # over the matrix peak of its instruction)
    hbm_frac = k_gbs / HBM_PEAK_GBPS
    mfma_frac = k_tf / kernel_peak(top)
    hbm_bound = hbm_frac > mfma_frac
    return {"bound": "hbm" if hbm_bound else "mfma", "kernel": top,
            "share_of_profiled_kernel_time": k["ms_per_step"] / ksum if ksum else 0.0,
            "achieved": k_gbs if hbm_bound else k_tf,
            "peak": HBM_PEAK_GBPS if hbm_bound else kernel_peak(top), "unit": "GB/s" if hbm_bound else "TFLOP/s",
            "frac": hbm_frac if hbm_bound else mfma_frac, "frac_mfma": mfma_frac, "frac_hbm": hbm_frac, "traffic": traffic,
            "avg_launch_us": us_frac, "avg_launch_us_event_pairs": us_ev, "avg_launch_us_rocprof": (rp or {}).get("avg_us"),
            "rocprof_source": (rp or {}).get("source"), "clock": clock,
            "launches_per_step": k["launches_per_step"],
            "event_pair_overhead_us": ov_us,
            "timing": "HIP event pairs on the engine's stream around every launch, minus the calibrated per-pair overhead "
                      "(sum of pairs - replayed pipeline time) / launches",
            "step": {"algorithmic_gflop": step_flops / 1e9, "achieved": step_tf, "frac": step_tf / FP32_MATRIX_PEAK_TFLOPS,
                     "what": "all algorithmic FLOPs of one step over the timed ms_per_step, against the f32 matrix peak "
                             "(157.3 TFLOP/s) in every matrix mode: > 1 is possible in mode bf16x3"},
            "families": families, "kernels": kernels, "stage_ms": stage_ms, "stage_tflops": stage_tf}


def rocprof_kernel_us(preset, B, kernel):
    """Average launch duration of `kernel` on rocprofv3's clock, from the newest committed `rocprofv3 --kernel-trace --stats`
    summary of this workload's command (profiles/rNN_{b1,b64,high_b1,high_b64}_kernel_stats.csv; scripts/collect_r06.sh
    writes them from the tree they are committed with). The in-process figure is HIP event pairs minus a calibrated pair
    overhead; for most kernels the two agree within 5 %, where they do not the line prices the kernel on the SLOWER clock."""
    import csv
    import glob
    import re
    if preset not in ("medium", "high") or B not in (1, 64):
        return None
    tag = ("high_" if preset == "high" else "") + ("b1" if B == 1 else "b64")
    want = kernel.replace(" ", "")
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{tag}_kernel_stats.csv")), reverse=True):
        try:
            for row in csv.DictReader(open(f)):
                name = re.sub(r"\(.*$", "", row["Name"]).replace("void ", "").replace("pe::", "").replace(" ", "")
                if name == want:
                    return {"avg_us": float(row["AverageNs"]) / 1e3, "calls": int(row["Calls"]), "source": os.path.basename(f)}
        except (OSError, ValueError, KeyError):
            continue
    return None


def pmc_traffic(preset, B, T, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command
    (profiles/*_pmc_traffic.json, written by scripts/pmc_traffic.py from separate --pmc FETCH_SIZE and
    --pmc WRITE_SIZE runs; (2*FETCH_SIZE + WRITE_SIZE) KiB per the gfx950 note in MI355X_MICROARCH.md).
    Counters cannot be read from inside the timed process, so this is null for a workload that has no
    committed pass."""
    import glob
    key = f"{preset}/b{B}/t{T}"
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
            k = d.get(key, {}).get("kernels", {}).get(kernel.replace(" ", ""))
            if k:
                return {"hbm_bytes_per_launch": k["hbm_bytes_per_launch"], "unit": "B",
                        "algorithmic_bytes_per_launch": k.get("algorithmic_bytes_per_launch"), "source": os.path.basename(f)}
        except (OSError, ValueError):
            continue
    return None


def stream_latency(eng, cfg, preset, T, steps, warmup, rank):
    """Time from the request to the first 45-frame chunk of PCM (encoder + durations + flow + one
    exact-halo vocoder window), like the "Latency" the reference's streaming script logs
    (infer_onnx_streaming.py:118-121), over >= 100 requests; plus the whole-utterance streaming rate."""
    from piper_amd import weights as W
    ids = W.synthetic_phoneme_ids(T, rank, id_max=min(cfg.n_vocab - 1, 129))
    first, total, samples = [], [], 0
    n = max(100, steps)
    for i in range(n + warmup):
        t0 = time.perf_counter()
        it = eng.stream(ids, SCALES, chunk_frames=45)
        a, _ = next(it)
        t1 = time.perf_counter()
        cnt = a.size + sum(c[0].size for c in it)
        t2 = time.perf_counter()
        if i >= warmup:
            first.append((t1 - t0) * 1e3)
            total.append((t2 - t0) * 1e3)
            samples = cnt
    first.sort()
    total.sort()
    return {"metric": "p50 first-chunk latency", "value": first[len(first) // 2], "unit": "ms", "higher_is_better": False,
            "p95_ms": first[int(len(first) * 0.95)], "n_gpus": 1, "steps": n, "warmup": warmup, "dtype": "f32",
            "data": "synthetic", "vs_baseline": None,
            "config": {"workload": f"BASELINE configs[4]: {preset} VITS voice, streaming decode, one {T}-id utterance, "
                                   f"45-frame chunks, halo {eng.stream_halo} frames, {eng.stream_frames} frames total"},
            "utterance_ms_p50": total[len(total) // 2],
            "streaming_samples_per_s": samples / (total[len(total) // 2] * 1e-3)}


def cpu_baseline(cfg, wts, ids, scales, noise_w, budget_s, preset="medium"):
    """CPU baseline on this box's host cores over a bounded sample: repeated B=1 synthesis of the same utterance,
    like piper.cpp's sequential loop. Preferred: onnxruntime's CPU EP with the reference's
    session options (piper.cpp:282-290, benchmark_onnx.py:39-53) -- probed here; it needs both the onnxruntime
    package and an exporter for this voice (torch.onnx + the reference's model code live only in the build
    container), so on a box without them the fallback is the oracle: a torch-CPU port of the reference graph,
    bit-identical to the reference's PyTorch module on the goldens (`kind: "port"` -- NOT the reference's ORT path)."""
    import torch
    from oracle import vits_oracle as O
    ncpu = os.cpu_count() or 1
    ort_probe = "not importable"
    try:
        import onnxruntime  # noqa: F401
        ort_probe = ("importable, but no .onnx of this synthetic voice can be produced on this box (the exporter "
                     "needs the reference's model code): torch port timed instead")
    except Exception:
        pass
    wt = O.to_torch(wts)
    rng = np.random.default_rng(99)
    nz = rng.standard_normal((cfg.inter, 16 * len(ids) + 64)).astype(np.float32)
    # torch's intra-op pool degrades badly when oversubscribed on many-core hosts: pick the thread
    # count that synthesizes this utterance fastest (bounded probe), then time the sample with it
    best, cores = None, 1
    for th in sorted({1, 8, 16, 32, min(64, ncpu)}):
        if th > ncpu:
            continue
        torch.set_num_threads(th)
        O.synthesize(wt, cfg, ids, scales, noise_w, nz)       # warm-up at this setting
        t = time.perf_counter()
        O.synthesize(wt, cfg, ids, scales, noise_w, nz)
        t = time.perf_counter() - t
        if best is None or t < best:
            best, cores = t, th
        if t > 20:
            break
    torch.set_num_threads(cores)
    n, samples, t0 = 0, 0, time.perf_counter()
    while True:
        r = O.synthesize(wt, cfg, ids, scales, noise_w, nz)
        samples += r["audio"].size
        n += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or n >= 200:
            break
    single = {"value": samples / dt, "cores": cores, "x_realtime": samples / dt / cfg.sample_rate}
    # ---- all host cores: P independent copies of that sequential program side by side (P * threads <= cores), the way
    # a CPU server would use the box; this is `value` when it beats the single program (it does on a 256-core host)
    ucores = usable_cores()
    allc = cpu_all_cores(preset, len(ids), ucores, budget_s)
    single["usable_cores"] = ucores
    if allc and allc["value"] > single["value"]:
        return {"value": allc["value"], "unit": "samples/s", "cores": allc["cores"], "kind": "port",
                "onnxruntime_probe": ort_probe, "x_realtime": allc["value"] / cfg.sample_rate,
                "single_program": single, "all_cores": {k: allc[k] for k in ("processes", "threads_each", "syntheses")},
                "note": "torch-CPU port of the reference graph (the oracle), not onnxruntime: the reference's own CPU "
                        "path cannot be built or imported on this box",
                "sample": f"{allc['syntheses']} B=1 syntheses of the same {len(ids)}-id utterance in {allc['seconds']:.1f} s by "
                          f"{allc['processes']} processes x {allc['threads_each']} thread (torch CPU fp32, {ucores} usable of {ncpu} cores); one "
                          f"program alone on {cores} threads: {single['value'] / 1e6:.2f} M samples/s"}
    return {"value": samples / dt, "unit": "samples/s", "cores": cores, "kind": "port", "onnxruntime_probe": ort_probe,
            "x_realtime": samples / dt / cfg.sample_rate, "usable_cores": ucores, "all_cores_attempt": allc,
            "note": "torch-CPU port of the reference graph (the oracle), not onnxruntime: the reference's own CPU path "
                    "cannot be built or imported on this box",
            "sample": f"{n} sequential B=1 syntheses of the same {len(ids)}-id utterance in {dt:.1f} s "
                      f"(torch CPU fp32, {cores} threads chosen by a probe over 1..64 on a {ncpu}-core host)"}


def usable_cores():
    """Cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (a container on a 256-core
    host can be limited to a few of them; os.cpu_count() does not say)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_all_cores(preset, T, ncpu, budget_s):
    """P single-threaded worker processes (oracle/cpu_worker.py; one thread each, so nothing spins when the box has fewer
    usable cores than it reports) over a shared window of ~budget_s/2 seconds."""
    threads = 1
    procs = max(1, min(64, ncpu))
    if procs < 2:
        return None
    seconds = max(4.0, min(10.0, budget_s / 2))
    start_at = time.time() + 25.0                     # imports + voice generation + one warm-up synthesis per worker
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_worker.py"), preset, str(T), str(threads), str(seconds)]
    ps = [subprocess.Popen(cmd + [str(start_at)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, text=True)
          for _ in range(procs)]
    rows, err = [], None
    for p in ps:
        try:
            o, e = p.communicate(timeout=120)
            rows.append(json.loads(o.strip().splitlines()[-1]))
        except Exception as ex:          # noqa: BLE001 -- a worker that failed simply does not count
            p.kill()
            err = err or f"{type(ex).__name__}: {ex}"
    if not rows:
        return {"value": 0.0, "error": err}
    t0, t1 = min(r["t0"] for r in rows), max(r["t1"] for r in rows)
    samples = sum(r["samples"] for r in rows)
    return {"value": samples / (t1 - t0), "cores": len(rows) * threads, "processes": len(rows), "threads_each": threads,
            "syntheses": sum(r["n"] for r in rows), "seconds": t1 - t0, "error": err}


if __name__ == "__main__":
    main()
