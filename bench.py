#!/usr/bin/env python3
"""Throughput of the synthesis hot path on MI355X (BASELINE.json metric: audio samples/s and x real-time).

A "step" is one pass of the hot path over one batch of synthetic input. Default (--config 2 = BASELINE.json
configs[1]): ONE utterance of 128 phoneme ids through the en_US-lessac-medium architecture, i.e. what one
piper::synthesize() call does. Phoneme ids and the duration noise are resident in HBM when the timed region starts;
a timed step is the whole device pipeline (`pe_run`: its mid-pipeline 4-byte read-back of the frame count included,
fresh prior noise drawn on the device every step) plus the delivery of the int16 PCM to pinned host memory
(`pe_fetch`). The rate of the full C-ABI call with host inputs (`pe_synthesize_batch`: ids H2D, float + int16 D2H,
the span the reference's inferSeconds covers) is reported beside it as `api_inclusive`.

    python bench.py                          # N=1, configs[1]
    python bench.py --config 3               # configs[2]: high, 64 x 128 ids
    python bench.py --config 4 --gpus 8      # configs[3]: medium, 64 utterances per GPU, 512 over 8 GPUs
    python bench.py --config 5               # configs[4]: streaming first-chunk latency
    python bench.py --gpus N                 # launches N ranks itself (torch.distributed.run, one process per GPU)

N>1: one process per GPU, every rank synthesizes its own utterances (weak scaling, no data-path collective); the
voice is parsed and packed by rank 0 only and broadcast over RCCL ("nccl" backend) before the timed region.
Prints ONE JSON line (rank 0).
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

# BASELINE.json configs (1-based like SURVEY.md section 8d): preset, utterances per GPU, ids per utterance
CONFIGS = {2: ("medium", 1, 128), 3: ("high", 64, 128), 4: ("medium", 64, 128), 5: ("high", 1, 128)}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS),
                    help="BASELINE.json configs[] entry (1-based): 2 medium B=1 (default), 3 high B=64, 4 medium "
                         "64 utterances per GPU, 5 streaming latency")
    ap.add_argument("--preset", default=None)
    ap.add_argument("--ids", type=int, default=None, help="phoneme ids per utterance")
    ap.add_argument("--batch", type=int, default=None, help="utterances per step (per GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the per-kernel event passes (profiling runs)")
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


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:
        self_launch(args)
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    preset, B, T = CONFIGS[args.config]
    preset = args.preset or preset
    B = args.batch or B
    T = args.ids or T
    if args.config == 5:
        args.stream_latency = True

    import torch
    from piper_amd import weights as W
    from piper_amd.engine import Engine

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    backend = os.environ.get("PIPER_BENCH_BACKEND", "nccl")   # gloo: single-GPU smoke test of the multi-process path only
    if world > ndev and backend == "nccl":
        raise SystemExit(f"--gpus {world} but only {ndev} GPU(s) visible")
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    dist = None
    cdev = torch.device("cuda", dev_index) if backend == "nccl" else torch.device("cpu")
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=cdev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus

    cfg = W.preset(preset)
    # ---- voice: rank 0 builds / parses / packs it; the others lay out an identical weight arena from the blob header and
    # receive the PACKED weights by one device-to-device broadcast into that arena (RCCL over xGMI)
    wts = W.synthetic_weights(cfg, 1234) if rank == 0 else None
    t_bcast, bcast_bytes = 0.0, 0
    if world > 1:
        from piper_amd.dist import load_sharded
        eng, t_bcast, bcast_bytes = load_sharded(W.pack_blob(cfg, wts) if rank == 0 else None, 0, dev_index)
    else:
        eng = Engine(blob=W.pack_blob(cfg, wts), device=dev_index)

    if args.stream_latency:
        stream_latency(eng, cfg, preset, T, args, rank)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- synthetic input, resident in HBM before timing
    id_max = min(cfg.n_vocab - 1, 129)
    id_lists = [W.synthetic_phoneme_ids(T, rank * B + i, id_max=id_max) for i in range(B)]
    scales = (0.667, 1.0, 0.8)
    rng = np.random.default_rng(1234 + rank)
    noise_w = rng.standard_normal((B, 2, T)).astype(np.float32)   # fixes the durations; the prior noise is drawn on device
    eng.set_seed(1234 + rank)
    eng.upload(id_lists, scales, noise_w=noise_w)

    def step():
        # device pipeline + delivery of the int16 PCM to pinned host memory (stream sync inside); the result views are
        # used as the C ABI hands them out -- no Python-side copy of the samples inside the timed region
        eng.run()
        return eng.fetch_views(False, True)

    for _ in range(max(1, args.warmup)):     # (at least one untimed step: graph capture, and the frame counts below)
        step()
    res = eng.fetch(False, True)             # the last warm-up step's result as numpy copies, for the bookkeeping
    torch.cuda.synchronize()
    frames = res.frames
    samples_per_step = int(frames.sum()) * eng.hop
    launches_per_step = eng.run_launches
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed_local = time.perf_counter() - t0
    elapsed, total_samples = elapsed_local, float(samples_per_step * args.steps)
    per_rank = [total_samples / elapsed_local]
    if dist is not None:
        t = torch.tensor([elapsed_local], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        s = torch.tensor([total_samples], dtype=torch.float64, device=cdev)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        total_samples = float(s.item())
        pr = [torch.zeros(1, dtype=torch.float64, device=cdev) for _ in range(world)]
        dist.all_gather(pr, torch.tensor([per_rank[0]], dtype=torch.float64, device=cdev))
        per_rank = [float(x.item()) for x in pr]

    # ---- sustained: keep the GPU busy for >= --min-seconds in total (same step), reported separately
    sustained = None
    if elapsed_local < args.min_seconds:
        n_more = int(min(200000, max(1, (args.min_seconds - elapsed_local) / (elapsed_local / args.steps))))
        t1 = time.perf_counter()
        for _ in range(n_more):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        sustained = {"steps": n_more, "seconds": dt, "value": samples_per_step * n_more / dt,
                     "ms_per_step": dt / n_more * 1e3}

    # ---- the full C-ABI call with host buffers (ids H2D, float + int16 D2H): what inferSeconds spans in the reference
    n_api = max(3, min(50, args.steps))
    eng.synthesize_batch(id_lists, scales, noise_w=noise_w)
    t1 = time.perf_counter()
    for _ in range(n_api):
        r_api = eng.synthesize_batch(id_lists, scales, noise_w=noise_w)
    dt_api = time.perf_counter() - t1
    api = {"value": sum(p.size for p in r_api.pcm) * n_api / dt_api, "unit": "samples/s", "calls": n_api,
           "ms_per_call": dt_api / n_api * 1e3,
           "what": "pe_synthesize_batch with host inputs and outputs (ids H2D, device pipeline, float + int16 D2H)"}
    # device pipeline only (no PCM delivery), for comparison with round 1's `value`
    eng.upload(id_lists, scales, noise_w=noise_w)
    n_dev = max(3, min(50, args.steps))
    eng.run(); eng.fetch(False, False)
    t1 = time.perf_counter()
    for _ in range(n_dev):
        eng.run()
    eng.fetch(False, False)
    dev_only_ms = (time.perf_counter() - t1) / n_dev * 1e3

    roof = None
    if rank == 0 and not args.no_roofline:
        roof = roofline(eng, cfg, preset, B, T, frames, id_lists, scales, noise_w, args, elapsed / args.steps)

    if rank == 0:
        value = total_samples / elapsed
        out = {
            "metric": "audio samples/sec",
            "value": value,
            "unit": "samples/s",
            "x_realtime": value / cfg.sample_rate,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (seeded random-weight voice of the named architecture, synthetic phoneme ids)",
            "config": {"workload": f"BASELINE configs[{args.config - 1}]: {preset} VITS voice ({cfg.sample_rate} Hz), {B} "
                                   f"utterance(s) x {T} phoneme ids per step per GPU, scales 0.667/1.0/0.8; step = pe_run "
                                   f"(device pipeline, inputs resident) + int16 PCM to host",
                       "frames_per_step": int(frames.sum()), "samples_per_step": samples_per_step,
                       "kernel_launches_per_step": launches_per_step,
                       "parallelism": f"utterance-parallel x{world}, one process per GPU, RCCL weight broadcast"},
            "api_inclusive": api,
            "device_pipeline_only_ms_per_step": dev_only_ms,
            "sustained": sustained,
            "per_rank_samples_per_s": per_rank,
            "weight_broadcast_s": t_bcast,
            "weight_broadcast_bytes": bcast_bytes,
        }
        if roof is not None:
            out["roofline"] = roof
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, wts, id_lists[0], scales, noise_w[0], args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def roofline(eng, cfg, preset, B, T, frames, id_lists, scales, noise_w, args, step_s):
    """HIP events on the engine's stream (pe_profile_enable): one pass with a pair per pipeline stage, one pass with
    a pair around every conv / attention / layer-norm / fused-stage launch. `kernel` is the kernel with the largest
    share of device time, whatever its bound; `achieved` its algorithmic FLOPs (2 * rows * Cin * taps per output
    column) over its summed launch durations. `step` prices the whole step the same way."""
    eng.upload(id_lists, scales, noise_w=noise_w)
    nprof = max(3, min(10, args.steps))
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
    krows = [r for r in eng.profile()[5:] if r["launches"]]
    eng.profile_enable(0)
    kernels = {r["name"]: {"ms_per_step": r["ms"] / nprof, "launches_per_step": r["launches"] / nprof,
                           "avg_launch_us": r["ms"] / r["launches"] * 1e3,
                           "tflops": (r["flops"] / (r["ms"] * 1e-3) / 1e12 if r["ms"] > 0 else 0.0),
                           "frac_of_mfma_peak": (r["flops"] / (r["ms"] * 1e-3) / 1e12 / FP32_MATRIX_PEAK_TFLOPS
                                                 if r["ms"] > 0 else 0.0),
                           "algorithmic_bytes_per_launch": (r["bytes"] / r["launches"] if r.get("bytes") else None)}
               for r in krows}
    ksum = sum(r["ms"] for r in krows)
    top = max(krows, key=lambda r: r["ms"])
    k = kernels[top["name"]]
    traffic = pmc_traffic(preset, B, T, top["name"])
    if traffic:
        traffic["algorithmic_bytes_per_launch"] = k["algorithmic_bytes_per_launch"]
    # the family view: all split-K launches / all tiled-GEMM launches together
    fam = {}
    for r in krows:
        f = r["name"].split("<")[0]
        d = fam.setdefault(f, {"ms": 0.0, "flops": 0.0, "launches": 0})
        d["ms"] += r["ms"]; d["flops"] += r["flops"]; d["launches"] += r["launches"]
    families = {f: {"share_of_profiled_kernel_time": d["ms"] / ksum, "launches_per_step": d["launches"] / nprof,
                    "tflops": d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0} for f, d in fam.items()}
    step_tf = step_flops / step_s / 1e12
    return {"bound": "mfma", "kernel": top["name"],
            "share_of_profiled_kernel_time": top["ms"] / ksum if ksum else 0.0,
            "achieved": k["tflops"], "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": k["tflops"] / FP32_MATRIX_PEAK_TFLOPS, "traffic": traffic,
            "avg_launch_us": k["avg_launch_us"], "launches_per_step": k["launches_per_step"],
            "step": {"algorithmic_gflop": step_flops / 1e9, "achieved": step_tf, "frac": step_tf / FP32_MATRIX_PEAK_TFLOPS,
                     "what": "all algorithmic FLOPs of one step over the timed ms_per_step"},
            "families": families, "kernels": kernels, "stage_ms": stage_ms, "stage_tflops": stage_tf}


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


def stream_latency(eng, cfg, preset, T, args, rank):
    """Time from the request to the first 45-frame chunk of PCM (encoder + durations + flow + one
    exact-halo vocoder window), like the "Latency" the reference's streaming script logs
    (infer_onnx_streaming.py:118-121), over >= 100 requests; plus the whole-utterance streaming rate."""
    from piper_amd import weights as W
    ids = W.synthetic_phoneme_ids(T, rank, id_max=min(cfg.n_vocab - 1, 129))
    scales = (0.667, 1.0, 0.8)
    first, total, samples = [], [], 0
    n = max(100, args.steps)
    for i in range(n + args.warmup):
        t0 = time.perf_counter()
        it = eng.stream(ids, scales, chunk_frames=45)
        a, _ = next(it)
        t1 = time.perf_counter()
        cnt = a.size + sum(c[0].size for c in it)
        t2 = time.perf_counter()
        if i >= args.warmup:
            first.append((t1 - t0) * 1e3)
            total.append((t2 - t0) * 1e3)
            samples = cnt
    first.sort()
    total.sort()
    if rank == 0:
        print(json.dumps({
            "metric": "p50 first-chunk latency", "value": first[len(first) // 2], "unit": "ms", "higher_is_better": False,
            "p95_ms": first[int(len(first) * 0.95)], "n_gpus": 1, "steps": n, "warmup": args.warmup, "dtype": "f32",
            "data": "synthetic", "vs_baseline": None,
            "config": {"workload": f"BASELINE configs[4]: {preset} VITS voice, streaming decode, one {T}-id utterance, "
                                   f"45-frame chunks, halo {eng.stream_halo} frames, {eng.stream_frames} frames total"},
            "utterance_ms_p50": total[len(total) // 2],
            "streaming_samples_per_s": samples / (total[len(total) // 2] * 1e-3)}), flush=True)


def cpu_baseline(cfg, wts, ids, scales, noise_w, budget_s):
    """CPU baseline on this box's host cores over a bounded sample: repeated B=1 synthesis of the same utterance,
    like piper.cpp's sequential loop. Preferred: onnxruntime's CPU EP with the reference's
    session options (piper.cpp:282-290, benchmark_onnx.py:39-53) -- probed here; it needs both the onnxruntime
    package and an exporter for this voice (torch.onnx + the reference's model code live only in the build
    container), so on a box without them the fallback is the oracle: a torch-CPU port of the reference graph,
    bit-identical to the reference's PyTorch module on the goldens (`kind: "port"`)."""
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
    return {"value": samples / dt, "unit": "samples/s", "cores": cores, "kind": "port", "onnxruntime_probe": ort_probe,
            "x_realtime": samples / dt / cfg.sample_rate,
            "sample": f"{n} sequential B=1 syntheses of the same {len(ids)}-id utterance in {dt:.1f} s "
                      f"(torch CPU fp32, {cores} threads chosen by a probe over 1..64 on a {ncpu}-core host)"}


if __name__ == "__main__":
    main()
