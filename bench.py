#!/usr/bin/env python3
"""Throughput of the synthesis hot path on MI355X.

A "step" is one pass of the hot path over one batch of synthetic input: by default ONE utterance
of 128 phoneme ids through the en_US-lessac-medium architecture (BASELINE.json configs[1]), i.e.
what one piper::synthesize() call does. Inputs (ids, duration noise) are resident in HBM when the
timed region starts; the timed region is the device pipeline only (`pe_run`), including its one
4-byte host read-back of the frame count. N>1: one process per GPU, every rank synthesizes its own
utterances (weak scaling), voice weights parsed on rank 0 and broadcast over RCCL.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MATRIX_PEAK_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--preset", default="medium")
    ap.add_argument("--ids", type=int, default=128, help="phoneme ids per utterance")
    ap.add_argument("--batch", type=int, default=1, help="utterances per step (per GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stream-latency", action="store_true",
                    help="BASELINE configs[4]: p50 time to the first chunk of a chunked (45-frame) decode, then exit")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    import torch
    from piper_amd import weights as W
    from piper_amd.engine import Engine

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dist = None
    # "nccl" is RCCL on ROCm. PIPER_BENCH_BACKEND=gloo exists only to smoke-test the multi-process path on
    # a single-GPU box (ranks then share the GPU and the collectives run on host tensors).
    backend = os.environ.get("PIPER_BENCH_BACKEND", "nccl")
    cdev = torch.device("cuda", dev_index) if backend == "nccl" else torch.device("cpu")
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=cdev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    cfg = W.preset(args.preset)
    # ---- voice weights: rank 0 builds the blob, RCCL broadcast to the others
    if rank == 0:
        wts = W.synthetic_weights(cfg, 1234)
        blob = W.pack_blob(cfg, wts)
    else:
        wts, blob = None, None
    t_bcast = 0.0
    if world > 1:
        from piper_amd.dist import broadcast_blob
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        blob = broadcast_blob(blob, 0, cdev)   # RCCL over xGMI
        torch.cuda.synchronize()
        t_bcast = time.perf_counter() - t0
    eng = Engine(blob=blob, device=dev_index)

    if args.stream_latency:
        stream_latency(eng, cfg, args, rank)
        return

    # ---- synthetic input, resident in HBM before timing
    B, T = args.batch, args.ids
    id_max = min(cfg.n_vocab - 1, 129)
    id_lists = [W.synthetic_phoneme_ids(T, rank * B + i, id_max=id_max) for i in range(B)]
    scales = (0.667, 1.0, 0.8)
    rng = np.random.default_rng(1234 + rank)
    noise_w = rng.standard_normal((B, 2, T)).astype(np.float32)   # fixes the durations; z noise is drawn on device
    eng.set_seed(1234 + rank)
    eng.upload(id_lists, scales, noise_w=noise_w)

    def sync():
        eng.fetch(False, False)          # stream sync, no copies
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.run()
    sync()
    frames = eng.fetch(False, False).frames
    samples_per_step = int(frames.sum()) * eng.hop
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.run()
    sync()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    total_samples = samples_per_step * args.steps
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        s = torch.tensor([total_samples], dtype=torch.float64, device=cdev)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        total_samples = float(s.item())

    # ---- host-buffer (PCIe-inclusive) rate of the full C-ABI call, for DESIGN.md -- never `value`
    t1 = time.perf_counter()
    n_api = max(3, min(10, args.steps))
    for _ in range(n_api):
        r_api = eng.synthesize_batch(id_lists, scales, noise_w=noise_w)
    api_rate = sum(p.size for p in r_api.pcm) * n_api / (time.perf_counter() - t1)

    # ---- roofline: HIP events on the engine's stream (pe_profile_enable): one pass with a pair per
    # pipeline stage, one pass with a pair around every conv/attention/layer-norm launch. The dominant
    # kernel is the conv_mfma_kernel instantiation with the largest share of device time; `achieved` is its
    # algorithmic FLOPs (2 * rows * Cin * taps per output column) over its summed duration.
    eng.upload(id_lists, scales, noise_w=noise_w)
    nprof = max(3, min(10, args.steps))
    eng.profile_enable(1)
    eng.profile_reset()
    for _ in range(nprof):
        eng.run()
    sync()
    rows = eng.profile()[:5]
    stage_ms = {r["name"]: r["ms"] / nprof for r in rows}
    stage_tf = {r["name"]: (r["flops"] / (r["ms"] * 1e-3) / 1e12 if r["ms"] > 0 else 0.0) for r in rows}
    eng.profile_enable(2)
    eng.profile_reset()
    for _ in range(nprof):
        eng.run()
    sync()
    krows = [r for r in eng.profile()[5:] if r["launches"]]
    eng.profile_enable(0)
    kernels = {r["name"]: {"ms_per_step": r["ms"] / nprof, "launches_per_step": r["launches"] / nprof,
                           "avg_launch_us": r["ms"] / r["launches"] * 1e3,
                           "tflops": (r["flops"] / (r["ms"] * 1e-3) / 1e12 if r["ms"] > 0 else 0.0),
                           "algorithmic_bytes_per_launch": (r["bytes"] / r["launches"] if r.get("bytes") else None)}
               for r in krows}
    # the MFMA-bound kernels are the tiled conv GEMM and the fused MRF stage (the HiFiGAN stage north_star
    # prices); conv_splitk_kernel / attention / DDS launches are latency chains and are listed in `kernels`
    convs = [r for r in krows if r["name"].startswith(("conv_mfma_kernel", "mrf_fused_kernel"))] or krows
    dom = max(convs, key=lambda r: r["ms"])
    achieved = kernels[dom["name"]]["tflops"]
    traffic = pmc_traffic(args, B, T, dom["name"])
    # for transparency: the kernel with the largest share of device time whatever its bound (at B=1 that is the
    # split-K conv kernel, a latency chain: tiny GEMMs, one 32x32 tile per workgroup)
    top = max(krows, key=lambda r: r["ms"])
    ksum = sum(r["ms"] for r in krows)
    by_time = {"kernel": top["name"], "share_of_profiled_kernel_time": top["ms"] / ksum if ksum else 0.0,
               "tflops": kernels[top["name"]]["tflops"],
               "frac_of_mfma_peak": kernels[top["name"]]["tflops"] / FP32_MATRIX_PEAK_TFLOPS,
               "avg_launch_us": kernels[top["name"]]["avg_launch_us"]}
    if traffic:
        traffic["algorithmic_bytes_per_launch"] = kernels[dom["name"]]["algorithmic_bytes_per_launch"]
    if traffic and dom["name"].startswith("mrf_fused_kernel<"):
        # one read of the stage input + one write of the MRF mean, fp32
        cp = int(dom["name"].split("<")[1].split(",")[0])
        ch, mult, alg = cfg.up_initial, 1, 0
        for rate in cfg.up_rates:
            ch //= 2
            mult *= rate
            if (ch <= 32) == (cp == 32) and ch <= 64:
                alg += 8 * ch * int(frames.sum()) * mult
        traffic["algorithmic_bytes_per_launch"] = alg

    out = None
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
            "config": {"workload": f"{args.preset} VITS voice ({cfg.sample_rate} Hz), {B} utterance(s) x {T} "
                                   f"phoneme ids per step per GPU, scales 0.667/1.0/0.8",
                       "frames_per_step": int(frames.sum()), "samples_per_step": samples_per_step,
                       "parallelism": f"utterance-parallel x{world}, RCCL weight broadcast"},
            "roofline": {"bound": "mfma", "kernel": dom["name"],
                         "achieved": achieved, "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP32_MATRIX_PEAK_TFLOPS, "traffic": traffic,
                         "avg_launch_us": kernels[dom["name"]]["avg_launch_us"],
                         "launches_per_step": kernels[dom["name"]]["launches_per_step"],
                         "largest_by_time": by_time,
                         "kernels": kernels, "stage_ms": stage_ms, "stage_tflops": stage_tf},
            "host_api_samples_per_s": api_rate,
            "weight_broadcast_s": t_bcast,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, wts, id_lists[0], scales, noise_w[0], args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(args, B, T, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command
    (profiles/*_pmc_traffic.json, written by scripts/pmc_traffic.py from separate --pmc FETCH_SIZE and
    --pmc WRITE_SIZE runs; (2*FETCH_SIZE + WRITE_SIZE) KiB per the gfx950 note in MI355X_MICROARCH.md).
    Counters cannot be read from inside the timed process, so this is null for a workload that has no
    committed pass."""
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    key = f"{args.preset}/b{B}/t{T}"
    for f in sorted(glob.glob(os.path.join(here, "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
            k = d.get(key, {}).get("kernels", {}).get(kernel.replace(" ", ""))
            if k:
                return {"hbm_bytes_per_launch": k["hbm_bytes_per_launch"], "unit": "B",
                        "algorithmic_bytes_per_launch": k.get("algorithmic_bytes_per_launch"), "source": os.path.basename(f)}
        except (OSError, ValueError):
            continue
    return None


def stream_latency(eng, cfg, args, rank):
    """Time from the request to the first 45-frame chunk of PCM (encoder + durations + flow + one
    exact-halo vocoder window), like the "Latency" the reference's streaming script logs
    (infer_onnx_streaming.py:118-121), over >= 100 requests; plus the whole-utterance streaming rate."""
    from piper_amd import weights as W
    ids = W.synthetic_phoneme_ids(args.ids, rank, id_max=min(cfg.n_vocab - 1, 129))
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
    print(json.dumps({
        "metric": "p50 first-chunk latency", "value": first[len(first) // 2], "unit": "ms", "higher_is_better": False,
        "p95_ms": first[int(len(first) * 0.95)], "n_gpus": 1, "steps": n, "warmup": args.warmup, "dtype": "f32",
        "data": "synthetic", "vs_baseline": None,
        "config": {"workload": f"{args.preset} VITS voice, streaming decode, one {args.ids}-id utterance, 45-frame chunks, "
                               f"halo {eng.stream_halo} frames, {eng.stream_frames} frames total"},
        "utterance_ms_p50": total[len(total) // 2],
        "streaming_samples_per_s": samples / (total[len(total) // 2] * 1e-3)}), flush=True)


def cpu_baseline(cfg, wts, ids, scales, noise_w, budget_s):
    """The oracle (a torch-CPU port of the reference graph, bit-identical to the reference's PyTorch
    module on the goldens) timed on this box's host cores over a bounded sample: repeated B=1
    synthesis of the same utterance, like piper.cpp's sequential loop."""
    import torch
    from oracle import vits_oracle as O
    ncpu = os.cpu_count() or 1
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
    return {"value": samples / dt, "unit": "samples/s", "cores": cores, "kind": "port",
            "x_realtime": samples / dt / cfg.sample_rate,
            "sample": f"{n} sequential B=1 syntheses of the same {len(ids)}-id utterance in {dt:.1f} s "
                      f"(torch CPU fp32, {cores} threads chosen by a probe over 1..64 on a {ncpu}-core host)"}


if __name__ == "__main__":
    main()
