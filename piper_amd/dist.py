"""Multi-GPU synthesis: one process per GPU, utterances sharded across ranks, no data-path
collective. The only communication is at load time: rank 0 parses the voice
(.onnx -> weight blob) and broadcasts the blob to the other ranks over RCCL (torch.distributed's
"nccl" backend on ROCm, xGMI on an MI355X node); every rank then builds its own engine from the blob.
Results are gathered to rank 0 in the caller's order.

The reference has no multi-device path at all (piper.cpp loops over phrases sequentially); this is
the MI355X-native extension BASELINE.json's configs[3] asks for."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np


MAX_PER_RANK = 4096     # utterances per engine call (Engine::upload)


def shard_indices(costs: Sequence[int], world: int) -> List[List[int]]:
    """Deterministic longest-processing-time assignment of utterances to ranks by cost (phoneme-id
    count, a proxy for frames): every rank computes the same table. Returns per-rank index lists,
    each sorted in the original order."""
    order = sorted(range(len(costs)), key=lambda i: (-int(costs[i]), i))
    load = [0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        # least-loaded rank that still has room: an engine call takes at most 4096 utterances (same rule as lpt() in
        # piper_amd/csrc/pe_api.cpp)
        room = [k for k in range(world) if len(out[k]) < MAX_PER_RANK]
        if not room:
            raise ValueError(f"more than {MAX_PER_RANK} utterances per rank")
        r = min(room, key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += int(costs[i])
    return [sorted(x) for x in out]


def broadcast_blob(blob: Optional[bytes], src: int = 0, device=None) -> bytes:
    """Broadcast the weight blob from rank `src`. With the nccl backend the payload travels GPU to GPU
    (RCCL over xGMI); with gloo (CPU tests) through host memory."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank()
    backend = dist.get_backend()
    dev = torch.device("cpu")
    if backend == "nccl":
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    n = torch.tensor([len(blob) if rank == src else 0], dtype=torch.int64, device=dev)
    dist.broadcast(n, src)
    if rank == src:
        buf = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
    else:
        buf = torch.empty(int(n.item()), dtype=torch.uint8, device=dev)
    dist.broadcast(buf, src)
    return blob if rank == src else buf.cpu().numpy().tobytes()


def blob_header(blob: bytes) -> bytes:
    """Architecture + tensor records of a PEBLOB01 (no data): all a rank needs to lay its weight arena out."""
    import struct
    from . import weights as W
    n = struct.unpack_from("<I", blob, 8 + 4 * W.ARCH_INTS)[0]
    return blob[: 8 + 4 * W.ARCH_INTS + 8 + n * W.REC_BYTES]


def load_sharded(blob: Optional[bytes], src: int = 0, device: Optional[int] = None, lib=None):
    """One engine per rank with ONE device-to-device broadcast of the PACKED weights (RCCL over xGMI with the nccl
    backend): rank `src` parses / packs / uploads the voice into its arena; every other rank receives only the small
    blob header (tensor shapes), lays out an identical arena without touching weight data, and gets the arena's content
    by `dist.broadcast` straight into it -- no host round trip, no packing outside `src`. The arena is a torch uint8
    tensor (so the collective can take it as is); the engine keeps it alive. Every rank checks that its arena layout
    (bytes used) equals the source's before the collective: the layout depends on the library build and on launcher
    knobs read from the environment, and a mismatch would otherwise hang the broadcast or scatter the weights.
    Returns (engine, seconds in the broadcast, bytes broadcast).
    """
    import time
    import torch
    import torch.distributed as dist
    from . import _lib as L
    from .engine import Engine
    real = lib is None or lib is L._lib          # the shipped GPU library (tests pass the emulator build explicitly)
    lib = lib if lib is not None else L.get_lib()
    rank = dist.get_rank()
    nccl = dist.get_backend() == "nccl"
    on_gpu = real and torch.cuda.is_available()
    if device is None:
        device = torch.cuda.current_device() if on_gpu else 0
    tdev = torch.device("cuda", device) if on_gpu else torch.device("cpu")
    hdr = [blob_header(blob) if rank == src else None]
    dist.broadcast_object_list(hdr, src)
    header = hdr[0]
    bound = C.c_size_t()
    if lib.pe_weights_bound(header, len(header), C.byref(bound)):
        raise RuntimeError(lib.pe_last_error().decode(errors="replace"))
    arena = torch.empty(bound.value + 256, dtype=torch.uint8, device=tdev)
    off = (-arena.data_ptr()) % 256
    arena = arena[off:off + bound.value]
    eng = Engine(blob=blob if rank == src else header, device=device, lib=lib,
                 arena=(arena.data_ptr(), bound.value), skeleton=rank != src)
    eng._arena = arena
    used = eng.weights_used()
    ref = [used if rank == src else None]
    dist.broadcast_object_list(ref, src)
    ok = torch.tensor([1 if ref[0] == used else 0], dtype=torch.int32, device=tdev if nccl else torch.device("cpu"))
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)             # every rank learns of a mismatch anywhere: nobody hangs
    if int(ok.item()) != 1:
        raise RuntimeError(f"rank {rank}: packed-weight arena uses {used} bytes but rank {src}'s uses {ref[0]}: the ranks "
                           "run different builds of libpiper_hip.so or different PIPER_HIP_* settings")
    if on_gpu:
        torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    if nccl or not on_gpu:
        dist.broadcast(arena[:used], src)          # nccl: device to device (RCCL over xGMI)
    else:                                          # gloo smoke test on a GPU box: the collective needs host tensors
        tmp = arena[:used].cpu()
        dist.broadcast(tmp, src)
        arena[:used].copy_(tmp)
    if on_gpu:
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank != src:
        eng.arena_ready()
    return eng, dt, used


def onnx_to_blob(onnx_path: str, lib=None) -> bytes:
    from . import _lib as L
    lib = lib if lib is not None else L.get_lib()
    blob, n = C.c_void_p(), C.c_size_t()
    if lib.pe_onnx_to_blob(str(onnx_path).encode(), C.byref(blob), C.byref(n)):
        raise RuntimeError(lib.pe_last_error().decode(errors="replace"))
    try:
        return C.string_at(blob, n.value)
    finally:
        lib.pe_free(blob)


class ShardedSynthesizer:
    """Utterance-parallel synthesis over the ranks of an initialised torch.distributed group."""

    def __init__(self, onnx_path: Optional[str] = None, blob: Optional[bytes] = None, device: Optional[int] = None,
                 lib=None):
        import torch.distributed as dist
        from .engine import Engine
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        if self.rank == 0:
            if blob is None:
                if onnx_path is None:
                    raise ValueError("rank 0 needs onnx_path or blob")
                blob = onnx_to_blob(onnx_path, lib)
        self.engine, self.broadcast_seconds, self.broadcast_bytes = load_sharded(blob if self.rank == 0 else None, 0,
                                                                                 device, lib)

    def synthesize(self, id_lists: Sequence[Sequence[int]], scales=(0.667, 1.0, 0.8), sids=None,
                   noise_w=None, noise_z=None) -> Optional[List[np.ndarray]]:
        """Every rank passes the same id_lists; returns the int16 PCM of every utterance on rank 0
        (None elsewhere). Optional injected noise is indexed like id_lists."""
        import torch.distributed as dist
        shards = shard_indices([len(x) for x in id_lists], self.world)
        mine = shards[self.rank]
        local = []
        if mine:
            kw = {}
            if sids is not None:
                kw["sids"] = [sids[i] for i in mine]
            if noise_w is not None:
                kw["noise_w"] = np.ascontiguousarray(np.asarray(noise_w)[mine])
            if noise_z is not None:
                kw["noise_z"] = np.ascontiguousarray(np.asarray(noise_z)[mine])
            r = self.engine.synthesize_batch([id_lists[i] for i in mine], scales, **kw)
            local = r.pcm
        gathered = [None] * self.world if self.rank == 0 else None
        dist.gather_object(list(zip(mine, local)), gathered, dst=0)
        if self.rank != 0:
            return None
        out: List[Optional[np.ndarray]] = [None] * len(id_lists)
        for part in gathered:
            for i, pcm in part:
                out[i] = pcm
        return out  # type: ignore[return-value]
