"""In-process multi-GPU synthesis (``pe_group_*`` of include/piper_hip.h): one engine, stream and worker thread per
device inside ONE process -- what a C++ caller of the library uses to reach the GPUs of a node without
``torch.distributed`` (for the one-process-per-GPU form see ``piper_amd.dist``). The voice is packed once on the first
device and reaches identically laid out arenas on the others by one RCCL broadcast (peer copies without librccl).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from . import _lib as L
from .engine import EngineError, Synthesis


class EngineGroup:
    def __init__(self, blob: bytes, devices: Sequence[int], lib: Optional[C.CDLL] = None):
        self._lib = lib if lib is not None else L.get_lib()
        self._blob = bytes(blob)
        self._h = C.c_void_p()
        dev = np.ascontiguousarray(devices, dtype=np.int32)
        rc = self._lib.pe_group_create(self._blob, len(self._blob), dev.ctypes.data_as(C.POINTER(C.c_int32)), len(dev),
                                       C.byref(self._h))
        self._check(rc)
        self.devices = [int(d) for d in dev]
        # "rccl" | "peer-copy (<why>)" | "same-device" | "none": how the packed weights reached the other devices
        self.broadcast_path = self._lib.pe_group_broadcast_path().decode()

    def _check(self, rc):
        if rc:
            raise EngineError(self._lib.pe_last_error().decode(errors="replace"))

    def __len__(self):
        return int(self._lib.pe_group_size(self._h))

    def engine(self, i: int):
        """Engine i of the group as a borrowed ``piper_amd.engine.Engine`` (durations / float waveform / profile of its
        share of the last call); the group keeps ownership."""
        from .engine import Engine
        h = self._lib.pe_group_engine(self._h, int(i))
        if not h:
            raise EngineError(f"no engine {i} in a group of {len(self)}")
        return Engine.borrowed(self._lib, h)

    def set_seed(self, seed: int):
        """Engine i draws from seed + i (independent noise streams per device)."""
        for i in range(len(self)):
            self._lib.pe_set_seed(C.c_void_p(self._lib.pe_group_engine(self._h, i)), int(seed) + i)

    def synthesize_batch(self, id_lists: Sequence[Sequence[int]], scales=(0.667, 1.0, 0.8),
                         sids: Optional[Sequence[int]] = None) -> Synthesis:
        if len(id_lists) == 0:
            raise EngineError("empty batch")
        ids = np.concatenate([np.asarray(x, dtype=np.int64) for x in id_lists])
        off = np.zeros(len(id_lists) + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(x) for x in id_lists])
        sc = np.asarray(scales, dtype=np.float32)
        sid = None if sids is None else np.ascontiguousarray(sids, dtype=np.int64)
        i64p, f32p = C.POINTER(C.c_int64), C.POINTER(C.c_float)
        res = L.PeResult()
        self._check(self._lib.pe_group_synthesize_batch(
            self._h, ids.ctypes.data_as(i64p), off.ctypes.data_as(i64p), len(id_lists), sc.ctypes.data_as(f32p),
            None if sid is None else sid.ctypes.data_as(i64p), C.byref(res)))
        B = res.batch
        so = np.ctypeslib.as_array(res.sample_offsets, shape=(B + 1,)).copy()
        frames = np.ctypeslib.as_array(res.frames, shape=(B,)).copy()
        total = int(so[-1])
        flat = np.ctypeslib.as_array(res.pcm, shape=(max(total, 1),))[:total].copy()
        pcm = [flat[so[i]:so[i + 1]] for i in range(B)]
        return Synthesis([], pcm, frames, float(res.infer_seconds))

    def assignment(self, n: int) -> List[int]:
        out = np.zeros(n, dtype=np.int32)
        self._check(self._lib.pe_group_assignment(self._h, out.ctypes.data_as(C.POINTER(C.c_int32)), n))
        return [int(x) for x in out]

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.pe_group_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Coalescer:
    """Dynamic batching of concurrent single-utterance requests on ONE engine (``pe_coalescer_*`` of include/piper_hip.h):
    ``synthesize`` is thread-safe and blocking; requests that are pending at the same moment run as one batched engine
    call. Every request gets what its own B=1 call computes (own noise draws, int16 peak-normalised over its own
    waveform). The engine must stay alive and must not be used directly while requests are in flight."""

    def __init__(self, engine, max_batch: int = 8, max_wait_us: int = 0):
        self._lib = engine._lib
        self._engine = engine                     # keeps the engine alive
        self._h = C.c_void_p()
        if self._lib.pe_coalescer_create(engine._h, int(max_batch), int(max_wait_us), C.byref(self._h)):
            raise EngineError(self._lib.pe_last_error().decode(errors="replace"))

    def synthesize(self, ids, scales=(0.667, 1.0, 0.8), sid: Optional[int] = None):
        """-> (int16 PCM, frames, seconds of the engine call that served the request, utterances in that call)"""
        a = np.ascontiguousarray(ids, dtype=np.int64)
        sc = (C.c_float * 3)(*[float(v) for v in scales])
        pcm, n, fr, secs, bs = C.POINTER(C.c_int16)(), C.c_int64(), C.c_int32(), C.c_double(), C.c_int32()
        rc = self._lib.pe_coalescer_synthesize(self._h, a.ctypes.data_as(C.POINTER(C.c_int64)), a.size, sc,
                                               -1 if sid is None else int(sid), C.byref(pcm), C.byref(n), C.byref(fr),
                                               C.byref(secs), C.byref(bs))
        if rc:
            raise EngineError(self._lib.pe_last_error().decode(errors="replace"))
        try:
            out = np.ctypeslib.as_array(pcm, shape=(max(int(n.value), 1),))[:int(n.value)].copy()
        finally:
            self._lib.pe_free(C.cast(pcm, C.c_void_p))
        return out, int(fr.value), float(secs.value), int(bs.value)

    @property
    def stats(self):
        """(engine calls, requests) so far"""
        a, b = C.c_int64(), C.c_int64()
        self._lib.pe_coalescer_stats(self._h, C.byref(a), C.byref(b))
        return int(a.value), int(b.value)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.pe_coalescer_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
