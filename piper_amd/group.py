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
