"""Python handle on the HIP engine (C ABI in include/piper_hip.h).

This is the host-side twin of what ``onnxruntime.InferenceSession`` is to the reference's
``PiperVoice`` (reference src/python_run/piper/voice.py:24-55,140-185): ``run()`` takes the same
feed (``input``, ``scales``, optional ``sid``) and returns the float waveform; everything is
computed on the GPU by libpiper_hip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from . import _lib as L


class EngineError(RuntimeError):
    pass


class Synthesis:
    """Result of one (batched) synthesis call. Arrays are copies (the C side reuses its buffers)."""

    def __init__(self, audio: List[np.ndarray], pcm: List[np.ndarray], frames: np.ndarray, infer_seconds: float):
        self.audio = audio
        self.pcm = pcm
        self.frames = frames
        self.infer_seconds = infer_seconds


class Engine:
    def __init__(self, *, onnx_path: Optional[str] = None, blob: Optional[bytes] = None, device: int = 0,
                 lib: Optional[C.CDLL] = None, arena=None, skeleton: bool = False):
        """``arena`` = (device pointer, bytes) of a caller-owned weight arena (multi-GPU loading: see
        piper_amd.dist.load_sharded); with ``skeleton`` the blob may be just the header of a PEBLOB01 and the arena's
        content is expected to arrive by broadcast before ``arena_ready()``."""
        self._lib = lib if lib is not None else L.get_lib()
        self._h = C.c_void_p()
        if (onnx_path is None) == (blob is None):
            raise ValueError("give exactly one of onnx_path / blob")
        if onnx_path is not None:
            rc = self._lib.pe_create(str(onnx_path).encode(), device, C.byref(self._h))
        elif arena is not None:
            self._blob = bytes(blob)
            rc = self._lib.pe_create_in_arena(self._blob, len(self._blob), device, C.c_void_p(int(arena[0])), int(arena[1]),
                                              int(bool(skeleton)), C.byref(self._h))
        else:
            self._blob = bytes(blob)
            rc = self._lib.pe_create_from_blob(self._blob, len(self._blob), device, C.byref(self._h))
        self._check(rc)
        sr, hop, nspk, nsym, wb = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
        self._check(self._lib.pe_get_info(self._h, C.byref(sr), C.byref(hop), C.byref(nspk), C.byref(nsym),
                                          C.byref(wb)))
        self.sample_rate, self.hop, self.num_speakers, self.num_symbols = sr.value, hop.value, nspk.value, nsym.value
        self.weight_bytes = wb.value

    @classmethod
    def borrowed(cls, lib: C.CDLL, handle: int) -> "Engine":
        """View of an engine somebody else owns (``pe_group_engine``): every method works, ``close()`` does not destroy."""
        self = cls.__new__(cls)
        self._lib, self._h, self._borrowed = lib, C.c_void_p(int(handle)), True
        sr, hop, nspk, nsym, wb = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
        self._check(lib.pe_get_info(self._h, C.byref(sr), C.byref(hop), C.byref(nspk), C.byref(nsym), C.byref(wb)))
        self.sample_rate, self.hop, self.num_speakers, self.num_symbols = sr.value, hop.value, nspk.value, nsym.value
        self.weight_bytes = wb.value
        return self

    def weights_used(self) -> int:
        n = C.c_size_t()
        self._check(self._lib.pe_weights_used(self._h, C.byref(n)))
        return int(n.value)

    def arena_ready(self):
        self._check(self._lib.pe_arena_ready(self._h))

    def _check(self, rc: int):
        if rc != 0:
            raise EngineError(self._lib.pe_last_error().decode(errors="replace"))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            if not getattr(self, "_borrowed", False):
                self._lib.pe_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- helpers
    @staticmethod
    def _pack(id_lists: Sequence[Sequence[int]]):
        offs = np.zeros(len(id_lists) + 1, np.int64)
        for i, s in enumerate(id_lists):
            offs[i + 1] = offs[i] + len(s)
        ids = np.concatenate([np.asarray(s, np.int64).reshape(-1) for s in id_lists]) if id_lists else \
            np.zeros(0, np.int64)
        return np.ascontiguousarray(ids), offs

    def _noise(self, noise_w, noise_z, keep):
        if noise_w is None and noise_z is None:
            return None
        n = L.PeNoise()
        if noise_w is not None:
            a = np.ascontiguousarray(noise_w, np.float32)      # [B][2][stride]
            keep.append(a)
            n.noise_w = a.ctypes.data_as(C.POINTER(C.c_float))
            n.w_stride = a.shape[-1]
        if noise_z is not None:
            a = np.ascontiguousarray(noise_z, np.float32)      # [B][inter][stride]
            keep.append(a)
            n.noise_z = a.ctypes.data_as(C.POINTER(C.c_float))
            n.z_stride = a.shape[-1]
        keep.append(n)
        return C.byref(n)

    def _collect(self, res: L.PeResult, want_audio=True, want_pcm=True) -> Synthesis:
        B = res.batch
        offs = np.frombuffer(C.string_at(res.sample_offsets, 8 * (B + 1)), np.int64)
        frames = np.frombuffer(C.string_at(res.frames, 4 * B), np.int32).copy()
        total = int(offs[-1])
        audio, pcm = [], []
        # (one copy out of the engine's pinned buffers through string_at; np.ctypeslib.as_array would build a new ctypes
        # array type for every distinct sample count -- ~0.7 ms per call when the frame counts change from call to call)
        if want_audio and total:
            a = np.frombuffer(bytearray(C.string_at(res.audio, 4 * total)), np.float32)
            audio = [a[offs[i]:offs[i + 1]] for i in range(B)]
        if want_pcm and total:
            p = np.frombuffer(bytearray(C.string_at(res.pcm, 2 * total)), np.int16)
            pcm = [p[offs[i]:offs[i + 1]] for i in range(B)]
        return Synthesis(audio, pcm, frames, res.infer_seconds)

    # ---- API
    def synthesize_batch(self, id_lists, scales=(0.667, 1.0, 0.8), sids=None, noise_w=None, noise_z=None) -> Synthesis:
        ids, offs = self._pack(id_lists)
        sc = (C.c_float * 3)(*[float(s) for s in scales])
        keep: list = []
        nz = self._noise(noise_w, noise_z, keep)
        sid_arr = None
        if sids is not None:
            sid_np = np.ascontiguousarray(sids, np.int64)
            keep.append(sid_np)
            sid_arr = sid_np.ctypes.data_as(C.POINTER(C.c_int64))
        res = L.PeResult()
        self._check(self._lib.pe_synthesize_batch(
            self._h, ids.ctypes.data_as(C.POINTER(C.c_int64)), offs.ctypes.data_as(C.POINTER(C.c_int64)),
            len(id_lists), sc, sid_arr, nz, C.byref(res)))
        return self._collect(res)

    def synthesize(self, ids, scales=(0.667, 1.0, 0.8), sid=None, noise_w=None, noise_z=None) -> Synthesis:
        nw = None if noise_w is None else np.asarray(noise_w, np.float32)[None]
        nz = None if noise_z is None else np.asarray(noise_z, np.float32)[None]
        return self.synthesize_batch([ids], scales, None if sid is None else [sid], nw, nz)

    def upload(self, id_lists, scales=(0.667, 1.0, 0.8), sids=None, noise_w=None, noise_z=None):
        ids, offs = self._pack(id_lists)
        sc = (C.c_float * 3)(*[float(s) for s in scales])
        keep: list = []
        nz = self._noise(noise_w, noise_z, keep)
        sid_arr = None
        if sids is not None:
            sid_np = np.ascontiguousarray(sids, np.int64)
            keep.append(sid_np)
            sid_arr = sid_np.ctypes.data_as(C.POINTER(C.c_int64))
        self._keep = keep          # noise_z is read during run()
        self._check(self._lib.pe_upload(self._h, ids.ctypes.data_as(C.POINTER(C.c_int64)),
                                        offs.ctypes.data_as(C.POINTER(C.c_int64)), len(id_lists), sc, sid_arr, nz))

    def pack_host(self, id_lists, scales=(0.667, 1.0, 0.8)):
        """The host-side inputs of a call as the C ABI takes them -- int64 ids, prefix offsets, float scales in host
        memory, what piper::synthesize wraps as Ort tensors (reference piper.cpp:342-365) -- built once, for callers that
        time whole calls (bench.py) without Python's list handling inside the timed region."""
        ids, offs = self._pack(id_lists)
        sc = (C.c_float * 3)(*[float(s) for s in scales])
        return (ids, offs, sc, ids.ctypes.data_as(C.POINTER(C.c_int64)), offs.ctypes.data_as(C.POINTER(C.c_int64)),
                len(id_lists))

    def upload_host(self, packed):
        """pe_upload of inputs prepared by pack_host: host ids -> device, the engine draws both noise sites."""
        self._keep = []
        self._check(self._lib.pe_upload(self._h, packed[3], packed[4], packed[5], packed[2], None, None))

    def run(self):
        self._check(self._lib.pe_run(self._h))

    def fetch(self, want_audio=True, want_pcm=True) -> Synthesis:
        res = L.PeResult()
        self._check(self._lib.pe_fetch(self._h, int(want_audio), int(want_pcm), C.byref(res)))
        return self._collect(res, want_audio, want_pcm)

    def fetch_views(self, want_audio=True, want_pcm=True) -> "L.PeResult":
        """Like fetch() but returns the C ABI's result views as they are (pointers into the engine's pinned host
        buffers, valid until the next call) -- what a C / C++ caller gets; no numpy copies."""
        res = L.PeResult()
        self._check(self._lib.pe_fetch(self._h, int(want_audio), int(want_pcm), C.byref(res)))
        return res

    def stream(self, ids, scales=(0.667, 1.0, 0.8), sid=None, chunk_frames: int = 45, noise_w=None, noise_z=None):
        """Generator over (float_audio, int16_pcm) chunks of one utterance: encoder/flow once, then the
        vocoder on exact-halo windows of `chunk_frames` frames (reference default 45). Concatenating
        the float chunks gives exactly the unchunked waveform; pcm is peak-normalised per chunk."""
        ids_np = np.ascontiguousarray(ids, np.int64)
        sc = (C.c_float * 3)(*[float(s) for s in scales])
        keep: list = []
        nw = None if noise_w is None else np.asarray(noise_w, np.float32)[None]
        nz = None if noise_z is None else np.asarray(noise_z, np.float32)[None]
        nref = self._noise(nw, nz, keep)
        frames, halo = C.c_int32(), C.c_int32()
        self._check(self._lib.pe_stream_begin(self._h, ids_np.ctypes.data_as(C.POINTER(C.c_int64)), ids_np.size, sc,
                                              -1 if sid is None else int(sid), nref, C.byref(frames), C.byref(halo)))
        self.stream_frames, self.stream_halo = frames.value, halo.value
        while True:
            a, p, n = C.POINTER(C.c_float)(), C.POINTER(C.c_int16)(), C.c_int64()
            self._check(self._lib.pe_stream_next(self._h, int(chunk_frames), C.byref(a), C.byref(p), C.byref(n)))
            if n.value == 0:
                return
            yield (np.ctypeslib.as_array(a, (n.value,)).copy(), np.ctypeslib.as_array(p, (n.value,)).copy())

    def durations(self) -> np.ndarray:
        n = C.c_int64()
        self._check(self._lib.pe_get_durations(self._h, None, 0, C.byref(n)))
        out = np.zeros(n.value, np.int32)
        self._check(self._lib.pe_get_durations(self._h, out.ctypes.data_as(C.POINTER(C.c_int32)), out.size,
                                               C.byref(n)))
        return out

    def set_seed(self, seed: int):
        self._lib.pe_set_seed(self._h, int(seed))

    def profile_enable(self, level=1):
        self._check(self._lib.pe_profile_enable(self._h, int(level)))

    def profile_reset(self):
        self._check(self._lib.pe_profile_reset(self._h))

    def profile(self):
        rows = []
        for i in range(self._lib.pe_profile_rows(self._h)):
            name, ms, fl, n = C.c_char_p(), C.c_double(), C.c_double(), C.c_int64()
            self._check(self._lib.pe_profile_get(self._h, i, C.byref(name), C.byref(ms), C.byref(fl), C.byref(n)))
            by = C.c_double()
            self._check(self._lib.pe_profile_bytes(self._h, i, C.byref(by)))
            rows.append({"name": name.value.decode(), "ms": ms.value, "flops": fl.value, "launches": n.value,
                         "bytes": by.value})
        return rows

    RNG_PITCH = 65536

    def debug_randn(self, site: int, call: int, n: int, row: int = 0) -> np.ndarray:
        """Test hook: n draws of the engine's own N(0,1) generator at sampling site 0/1, run counter `call`, from
        logical row `row` of the site's [row][65536] stream (row = utterance * channels + channel, column = id / frame)."""
        out = np.zeros(int(n), np.float32)
        self._check(self._lib.pe_debug_randn(self._h, int(site), int(call), int(row), int(n),
                                             out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    @property
    def run_launches(self) -> int:
        """Kernel launches (graph nodes) of the last run: the dependent launch chain of one step."""
        return int(self._lib.pe_run_launches(self._h))

    @property
    def speculation_stats(self):
        """(runs, misses) of the speculative stage-B sizing since the engine was created (include/piper_hip.h)."""
        runs, miss = C.c_int64(), C.c_int64()
        self._check(self._lib.pe_speculation_stats(self._h, C.byref(runs), C.byref(miss)))
        return int(runs.value), int(miss.value)

    def warmup(self, max_batch: int = 1, max_ids: int = 256, frames_per_id: float = 0.0, scales=None, sample_ids=None):
        """Pre-size the workspaces and (given a sample utterance) capture the single-utterance graphs of every id bucket
        up to max_ids -- include/piper_hip.h: pe_warmup."""
        sc = None if scales is None else (C.c_float * 3)(*[float(v) for v in scales])
        if sample_ids is None:
            self._check(self._lib.pe_warmup(self._h, int(max_batch), int(max_ids), float(frames_per_id), sc, None, 0))
            return
        ids = np.ascontiguousarray(sample_ids, np.int64)
        self._check(self._lib.pe_warmup(self._h, int(max_batch), int(max_ids), float(frames_per_id), sc,
                                        ids.ctypes.data_as(C.POINTER(C.c_int64)), ids.size))

    @property
    def graph_stats(self):
        """(graphs cached, captures since the engine was created) -- include/piper_hip.h: pe_graph_stats."""
        a, b = C.c_int64(), C.c_int64()
        self._check(self._lib.pe_graph_stats(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    @property
    def xcc_pattern(self):
        """(XCC id of workgroups 0..63 of a probe launch at engine creation, round-robin period or 0) -- include/piper_hip.h."""
        xs = (C.c_int32 * 64)()
        per = C.c_int32()
        self._check(self._lib.pe_xcc_pattern(self._h, xs, C.byref(per)))
        return [int(v) for v in xs], int(per.value)

    @property
    def rng_calls(self) -> int:
        return int(self._lib.pe_rng_calls(self._h))

    @property
    def hip_stream(self) -> int:
        return int(self._lib.pe_stream(self._h) or 0)

    def debug_tensor(self, name: str, b: int = 0, capacity: int = 1 << 24) -> np.ndarray:
        out = np.zeros(capacity, np.float32)
        r, c = C.c_int32(), C.c_int32()
        self._check(self._lib.pe_debug_tensor(self._h, name.encode(), b, out.ctypes.data_as(C.POINTER(C.c_float)),
                                              capacity, C.byref(r), C.byref(c)))
        return out[: r.value * c.value].reshape(r.value, c.value).copy()
