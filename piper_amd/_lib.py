"""ctypes binding of include/piper_hip.h (libpiper_hip.so, built by `make` / __graft_entry__.build()).

The product path loads exactly one library: ``piper_amd/libpiper_hip.so`` (hipcc, gfx950). If it is
missing or cannot be loaded this module raises -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpiper_hip.so")

SYMBOLS = [
    "pe_create", "pe_create_from_blob", "pe_weights_bound", "pe_create_in_arena", "pe_weights_used", "pe_arena_ready",
    "pe_onnx_to_blob", "pe_free", "pe_synthesize",
    "pe_synthesize_batch", "pe_upload", "pe_run", "pe_fetch", "pe_stream_begin", "pe_stream_next",
    "pe_get_durations", "pe_get_info",
    "pe_set_seed", "pe_profile_enable", "pe_profile_reset", "pe_profile_rows", "pe_profile_get", "pe_profile_bytes",
    "pe_stream", "pe_debug_tensor", "pe_debug_randn", "pe_rng_calls", "pe_run_launches", "pe_speculation_stats", "pe_warmup", "pe_graph_stats", "pe_xcc_pattern", "pe_device_pci_bus_id", "pe_policy_describe", "pe_last_error", "pe_destroy",
    "pe_group_create", "pe_group_broadcast_path", "pe_group_size", "pe_group_engine", "pe_group_synthesize_batch", "pe_group_assignment",
    "pe_group_destroy",
    "pe_coalescer_create", "pe_coalescer_synthesize", "pe_coalescer_stats", "pe_coalescer_destroy",
]


class PeNoise(C.Structure):
    _fields_ = [("noise_w", C.POINTER(C.c_float)), ("w_stride", C.c_int64),
                ("noise_z", C.POINTER(C.c_float)), ("z_stride", C.c_int64)]


class PeResult(C.Structure):
    _fields_ = [("batch", C.c_int32), ("sample_offsets", C.POINTER(C.c_int64)),
                ("audio", C.POINTER(C.c_float)), ("pcm", C.POINTER(C.c_int16)),
                ("frames", C.POINTER(C.c_int32)), ("infer_seconds", C.c_double)]


def bind(path: str) -> C.CDLL:
    """dlopen + declare prototypes. ``path`` is the shipped library for the product; tests may pass
    the path of the emulator build (tests/emu) explicitly."""
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: build the HIP extension first (`make` or __graft_entry__.build()); "
            "piper_amd has no CPU fallback")
    lib = C.CDLL(path)
    vp, i64p, f32p, i32p = C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_float), C.POINTER(C.c_int32)
    lib.pe_last_error.restype = C.c_char_p
    lib.pe_create.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
    lib.pe_create_from_blob.argtypes = [vp, C.c_size_t, C.c_int, C.POINTER(vp)]
    lib.pe_weights_bound.argtypes = [vp, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.pe_create_in_arena.argtypes = [vp, C.c_size_t, C.c_int, vp, C.c_size_t, C.c_int, C.POINTER(vp)]
    lib.pe_weights_used.argtypes = [vp, C.POINTER(C.c_size_t)]
    lib.pe_arena_ready.argtypes = [vp]
    lib.pe_onnx_to_blob.argtypes = [C.c_char_p, C.POINTER(vp), C.POINTER(C.c_size_t)]
    lib.pe_free.argtypes = [vp]
    lib.pe_free.restype = None
    lib.pe_synthesize.argtypes = [vp, i64p, C.c_int64, f32p, C.c_int64, C.POINTER(PeNoise), C.POINTER(PeResult)]
    lib.pe_synthesize_batch.argtypes = [vp, i64p, i64p, C.c_int32, f32p, i64p, C.POINTER(PeNoise),
                                        C.POINTER(PeResult)]
    lib.pe_upload.argtypes = [vp, i64p, i64p, C.c_int32, f32p, i64p, C.POINTER(PeNoise)]
    lib.pe_run.argtypes = [vp]
    lib.pe_fetch.argtypes = [vp, C.c_int, C.c_int, C.POINTER(PeResult)]
    lib.pe_stream_begin.argtypes = [vp, i64p, C.c_int64, f32p, C.c_int64, C.POINTER(PeNoise), i32p, i32p]
    lib.pe_stream_next.argtypes = [vp, C.c_int32, C.POINTER(f32p), C.POINTER(C.POINTER(C.c_int16)), i64p]
    lib.pe_get_durations.argtypes = [vp, i32p, C.c_int64, i64p]
    lib.pe_get_info.argtypes = [vp, i32p, i32p, i32p, i32p, i64p]
    lib.pe_set_seed.argtypes = [vp, C.c_uint64]
    lib.pe_set_seed.restype = None
    lib.pe_profile_enable.argtypes = [vp, C.c_int]
    lib.pe_profile_reset.argtypes = [vp]
    lib.pe_profile_rows.argtypes = [vp]
    lib.pe_profile_get.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_double),
                                   C.POINTER(C.c_double), i64p]
    lib.pe_profile_bytes.argtypes = [vp, C.c_int, C.POINTER(C.c_double)]
    lib.pe_stream.argtypes = [vp]
    lib.pe_stream.restype = vp
    lib.pe_debug_tensor.argtypes = [vp, C.c_char_p, C.c_int32, f32p, C.c_int64, i32p, i32p]
    lib.pe_debug_randn.argtypes = [vp, C.c_int32, C.c_uint64, C.c_int64, C.c_int64, f32p]
    lib.pe_rng_calls.argtypes = [vp]
    lib.pe_rng_calls.restype = C.c_uint64
    lib.pe_run_launches.argtypes = [vp]
    lib.pe_run_launches.restype = C.c_int64
    lib.pe_speculation_stats.argtypes = [vp, i64p, i64p]
    lib.pe_xcc_pattern.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.pe_policy_describe.argtypes = []
    lib.pe_policy_describe.restype = C.c_char_p
    lib.pe_device_pci_bus_id.argtypes = [C.c_int, C.c_char_p, C.c_int32]
    lib.pe_warmup.argtypes = [vp, C.c_int32, C.c_int32, C.c_float, f32p, i64p, C.c_int64]
    lib.pe_graph_stats.argtypes = [vp, i64p, i64p]
    lib.pe_group_create.argtypes = [vp, C.c_size_t, i32p, C.c_int32, C.POINTER(vp)]
    lib.pe_group_broadcast_path.argtypes = []
    lib.pe_group_broadcast_path.restype = C.c_char_p
    lib.pe_group_size.argtypes = [vp]
    lib.pe_group_size.restype = C.c_int32
    lib.pe_group_engine.argtypes = [vp, C.c_int32]
    lib.pe_group_engine.restype = vp
    lib.pe_group_synthesize_batch.argtypes = [vp, i64p, i64p, C.c_int32, f32p, i64p, C.POINTER(PeResult)]
    lib.pe_group_assignment.argtypes = [vp, i32p, C.c_int64]
    lib.pe_group_destroy.argtypes = [vp]
    lib.pe_group_destroy.restype = None
    lib.pe_coalescer_create.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(vp)]
    lib.pe_coalescer_synthesize.argtypes = [vp, i64p, C.c_int64, f32p, C.c_int64, C.POINTER(C.POINTER(C.c_int16)), i64p,
                                            i32p, C.POINTER(C.c_double), i32p]
    lib.pe_coalescer_stats.argtypes = [vp, i64p, i64p]
    lib.pe_coalescer_destroy.argtypes = [vp]
    lib.pe_coalescer_destroy.restype = None
    lib.pe_destroy.argtypes = [vp]
    lib.pe_destroy.restype = None
    return lib


def device_pci_bus_id(device: int, lib=None) -> str:
    """PCI bus id of HIP device `device` (include/piper_hip.h: pe_device_pci_bus_id)."""
    lib = lib if lib is not None else get_lib()
    buf = C.create_string_buffer(64)
    if lib.pe_device_pci_bus_id(int(device), buf, 64):
        raise RuntimeError(lib.pe_last_error().decode(errors="replace"))
    return buf.value.decode()


_lib = None


def get_lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = bind(LIB_PATH)
    return _lib
