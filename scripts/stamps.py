"""Phase timestamps of the latency-critical small kernels at batch 1 (tuning aid, GPU box only).

    make stamps && python scripts/stamps.py [preset] [ids]

Loads piper_amd/libpiper_hip_stamps.so (the product library built with -DPE_STAMPS: thread 0 of workgroup (0,0,0) of
the instrumented kernels stores the 100 MHz wall clock at phase boundaries, pe_rt.h PE_STAMP), replays one utterance a
few times and prints, for the LAST launch of each instrumented kernel in the step, the time between stamps in us.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from piper_amd import _lib as L, weights as W      # noqa: E402
from piper_amd.engine import Engine                # noqa: E402

KERNELS = {0: "attn_kernel / attno_kernel (0 entry, 1 operands in LDS, 2 scores, 3 band, 4 softmax, 5 V P^T, 6 O in LDS, 7 conv_o, 8 LayerNorm + store)", 1: "conv_splitk_kernel (last launch)", 2: "dds_layer16_kernel (last plain layer)",
           3: "colchain_kernel mode 0 (conv_o + LN)", 4: "conv_post_kernel", 5: "ln_kernel / gate4_kernel (0 entry, 1 loads requested + length known, 2 window in LDS, 3 MFMAs done, 4 barrier, 5 epilogue done)",
           6: "colchain_kernel mode 1 (post + pre)", 7: "colchain_kernel mode 1 (post only)"}


def main():
    preset = sys.argv[1] if len(sys.argv) > 1 else "medium"
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    lib = L.bind(os.environ.get("PIPER_STAMPS_LIB") or os.path.join(ROOT, "piper_amd", "libpiper_hip_stamps.so"))
    lib.pe_debug_stamps.argtypes = [C.POINTER(C.c_longlong)]
    cfg = W.preset(preset)
    eng = Engine(blob=W.pack_blob(cfg, W.synthetic_weights(cfg, 1234)), lib=lib)
    ids = [W.synthetic_phoneme_ids(T, 7, id_max=129)]
    eng.upload(ids, (0.667, 1.0, 0.8))
    for _ in range(20):
        eng.run()
    buf = (C.c_longlong * (8 * 24))()
    rc = lib.pe_debug_stamps(buf)
    assert rc == 0, rc
    st = np.array(buf[:], dtype=np.int64).reshape(8, 24)
    for k, name in KERNELS.items():
        row = st[k]
        idx = [i for i in range(24) if row[i] != 0]
        if not idx:
            continue
        t0 = row[idx[0]]
        # stamps of an earlier launch can survive in slots the last launch did not reach: keep the increasing run
        parts, prev = [], t0
        for i in idx[1:]:
            if row[i] < prev:
                continue
            parts.append("%d:+%.2f" % (i, (row[i] - prev) / 100.0))
            prev = row[i]
        print("%-44s total %.2f us   %s" % (name, (prev - t0) / 100.0, "  ".join(parts)))
    trace(lib, eng)
    eng.close()


NAMES = {0: "attn", 1: "conv_splitk", 2: "dds_layer16", 3: "colchain", 4: "conv_splitk16", 5: "ln", 10: "embed",
         11: "conv_mfma", 13: "duration", 6: "conv_splitk_group", 7: "lngemm", 8: "conv_splitk_sum", 14: "dp_persist", 15: "randn", 16: "regulate", 17: "conv_post",
         18: "pcm16", 20: "mrf", 12: "conv_bf3", 21: "spline_inverse"}


def trace(lib, eng):
    """One replayed step, launch by launch: time inside workgroup (0,0,0), gap from the previous launch's exit to this
    one's entry (launch overhead + the other workgroups' tails), and the shader clock while it ran."""
    lib.pe_debug_trace.argtypes = [C.POINTER(C.c_longlong), C.POINTER(C.c_uint)]
    buf = (C.c_longlong * (2048 * 5))()
    cnt = C.c_uint(0)
    assert lib.pe_debug_trace(buf, C.byref(cnt)) == 0      # drop what the warm-up runs recorded
    eng.run()
    assert lib.pe_debug_trace(buf, C.byref(cnt)) == 0
    n = min(cnt.value, 2048)
    tr = np.array(buf[:], dtype=np.int64).reshape(2048, 5)[:n]
    tr = tr[np.argsort(tr[:, 1])]
    print("\nper-launch trace of one step (%d launches): in-WG us, gap before us, MHz" % n)
    t0 = tr[0, 1]
    prev_end = None
    tot_in = tot_gap = 0.0
    agg = {}
    for r in tr:
        kid, w0, w1, c0, c1 = (int(v) for v in r)
        dur = (w1 - w0) / 100.0 if w1 else float("nan")
        gap = (w0 - prev_end) / 100.0 if prev_end else 0.0
        mhz = (c1 - c0) / max(w1 - w0, 1) * 100.0 if w1 else 0.0
        print("%8.2f  %-16s in %6.2f  gap %6.2f  %5.0f MHz" % ((w0 - t0) / 100.0, NAMES.get(kid, str(kid)), dur, gap, mhz))
        if w1:
            prev_end = w1
            tot_in += dur
            tot_gap += gap
            a = agg.setdefault(kid, [0, 0.0, 0.0])
            a[0] += 1; a[1] += dur; a[2] += gap
    print("sum in-WG %.1f us, sum gaps %.1f us, span %.1f us" % (tot_in, tot_gap, (tr[-1, 2] - t0) / 100.0))
    for kid, (c, d, g) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        print("  %-16s x%-3d in %7.2f (avg %5.2f)  gap-before %7.2f (avg %5.2f)" % (NAMES.get(kid, str(kid)), c, d, d / c, g, g / c))


if __name__ == "__main__":
    main()
