"""Phase timestamps of the fused MRF stage kernels (tuning aid, GPU box only).

    make stamps && python scripts/stamps_mrf.py [preset] [batch] [ids]

Workgroup (0,0,0) of the last mrf_kernel launch per channel width: entry, window staged, then per phase the start of the K
loop, its end and the end of the epilogue (kernels/mrf.h PE_STAMP), in us, beside the K loop's matrix time at the pipe's
rate (steps x MFMAs per step x 32 clocks x 2 waves per SIMD at 2.4 GHz).
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from piper_amd import _lib as L, weights as W      # noqa: E402
from piper_amd.engine import Engine                # noqa: E402


def main():
    preset = sys.argv[1] if len(sys.argv) > 1 else "medium"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    lib = L.bind(os.environ.get("PIPER_STAMPS_LIB") or os.path.join(ROOT, "piper_amd", "libpiper_hip_stamps.so"))
    lib.pe_debug_stamps.argtypes = [C.POINTER(C.c_longlong)]
    cfg = W.preset(preset)
    eng = Engine(blob=W.pack_blob(cfg, W.synthetic_weights(cfg, 1234)), lib=lib)
    ids = [W.synthetic_phoneme_ids(T, 7 + i, id_max=129) for i in range(B)]
    eng.upload(ids, (0.667, 1.0, 0.8))
    for _ in range(5):
        eng.run()
    buf = (C.c_longlong * (8 * 24))()
    rc = lib.pe_debug_stamps(buf)
    assert rc == 0, rc
    st = np.array(buf[:], dtype=np.int64).reshape(8, 24)
    for k, name in ((6, "mrf_kernel<64,...>"), (7, "mrf_kernel<32,...>")):
        row = st[k]
        if row[0] == 0:
            continue
        t = [(row[i] - row[0]) / 100.0 for i in range(24)]
        print("%s  B=%d: window staged at +%.2f us" % (name, B, t[1]))
        for ph in range(7):
            a, b, c = row[2 + 3 * ph], row[3 + 3 * ph], row[4 + 3 * ph]
            if a == 0 or a < row[0]:
                break
            print("  phase %d: K loop starts +%.2f  K loop %.2f us  epilogue %.2f us" % (ph, (a - row[0]) / 100.0, (b - a) / 100.0, (c - b) / 100.0))
        print("  phase 4 epilogue, after unit 0..3: " + " ".join("+%.2f" % ((row[20 + u] - row[3 + 3 * 4]) / 100.0) if row[20 + u] else "-" for u in range(4)))
        last = max(row)
        print("  last stamp +%.2f us" % ((last - row[0]) / 100.0))
    eng.close()


if __name__ == "__main__":
    main()
