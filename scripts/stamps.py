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

KERNELS = {0: "attn_kernel", 1: "conv_splitk_kernel (last launch)", 2: "dds_layer16_kernel (last plain layer)",
           3: "colchain_kernel mode 0 (conv_o + LN)", 4: "conv_splitk16_kernel (end only)", 5: "ln_kernel",
           6: "colchain_kernel mode 1 (post + pre)", 7: "colchain_kernel mode 1 (post only)"}


def main():
    preset = sys.argv[1] if len(sys.argv) > 1 else "medium"
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    lib = L.bind(os.path.join(ROOT, "piper_amd", "libpiper_hip_stamps.so"))
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
    eng.close()


if __name__ == "__main__":
    main()
