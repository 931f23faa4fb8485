"""Why does pe_synthesize_batch (float + int16 out) cost more than pe_upload + pe_run + pe_fetch(int16)? Times the C-ABI calls
of one 128-id utterance in several forms (GPU box only):  python scripts/exp_api2.py"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from piper_amd import _lib as L                    # noqa: E402
from piper_amd import weights as W                 # noqa: E402
from piper_amd.engine import Engine                # noqa: E402


def med(f, n=200, warm=20):
    for _ in range(warm):
        f()
    t = []
    for _ in range(n):
        a = time.perf_counter_ns()
        f()
        t.append((time.perf_counter_ns() - a) / 1e3)
    t.sort()
    return t[len(t) // 2], t[int(len(t) * 0.9)]


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "torch":
        import torch                                   # bench.py's context: torch's HIP runtime state in the process
        torch.cuda.set_device(0)
        torch.cuda.synchronize()
        print("torch imported, device synchronised")
    cfg = W.preset("medium")
    eng = Engine(blob=W.pack_blob(cfg, W.synthetic_weights(cfg, 1234)))
    ids = [W.synthetic_phoneme_ids(128, 0, id_max=129)]
    pk = eng.pack_host(ids)
    rng = np.random.default_rng(1234)
    nw = rng.standard_normal((1, 2, 128)).astype(np.float32)
    lib, h = eng._lib, eng._h
    res = L.PeResult()

    def split(audio, pcm):
        def f():
            eng.upload_host(pk); eng.run(); eng.fetch_views(audio, pcm)
        return f

    def one_call():
        lib.pe_synthesize_batch(h, pk[3], pk[4], 1, pk[2], None, None, C.byref(res))

    print("upload+run+fetch(int16)        p50 %.1f us  p90 %.1f" % med(split(False, True)))
    print("upload+run+fetch(float)        p50 %.1f us  p90 %.1f" % med(split(True, False)))
    print("upload+run+fetch(float+int16)  p50 %.1f us  p90 %.1f" % med(split(True, True)))
    print("pe_synthesize_batch (C call)   p50 %.1f us  p90 %.1f" % med(one_call))
    print("Engine.synthesize_batch (py)   p50 %.1f us  p90 %.1f" % med(lambda: eng.synthesize_batch(ids)))
    print("  ... with injected noise_w    p50 %.1f us  p90 %.1f" % med(lambda: eng.synthesize_batch(ids, noise_w=nw)))
    print("upload+run+fetch(int16) again  p50 %.1f us  p90 %.1f" % med(split(False, True)))
    if len(sys.argv) > 1 and sys.argv[1] == "torch":
        import torch
        t0 = time.perf_counter()
        for _ in range(50):
            eng.synthesize_batch(ids)
        print("50 x Engine.synthesize_batch inside a torch process: %.1f us per call" % ((time.perf_counter() - t0) / 50 * 1e6))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            eng.synthesize_batch(ids)
        print("  ... after torch.cuda.synchronize(): %.1f us per call" % ((time.perf_counter() - t0) / 50 * 1e6))
    print("speculation", eng.speculation_stats, "graphs", eng.graph_stats)
    eng.close()


if __name__ == "__main__":
    main()
