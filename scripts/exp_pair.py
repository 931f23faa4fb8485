"""Experiment (round 4, call 22): configs[3]'s per-GPU share as resident-input steps of ONE engine (64 utterances) against
TWO engines on the same GPU stepping 32 utterances each concurrently (two host threads; the ctypes calls release the GIL)."""
import sys, time, os
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from piper_amd import weights as W
from piper_amd.engine import Engine

preset, B, T = (sys.argv[1] if len(sys.argv) > 1 else "medium"), int(sys.argv[2]) if len(sys.argv) > 2 else 64, 128
SCALES = (0.667, 1.0, 0.8)
cfg = W.preset(preset)
blob = W.pack_blob(cfg, W.synthetic_weights(cfg, 1234))
texts = [W.synthetic_phoneme_ids(T, i, id_max=min(cfg.n_vocab - 1, 129)) for i in range(B)]
nw = np.random.default_rng(1234).standard_normal((B, 2, T)).astype(np.float32)
steps = 10 if preset == "medium" else 3

def one(eng):
    eng.run()
    return eng.fetch_views(False, True)

for n in (1, 2, 3):
    engs = [Engine(blob=blob, device=0) for _ in range(n)]
    parts = [list(range(i, B, n)) for i in range(n)]
    for e, p in zip(engs, parts):
        e.set_seed(1234)
        e.upload([texts[i] for i in p], SCALES, noise_w=nw[p])
    pool = ThreadPoolExecutor(n)
    def step():
        if n == 1:
            return [one(engs[0])]
        return list(pool.map(one, engs))
    for _ in range(3):
        step()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ms = (time.perf_counter() - t0) / steps * 1e3
    samples = sum(int(e.fetch(False, True).frames.sum()) * e.hop for e in engs)
    print(f"{preset} B={B}: {n} engine(s) x {B // n} utterances: {ms:.3f} ms per step, {samples / ms / 1e3:.1f} M samples/s", flush=True)
    for e in engs:
        e.close()
