"""Experiment (round 4, call 14): configs[3]'s per-GPU share (medium, 64 x 128 ids) as ONE engine call against the same
64 utterances dealt to N engines that share the GPU (pe_group_*: own stream, worker thread and workspaces each).
Whole C-ABI calls with host inputs and outputs in both cases."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from piper_amd import weights as W
from piper_amd.engine import Engine
from piper_amd.group import EngineGroup

preset = sys.argv[1] if len(sys.argv) > 1 else "medium"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
T = 128
SCALES = (0.667, 1.0, 0.8)
cfg = W.preset(preset)
blob = W.pack_blob(cfg, W.synthetic_weights(cfg, 1234))
texts = [W.synthetic_phoneme_ids(T, 1234 + i, id_max=min(cfg.n_vocab - 1, 129)) for i in range(B)]
calls = 8 if preset == "medium" else 3

def timeit(fn):
    for _ in range(2):
        fn()
    ms = []
    for _ in range(calls):
        t = time.perf_counter(); r = fn(); ms.append((time.perf_counter() - t) * 1e3)
    ms.sort()
    return ms[len(ms) // 2], ms[0], sum(p.size for p in r.pcm)

eng = Engine(blob=blob, device=0)
eng.set_seed(1234)
p50, mn, S = timeit(lambda: eng.synthesize_batch(texts, SCALES))
print(f"{preset} B={B} single engine call: p50 {p50:.3f} ms  min {mn:.3f}  {S / p50 / 1e3:.1f} M samples/s", flush=True)
eng.close()
for n in (1, 2, 3, 4, 8):
    grp = EngineGroup(blob, [0] * n)
    grp.set_seed(1234)
    p50, mn, S = timeit(lambda: grp.synthesize_batch(texts, SCALES))
    print(f"{preset} B={B} group of {n} engines:  p50 {p50:.3f} ms  min {mn:.3f}  {S / p50 / 1e3:.1f} M samples/s", flush=True)
    grp.close()
