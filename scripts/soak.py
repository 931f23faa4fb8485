"""Soak of the single-utterance / small-batch path (GPU box only): thousands of calls with random texts and batch sizes
through the calls piper::synthesize makes, watching host RSS, free device memory, the graph cache and the speculation
counters -- evicted graphs and regrown workspaces must not leak.
    python scripts/soak.py [calls] [preset]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from piper_amd import weights as W                 # noqa: E402
from piper_amd.engine import Engine                # noqa: E402


def rss_mb():
    with open("/proc/self/status") as f:
        for line in f:
            if line.startswith("VmRSS"):
                return int(line.split()[1]) / 1024.0
    return 0.0


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    preset = sys.argv[2] if len(sys.argv) > 2 else "medium"
    import torch
    cfg = W.preset(preset)
    eng = Engine(blob=W.pack_blob(cfg, W.synthetic_weights(cfg, 1234)))
    rng = np.random.default_rng(7)
    id_max = min(cfg.n_vocab - 1, 129)
    t0 = time.perf_counter()
    samples = 0
    marks = []
    for i in range(n):
        B = int(rng.choice([1, 1, 1, 2, 3, 4]))
        texts = [W.synthetic_phoneme_ids(int(rng.integers(5, 300)), 5000 + 7 * i + j, id_max=id_max) for j in range(B)]
        scales = (0.667, float(rng.choice([0.9, 1.0, 1.2])), 0.8)
        eng.upload(texts, scales)
        eng.run()
        r = eng.fetch(False, True)
        samples += sum(p.size for p in r.pcm)
        assert all(p.size == int(f) * 256 for p, f in zip(r.pcm, r.frames))
        if (i + 1) % (n // 8) == 0:
            free, total = torch.cuda.mem_get_info()
            runs, misses = eng.speculation_stats
            cached, captures = eng.graph_stats
            marks.append((i + 1, rss_mb(), (total - free) / 2**20, cached, captures, runs, misses))
            print("call %6d: host RSS %7.1f MB, device memory in use %8.1f MB, graphs cached %3d / captured %4d, speculative runs %5d misses %3d"
                  % marks[-1], flush=True)
    dt = time.perf_counter() - t0
    print("%d calls in %.1f s: %.2f ms per call, %.1f M samples/s" % (n, dt, dt / n * 1e3, samples / dt / 1e6))
    half = marks[len(marks) // 2]
    last = marks[-1]
    print("growth over the second half: host RSS %+.1f MB, device %+.1f MB" % (last[1] - half[1], last[2] - half[2]))
    eng.close()


if __name__ == "__main__":
    main()
