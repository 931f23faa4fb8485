"""Where a whole single-utterance call spends its time on the host side (GPU box only): pe_upload / pe_run / pe_fetch timed
separately around 300 calls of one 128-id utterance, beside back-to-back replays of the same graph (device time per step).
    python scripts/exp_api.py [preset] [ids]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from piper_amd import weights as W                 # noqa: E402
from piper_amd.engine import Engine                # noqa: E402


def main():
    preset = sys.argv[1] if len(sys.argv) > 1 else "medium"
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    cfg = W.preset(preset)
    eng = Engine(blob=W.pack_blob(cfg, W.synthetic_weights(cfg, 1234)))
    ids = [W.synthetic_phoneme_ids(T, 7, id_max=129)]
    for _ in range(30):
        eng.upload(ids)
        eng.run()
        eng.fetch_views(False, True)
    n = 300
    t = np.zeros((n, 4))
    for i in range(n):
        a = time.perf_counter_ns()
        eng.upload(ids)
        b = time.perf_counter_ns()
        eng.run()
        c = time.perf_counter_ns()
        eng.fetch_views(False, True)
        d = time.perf_counter_ns()
        t[i] = (b - a, c - b, d - c, d - a)
    med = np.median(t, axis=0) / 1e3
    print("env HSA_ENABLE_INTERRUPT=%s: upload %.1f us, run (enqueue) %.1f us, fetch (wait + views) %.1f us, whole call %.1f us (p10 %.1f, p90 %.1f)"
          % (os.environ.get("HSA_ENABLE_INTERRUPT", "-"), med[0], med[1], med[2], med[3],
             np.percentile(t[:, 3], 10) / 1e3, np.percentile(t[:, 3], 90) / 1e3))
    eng.upload(ids)
    a = time.perf_counter_ns()
    for i in range(n):
        eng.run()
    eng.fetch_views(False, True)
    print("  back-to-back replays: %.1f us per step" % ((time.perf_counter_ns() - a) / 1e3 / n))
    eng.close()


if __name__ == "__main__":
    main()
