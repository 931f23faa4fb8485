"""Extreme call shapes through the C ABI (GPU box only): very large batches of short texts, a few very long texts, a long
streamed utterance -- each must either run (PCM lengths = frames x hop, finite audio) or fail with a clean error.
    python scripts/exp_limits.py [preset]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from piper_amd import weights as W                 # noqa: E402
from piper_amd.engine import Engine, EngineError   # noqa: E402


def main():
    preset = sys.argv[1] if len(sys.argv) > 1 else "medium"
    import torch
    cfg = W.preset(preset)
    eng = Engine(blob=W.pack_blob(cfg, W.synthetic_weights(cfg, 1234)))
    id_max = min(cfg.n_vocab - 1, 129)
    rng = np.random.default_rng(3)
    shapes = [("512 x 32 ids", [32] * 512), ("2048 x 8 ids", [8] * 2048), ("4096 x 4 ids", [4] * 4096),
              ("4097 x 4 ids (one too many)", [4] * 4097),
              ("ragged 300 utterances, 1..200 ids", [int(v) for v in rng.integers(1, 200, 300)]),
              ("2 x 3000 ids", [3000, 2500]), ("1 x 8000 ids", [8000]), ("1 x 20000 ids (beyond the frame limit)", [20000]),
              ("64 x 512 ids", [512] * 64), ("back to 1 x 128 ids", [128])]
    for name, lens in shapes:
        texts = [W.synthetic_phoneme_ids(T, 11 + i, id_max=id_max) for i, T in enumerate(lens)]
        t0 = time.perf_counter()
        try:
            eng.upload(texts, (0.667, 1.0, 0.8))
            eng.run()
            r = eng.fetch(False, True)
            dt = time.perf_counter() - t0
            ok = all(p.size == int(f) * 256 for p, f in zip(r.pcm, r.frames)) and all(np.abs(p).max() <= 32767 for p in r.pcm)
            peak = sum(int(np.abs(p).max()) == 32767 for p in r.pcm)
            free, total = torch.cuda.mem_get_info()
            print("%-42s ok=%s  %8.1f ms (first call)  frames %d..%d  utterances at full scale %d / %d  device memory in use %.1f GB"
                  % (name, ok, dt * 1e3, int(min(r.frames)), int(max(r.frames)), peak, len(lens), (total - free) / 2**30), flush=True)
        except EngineError as e:
            print("%-42s clean error: %s" % (name, str(e)[:150]), flush=True)
    # a long streamed utterance: chunks concatenate to the frame count announced by pe_stream_begin
    ids = W.synthetic_phoneme_ids(1200, 5, id_max=id_max)
    n = 0
    t0 = time.perf_counter()
    for a, p in eng.stream(ids, (0.667, 1.0, 0.8), chunk_frames=45):
        n += p.size
    print("stream of 1200 ids: %d samples in %.1f ms" % (n, (time.perf_counter() - t0) * 1e3))
    eng.close()


if __name__ == "__main__":
    main()
