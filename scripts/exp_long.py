"""Very long single utterances against the oracle (GPU box only): tiny voice at 3000 / 8000 ids, medium at 3000 ids."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import vits_oracle as O                 # noqa: E402
from piper_amd import weights as W                  # noqa: E402
from piper_amd.engine import Engine                 # noqa: E402

for preset, T in (("tiny", 3000), ("tiny", 8000), ("medium", 3000)):
    cfg = W.preset(preset)
    w = W.synthetic_weights(cfg, 1234)
    eng = Engine(blob=W.pack_blob(cfg, w))
    ids = W.synthetic_phoneme_ids(T, 5, id_max=min(cfg.n_vocab - 1, 129))
    nw = np.random.default_rng(1).standard_normal((2, T)).astype(np.float32)
    r = eng.synthesize(ids, (0.0, 1.0, 0.8), noise_w=nw)
    t0 = time.perf_counter()
    o = O.synthesize(w, cfg, ids, (0.0, 1.0, 0.8), nw)
    same = bool(np.array_equal(eng.durations(), o["durations"]))
    print("%s T=%d: frames %d (oracle %d, %.1f s), durations equal %s, float peak %.4f (oracle %.4f), pcm peak %d (oracle %d), max |d audio| %.3g"
          % (preset, T, int(r.frames[0]), o["frames"], time.perf_counter() - t0, same, float(np.abs(r.audio[0]).max()), float(np.abs(o["audio"]).max()),
             int(np.abs(r.pcm[0].astype(np.int32)).max()), int(np.abs(o["pcm"].astype(np.int32)).max()),
             float(np.abs(r.audio[0] - o["audio"]).max()) if r.audio[0].shape == o["audio"].shape else -1), flush=True)
    eng.close()
