import sys, numpy as np
sys.path.insert(0, ".")
from piper_amd import weights as W
from piper_amd.engine import Engine
cfg = W.preset("medium"); w = W.synthetic_weights(cfg, 1234)
eng = Engine(blob=W.pack_blob(cfg, w), device=0)
ids = W.synthetic_phoneme_ids(128, 0, id_max=129)
for _ in range(4): eng.synthesize(ids)
