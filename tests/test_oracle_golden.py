"""Pins oracle/vits_oracle.py against tests/golden/*.npz -- outputs of the reference's own
PyTorch graph (SynthesizerTrn.infer, reference models.py:681-722) produced by oracle/make_golden.py.
The reference holds no golden vectors for this path, so these are the pin."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import vits_oracle as O
from piper_amd import weights as W

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
_wcache = {}


def load_case(path):
    g = np.load(path)
    preset = str(g["preset"])
    cfg = W.preset(preset)
    key = (preset, int(g["weight_seed"]))
    if key not in _wcache:
        _wcache.clear()
        _wcache[key] = W.synthetic_weights(cfg, key[1])
    T = len(g["ids"])
    rng = np.random.default_rng(int(g["noise_seed"]))
    nw = rng.standard_normal((2, T)).astype(np.float32)
    nz = rng.standard_normal((cfg.inter, 32 * T + 64)).astype(np.float32)
    sid = int(g["sid"])
    return cfg, _wcache[key], g, nw, nz, (None if sid < 0 else sid)


def test_goldens_present():
    assert len(GOLD) >= 8


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_oracle_matches_reference(path):
    cfg, w, g, nw, nz, sid = load_case(path)
    o = O.synthesize(w, cfg, g["ids"], tuple(g["scales"]), nw, nz, sid=sid, keep=True)
    # integer durations first (SURVEY.md section 7 hard part B), then the waveform
    assert np.array_equal(o["durations"], g["durations"])
    assert o["frames"] == int(g["frames"])
    assert o["audio"].shape == g["audio"].shape
    # same op sequence on the same torch CPU kernels: expect (near) bit equality
    assert np.max(np.abs(o["z"] - g["z"])) <= 1e-5
    assert np.max(np.abs(o["audio"] - g["audio"])) <= 1e-5
    pcm_ref = O.audio_float_to_int16(g["audio"])
    rms = np.sqrt(np.mean(((o["pcm"].astype(np.float64) - pcm_ref) / 32767.0) ** 2))
    assert rms <= 1e-4


def test_fp64_truth_close_to_fp32():
    """fp32 round-off of the whole graph, measured against an fp64 run of the same restatement:
    sets the scale for the 1e-3 RMS tolerance used on the GPU path."""
    cfg, w, g, nw, nz, sid = load_case(GOLD[[os.path.basename(p) for p in GOLD].index("tiny_noise.npz")])
    o32 = O.synthesize(w, cfg, g["ids"], tuple(g["scales"]), nw, nz)
    o64 = O.synthesize(w, cfg, g["ids"], tuple(g["scales"]), nw, nz, dtype=torch.float64)
    assert np.array_equal(o32["durations"], o64["durations"])
    assert np.max(np.abs(o32["audio"] - o64["audio"])) < 1e-5


def test_int16_conversion_matches_reference_rule():
    # piper.cpp:410-431: floor 0.01 on the peak, scale 32767/peak, clamp, truncate toward zero
    a = np.array([0.0, 0.005, -0.005], np.float32)
    assert O.audio_float_to_int16(a).tolist() == [0, 16383, -16383]
    a = np.array([0.5, -0.25, 0.1249], np.float32)
    assert O.audio_float_to_int16(a).tolist() == [32767, -16383, int(0.1249 * 65534)]
    assert O.audio_float_to_int16(np.zeros(0, np.float32)).size == 0


def test_synthetic_ids_shape():
    ids = W.synthetic_phoneme_ids(64, 3)
    assert len(ids) == 64 and ids[0] == 1 and ids[1] == 0 and ids[-1] == 2
    assert all(ids[i] == 0 for i in range(1, 62, 2))


def test_blob_roundtrip():
    cfg = W.preset("tiny-high")
    w = W.synthetic_weights(cfg, 5)
    cfg2, w2 = W.unpack_blob(W.pack_blob(cfg, w))
    assert cfg2 == cfg and set(w2) == set(w)
    assert all(np.array_equal(w[k], w2[k]) for k in w)
