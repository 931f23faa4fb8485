"""ORACLE support -- generates tests/golden/*.npz by running the REFERENCE's own PyTorch graph
(/root/reference, build container only). Committed together with its outputs so the oracle can
be re-pinned: ``python oracle/make_golden.py``.

Each golden holds: arch preset name, weight seed, ids, scales, the injected noise seed, and the
reference's float audio / per-id durations / latent z. Weights are NOT stored: they are
regenerated from the seed by ``piper_amd.weights.synthetic_weights`` (numpy Generator streams are
stable across numpy versions for ``standard_normal``).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from piper_amd import weights as W  # noqa: E402
from oracle import ref_harness as R  # noqa: E402

CASES = [
    # name,        preset,      T,  scales,              sid
    ("tiny_zero",  "tiny",      24, (0.0, 1.0, 0.0),     None),
    ("tiny_noise", "tiny",      24, (0.667, 1.0, 0.8),   None),
    ("tiny_len",   "tiny",      9,  (0.667, 1.7, 0.8),   None),
    ("tinyhigh_noise", "tiny-high", 24, (0.667, 1.0, 0.8), None),
    ("tinyms_noise", "tiny-ms", 24, (0.667, 1.0, 0.8),   2),
    ("xlow_cfg1",  "x-low",     64, (0.667, 1.0, 0.8),   None),   # BASELINE configs[0] shape
    ("medium_t48", "medium",    48, (0.667, 1.0, 0.8),   None),
    ("high_t32",   "high",      32, (0.667, 1.0, 0.8),   None),
]
WEIGHT_SEED = 1234
NOISE_SEED = 7


def noise_for(cfg, T, seed=NOISE_SEED):
    rng = np.random.default_rng(seed)
    nw = rng.standard_normal((2, T)).astype(np.float32)
    nz = rng.standard_normal((cfg.inter, 32 * T + 64)).astype(np.float32)
    return nw, nz


def main():
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    models = {}
    for name, preset, T, scales, sid in CASES:
        cfg = W.preset(preset)
        if preset not in models:
            models[preset] = R.build_reference_model(cfg, W.synthetic_weights(cfg, WEIGHT_SEED))
        ids = W.synthetic_phoneme_ids(T, 0, id_max=min(cfg.n_vocab - 1, 129))
        nw, nz = noise_for(cfg, T)
        ref = R.reference_infer(models[preset], ids, scales, nw, nz, sid=sid)
        assert ref["frames"] <= nz.shape[1]
        np.savez_compressed(
            os.path.join(out_dir, name + ".npz"), preset=preset, weight_seed=WEIGHT_SEED,
            noise_seed=NOISE_SEED, ids=ids, scales=np.asarray(scales, np.float32),
            sid=-1 if sid is None else sid, audio=ref["audio"].astype(np.float32),
            durations=ref["durations"], frames=ref["frames"], z=ref["z"].astype(np.float32))
        print(f"{name}: T={T} frames={ref['frames']} samples={ref['audio'].size} "
              f"peak={np.abs(ref['audio']).max():.4f}")


if __name__ == "__main__":
    main()
