"""ORACLE support -- test infrastructure only; needs /root/reference, so it runs in the build
container only (never on the GPU box, never from the product path).

Imports the reference's own PyTorch definition of the synthesis graph
(``/root/reference/src/python/piper_train/vits/models.py``: ``SynthesizerTrn``), loads a canonical
weight dict (``piper_amd/weights.py``) into it, and runs ``infer()`` with the two ``randn`` sites
replaced by injected noise. Used by ``oracle/make_golden.py`` to pin ``oracle/vits_oracle.py``.
Recipe: SURVEY.md section 8c.
"""
from __future__ import annotations

import contextlib
import os
import sys
import types
import warnings
from typing import Dict

import numpy as np
import torch

REFERENCE_SRC = "/root/reference/src/python"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_SRC, "piper_train", "vits"))


def import_reference():
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    name = "piper_train.vits.monotonic_align"
    if name not in sys.modules:
        # training-only Cython module (models.py:647); stub so the import succeeds
        stub = types.ModuleType(name)
        stub.maximum_path = None
        sys.modules[name] = stub
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from piper_train.vits.models import SynthesizerTrn  # noqa
    return SynthesizerTrn


def build_reference_model(cfg, weights: Dict[str, np.ndarray]):
    """Reference module with the canonical weights loaded; weight_norm removed everywhere (the
    exporter removes it on ``dec`` -- export_onnx.py:51-52 -- and ONNX constant-folds it on the
    flow), so the conv weights are exactly the canonical tensors."""
    SynthesizerTrn = import_reference()
    with warnings.catch_warnings(), contextlib.redirect_stdout(open(os.devnull, "w")):
        warnings.simplefilter("ignore")
        m = SynthesizerTrn(
            n_vocab=cfg.n_vocab, spec_channels=513, segment_size=32, inter_channels=cfg.inter,
            hidden_channels=cfg.hidden, filter_channels=cfg.filter, n_heads=cfg.n_heads,
            n_layers=cfg.n_layers, kernel_size=cfg.kernel_size, p_dropout=0.1,
            resblock=str(cfg.resblock), resblock_kernel_sizes=cfg.rb_kernel_sizes,
            resblock_dilation_sizes=cfg.rb_dilations, upsample_rates=cfg.up_rates,
            upsample_initial_channel=cfg.up_initial, upsample_kernel_sizes=cfg.up_kernel_sizes,
            n_speakers=cfg.n_speakers, gin_channels=cfg.gin, use_sdp=True).eval()
        m.dec.remove_weight_norm()
        for f in m.flow.flows:
            if hasattr(f, "enc"):
                f.enc.remove_weight_norm()
    sd = m.state_dict()
    missing = []
    for k, v in weights.items():
        if k not in sd:
            missing.append(k)
            continue
        assert tuple(sd[k].shape) == tuple(v.shape), (k, sd[k].shape, v.shape)
        sd[k] = torch.as_tensor(v)
    assert not missing, missing
    m.load_state_dict(sd)
    return m


@contextlib.contextmanager
def injected_noise(noise_w, noise_z):
    """Replace torch.randn (models.py:111) and torch.randn_like (models.py:718) by fixed draws."""
    orig_randn, orig_like = torch.randn, torch.randn_like

    def randn(*size, **kw):
        t = size[-1] if len(size) == 3 else None
        assert len(size) == 3 and size[0] == 1 and size[1] == 2, size
        if noise_w is None:
            return torch.zeros(*size)
        return torch.as_tensor(np.asarray(noise_w), dtype=torch.float32).view(1, 2, -1)[:, :, :t]

    def randn_like(x, **kw):
        if noise_z is None:
            return torch.zeros_like(x)
        n = torch.as_tensor(np.asarray(noise_z), dtype=x.dtype)
        return n[None, :, :x.shape[2]]

    torch.randn, torch.randn_like = randn, randn_like
    try:
        yield
    finally:
        torch.randn, torch.randn_like = orig_randn, orig_like


@torch.no_grad()
def reference_infer(model, ids, scales, noise_w=None, noise_z=None, sid=None):
    """Runs the reference graph exactly as export_onnx.py:56-69 wraps it (B=1)."""
    x = torch.as_tensor(np.asarray(ids), dtype=torch.long).view(1, -1)
    xl = torch.tensor([x.shape[1]], dtype=torch.long)
    sid_t = None if sid is None else torch.tensor([int(sid)], dtype=torch.long)
    with injected_noise(noise_w, noise_z):
        o, attn, y_mask, (z, z_p, m_p, logs_p) = model.infer(
            x, xl, sid=sid_t, noise_scale=float(scales[0]), length_scale=float(scales[1]),
            noise_scale_w=float(scales[2]))
    dur = attn.sum(2)[0, 0].round().to(torch.int64)      # per-id frame counts
    return {"audio": o[0, 0].numpy(), "durations": dur.numpy(), "frames": int(y_mask.shape[2]),
            "z": z[0].numpy(), "z_p": z_p[0].numpy()}
