"""CPU study (torch, no GPU): what the split-operand matrix modes cost in accuracy at full size, before any kernel is
written. The flow + generator convs of the oracle are re-run with both operands split into low-precision terms and only
the listed term products kept (each product exact, summed in f64, rounded to f32 -- the accumulation rounding of the f32
MFMA accumulate is the f32 path's own and is not modelled):
    bf16x3   v = h + l (bf16),      products hh, hl, lh                (16 significand bits per operand)
    bf16x6   v = h + m + l (bf16),  products hh, hm, mh, hl, lh, mm    (24 bits: the whole f32 significand)
    f16x3    v = h + l (f16),       products hh, hl, lh                (22 bits; |v| <= 65504, terms below 2^-24 flush)
Usage: python scripts/split_study.py [medium|high] [T] [family]
Prints max|d audio| and PCM RMS against the plain f32 oracle on the same inputs (the gate of the f32 HIP path is
max|d audio| < 2e-4 against the oracle, tests/test_gpu_batched.py)."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vits_oracle as O          # noqa: E402  (test / study infrastructure only)
from piper_amd import weights as W           # noqa: E402

MODE = {"on": False, "kind": None}


def terms(v, kind):
    if kind.startswith("bf16"):
        n = 3 if kind == "bf16x6" else 2
        out, r = [], v.clone()
        for _ in range(n):
            t = r.to(torch.bfloat16).to(torch.float32)
            out.append(t)
            r = r - t
        return out
    h = v.clamp(-65504.0, 65504.0).to(torch.float16).to(torch.float32)
    l = (v - h).to(torch.float16).to(torch.float32)
    return [h, l]


def pairs(kind):
    if kind == "bf16x6":
        return [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)]
    return [(0, 0), (0, 1), (1, 0)]


def split_op(fn, x, wt, bias, **kw):
    xs, ws = terms(x, MODE["kind"]), terms(wt, MODE["kind"])
    acc = None
    for i, j in pairs(MODE["kind"]):
        y = fn(xs[i].double(), ws[j].double(), None, **kw)
        acc = y if acc is None else acc + y
    if bias is not None:
        acc = acc + bias.double().view(1, -1, 1)
    return acc.float()


_conv0 = O._conv


def _conv(w, name, x, *, dilation=1, padding=0, groups=1):
    if not MODE["on"] or groups != 1:
        return _conv0(w, name, x, dilation=dilation, padding=padding, groups=groups)
    return split_op(F.conv1d, x, w[name + ".weight"], w.get(name + ".bias"), dilation=dilation, padding=padding)


_ct0 = F.conv_transpose1d


def _ct(x, wt, bias=None, stride=1, padding=0, **kw):
    if not MODE["on"]:
        return _ct0(x, wt, bias, stride=stride, padding=padding, **kw)
    return split_op(lambda a, b, c, **k: _ct0(a, b, c, **k), x, wt, bias, stride=stride, padding=padding)


def wrap(fn):
    def inner(*a, **k):
        MODE["on"] = MODE["kind"] is not None
        try:
            return fn(*a, **k)
        finally:
            MODE["on"] = False
    return inner


def main():
    preset = sys.argv[1] if len(sys.argv) > 1 else "medium"
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    fam = sys.argv[3] if len(sys.argv) > 3 else "gauss"
    cfg = W.preset(preset)
    try:
        wts = W.synthetic_weights(cfg, 1234, family=fam)
    except TypeError:
        wts = W.synthetic_weights(cfg, 1234)
    wt = O.to_torch(wts)
    ids = W.synthetic_phoneme_ids(T, 0, id_max=min(cfg.n_vocab - 1, 129))
    rng = np.random.default_rng(5)
    nw = rng.standard_normal((2, T)).astype(np.float32)
    nz = rng.standard_normal((cfg.inter, 16 * T + 64)).astype(np.float32)
    O._conv = _conv
    O.F.conv_transpose1d = _ct
    O.flow_reverse = wrap(O.flow_reverse)
    O.generator = wrap(O.generator)
    torch.set_num_threads(8)
    with torch.no_grad():
        MODE["kind"] = None
        ref = O.synthesize(wt, cfg, ids, (0.667, 1.0, 0.8), nw, nz)
        r64 = O.synthesize(O.to_torch(wts, torch.float64), cfg, ids, (0.667, 1.0, 0.8), nw, nz, dtype=torch.float64)
        print(f"{preset} T={T} frames={ref['frames']} family={fam}  peak|audio|={np.abs(ref['audio']).max():.3f}")
        d = ref["audio"].astype(np.float64) - r64["audio"]
        print(f"  f32 oracle vs f64 oracle: max|d| {np.abs(d).max():.3e}  rms {np.sqrt((d*d).mean()):.3e}")
        for kind in ("bf16x3", "f16x3", "bf16x6"):
            MODE["kind"] = kind
            r = O.synthesize(wt, cfg, ids, (0.667, 1.0, 0.8), nw, nz)
            assert np.array_equal(r["durations"], ref["durations"])
            d = r["audio"].astype(np.float64) - ref["audio"].astype(np.float64)
            d64 = r["audio"].astype(np.float64) - r64["audio"]
            p = (r["pcm"].astype(np.float64) - ref["pcm"].astype(np.float64)) / 32767.0
            print(f"  {kind:7s} vs f32 oracle: max|d audio| {np.abs(d).max():.3e}  rms {np.sqrt((d*d).mean()):.3e}  "
                  f"pcm rms {np.sqrt((p*p).mean()):.3e} | vs f64: max {np.abs(d64).max():.3e}", flush=True)


if __name__ == "__main__":
    main()
