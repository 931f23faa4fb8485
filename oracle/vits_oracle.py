"""ORACLE -- test infrastructure only. A CPU restatement of the reference's synthesis arithmetic.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this file; the product path (``piper_amd``) never does.

What it restates: ``SynthesizerTrn.infer`` (reference ``src/python/piper_train/vits/models.py:681-722``)
as exported by ``export_onnx.py:56-69`` -- i.e. the graph that ``Ort::Session::Run`` executes inside
``piper::synthesize`` (``src/cpp/piper.cpp:386-388``) -- plus the int16 conversion of
``src/cpp/piper.cpp:410-431`` / ``src/python_run/piper/util.py:5-12``. It is written as plain
functions over a flat ``{canonical name -> tensor}`` dict (names: ``piper_amd/weights.py``), with
torch CPU ops as the floating-point kernel library (fp32 like the reference, or fp64 as a
"truth" run). The two ``torch.randn`` sites of the reference (models.py:111 and :718) take
injected noise so runs are reproducible.

Pinning: the reference publishes no golden vectors for this path, so the
oracle is pinned against outputs of the reference's own PyTorch module run in the build container:
``oracle/make_golden.py`` writes ``tests/golden/*.npz`` and ``tests/test_oracle_golden.py`` checks
this file against them.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.1          # modules.py:229,331 ; models.py:317
TAIL_BOUND = 5.0           # modules.py:504
MIN_BIN = 1e-3             # transforms.py:5-7
MIN_DERIV = 1e-3


def _w(w: Dict[str, torch.Tensor], name: str) -> torch.Tensor:
    return w[name]


def _conv(w, name, x, *, dilation=1, padding=0, groups=1):
    b = w.get(name + ".bias")
    return F.conv1d(x, w[name + ".weight"], b, dilation=dilation, padding=padding, groups=groups)


def layer_norm(w, name, x):
    """modules.py:23-26 -- LayerNorm over channels of a [B,C,T] tensor, eps 1e-5."""
    C = x.shape[1]
    y = F.layer_norm(x.transpose(1, -1), (C,), w[name + ".gamma"], w[name + ".beta"], 1e-5)
    return y.transpose(1, -1)


def sequence_mask(length, max_length):
    """commons.py:109-113"""
    x = torch.arange(max_length, dtype=length.dtype)
    return x.unsqueeze(0) < length.unsqueeze(1)


# ----------------------------------------------------------------------------- text encoder

def _get_relative_embeddings(rel, length, window):
    """attentions.py:292-309"""
    pad_length = max(length - (window + 1), 0)
    start = max((window + 1) - length, 0)
    end = start + 2 * length - 1
    if pad_length > 0:
        rel = F.pad(rel, (0, 0, pad_length, pad_length, 0, 0))
    return rel[:, start:end]


def _rel_to_abs(x):
    """attentions.py:311-330"""
    b, h, l, _ = x.size()
    x = F.pad(x, (0, 1, 0, 0, 0, 0, 0, 0))
    x_flat = x.view(b, h, l * 2 * l)
    x_flat = F.pad(x_flat, (0, l - 1, 0, 0, 0, 0))
    return x_flat.view(b, h, l + 1, 2 * l - 1)[:, :, :l, l - 1:]


def _abs_to_rel(x):
    """attentions.py:332-348"""
    b, h, l, _ = x.size()
    x = F.pad(x, (0, l - 1, 0, 0, 0, 0, 0, 0))
    x_flat = x.view(b, h, l * l + l * (l - 1))
    x_flat = F.pad(x_flat, (l, 0, 0, 0, 0, 0))
    return x_flat.view(b, h, l, 2 * l)[:, :, :, 1:]


def attention(w, cfg, prefix, x, attn_mask):
    """MultiHeadAttention.forward/attention with window_size=4, heads_share=True
    (attentions.py:215-272)."""
    nh, H = cfg.n_heads, cfg.hidden
    dk = H // nh
    q = _conv(w, prefix + ".conv_q", x)
    k = _conv(w, prefix + ".conv_k", x)
    v = _conv(w, prefix + ".conv_v", x)
    b, d, t = k.shape
    q = q.view(b, nh, dk, t).transpose(2, 3)
    k = k.view(b, nh, dk, t).transpose(2, 3)
    v = v.view(b, nh, dk, t).transpose(2, 3)
    qs = q / math.sqrt(dk)
    scores = torch.matmul(qs, k.transpose(-2, -1))
    rel_k = _get_relative_embeddings(w[prefix + ".emb_rel_k"], t, cfg.window)
    rel_logits = torch.matmul(qs, rel_k.unsqueeze(0).transpose(-2, -1))
    scores = scores + _rel_to_abs(rel_logits)
    scores = scores.masked_fill(attn_mask == 0, -1e4)
    p = F.softmax(scores, dim=-1)
    out = torch.matmul(p, v)
    rel_w = _abs_to_rel(p)
    rel_v = _get_relative_embeddings(w[prefix + ".emb_rel_v"], t, cfg.window)
    out = out + torch.matmul(rel_w, rel_v.unsqueeze(0))
    out = out.transpose(2, 3).contiguous().view(b, d, t)
    return _conv(w, prefix + ".conv_o", out)


def ffn(w, cfg, prefix, x, x_mask):
    """attentions.py:386-427 (non-causal, ReLU)."""
    k = cfg.kernel_size
    pl, pr = (k - 1) // 2, k // 2
    y = _conv(w, prefix + ".conv_1", F.pad(x * x_mask, (pl, pr)))
    y = torch.relu(y)
    y = _conv(w, prefix + ".conv_2", F.pad(y * x_mask, (pl, pr)))
    return y * x_mask


def text_encoder(w, cfg, ids, lengths):
    """TextEncoder.forward + attentions.Encoder.forward (models.py:198-209, attentions.py:60-74)."""
    H = cfg.hidden
    x = F.embedding(ids, w["enc_p.emb.weight"]) * math.sqrt(H)
    x = x.transpose(1, -1)
    x_mask = sequence_mask(lengths, x.size(2)).unsqueeze(1).to(x.dtype)
    attn_mask = x_mask.unsqueeze(2) * x_mask.unsqueeze(-1)
    x = x * x_mask
    x = x * x_mask
    for l in range(cfg.n_layers):
        y = attention(w, cfg, f"enc_p.encoder.attn_layers.{l}", x, attn_mask)
        x = layer_norm(w, f"enc_p.encoder.norm_layers_1.{l}", x + y)
        y = ffn(w, cfg, f"enc_p.encoder.ffn_layers.{l}", x, x_mask)
        x = layer_norm(w, f"enc_p.encoder.norm_layers_2.{l}", x + y)
    x = x * x_mask
    stats = _conv(w, "enc_p.proj", x) * x_mask
    m, logs = torch.split(stats, cfg.inter, dim=1)
    return x, m, logs, x_mask


# ----------------------------------------------------------------------------- duration predictor

def dds_conv(w, cfg, prefix, x, x_mask, g=None):
    """DDSConv.forward (modules.py:117-129): depthwise dilated conv -> LN -> GELU(erf) -> 1x1 ->
    LN -> GELU -> residual."""
    k = cfg.kernel_size
    if g is not None:
        x = x + g
    for i in range(cfg.dds_layers):
        dil = k ** i
        pad = (k * dil - dil) // 2
        y = _conv(w, f"{prefix}.convs_sep.{i}", x * x_mask, dilation=dil, padding=pad,
                  groups=x.shape[1])
        y = layer_norm(w, f"{prefix}.norms_1.{i}", y)
        y = F.gelu(y)
        y = _conv(w, f"{prefix}.convs_1x1.{i}", y)
        y = layer_norm(w, f"{prefix}.norms_2.{i}", y)
        y = F.gelu(y)
        x = x + y
    return x * x_mask


def rq_spline_inverse(inputs, uw, uh, ud):
    """unconstrained_rational_quadratic_spline(inverse=True, tails='linear', tail_bound=5)
    (transforms.py:50-98) over rational_quadratic_spline (transforms.py:101-191).
    inputs [...], uw/uh [...,nb], ud [...,nb-1]. Dense restatement: every position is evaluated,
    positions outside [-5,5] keep their input (the reference uses boolean-mask scatter)."""
    nb = uw.shape[-1]
    left = bottom = -TAIL_BOUND
    right = top = TAIL_BOUND
    inside = (inputs >= -TAIL_BOUND) & (inputs <= TAIL_BOUND)

    ud = F.pad(ud, (1, 1))
    constant = math.log(math.exp(1 - MIN_DERIV) - 1)
    ud[..., 0] = constant
    ud[..., -1] = constant

    widths = F.softmax(uw, dim=-1)
    widths = MIN_BIN + (1 - MIN_BIN * nb) * widths
    cumwidths = F.pad(torch.cumsum(widths, dim=-1), (1, 0), value=0.0)
    cumwidths = (right - left) * cumwidths + left
    cumwidths[..., 0] = left
    cumwidths[..., -1] = right
    widths = cumwidths[..., 1:] - cumwidths[..., :-1]

    derivatives = MIN_DERIV + F.softplus(ud)

    heights = F.softmax(uh, dim=-1)
    heights = MIN_BIN + (1 - MIN_BIN * nb) * heights
    cumheights = F.pad(torch.cumsum(heights, dim=-1), (1, 0), value=0.0)
    cumheights = (top - bottom) * cumheights + bottom
    cumheights[..., 0] = bottom
    cumheights[..., -1] = top
    heights = cumheights[..., 1:] - cumheights[..., :-1]

    # searchsorted (transforms.py:44-47): last edge += 1e-6, count of edges <= input, minus 1
    edges = cumheights.clone()
    edges[..., -1] += 1e-6
    x_in = torch.where(inside, inputs, torch.zeros_like(inputs))
    bin_idx = (torch.sum(x_in[..., None] >= edges, dim=-1) - 1)[..., None]

    def g(t):
        return t.gather(-1, bin_idx)[..., 0]

    in_cumw, in_w = g(cumwidths), g(widths)
    in_cumh, in_h = g(cumheights), g(heights)
    delta = heights / widths
    in_delta = g(delta)
    in_d = g(derivatives)
    in_d1 = g(derivatives[..., 1:])

    y = x_in - in_cumh
    s = in_d + in_d1 - 2 * in_delta
    a = y * s + in_h * (in_delta - in_d)
    b = in_h * in_d - y * s
    c = -in_delta * y
    disc = b.pow(2) - 4 * a * c
    root = (2 * c) / (-b - torch.sqrt(disc))
    out = root * in_w + in_cumw
    return torch.where(inside, out, inputs)


def conv_flow_reverse(w, cfg, prefix, z, x_mask, g):
    """ConvFlow.forward(reverse=True) (modules.py:496-527), in_channels=2 so half_channels=1."""
    H, nb = cfg.hidden, cfg.num_bins
    x0, x1 = torch.split(z, [1, 1], 1)
    h = _conv(w, prefix + ".pre", x0)
    h = dds_conv(w, cfg, prefix + ".convs", h, x_mask, g=g)
    h = _conv(w, prefix + ".proj", h) * x_mask
    b, c, t = x0.shape
    h = h.reshape(b, c, -1, t).permute(0, 1, 3, 2)
    uw = h[..., :nb] / math.sqrt(H)
    uh = h[..., nb:2 * nb] / math.sqrt(H)
    ud = h[..., 2 * nb:]
    x1 = rq_spline_inverse(x1, uw, uh, ud)
    return torch.cat([x0, x1], 1) * x_mask


def sdp_reverse(w, cfg, x, x_mask, noise_w, noise_scale_w, g=None):
    """StochasticDurationPredictor.forward(reverse=True) (models.py:63-71,108-117).
    noise_w: the N(0,1) draw of models.py:111, shape [B,2,T]."""
    from piper_amd.weights import dp_flow_indices
    x = _conv(w, "dp.pre", x)
    if g is not None:
        x = x + _conv(w, "dp.cond", g)
    x = dds_conv(w, cfg, "dp.convs", x, x_mask)
    x = _conv(w, "dp.proj", x) * x_mask
    z = noise_w.to(x.dtype) * noise_scale_w
    for fi in dp_flow_indices(cfg):
        z = torch.flip(z, [1])                                   # Flip (modules.py:385-391)
        z = conv_flow_reverse(w, cfg, f"dp.flows.{fi}", z, x_mask, g=x)
    z = torch.flip(z, [1])
    z = (z - w["dp.flows.0.m"]) * torch.exp(-w["dp.flows.0.logs"]) * x_mask  # modules.py:407-409
    return z[:, 0:1]


# ----------------------------------------------------------------------------- flow

def wn(w, cfg, prefix, x, x_mask, g=None):
    """WN.forward (modules.py:184-209), dilation_rate=1, with fused_add_tanh_sigmoid_multiply
    (commons.py:99-106)."""
    H, k = cfg.hidden, cfg.wn_kernel
    out = torch.zeros_like(x)
    if g is not None:
        g = _conv(w, prefix + ".cond_layer", g)
    for i in range(cfg.wn_layers):
        x_in = _conv(w, f"{prefix}.in_layers.{i}", x, padding=(k - 1) // 2)
        if g is not None:
            x_in = x_in + g[:, i * 2 * H:(i + 1) * 2 * H, :]
        acts = torch.tanh(x_in[:, :H]) * torch.sigmoid(x_in[:, H:])
        rs = _conv(w, f"{prefix}.res_skip_layers.{i}", acts)
        if i < cfg.wn_layers - 1:
            x = (x + rs[:, :H]) * x_mask
            out = out + rs[:, H:]
        else:
            out = out + rs
    return out * x_mask


def flow_reverse(w, cfg, z, y_mask, g=None):
    """ResidualCouplingBlock.forward(reverse=True) (models.py:247-254) over
    ResidualCouplingLayer(mean_only=True) reverse (modules.py:447-466) and Flip."""
    half = cfg.inter // 2
    for f in range(cfg.flow_n - 1, -1, -1):
        z = torch.flip(z, [1])
        p = f"flow.flows.{2 * f}"
        x0, x1 = torch.split(z, [half, half], 1)
        h = _conv(w, p + ".pre", x0) * y_mask
        h = wn(w, cfg, p + ".enc", h, y_mask, g=g)
        m = _conv(w, p + ".post", h) * y_mask
        x1 = (x1 - m) * y_mask
        z = torch.cat([x0, x1], 1)
    return z


# ----------------------------------------------------------------------------- HiFiGAN

def _same(k, d):
    return int((k * d - d) / 2)            # commons.py:17-18


def resblock(w, cfg, prefix, j, x):
    """ResBlock1.forward / ResBlock2.forward with x_mask=None (modules.py:301-314,355-364)."""
    ks = cfg.rb_kernel_sizes[j]
    dils = cfg.rb_dilations[j]
    if cfg.resblock == 1:
        for d, dil in enumerate(dils):
            xt = F.leaky_relu(x, LRELU_SLOPE)
            xt = _conv(w, f"{prefix}.convs1.{d}", xt, dilation=dil, padding=_same(ks, dil))
            xt = F.leaky_relu(xt, LRELU_SLOPE)
            xt = _conv(w, f"{prefix}.convs2.{d}", xt, dilation=1, padding=_same(ks, 1))
            x = xt + x
    else:
        for d, dil in enumerate(dils):
            xt = F.leaky_relu(x, LRELU_SLOPE)
            xt = _conv(w, f"{prefix}.convs.{d}", xt, dilation=dil, padding=_same(ks, dil))
            x = xt + x
    return x


def generator(w, cfg, x, g=None):
    """Generator.forward (models.py:348-368). Note the final leaky_relu uses PyTorch's default
    slope 0.01 (models.py:364) and conv_post has no bias (models.py:342)."""
    x = _conv(w, "dec.conv_pre", x, padding=3)
    if g is not None:
        x = x + _conv(w, "dec.cond", g)
    nk = len(cfg.rb_kernel_sizes)
    for i, (r, uk) in enumerate(zip(cfg.up_rates, cfg.up_kernel_sizes)):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = F.conv_transpose1d(x, w[f"dec.ups.{i}.weight"], w[f"dec.ups.{i}.bias"], stride=r,
                               padding=(uk - r) // 2)
        xs = None
        for j in range(nk):
            y = resblock(w, cfg, f"dec.resblocks.{i * nk + j}", j, x)
            xs = y if xs is None else xs + y
        x = xs / nk
    x = F.leaky_relu(x)
    x = _conv(w, "dec.conv_post", x, padding=3)
    return torch.tanh(x)


# ----------------------------------------------------------------------------- whole path

def generate_path(duration, mask):
    """commons.py:116-129"""
    b, _, t_y, t_x = mask.shape
    cum = torch.cumsum(duration, -1).view(b * t_x)
    path = sequence_mask(cum, t_y).to(mask.dtype).view(b, t_x, t_y)
    path = path - F.pad(path, (0, 0, 1, 0, 0, 0))[:, :-1]
    return path.unsqueeze(1).transpose(2, 3) * mask


@torch.no_grad()
def stream_chunks(w, cfg, z, chunk_frames: int, pad_frames: int, sid: Optional[int] = None):
    """The reference's chunked decode (infer_onnx_streaming.py:76-124): the latent z [inter, F] is cut into chunks of
    `chunk_frames` frames, every chunk is decoded together with `pad_frames` frames of its neighbours (as many as exist)
    and the padding's samples are trimmed; each chunk is converted to int16 on its own peak (:122). Restated without the
    script's end-pad bug (`audio[start:-wav_end_pad]` reuses the previous chunk's end pad on the LAST chunk, :107-110)
    and without its "too short to stream" shortcut, so that every chunk goes through the same path.
    Returns [(float chunk, int16 chunk), ...]."""
    if not isinstance(next(iter(w.values())), torch.Tensor):
        w = to_torch(w)
    zt = torch.as_tensor(np.asarray(z), dtype=w["enc_p.emb.weight"].dtype)[None]
    g = None
    if cfg.n_speakers > 1:
        g = F.embedding(torch.tensor([int(sid or 0)]), w["emb_g.weight"]).unsqueeze(-1)
    Fr = zt.shape[2]
    hop = int(np.prod(cfg.up_rates))
    out = []
    for s in range(0, Fr, chunk_frames):
        e = min(Fr, s + chunk_frames)
        ps, pe = min(pad_frames, s), min(pad_frames, Fr - e)
        a = generator(w, cfg, zt[:, :, s - ps:e + pe], g=g)[0, 0].numpy()
        a = a[ps * hop:a.size - pe * hop]
        out.append((a, audio_float_to_int16(a)))
    return out


@torch.no_grad()
def durations_only(w, cfg, ids, scales, noise_w=None, sid: Optional[int] = None, return_w: bool = False):
    """The integer durations ceil(exp(logw) * length_scale) of one utterance (models.py:688-703): text encoder +
    stochastic duration predictor only -- cheap enough to check EVERY utterance of a large batch. With `return_w` also
    the values in front of the ceil (a caller that checks tens of thousands of ids needs them: a value within an ulp or
    two of an integer can land on either side of it in another summation order)."""
    if not isinstance(next(iter(w.values())), torch.Tensor):
        w = to_torch(w)
    dtype = w["enc_p.emb.weight"].dtype
    ids_t = torch.as_tensor(np.asarray(ids), dtype=torch.long).view(1, -1)
    T = ids_t.shape[1]
    x, _, _, x_mask = text_encoder(w, cfg, ids_t, torch.tensor([T], dtype=torch.long))
    g = None
    if cfg.n_speakers > 1:
        g = F.embedding(torch.tensor([int(sid or 0)]), w["emb_g.weight"]).unsqueeze(-1)
    nw = torch.zeros(1, 2, T, dtype=dtype) if noise_w is None else \
        torch.as_tensor(np.asarray(noise_w), dtype=dtype).view(1, 2, -1)[:, :, :T]
    logw = sdp_reverse(w, cfg, x, x_mask, nw, float(scales[2]), g=g)
    wv = torch.exp(logw) * x_mask * float(scales[1])
    d = torch.ceil(wv)[0, 0].to(torch.int64).numpy()
    return (d, wv[0, 0].numpy()) if return_w else d


def infer_one(w, cfg, ids, scales, noise_w=None, noise_z=None, sid: Optional[int] = None,
              keep=False) -> Dict[str, torch.Tensor]:
    """One utterance through SynthesizerTrn.infer as wrapped by export_onnx.py:56-69 (B=1, like
    ``piper::synthesize``). ids: 1-D int64. scales = (noise_scale, length_scale, noise_scale_w).
    noise_w: [2,T] N(0,1) (models.py:111); noise_z: [inter, >=F] N(0,1) (models.py:718), column f
    is used for frame f. None -> zeros."""
    dtype = w["enc_p.emb.weight"].dtype
    noise_scale, length_scale, noise_scale_w = (float(s) for s in scales)
    ids_t = torch.as_tensor(np.asarray(ids), dtype=torch.long).view(1, -1)
    T = ids_t.shape[1]
    lengths = torch.tensor([T], dtype=torch.long)
    x, m_p, logs_p, x_mask = text_encoder(w, cfg, ids_t, lengths)
    g = None
    if cfg.n_speakers > 1:
        g = F.embedding(torch.tensor([int(sid or 0)]), w["emb_g.weight"]).unsqueeze(-1)
    nw = torch.zeros(1, 2, T, dtype=dtype) if noise_w is None else \
        torch.as_tensor(np.asarray(noise_w), dtype=dtype).view(1, 2, -1)[:, :, :T]
    logw = sdp_reverse(w, cfg, x, x_mask, nw, noise_scale_w, g=g)
    wdur = torch.exp(logw) * x_mask * length_scale
    w_ceil = torch.ceil(wdur)
    y_len = torch.clamp_min(torch.sum(w_ceil, [1, 2]), 1).long()
    Fr = int(y_len.max())
    y_mask = sequence_mask(y_len, Fr).unsqueeze(1).to(dtype)
    attn_mask = x_mask.unsqueeze(2) * y_mask.unsqueeze(-1)
    attn = generate_path(w_ceil, attn_mask)
    m_pe = torch.matmul(attn.squeeze(1), m_p.transpose(1, 2)).transpose(1, 2)
    logs_pe = torch.matmul(attn.squeeze(1), logs_p.transpose(1, 2)).transpose(1, 2)
    nz = torch.zeros(1, cfg.inter, Fr, dtype=dtype) if noise_z is None else \
        torch.as_tensor(np.asarray(noise_z), dtype=dtype)[None, :, :Fr]
    z_p = m_pe + nz * torch.exp(logs_pe) * noise_scale
    z = flow_reverse(w, cfg, z_p, y_mask, g=g)
    o = generator(w, cfg, z * y_mask, g=g)
    res = {"audio": o[0, 0], "durations": w_ceil[0, 0].to(torch.int64), "frames": Fr}
    if keep:
        res.update(x_enc=x[0], m_p=m_p[0], logs_p=logs_p[0], logw=logw[0, 0], z_p=z_p[0], z=z[0])
    return res


def audio_float_to_int16(audio: np.ndarray) -> np.ndarray:
    """piper.cpp:410-431 == util.py:5-12: peak-normalise with floor 0.01, clamp, truncating cast."""
    audio = np.asarray(audio, dtype=np.float32)
    mx = np.float32(0.01)
    if audio.size:
        mx = max(mx, np.float32(np.max(np.abs(audio))))
    scale = np.float32(32767.0) / mx
    y = np.clip(audio * scale, np.float32(-32768.0), np.float32(32767.0))
    return y.astype(np.int16)


def to_torch(weights: Dict[str, np.ndarray], dtype=torch.float32) -> Dict[str, torch.Tensor]:
    return {k: torch.as_tensor(np.asarray(v)).to(dtype) for k, v in weights.items()}


@torch.no_grad()
def synthesize(weights, cfg, ids, scales=(0.667, 1.0, 0.8), noise_w=None, noise_z=None, sid=None,
               dtype=torch.float32, keep=False):
    """Convenience wrapper: numpy in, numpy out. Returns dict with float audio, int16 pcm,
    integer durations and (keep=True) the per-stage tensors."""
    w = weights if isinstance(next(iter(weights.values())), torch.Tensor) else to_torch(weights, dtype)
    r = infer_one(w, cfg, ids, scales, noise_w, noise_z, sid, keep=keep)
    out = {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in r.items()}
    out["pcm"] = audio_float_to_int16(out["audio"].astype(np.float32))
    return out
