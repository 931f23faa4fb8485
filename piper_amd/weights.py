"""Voice weights: architecture description, canonical tensor names, synthetic voices, weight blob.

The canonical tensor names are the reference's ``SynthesizerTrn.state_dict()`` keys for the
modules reachable from ``infer()`` (reference ``src/python/piper_train/vits/models.py:681-722``)
with every ``weight_norm`` folded into a plain ``.weight`` (the exporter does this for ``dec`` --
``export_onnx.py:51-52`` -- and ONNX constant folding does it for the flow's ``WN`` layers).

Numpy only: this module travels to the GPU box and must not need the reference or torch.

Blob layout (little endian), parsed by ``piper_amd/csrc/weights.cpp``::

    char     magic[8] = "PEBLOB01"
    int32    arch[64]                 (ARCH_* indices below)
    uint32   n_tensors, uint32 reserved
    n_tensors x { char name[96]; int32 ndim; int32 dims[4]; uint32 pad; uint64 offset; uint64 numel }
    raw float32 data, every tensor 64-byte aligned (offset counted from the blob start)
"""
from __future__ import annotations

import dataclasses
import struct
from typing import Dict, List, Sequence, Tuple

import numpy as np

MAGIC = b"PEBLOB01"
ARCH_INTS = 64
NAME_BYTES = 96
REC_BYTES = NAME_BYTES + 4 + 16 + 4 + 8 + 8  # 136
MAX_UPS = 8
MAX_RB = 4
MAX_DIL = 4

# ints in the arch header
(A_NVOCAB, A_HIDDEN, A_INTER, A_FILTER, A_NHEADS, A_NLAYERS, A_KSIZE, A_WINDOW, A_RESBLOCK,
 A_NRB) = range(10)
A_RBK0 = 10          # [4]
A_NDIL = 14
A_RBDIL0 = 15        # [4][4]
A_NUPS = 31
A_UPR0 = 32          # [8]
A_UPK0 = 40          # [8]
A_UPINIT = 48
A_NSPK = 49
A_GIN = 50
A_SR = 51
A_DPFLOWS = 52
A_DDSLAYERS = 53
A_NBINS = 54
A_FLOWN = 55
A_WNLAYERS = 56
A_WNK = 57


@dataclasses.dataclass(frozen=True)
class ArchConfig:
    """Hyper-parameters of one voice (reference ``vits/lightning.py:26-58``, ``__main__.py:68-82``)."""

    n_vocab: int = 256
    hidden: int = 192          # hidden_channels (text encoder, dp, WN)
    inter: int = 192           # inter_channels (latent z)
    filter: int = 768          # FFN filter_channels
    n_heads: int = 2
    n_layers: int = 6
    kernel_size: int = 3       # FFN / DDSConv kernel
    window: int = 4            # relative attention window (attentions.py:22)
    resblock: int = 2          # 1 = ResBlock1 (high), 2 = ResBlock2
    rb_kernel_sizes: Tuple[int, ...] = (3, 5, 7)
    rb_dilations: Tuple[Tuple[int, ...], ...] = ((1, 2), (2, 6), (3, 12))
    up_rates: Tuple[int, ...] = (8, 8, 4)
    up_kernel_sizes: Tuple[int, ...] = (16, 16, 8)
    up_initial: int = 256
    n_speakers: int = 1
    gin: int = 0
    sample_rate: int = 22050
    # fixed by the reference's constructors (models.py:594-608, 43-55)
    dp_flows: int = 4
    dds_layers: int = 3
    num_bins: int = 10
    flow_n: int = 4
    wn_layers: int = 4
    wn_kernel: int = 5

    @property
    def hop(self) -> int:
        h = 1
        for r in self.up_rates:
            h *= r
        return h

    def to_ints(self) -> List[int]:
        a = [0] * ARCH_INTS
        a[A_NVOCAB] = self.n_vocab
        a[A_HIDDEN] = self.hidden
        a[A_INTER] = self.inter
        a[A_FILTER] = self.filter
        a[A_NHEADS] = self.n_heads
        a[A_NLAYERS] = self.n_layers
        a[A_KSIZE] = self.kernel_size
        a[A_WINDOW] = self.window
        a[A_RESBLOCK] = self.resblock
        a[A_NRB] = len(self.rb_kernel_sizes)
        for i, k in enumerate(self.rb_kernel_sizes):
            a[A_RBK0 + i] = k
        a[A_NDIL] = len(self.rb_dilations[0])
        for i, ds in enumerate(self.rb_dilations):
            assert len(ds) == a[A_NDIL]
            for j, d in enumerate(ds):
                a[A_RBDIL0 + i * MAX_DIL + j] = d
        a[A_NUPS] = len(self.up_rates)
        for i, (r, k) in enumerate(zip(self.up_rates, self.up_kernel_sizes)):
            a[A_UPR0 + i] = r
            a[A_UPK0 + i] = k
        a[A_UPINIT] = self.up_initial
        a[A_NSPK] = self.n_speakers
        a[A_GIN] = self.gin
        a[A_SR] = self.sample_rate
        a[A_DPFLOWS] = self.dp_flows
        a[A_DDSLAYERS] = self.dds_layers
        a[A_NBINS] = self.num_bins
        a[A_FLOWN] = self.flow_n
        a[A_WNLAYERS] = self.wn_layers
        a[A_WNK] = self.wn_kernel
        return a

    @staticmethod
    def from_ints(a: Sequence[int]) -> "ArchConfig":
        nrb, ndil, nups = a[A_NRB], a[A_NDIL], a[A_NUPS]
        return ArchConfig(
            n_vocab=a[A_NVOCAB], hidden=a[A_HIDDEN], inter=a[A_INTER], filter=a[A_FILTER],
            n_heads=a[A_NHEADS], n_layers=a[A_NLAYERS], kernel_size=a[A_KSIZE], window=a[A_WINDOW],
            resblock=a[A_RESBLOCK],
            rb_kernel_sizes=tuple(a[A_RBK0 + i] for i in range(nrb)),
            rb_dilations=tuple(tuple(a[A_RBDIL0 + i * MAX_DIL + j] for j in range(ndil))
                               for i in range(nrb)),
            up_rates=tuple(a[A_UPR0 + i] for i in range(nups)),
            up_kernel_sizes=tuple(a[A_UPK0 + i] for i in range(nups)),
            up_initial=a[A_UPINIT], n_speakers=a[A_NSPK], gin=a[A_GIN], sample_rate=a[A_SR],
            dp_flows=a[A_DPFLOWS], dds_layers=a[A_DDSLAYERS], num_bins=a[A_NBINS],
            flow_n=a[A_FLOWN], wn_layers=a[A_WNLAYERS], wn_kernel=a[A_WNK])


def preset(name: str, **over) -> ArchConfig:
    """Named architectures. 'medium'/'high'/'x-low' are the reference's qualities
    (``piper_train/__main__.py:68-82``); 'tiny' is a test-only shrink of 'medium'."""
    if name == "medium":
        c = ArchConfig()
    elif name == "x-low":
        c = ArchConfig(hidden=96, inter=96, filter=384, n_vocab=130, sample_rate=16000)
    elif name == "high":
        c = ArchConfig(resblock=1, rb_kernel_sizes=(3, 7, 11),
                       rb_dilations=((1, 3, 5), (1, 3, 5), (1, 3, 5)),
                       up_rates=(8, 8, 2, 2), up_kernel_sizes=(16, 16, 4, 4), up_initial=512)
    elif name == "tiny":
        c = ArchConfig(n_vocab=40, hidden=32, inter=32, filter=64, n_layers=2, up_initial=64,
                       sample_rate=16000)
    elif name == "tiny-high":
        c = ArchConfig(n_vocab=40, hidden=32, inter=32, filter=64, n_layers=2, up_initial=64,
                       resblock=1, rb_kernel_sizes=(3, 7, 11),
                       rb_dilations=((1, 3, 5), (1, 3, 5), (1, 3, 5)),
                       up_rates=(8, 8, 2, 2), up_kernel_sizes=(16, 16, 4, 4), sample_rate=16000)
    elif name == "tiny-high-ms":
        c = ArchConfig(n_vocab=40, hidden=32, inter=32, filter=64, n_layers=2, up_initial=64,
                       resblock=1, rb_kernel_sizes=(3, 7, 11),
                       rb_dilations=((1, 3, 5), (1, 3, 5), (1, 3, 5)),
                       up_rates=(8, 8, 2, 2), up_kernel_sizes=(16, 16, 4, 4), n_speakers=4, gin=16,
                       sample_rate=16000)
    elif name == "tiny-ms":
        c = ArchConfig(n_vocab=40, hidden=32, inter=32, filter=64, n_layers=2, up_initial=64,
                       n_speakers=4, gin=16, sample_rate=16000)
    else:
        raise ValueError(f"unknown preset {name!r}")
    return dataclasses.replace(c, **over) if over else c


# ---------------------------------------------------------------------------------------------
# canonical tensor list
# ---------------------------------------------------------------------------------------------

def dp_flow_indices(cfg: ArchConfig) -> List[int]:
    """Module indices of the ConvFlows that ``StochasticDurationPredictor.forward(reverse=True)``
    actually runs, in execution order (models.py:108-110: flows reversed, "useless vflow" dropped):
    for n_flows=4 -> dp.flows.{7,5,3}."""
    return [2 * i + 1 for i in range(cfg.dp_flows - 1, 0, -1)]


def tensor_specs(cfg: ArchConfig) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(name, shape, init-kind) for every tensor the synthesis path reads, in canonical order."""
    H, C, FC, k = cfg.hidden, cfg.inter, cfg.filter, cfg.kernel_size
    dk = H // cfg.n_heads
    s: List[Tuple[str, Tuple[int, ...], str]] = []

    def conv(name, co, ci, ks, bias=True, kind="conv"):
        s.append((name + ".weight", (co, ci, ks), kind))
        if bias:
            s.append((name + ".bias", (co,), "bias"))

    def ln(name, ch):
        s.append((name + ".gamma", (ch,), "gamma"))
        s.append((name + ".beta", (ch,), "beta"))

    def dds(prefix):
        for i in range(cfg.dds_layers):
            s.append((f"{prefix}.convs_sep.{i}.weight", (H, 1, k), "dw"))
            s.append((f"{prefix}.convs_sep.{i}.bias", (H,), "bias"))
            conv(f"{prefix}.convs_1x1.{i}", H, H, 1)
            ln(f"{prefix}.norms_1.{i}", H)
            ln(f"{prefix}.norms_2.{i}", H)

    # --- text encoder (models.py:168-209, attentions.py:12-74)
    s.append(("enc_p.emb.weight", (cfg.n_vocab, H), "emb"))
    for l in range(cfg.n_layers):
        p = f"enc_p.encoder.attn_layers.{l}"
        s.append((p + ".emb_rel_k", (1, 2 * cfg.window + 1, dk), "rel"))
        s.append((p + ".emb_rel_v", (1, 2 * cfg.window + 1, dk), "rel"))
        for n in ("conv_q", "conv_k", "conv_v", "conv_o"):
            conv(f"{p}.{n}", H, H, 1)
        ln(f"enc_p.encoder.norm_layers_1.{l}", H)
        conv(f"enc_p.encoder.ffn_layers.{l}.conv_1", FC, H, k)
        conv(f"enc_p.encoder.ffn_layers.{l}.conv_2", H, FC, k)
        ln(f"enc_p.encoder.norm_layers_2.{l}", H)
    conv("enc_p.proj", 2 * C, H, 1)

    # --- speaker embedding (models.py:614-615)
    if cfg.n_speakers > 1:
        s.append(("emb_g.weight", (cfg.n_speakers, cfg.gin), "emb_g"))

    # --- stochastic duration predictor, reverse path only (models.py:63-71,108-117)
    conv("dp.pre", H, H, 1)
    if cfg.gin:
        conv("dp.cond", H, cfg.gin, 1)
    dds("dp.convs")
    conv("dp.proj", H, H, 1)
    s.append(("dp.flows.0.m", (2, 1), "ea_m"))
    s.append(("dp.flows.0.logs", (2, 1), "ea_logs"))
    for fi in dp_flow_indices(cfg):
        p = f"dp.flows.{fi}"
        conv(p + ".pre", H, 1, 1)
        dds(p + ".convs")
        conv(p + ".proj", 3 * cfg.num_bins - 1, H, 1, kind="spline_proj")

    # --- residual coupling flow (models.py:212-254, modules.py:412-466,132-209)
    for f in range(cfg.flow_n):
        p = f"flow.flows.{2 * f}"
        conv(p + ".pre", H, C // 2, 1)
        if cfg.gin:
            conv(p + ".enc.cond_layer", 2 * H * cfg.wn_layers, cfg.gin, 1)
        for i in range(cfg.wn_layers):
            conv(f"{p}.enc.in_layers.{i}", 2 * H, H, cfg.wn_kernel)
            rs = 2 * H if i < cfg.wn_layers - 1 else H
            conv(f"{p}.enc.res_skip_layers.{i}", rs, H, 1)
        conv(p + ".post", C // 2, H, 1, kind="post")

    # --- HiFiGAN generator (models.py:299-375, modules.py:220-368)
    U = cfg.up_initial
    conv("dec.conv_pre", U, C, 7)
    if cfg.gin:
        conv("dec.cond", U, cfg.gin, 1)
    ch = U
    for i, (r, uk) in enumerate(zip(cfg.up_rates, cfg.up_kernel_sizes)):
        cin, ch = U // (2 ** i), U // (2 ** (i + 1))
        s.append((f"dec.ups.{i}.weight", (cin, ch, uk), "convT"))   # ConvTranspose1d: [Cin,Cout,k]
        s.append((f"dec.ups.{i}.bias", (ch,), "bias"))
        for j, ks in enumerate(cfg.rb_kernel_sizes):
            rb = f"dec.resblocks.{i * len(cfg.rb_kernel_sizes) + j}"
            nd = len(cfg.rb_dilations[j])
            if cfg.resblock == 1:
                for d in range(nd):
                    conv(f"{rb}.convs1.{d}", ch, ch, ks)
                for d in range(nd):
                    conv(f"{rb}.convs2.{d}", ch, ch, ks)
            else:
                for d in range(nd):
                    conv(f"{rb}.convs.{d}", ch, ch, ks)
    conv("dec.conv_post", 1, ch, 7, bias=False)
    return s


def synthetic_weights(cfg: ArchConfig, seed: int = 1234, family: str = "gauss") -> Dict[str, np.ndarray]:
    """Seeded random voice with every tensor non-trivial (the reference zero-initialises
    ``post``/``proj`` convs -- modules.py:443-444,519-520 -- which would make parity tests blind
    to the coupling and spline arithmetic). Scales are chosen so activations stay O(1) and
    durations land near the 2.5-3 frames/id of real voices.

    ``family="heavy"`` (parity tests only): the same layer variances drawn from a heavy-tailed law (Student-t, 3
    degrees of freedom: weights of 10+ standard deviations occur in every large tensor) with a log-normal gain per
    output channel (sigma 0.6, i.e. channels a factor ~4 apart) and 4x larger biases -- closer to the dynamic range
    of trained, weight-normed layers than i.i.d. Gaussians are. Real voices are not available offline."""
    if family not in ("gauss", "heavy"):
        raise ValueError(f"unknown weight family {family!r}")
    rng = np.random.default_rng(seed)
    heavy = family == "heavy"

    def draw(shape, std):
        if not heavy:
            return rng.standard_normal(shape) * std
        a = rng.standard_t(3, size=shape) / np.sqrt(3.0)               # unit variance
        if len(shape) >= 2:                                            # per-output-channel gain, mean square 1
            g = np.exp(0.6 * rng.standard_normal(shape[0]))
            g /= np.sqrt(np.mean(g * g))
            a = a * g.reshape((-1,) + (1,) * (len(shape) - 1))
        return a * std

    w: Dict[str, np.ndarray] = {}
    for name, shape, kind in tensor_specs(cfg):
        if kind in ("conv", "dw", "convT"):
            if kind == "convT":
                fan_in = shape[0] * shape[2] / max(1, _stride_of(cfg, name))
            else:
                fan_in = shape[1] * shape[2]
            a = draw(shape, 0.8 / np.sqrt(fan_in))
        elif kind == "post":
            a = draw(shape, 0.5 / np.sqrt(shape[1]))
        elif kind == "spline_proj":
            a = draw(shape, 2.0 / np.sqrt(shape[1]))
        elif kind == "bias":
            a = rng.standard_normal(shape) * (0.2 if heavy else 0.05)
        elif kind == "gamma":
            a = 1.0 + rng.standard_normal(shape) * (0.3 if heavy else 0.1)
        elif kind == "beta":
            a = rng.standard_normal(shape) * (0.4 if heavy else 0.1)
        elif kind == "emb":
            a = draw(shape, shape[1] ** -0.5)
        elif kind == "emb_g":
            a = rng.standard_normal(shape) * 0.3
        elif kind == "rel":
            a = rng.standard_normal(shape) * (shape[2] ** -0.5)
        elif kind == "ea_m":
            a = np.array([[-2.0], [0.2]])
        elif kind == "ea_logs":
            a = np.array([[1.0], [-0.1]])
        else:
            raise AssertionError(kind)
        w[name] = np.ascontiguousarray(a, dtype=np.float32)
    return w


def _stride_of(cfg: ArchConfig, name: str) -> int:
    i = int(name.split(".")[2])
    return cfg.up_rates[i]


# ---------------------------------------------------------------------------------------------
# blob
# ---------------------------------------------------------------------------------------------

def pack_blob(cfg: ArchConfig, weights: Dict[str, np.ndarray]) -> bytes:
    specs = tensor_specs(cfg)
    n = len(specs)
    head = len(MAGIC) + 4 * ARCH_INTS + 8 + n * REC_BYTES
    off = (head + 63) // 64 * 64
    recs = []
    chunks = []
    for name, shape, _ in specs:
        a = np.ascontiguousarray(weights[name], dtype=np.float32)
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"{name}: shape {a.shape} != expected {shape}")
        nb = name.encode()
        if len(nb) >= NAME_BYTES:
            raise ValueError("tensor name too long: " + name)
        dims = list(shape) + [1] * (4 - len(shape))
        recs.append(struct.pack(f"<{NAME_BYTES}si4iIQQ", nb, len(shape), *dims, 0, off, a.size))
        chunks.append((off, a.tobytes()))
        off = (off + a.nbytes + 63) // 64 * 64
    out = bytearray(off)
    p = 0
    out[p:p + 8] = MAGIC
    p += 8
    out[p:p + 4 * ARCH_INTS] = struct.pack(f"<{ARCH_INTS}i", *cfg.to_ints())
    p += 4 * ARCH_INTS
    out[p:p + 8] = struct.pack("<II", n, 0)
    p += 8
    for r in recs:
        out[p:p + REC_BYTES] = r
        p += REC_BYTES
    for o, b in chunks:
        out[o:o + len(b)] = b
    return bytes(out)


def unpack_blob(blob: bytes) -> Tuple[ArchConfig, Dict[str, np.ndarray]]:
    if blob[:8] != MAGIC:
        raise ValueError("not a PEBLOB01 blob")
    p = 8
    arch = struct.unpack_from(f"<{ARCH_INTS}i", blob, p)
    p += 4 * ARCH_INTS
    n, _ = struct.unpack_from("<II", blob, p)
    p += 8
    cfg = ArchConfig.from_ints(arch)
    w: Dict[str, np.ndarray] = {}
    for _ in range(n):
        nb, ndim, d0, d1, d2, d3, _pad, off, numel = struct.unpack_from(
            f"<{NAME_BYTES}si4iIQQ", blob, p)
        p += REC_BYTES
        shape = (d0, d1, d2, d3)[:ndim]
        w[nb.rstrip(b"\0").decode()] = np.frombuffer(blob, np.float32, numel, off).reshape(shape).copy()
    return cfg, w


def synthetic_phoneme_ids(T: int, index: int = 0, id_max: int = 129) -> np.ndarray:
    """Fixed-length id sequence shaped like piper-phonemize output, ``[1,0,p1,0,p2,0,...,2]``
    (BOS=1, PAD=0 interspersed, EOS=2 -- reference piper.hpp:44-47)."""
    rng = np.random.default_rng(1234 + index)
    ids = [1, 0]
    while len(ids) < T - 1:
        ids.append(int(rng.integers(3, id_max + 1)))
        ids.append(0)
    ids = ids[:T - 1]
    ids.append(2)
    return np.asarray(ids, dtype=np.int64)
