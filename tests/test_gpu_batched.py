"""GPU parity tests of the BATCHED / TILED kernel family (run with -m gpu on an MI355X).

tests/test_gpu_parity.py runs one utterance (or a few tiny ones) per call, which the launcher routes to the
split-K kernels; the throughput configurations of BASELINE.json (configs[2]: high, 64 x 128 ids; configs[3]: medium,
64 utterances per GPU) run through the tiled conv_mfma_kernel instantiations instead. Every test here

  * compares the HIP path (through the C ABI) with the CPU oracle on sampled utterances of the batch:
    integer durations exact, float waveform max |d| < 2e-4, int16 PCM RMS <= 1e-3 (north_star tolerance);
  * records which kernel instantiations the engine launched (level-2 profile rows carry the instantiation name as
    rocprofv3 prints it) so that the last test can assert that every instantiation listed in the committed
    profiles/*kernel_stats.csv has been compared with the oracle in this session.
"""
import csv
import glob
import json
import os

import numpy as np
import pytest

from piper_amd import weights as W

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RMS_TOL = 1e-3
TIGHT_AUDIO_TOL = 2e-4
SCALES = (0.667, 1.0, 0.8)

SEEN = set()          # kernel instantiations whose output has been compared with the oracle in this session
_weights = {}


def voice(preset, seed=1234, **over):
    key = (preset, seed, tuple(sorted(over.items())))
    if key not in _weights:
        cfg = W.preset(preset, **over)
        _weights[key] = (cfg, W.synthetic_weights(cfg, seed))
    return _weights[key]


def make_engine(monkeypatch, cfg, w, env=None):
    from piper_amd.engine import Engine
    import json
    from piper_amd import _lib as L
    # every launch-policy knob the engine reads (piper_amd/csrc/policy.h) + the string-valued matrix mode
    for k in [x["env"] for x in json.loads(L.get_lib().pe_policy_describe().decode())] + ["PIPER_HIP_MATRIX"]:
        monkeypatch.delenv(k, raising=False)
    for k, v in (env or {}).items():
        monkeypatch.setenv(k, str(v))
    return Engine(blob=W.pack_blob(cfg, w), device=0)     # the knobs are read once, at engine creation


def batch_inputs(cfg, lens, seed, zcols=None):
    id_max = min(cfg.n_vocab - 1, 129)
    ids = [W.synthetic_phoneme_ids(T, 100 * seed + i, id_max=id_max) if T > 2 else np.array([1, 2][:T], np.int64)
           for i, T in enumerate(lens)]
    Tm = max(lens)
    rng = np.random.default_rng(seed)
    nw = rng.standard_normal((len(lens), 2, Tm)).astype(np.float32)
    nz = rng.standard_normal((len(lens), cfg.inter, zcols or (6 * Tm + 64))).astype(np.float32)
    return ids, nw, nz


def pcm_rms(a, b):
    d = (a.astype(np.float64) - b.astype(np.float64)) / 32767.0
    return float(np.sqrt(np.mean(d * d))) if d.size else 0.0


_ORACLE = {}          # (cache_key, utterance) -> oracle result: the split-mode tests compare the SAME inputs as the f32 ones


def run_and_check(eng, cfg, w, ids, nw, nz, sample, scales=SCALES, sids=None, audio_tol=TIGHT_AUDIO_TOL, stats=None,
                  cache_key=None):
    """One profiled batched call; utterances `sample` are compared with the oracle, all of them with the
    size-independent properties (sample count = frames * hop = sum of durations * hop, peak-normalised PCM)."""
    from oracle import vits_oracle as O
    eng.profile_enable(2)
    eng.profile_reset()
    r = eng.synthesize_batch(ids, scales, sids=sids, noise_w=nw, noise_z=nz)
    names = {row["name"] for row in eng.profile()[5:] if row["launches"]}
    eng.profile_enable(0)
    durs = eng.durations()
    off = np.concatenate([[0], np.cumsum([len(x) for x in ids])])
    wt = O.to_torch(w)
    for i in range(len(ids)):
        d = durs[off[i]:off[i + 1]]
        assert int(r.frames[i]) == max(int(d.sum()), 1)
        assert r.pcm[i].size == int(r.frames[i]) * eng.hop == r.audio[i].size
        # integer work is bit-exact, for EVERY utterance: the durations against the oracle's encoder + duration predictor
        # (f32 there and here, also in matrix mode bf16x3), and the int16 conversion of the engine's own float waveform
        # against the oracle's restatement of piper.cpp:410-431 -- 0 LSB
        assert np.array_equal(d, O.durations_only(wt, cfg, ids[i], scales, nw[i], None if sids is None else sids[i])), \
            f"utterance {i}: durations differ from the oracle"
        assert np.array_equal(O.audio_float_to_int16(r.audio[i]), r.pcm[i]), f"utterance {i}: int16 conversion not bit-exact"
        # peak-normalised (piper.cpp:410-431): the loudest sample maps to 32767 (32766 when peak * (32767 / peak)
        # rounds just below 32767 before the truncating cast)
        assert np.max(np.abs(r.pcm[i].astype(np.int32))) >= 32766 or np.max(np.abs(r.audio[i])) < 0.01
    worst = 0.0
    for i in sample:
        o = _ORACLE.get((cache_key, i)) if cache_key else None
        if o is None:
            o = O.synthesize(wt, cfg, ids[i], scales, nw[i], nz[i], sid=None if sids is None else sids[i])
            if cache_key:
                _ORACLE[(cache_key, i)] = {k: o[k] for k in ("durations", "audio", "pcm")}
        assert np.array_equal(durs[off[i]:off[i + 1]], o["durations"]), f"utterance {i}: durations differ"
        assert r.audio[i].shape == o["audio"].shape
        d = float(np.max(np.abs(r.audio[i] - o["audio"])))
        worst = max(worst, d)
        assert d < audio_tol, f"utterance {i}: max |d audio| = {d}"
        rms = pcm_rms(r.pcm[i], o["pcm"])
        assert rms <= RMS_TOL
        if stats is not None:
            stats.append((d, rms))
    SEEN.update(names)
    return names, worst


def test_high_b64_t128_matches_oracle(monkeypatch):
    """BASELINE.json configs[2]: en_US-lessac-high architecture, batch of 64 fixed 128-id utterances."""
    cfg, w = voice("high")
    eng = make_engine(monkeypatch, cfg, w)
    ids, nw, nz = batch_inputs(cfg, [128] * 64, seed=31)
    names, worst = run_and_check(eng, cfg, w, ids, nw, nz, sample=range(64), cache_key="high64")
    eng.close()
    assert any(n.startswith("conv_mfma_kernel<") and ",true," in n for n in names), names    # tiled WN gate conv
    assert any(n.startswith("conv_mfma_kernel<2,2,1,1,16,false,") for n in names), names
    print("high B=64 kernels:", sorted(names), "worst |d audio| %.2e" % worst)


def test_medium_b64_t128_matches_oracle(monkeypatch):
    """BASELINE.json configs[3], one GPU's share: en_US-lessac-medium architecture, 64 utterances x 128 ids."""
    cfg, w = voice("medium")
    eng = make_engine(monkeypatch, cfg, w)
    ids, nw, nz = batch_inputs(cfg, [128] * 64, seed=32)
    names, worst = run_and_check(eng, cfg, w, ids, nw, nz, sample=range(64), cache_key="medium64")
    eng.close()
    assert "conv_mfma_kernel<2,2,2,1,16,true,64>" in names, names
    assert {"conv_mfma_kernel<2,2,1,1,16,false,64>", "conv_mfma_kernel<2,2,1,1,16,false,128>"} <= names, names
    print("medium B=64 kernels:", sorted(names), "worst |d audio| %.2e" % worst)


SPLIT_MODES = {"bf16x3": 0, "f16x3": 1, "bf16x6": 2}          # PIPER_HIP_MATRIX -> conv_split_kernel's SM


@pytest.mark.parametrize("mode", sorted(SPLIT_MODES))
@pytest.mark.parametrize("preset,lens,seed", [
    ("medium", [128, 3, 77, 128, 1, 50, 128, 19, 101, 64, 128, 33, 90, 2, 128, 111], 41),
    ("high", [128, 40, 128, 97, 128, 5, 128, 128], 42),
    ("x-low", [64, 128, 9, 128, 77, 128, 128, 30, 128, 128, 128, 128], 43),
])
def test_split_matrix_modes_match_oracle(monkeypatch, preset, lens, seed, mode):
    """Opt-in matrix modes PIPER_HIP_MATRIX=bf16x3 | f16x3 | bf16x6 (kernels/conv_bf3.h: conv_split_kernel): the tiled flow /
    generator convs as 3 / 3 / 6 sixteen-bit MFMAs on split f32 operands. Gate = the f32 path's OWN:
    integer durations EQUAL to the oracle's (the text encoder and duration predictor stay f32), max |d audio| < 2e-4 on the
    float waveform, int16 PCM within 1e-3 RMS -- for all three modes (bf16x3 keeps 16 significand bits per operand and lands
    near 1e-5; f16x3 keeps 22, bf16x6 all 24: both land at the f32 kernels' own ~1e-6). The fused stage kernels are switched
    off (PIPER_HIP_BF3_MINF=0, PIPER_HIP_MRF_SPLIT=0) so that every generator stage goes through conv_split_kernel."""
    cfg, w = voice(preset)
    eng = make_engine(monkeypatch, cfg, w, {"PIPER_HIP_MATRIX": mode, "PIPER_HIP_BF3_MINF": 0, "PIPER_HIP_MRF_SPLIT": 0})
    ids, nw, nz = batch_inputs(cfg, lens, seed=seed)
    stats = []
    sample = [i for i in range(len(lens))][:8]
    names, worst = run_and_check(eng, cfg, w, ids, nw, nz, sample=sample, audio_tol=TIGHT_AUDIO_TOL, stats=stats,
                                 cache_key=f"split-{preset}")
    eng.close()
    pre = f"conv_split_kernel<{SPLIT_MODES[mode]},"
    assert any(n.startswith(pre) and ",true," in n for n in names), names
    assert any(n.startswith(pre) and ",false," in n for n in names), names
    assert not any(n.startswith("mrf_") and "sum" not in n for n in names), names
    print(preset, mode, "kernels:", sorted(n for n in names if "split" in n),
          "worst |d audio| %.2e, worst pcm rms %.2e" % (worst, max(s[1] for s in stats)))


@pytest.mark.parametrize("mode", ["f16x3", "bf16x6"])
@pytest.mark.parametrize("preset,T,chunk", [("medium", 96, 45), ("high", 40, 45)])
def test_split_matrix_modes_streaming_chunks_equal_unchunked(monkeypatch, preset, T, chunk, mode):
    """BASELINE configs[4] in the near-exact matrix modes: the exact-halo chunked decode (pe_stream_*) still concatenates to
    the unchunked waveform of the same mode (every output column is the same split arithmetic whatever window it sits in),
    and that waveform is inside the f32 gate against the oracle."""
    from oracle import vits_oracle as O
    cfg, w = voice(preset)
    eng = make_engine(monkeypatch, cfg, w, {"PIPER_HIP_MATRIX": mode})
    ids = W.synthetic_phoneme_ids(T, 5, id_max=min(cfg.n_vocab - 1, 129))
    rng = np.random.default_rng(13)
    nw = rng.standard_normal((2, T)).astype(np.float32)
    nz = rng.standard_normal((cfg.inter, 16 * T + 64)).astype(np.float32)
    full = eng.synthesize(ids, SCALES, noise_w=nw, noise_z=nz)
    chunks = list(eng.stream(ids, SCALES, chunk_frames=chunk, noise_w=nw, noise_z=nz))
    eng.close()
    cat = np.concatenate([c[0] for c in chunks])
    assert cat.shape == full.audio[0].shape and len(chunks) == -(-int(full.frames[0]) // chunk)
    assert np.max(np.abs(cat - full.audio[0])) < 2e-5
    o = O.synthesize(w, cfg, ids, SCALES, nw, nz)
    assert np.array_equal(o["durations"].sum(), int(full.frames[0])) or int(o["frames"]) == int(full.frames[0])
    assert np.max(np.abs(full.audio[0] - o["audio"])) < TIGHT_AUDIO_TOL


@pytest.mark.parametrize("mode", sorted(SPLIT_MODES))
@pytest.mark.parametrize("preset,lens", [("medium", [128, 61, 9]), ("high", [64, 17])])
def test_split_matrix_modes_on_heavy_tailed_weights(monkeypatch, preset, lens, mode):
    """The split modes on the heavy-tailed weight family (Student-t weights of 10+ standard deviations, per-channel gains a
    factor ~4 apart, 4x biases: the dynamic range of trained, weight-normed layers rather than of i.i.d. Gaussians) -- what
    f16x3's power-of-two weight scaling and its +-65504 activation clamp exist for: same gate as the f32 path."""
    cfg = W.preset(preset)
    w = W.synthetic_weights(cfg, 4321, family="heavy")
    eng = make_engine(monkeypatch, cfg, w, {"PIPER_HIP_MATRIX": mode})
    ids, nw, nz = batch_inputs(cfg, lens, seed=77)
    stats = []
    names, worst = run_and_check(eng, cfg, w, ids, nw, nz, sample=range(len(lens)), audio_tol=TIGHT_AUDIO_TOL, stats=stats,
                                 cache_key=f"heavy-{preset}")
    eng.close()
    assert any("_split_kernel<" in n for n in names), names
    print(preset, "heavy family", mode, "worst |d audio| %.2e, worst pcm rms %.2e" % (worst, max(s[1] for s in stats)))


@pytest.mark.parametrize("mode", sorted(SPLIT_MODES))
@pytest.mark.parametrize("preset,seed,key", [("high", 31, "high64"), ("medium", 32, "medium64")])
def test_split_matrix_modes_at_baseline_sizes(monkeypatch, preset, seed, key, mode):
    """The same gate at BASELINE.json configs[2] (high, 64 x 128 ids) and configs[3]'s per-GPU share (medium, 64 x 128): ALL
    64 waveforms of each against the oracle at the f32 path's tolerance, in the engine's default policy for the mode (what
    the bench legs time)."""
    cfg, w = voice(preset)
    eng = make_engine(monkeypatch, cfg, w, {"PIPER_HIP_MATRIX": mode})
    ids, nw, nz = batch_inputs(cfg, [128] * 64, seed=seed)
    stats = []
    names, worst = run_and_check(eng, cfg, w, ids, nw, nz, sample=range(64), audio_tol=TIGHT_AUDIO_TOL, stats=stats, cache_key=key)
    eng.close()
    assert any(n.startswith(f"conv_split_kernel<{SPLIT_MODES[mode]},") for n in names), names
    print(preset, "B=64", mode, "worst |d audio| %.2e, worst pcm rms %.2e" % (worst, max(s[1] for s in stats)))


def test_medium_b16_ragged_matches_oracle(monkeypatch):
    """The profiled B=16 configuration with ragged lengths (masking of every utterance inside shared tiles)."""
    cfg, w = voice("medium")
    eng = make_engine(monkeypatch, cfg, w)
    lens = [128, 3, 77, 128, 1, 50, 128, 19, 101, 64, 128, 33, 90, 2, 128, 111]
    ids, nw, nz = batch_inputs(cfg, lens, seed=33)
    names, worst = run_and_check(eng, cfg, w, ids, nw, nz, sample=[1, 2, 4, 7, 13, 15], scales=(0.5, 1.2, 0.9))
    eng.close()
    print("medium ragged B=16 kernels:", sorted(names), "worst |d audio| %.2e" % worst)


@pytest.mark.parametrize("preset,lens", [
    ("x-low", [64, 128, 9, 128, 77, 128, 128, 30, 128, 128, 128, 128]),   # 48-channel coupling halves: partial 32-channel chunks / row tiles
    ("x-low", [100]),
    ("medium", [128]),
    ("medium", [128, 3, 77, 128, 1, 50, 128, 19, 101, 64, 128, 33, 90, 2, 128, 111]),
    ("high", [90, 128]),
])
def test_no_kernel_reads_what_the_call_did_not_write(monkeypatch, preset, lens):
    """ADVICE r4 (conv1x1.h): PIPER_HIP_DEBUG_POISON=1 fills every activation workspace with NaN bit patterns at
    allocation. On gfx9 / CDNA the SGPR offset of a buffer access is outside the hardware's range check, so a kernel that
    puts a channel row's offset there and relies on the check for rows beyond the tensor reads whatever lies behind it --
    finite in practice, NaN here. Results must equal the oracle's, twice (the second call replays on used buffers)."""
    cfg, w = voice(preset)
    eng = make_engine(monkeypatch, cfg, w, {"PIPER_HIP_DEBUG_POISON": 1})
    ids, nw, nz = batch_inputs(cfg, lens, seed=91)
    for rep in range(2):
        names, worst = run_and_check(eng, cfg, w, ids, nw, nz, sample=range(min(len(lens), 6)))
    r = eng.synthesize_batch(ids, SCALES, noise_w=nw, noise_z=nz)
    assert all(np.all(np.isfinite(a)) for a in r.audio)
    eng.close()
    print(preset, len(lens), "poisoned workspaces: worst |d audio| %.2e" % worst, sorted(names))


# Launcher knobs (read at engine creation) that force each tile configuration of conv_mfma_kernel / each split-K
# variant on shapes where the default heuristics would pick another one. `expect`: instantiations that must run.
FORCED = [
    # every conv of a single utterance through the TILED kernels (gate epilogue, 32-row and 64-row tiles, both halos)
    ("medium", [128], {"PIPER_HIP_SPLITK_MAX": 0, "PIPER_HIP_MRF": 0, "PIPER_HIP_GROUP_TILED": 0},
     {"conv_mfma_kernel<2,2,2,1,16,true,64>", "conv_mfma_kernel<2,2,1,1,16,false,64>",
      "conv_mfma_kernel<1,4,1,1,16,false,64>", "conv_mfma_kernel<2,2,1,1,16,false,128>",
      "conv_mfma_kernel<1,4,1,1,16,false,128>"}),
    # ... and with the sibling resblock convs of every stage as grouped launches of the tiled kernel (both tile shapes, both halos)
    ("medium", [128], {"PIPER_HIP_SPLITK_MAX": 0, "PIPER_HIP_MRF": 0},
     {"conv_mfma_group_kernel<2,2,1,1,16,64>", "conv_mfma_group_kernel<2,2,1,1,16,128>",
      "conv_mfma_group_kernel<1,4,1,1,16,64>", "conv_mfma_group_kernel<1,4,1,1,16,128>", "mrf_sum_kernel"}),
    # the high voice's 128- / 64-channel stages of one and of two utterances: the grouped tiled form by default; one launch per conv
    ("high", [128], {}, {"conv_mfma_group_kernel<2,2,1,1,16,64>", "mrf_sum_kernel"}),
    ("high", [100, 128], {}, {"conv_mfma_group_kernel<2,2,1,1,16,64>"}),
    ("high", [128], {"PIPER_HIP_GROUP_TILED": 0}, {"conv_mfma_kernel<2,2,1,1,16,false,64>"}),
    # the one-tap convs (q/k/v, conv_o, res/skip, pre / post, proj) on the batched route: B operand straight from global
    # memory (conv1x1_kernel, the default), and through the tiled kernel's LDS slabs
    ("medium", [128, 70], {"PIPER_HIP_SPLITK_MAX": 0, "PIPER_HIP_MRF": 0, "PIPER_HIP_COLCHAIN": 0}, {"conv1x1_kernel<1>"}),
    ("medium", [128, 70], {"PIPER_HIP_SPLITK_MAX": 0, "PIPER_HIP_MRF": 0, "PIPER_HIP_COLCHAIN": 0, "PIPER_HIP_CONV1X1": 0},
     {"conv_mfma_kernel<1,4,1,1,16,false,64>"}),
    # the x-low voice's gate conv at batch: the 64 x 128 gate tile
    ("x-low", [64] * 32, {"PIPER_HIP_MRF": 0}, {"conv_mfma_kernel<1,4,2,1,16,true,64>"}),
    # several column tiles per workgroup (in-kernel slab pipeline across tiles)
    ("medium", [128, 40], {"PIPER_HIP_SPLITK_MAX": 0, "PIPER_HIP_TPB": 3, "PIPER_HIP_MRF": 0}, set()),
    # the duration predictor with ConvFlow.pre / proj / spline as separate launches (default: fused into the DDSConv layers)
    ("medium", [128, 31], {"PIPER_HIP_FUSE_DP": 0}, {"conv_splitk_kernel<1,false,8,4>"}),
    # conv_o + LN and coupling post + next pre as single launches (colchain_kernel): forced on for a batch, and off
    ("medium", [128, 77, 16, 33], {"PIPER_HIP_COLCHAIN": 2, "PIPER_HIP_COL4": 0}, {"colchain_kernel<6>", "lngemm_kernel<6>"}),
    ("medium", [128, 77, 16, 33], {"PIPER_HIP_COLCHAIN": 2}, {"colchain4_kernel<false>", "lngemm4_kernel", "dds_layer4_kernel"}),
    ("medium", [128, 31], {"PIPER_HIP_COLCHAIN": 0}, {"ln_kernel", "conv_splitk_kernel<1,false,4,4>"}),
    # sibling resblock convs of the 128-channel stage as grouped launches (64- and 128-column slabs), and one by one
    # (the last step, whose outputs the MRF only sums, is one GEMM over the concatenated K; GROUP_MRF=2: kept apart)
    ("medium", [117], {}, {"conv_splitk_group_kernel<4,2,64>", "conv_splitk_sum_kernel<4,2>"}),
    ("medium", [100], {"PIPER_HIP_GROUP_MRF": 2}, {"conv_splitk_group_kernel<4,2,64>", "conv_splitk_group_kernel<4,2,128>"}),
    ("high", [40], {}, {"conv_splitk_group_kernel<4,2,64>", "conv_splitk_sum_kernel<4,2>"}),
    # past PIPER_HIP_GROUP_MAXB tiles per conv the first stage of one utterance takes the TILED grouped launches (the high voice's
    # 256-channel stage at 128 ids, the medium voice's 128-channel stage at 256 ids); the split-K grouped launches forced back on
    ("high", [128], {"PIPER_HIP_GROUP_MAXB": 700}, {"conv_splitk_group_kernel<4,2,64>", "conv_splitk_group_kernel<4,2,128>", "conv_splitk_sum_kernel<4,2>"}),
    ("medium", [256], {}, {"conv_mfma_group_kernel<2,2,1,1,16,64>", "conv_mfma_group_kernel<2,2,1,1,16,128>", "mrf_sum_kernel"}),
    ("medium", [256], {"PIPER_HIP_GROUP_MAXB": 700}, {"conv_splitk_group_kernel<4,2,64>", "conv_splitk_sum_kernel<4,2>"}),
    ("medium", [128], {"PIPER_HIP_GROUP_MRF": 0}, {"conv_splitk_kernel<1,false,4,4>"}),
    # split-K variants: 4/8-wave only (no 16-column form), 12-wave everywhere, 16-column form everywhere
    ("medium", [128], {"PIPER_HIP_SPLITK16": 0, "PIPER_HIP_WIDE_SPLITK": 0, "PIPER_HIP_COL4": 0, "PIPER_HIP_GATE4": 0},
     {"conv_splitk_kernel<2,true,8,3>", "conv_splitk_kernel<1,false,8,4>", "conv_splitk_kernel<1,false,4,4>"}),
    ("medium", [96], {"PIPER_HIP_SPLITK16": 0, "PIPER_HIP_WIDE_SPLITK": 2, "PIPER_HIP_GATE4": 0},
     {"conv_splitk_kernel<2,true,12,2>", "conv_splitk_kernel<1,false,12,4>"}),
    ("x-low", [64], {"PIPER_HIP_SPLITK16": 0, "PIPER_HIP_WIDE_SPLITK": 0}, {"conv_splitk_kernel<2,true,4,3>"}),
    ("medium", [128, 17], {"PIPER_HIP_SPLITK16": 3, "PIPER_HIP_GATE4": 0},
     {"conv_splitk16_kernel<true,12,2,4>", "conv_splitk16_kernel<false,8,4,4>"}),
    # enc_p.proj + dp.pre as one launch over the stacked matrix is the default of small calls (every case above with the
    # 4-column chains on; multi-speaker: test_full_size_multi_speaker_matches_oracle); here as two launches
    ("medium", [128, 31], {"PIPER_HIP_STACK_PRE": 0}, {"colchain4_kernel<false>", "lngemm4_kernel"}),
    # attention with the score slabs in global memory (the form of utterances beyond ~830 ids) forced on short ragged batches
    ("medium", [128, 31], {"PIPER_HIP_ATTN_LONG": 1}, {"attn_long_kernel<96>"}),
    ("x-low", [64, 20, 33], {"PIPER_HIP_ATTN_LONG": 1}, {"attn_long_kernel<48>"}),
    ("tiny", [50, 7], {"PIPER_HIP_ATTN_LONG": 1}, {"attn_long_kernel<0>"}),
    # ... and chosen by the engine: a ragged batch padded to 900 ids
    ("medium", [900, 40], {}, {"attn_long_kernel<96>"}),
    # the last WN layer's res/skip conv in front of the post + pre chain launch (default) and as a launch of its own
    ("medium", [128, 31], {}, {"colchain4_kernel<true>", "colchain4_kernel<false>"}),
    ("medium", [128, 31], {"PIPER_HIP_CHAIN_RS": 0}, {"colchain4_kernel<false>"}),
    # a short utterance: the gate conv on half channel groups (six waves, twice the workgroups), and forced back to whole groups
    ("medium", [48], {"PIPER_HIP_GATE4": 0}, {"conv_splitk16_kernel<true,6,5,2>"}),
    ("medium", [48], {"PIPER_HIP_GATE_HALF": 0, "PIPER_HIP_GATE4": 0}, {"conv_splitk16_kernel<true,12,2,4>"}),
    ("high", [48], {"PIPER_HIP_SPLITK_MAX": 0}, {"conv_mfma_kernel<2,2,2,1,16,true,64>"}),
    # the 192-channel small-call chains (DDSConv layers, colchain, lngemm): the 4-column forms on the 4x4x1 MFMA (default for small calls) forced
    # on for a ragged batch beyond its column limit, and off (the 16-column form)
    ("medium", [128, 77, 16, 33, 3, 1, 128, 90, 128, 128], {"PIPER_HIP_COL4": 2},
     {"dds_layer4_kernel", "colchain4_kernel<false>", "lngemm4_kernel"}),
    ("medium", [128, 31], {"PIPER_HIP_COL4": 0}, {"dds_layer16_kernel<6>", "colchain_kernel<6>", "lngemm_kernel<6>"}),
    # the encoder FFN as one launch with partial outputs per 48-row slice of the hidden dimension (default for small
    # calls), and conv by conv behind the 4-column chains
    ("medium", [128, 13, 14, 15, 29], {}, {"ffn_kernel", "lngemm4_kernel"}),
    ("medium", [128, 31], {"PIPER_HIP_FFN": 0}, {"lngemm4_kernel", "conv_splitk16_kernel<false,8,4,4>"}),
    # attention + conv_o + norm_layers_1 as one launch (default for small calls): on for a ragged batch incl. lengths on
    # both softmax paths, and off (attn_kernel + colchain4_kernel)
    # -- on 4-query workgroups while the call's longest utterance has up to 256 ids (attn4_kernel: the default for these
    # shapes; K units / V chunks beyond the first of a wave from 129 ids on), on 16-query workgroups beyond and with
    # PIPER_HIP_ATTN4=0 (attno_kernel), and as two launches
    ("medium", [128, 13, 1, 129], {}, {"attn4_kernel<96,true>", "lngemm4_kernel", "ffn_kernel"}),
    ("medium", [128, 13, 1, 77], {}, {"attn4_kernel<96,false>"}),
    ("high", [96, 40], {}, {"attn4_kernel<96,false>"}),
    ("medium", [256, 70, 200], {}, {"attn4_kernel<96,true>"}),
    ("medium", [500], {}, {"attn4_kernel<96,true>"}),
    ("medium", [128, 13, 1, 129], {"PIPER_HIP_ATTN4": 0}, {"attno_kernel<96>", "lngemm4_kernel", "ffn_kernel"}),
    ("high", [96, 40], {"PIPER_HIP_ATTN4": 0}, {"attno_kernel<96>"}),
    ("medium", [600], {}, {"attno_kernel<96>"}),
    ("medium", [700], {"PIPER_HIP_ATTN4": 2}, {"attn4_kernel<96,true>"}),
    # the WN gate conv of one-utterance-sized calls on 64-row x 12-column workgroups (gate4_kernel: the default), and the
    # 16-column split-K forms it replaces
    ("medium", [128], {}, {"gate4_kernel"}),
    ("high", [77], {}, {"gate4_kernel"}),
    ("medium", [128], {"PIPER_HIP_GATE4": 0}, {"conv_splitk16_kernel<true,12,2,4>"}),
    ("medium", [40, 50], {"PIPER_HIP_GATE4": 2}, {"gate4_kernel"}),
    ("medium", [128, 31], {"PIPER_HIP_ATTNO": 0}, {"attn_kernel<96>", "colchain4_kernel<false>"}),
    # the up-convs' tiles stored one 4-byte piece per phase (default: 16- / 8-byte pieces of consecutive samples straight
    # from the accumulators: strides 8 and 4 on the medium voice, 8 and 2 on the high one), B = 1 and ragged batches
    ("medium", [128], {"PIPER_HIP_CONVT_VEC": 0}, {"conv_mfma_kernel<2,2,1,1,16,false,64>"}),
    ("medium", [128, 40, 77], {"PIPER_HIP_CONVT_VEC": 0}, set()),
    ("high", [70, 128, 9, 128, 128, 33], {"PIPER_HIP_CONVT_VEC": 0}, set()),
    ("high", [70, 128, 9, 128, 128, 33], {}, set()),
    # tiles of the 4-column kernels in workgroup order (no XCD-contiguous runs)
    ("medium", [128, 31], {"PIPER_HIP_XCD": 0}, {"colchain4_kernel<false>", "lngemm4_kernel", "dds_layer4_kernel"}),
]


@pytest.mark.parametrize("preset,lens,env,expect", FORCED,
                         ids=[f"{p}-B{len(l)}-" + ("-".join(f"{k[10:]}{v}" for k, v in e.items()) or "default") for p, l, e, _ in FORCED])
def test_forced_kernel_variants_match_oracle(monkeypatch, preset, lens, env, expect):
    cfg, w = voice(preset)
    eng = make_engine(monkeypatch, cfg, w, env)
    ids, nw, nz = batch_inputs(cfg, lens, seed=41 + len(lens))
    sample = sorted({0, len(lens) // 2, len(lens) - 1})
    names, worst = run_and_check(eng, cfg, w, ids, nw, nz, sample=sample)
    eng.close()
    assert expect <= names, f"missing {sorted(expect - names)}; launched {sorted(names)}"
    print(preset, env, "kernels:", sorted(names), "worst |d audio| %.2e" % worst)


@pytest.mark.parametrize("preset,T", [("medium", 128), ("high", 96), ("x-low", 64)])
def test_intermediate_tensors_match_oracle(monkeypatch, preset, T):
    """Every stage boundary of SynthesizerTrn.infer (models.py:681-722) against the oracle, not only the audio:
    x (encoder output), m_p / logs_p, logw (the SDP output BEFORE ceil, which the duration compare hides), z_p (length
    regulator + prior sample), z (flow output)."""
    from oracle import vits_oracle as O
    cfg, w = voice(preset)
    eng = make_engine(monkeypatch, cfg, w, {"PIPER_HIP_DEBUG_KEEP": 1})
    ids, nw, nz = batch_inputs(cfg, [T, T // 2 + 1], seed=51)
    r = eng.synthesize_batch(ids, SCALES, noise_w=nw, noise_z=nz)
    for b in range(2):
        o = O.synthesize(w, cfg, ids[b], SCALES, nw[b], nz[b], keep=True)
        stats = eng.debug_tensor("stats", b)
        C = cfg.inter
        got = {"x_enc": eng.debug_tensor("x_enc", b), "m_p": stats[:C], "logs_p": stats[C:],
               "logw": eng.debug_tensor("logw", b)[0], "z_p": eng.debug_tensor("z_p", b), "z": eng.debug_tensor("z", b)}
        for name, g in got.items():
            ref = o[name]
            assert g.shape == ref.shape, (name, g.shape, ref.shape)
            tol = 1e-4 * max(1.0, float(np.abs(ref).max()))
            err = float(np.max(np.abs(g - ref)))
            assert err < tol, f"{preset} utt {b} {name}: max |d| {err} (tol {tol})"
        assert np.max(np.abs(r.audio[b] - o["audio"])) < TIGHT_AUDIO_TOL
    eng.close()


def test_randn_moments_both_sites(monkeypatch):
    """The engine's own N(0,1) generator (Philox-4x32-10 + Box-Muller, randn_kernel) -- what the product path and
    bench.py draw from when no noise is injected: 2^20 draws per site, mean / variance / skew / kurtosis / tails,
    independence of the two sites and of consecutive runs."""
    cfg, w = voice("tiny")
    eng = make_engine(monkeypatch, cfg, w)
    eng.set_seed(2026)
    n = 1 << 20
    a = eng.debug_randn(0, 1, n).astype(np.float64)
    b = eng.debug_randn(1, 1, n).astype(np.float64)
    c = eng.debug_randn(0, 2, n).astype(np.float64)
    for x in (a, b, c):
        assert np.all(np.isfinite(x))
        assert abs(x.mean()) < 5e-3
        assert abs(x.var() - 1.0) < 1e-2
        assert abs(np.mean(x ** 3)) < 2e-2                    # skewness
        assert abs(np.mean(x ** 4) - 3.0) < 5e-2              # kurtosis
        assert abs(np.mean(np.abs(x) > 1.959964) - 0.05) < 2e-3
        assert abs(np.mean(np.abs(x) > 3.0) - 0.0026998) < 4e-4
        assert 4.0 < np.abs(x).max() < 7.0
        assert abs(np.corrcoef(x[:-1], x[1:])[0, 1]) < 5e-3   # neighbouring draws (Box-Muller pairs included)
    assert abs(np.corrcoef(a, b)[0, 1]) < 5e-3 and abs(np.corrcoef(a, c)[0, 1]) < 5e-3
    assert not np.array_equal(a, c)
    eng.close()


def test_product_path_draws_from_the_tested_generator(monkeypatch):
    """Without injected noise the pipeline's two sampling sites hold exactly what pe_debug_randn reports for the
    run counter, and the counter advances on every run (a replayed hipGraph draws fresh noise)."""
    cfg, w = voice("tiny")
    eng = make_engine(monkeypatch, cfg, w, {"PIPER_HIP_DEBUG_KEEP": 1, "PIPER_HIP_SPEC": 0})   # the prior noise is drawn inside regulate_kernel
    eng.set_seed(77)
    ids = [W.synthetic_phoneme_ids(40, 3, id_max=cfg.n_vocab - 1), W.synthetic_phoneme_ids(25, 4, id_max=cfg.n_vocab - 1)]
    eng.upload(ids, SCALES)
    audio = []
    for run in (1, 2, 3):
        eng.run()
        res = eng.fetch(True, False)
        assert eng.rng_calls == run
        for b in range(2):
            # logical row = utterance * channels + channel, column = id / frame (include/piper_hip.h: pe_debug_randn)
            nw = eng.debug_tensor("noise_w", b)
            nz = eng.debug_tensor("noise_z", b)
            assert np.array_equal(nw, np.stack([eng.debug_randn(0, run, nw.shape[1], row=2 * b + c) for c in range(2)]))
            assert np.array_equal(nz, np.stack([eng.debug_randn(1, run, nz.shape[1], row=cfg.inter * b + c)
                                                for c in range(cfg.inter)]))
        audio.append(res.audio[0])
    assert audio[0].shape != audio[1].shape or not np.array_equal(audio[0], audio[1])
    eng.close()


def test_noise_does_not_depend_on_capacity_or_speculation(monkeypatch):
    """For a given (seed, run counter) the engine's own noise for (utterance, channel, column) is the same whatever the
    workspace capacity and whether stage B was sized speculatively: an engine that has grown its workspaces
    on a long utterance and runs without speculation draws what a fresh speculating engine draws at the same counter."""
    cfg, w = voice("tiny")
    ids = W.synthetic_phoneme_ids(30, 5, id_max=cfg.n_vocab - 1)
    long_ids = W.synthetic_phoneme_ids(400, 6, id_max=cfg.n_vocab - 1)
    a = make_engine(monkeypatch, cfg, w, {"PIPER_HIP_DEBUG_KEEP": 1})
    b = make_engine(monkeypatch, cfg, w, {"PIPER_HIP_DEBUG_KEEP": 1, "PIPER_HIP_SPEC": 0})
    a.set_seed(31)
    b.set_seed(31)
    a.synthesize(ids, SCALES)                              # run 1 (sets the frames-per-id estimate)
    b.synthesize(long_ids, SCALES)                         # run 1: grows b's capacities (other row strides)
    for run in (2, 3):
        ra = a.synthesize(ids, SCALES)                     # speculative from run 2 on
        nza, nwa = a.debug_tensor("noise_z", 0), a.debug_tensor("noise_w", 0)
        rb = b.synthesize(ids, SCALES)
        assert a.rng_calls == b.rng_calls == run
        assert np.array_equal(nwa, b.debug_tensor("noise_w", 0))
        assert np.array_equal(nza, b.debug_tensor("noise_z", 0))
        assert np.array_equal(ra.frames, rb.frames)
        assert np.max(np.abs(ra.audio[0] - rb.audio[0])) < 1e-5      # (other shape buckets: other kernel routes)
    assert a.speculation_stats[0] >= 2
    assert b.speculation_stats == (0, 0)
    a.close()
    b.close()


def test_reference_test_sentences_medium(monkeypatch):
    """the 7 rows of the reference's etc/test_sentences/test_en-us.jsonl (their
    phoneme_ids as produced by piper-phonemize, 113..381 ids; fixture tests/golden/phoneme_ids_en-us.json) on the
    medium architecture, each as its own B=1 call like piper.cpp, and all 7 as one ragged batch."""
    from oracle import vits_oracle as O
    rows = json.load(open(os.path.join(ROOT, "tests", "golden", "phoneme_ids_en-us.json"), encoding="utf-8"))["rows"]
    id_lists = [np.asarray(r["phoneme_ids"], np.int64) for r in rows]
    assert len(id_lists) == 7 and min(map(len, id_lists)) >= 100
    cfg, w = voice("medium")
    eng = make_engine(monkeypatch, cfg, w)
    Tm = max(map(len, id_lists))
    rng = np.random.default_rng(61)
    nw = rng.standard_normal((7, 2, Tm)).astype(np.float32)
    nz = rng.standard_normal((7, cfg.inter, 6 * Tm)).astype(np.float32)
    wt = O.to_torch(w)
    singles = []
    for i, ids in enumerate(id_lists):
        o = O.synthesize(wt, cfg, ids, SCALES, nw[i], nz[i])
        eng.profile_enable(2)
        r = eng.synthesize(ids, SCALES, noise_w=nw[i], noise_z=nz[i])
        SEEN.update(row["name"] for row in eng.profile()[5:] if row["launches"])
        eng.profile_enable(0)
        assert np.array_equal(eng.durations(), o["durations"])
        assert np.max(np.abs(r.audio[0] - o["audio"])) < TIGHT_AUDIO_TOL
        assert pcm_rms(r.pcm[0], o["pcm"]) <= RMS_TOL
        singles.append(r.pcm[0])
    rb = eng.synthesize_batch(id_lists, SCALES, noise_w=nw, noise_z=nz)
    for i in range(7):
        assert rb.pcm[i].shape == singles[i].shape
        assert np.max(np.abs(rb.pcm[i].astype(np.int32) - singles[i].astype(np.int32))) <= 2
    eng.close()


def test_full_size_multi_speaker_matches_oracle(monkeypatch):
    """Speaker conditioning at the catalogue's real dimensions (gin_channels 512, e.g. en_US-libritts-high's graph
    shape on the medium vocoder): emb_g -> dp.cond / WN.cond_layer / dec.cond (models.py:692-696)."""
    cfg, w = voice("medium", n_speakers=12, gin=512)
    eng = make_engine(monkeypatch, cfg, w)
    ids, nw, nz = batch_inputs(cfg, [90, 128, 31], seed=71)
    sids = [11, 0, 5]
    run_and_check(eng, cfg, w, ids, nw, nz, sample=[0, 1, 2], sids=sids)
    # a different speaker changes the audio; the same speaker alone reproduces its batched result
    r1 = eng.synthesize(ids[0], SCALES, sid=11, noise_w=nw[0], noise_z=nz[0])
    r2 = eng.synthesize(ids[0], SCALES, sid=3, noise_w=nw[0], noise_z=nz[0])
    rb = eng.synthesize_batch(ids, SCALES, sids=sids, noise_w=nw, noise_z=nz)
    assert np.max(np.abs(r1.pcm[0].astype(np.int32) - rb.pcm[0].astype(np.int32))) <= 2
    assert r1.audio[0].shape != r2.audio[0].shape or np.max(np.abs(r1.audio[0] - r2.audio[0])) > 1e-3
    eng.close()


def test_randomised_stress_sweep(monkeypatch):
    """Seeded random lengths / batch sizes / scales / speakers over the medium, high, x-low and multi-speaker
    architectures (the longer sweep is scripts/stress_parity.py, part of the round's collection: profiles/r04_stress_parity.log)."""
    rng = np.random.default_rng(2026)
    worst = 0.0
    for preset, tmax, cases in (("medium", 220, 6), ("high", 90, 2), ("tiny-high-ms", 60, 4), ("x-low", 120, 3)):
        cfg, w = voice(preset, seed=99)
        eng = make_engine(monkeypatch, cfg, w)
        for c in range(cases):
            B = int(rng.integers(1, 7))
            lens = [int(rng.integers(1, tmax)) for _ in range(B)]
            ids, nw, nz = batch_inputs(cfg, lens, seed=1000 + 17 * c, zcols=8 * max(lens) + 64)
            scales = (float(rng.uniform(0, 1)), float(rng.uniform(0.6, 1.5)), float(rng.uniform(0, 1)))
            sids = [int(rng.integers(0, cfg.n_speakers)) for _ in range(B)] if cfg.n_speakers > 1 else None
            _, d = run_and_check(eng, cfg, w, ids, nw, nz, sample=range(B), scales=scales, sids=sids)
            worst = max(worst, d)
        eng.close()
    print("stress sweep worst |d audio| = %.2e" % worst)


@pytest.mark.parametrize("preset,lens,ou", [("medium", [128, 37], 0), ("medium", [100] * 12, 0), ("medium", [77, 128], 1),
                                            ("medium", [128], 2), ("medium", [90, 31], 3), ("medium", [128, 60], 4), ("high", [64, 9], 0),
                                            ("high", [40], 1), ("high", [33, 20, 50], 3), ("x-low", [64], 0)])
def test_fused_mrf_stage_kernel_matches_unfused_and_oracle(monkeypatch, preset, lens, ou):
    """mrf_kernel (one launch per <= 64-channel MRF stage: every resblock conv out of LDS, activated tensors, residuals
    and the MRF sum in registers, weights prefetched from L2 into registers) against the conv-by-conv schedule
    (PIPER_HIP_MRF=0) and the oracle, for every window geometry (`ou` output units per wave; 0 = the launcher's cost
    model)."""
    cfg, w = voice(preset)
    ids, nw, nz = batch_inputs(cfg, lens, seed=81)
    env = {"PIPER_HIP_MRF": 2}                                           # 2 = wherever the kernel applies (default: measured policy)
    if ou:
        env["PIPER_HIP_MRF_OU"] = ou
    fused = make_engine(monkeypatch, cfg, w, env)
    names, worst = run_and_check(fused, cfg, w, ids, nw, nz, sample=sorted({0, len(lens) - 1}))
    assert any(n.startswith("mrf_kernel<") for n in names), names
    a = fused.synthesize_batch(ids, SCALES, noise_w=nw, noise_z=nz)
    fused.close()
    plain = make_engine(monkeypatch, cfg, w, {"PIPER_HIP_MRF": 0})
    names0, _ = run_and_check(plain, cfg, w, ids, nw, nz, sample=[0])
    assert not any(n.startswith("mrf_kernel<") for n in names0)
    b = plain.synthesize_batch(ids, SCALES, noise_w=nw, noise_z=nz)
    plain.close()
    for i in range(len(lens)):
        assert a.audio[i].shape == b.audio[i].shape
        assert np.max(np.abs(a.audio[i] - b.audio[i])) < 2e-5
    print(preset, "fused stage kernels:", sorted(n for n in names if n.startswith("mrf")), "worst |d audio| %.2e" % worst)


@pytest.mark.parametrize("mode", ["bf16x3", "f16x3"])
@pytest.mark.parametrize("preset,lens,ou", [("medium", [128, 37], 0), ("medium", [100] * 12, 0), ("medium", [77, 128], 1),
                                            ("medium", [128], 0), ("medium", [128], 2), ("medium", [90, 31], 3), ("medium", [128, 60], 4),
                                            ("high", [64, 9], 0), ("high", [33, 20, 50], 3), ("x-low", [64, 128], 0)])
def test_split_fused_mrf_stage_kernel_matches_oracle(monkeypatch, preset, lens, ou, mode):
    """mrf_split_kernel (kernels/mrf_split.h: the fused MRF stage on the 16-bit matrix pipe, activations split once by
    their producer, in LDS) in the two-term matrix modes, every window geometry (`ou` = output units per wave, 0 = the
    launcher's cost model), ResBlock2 (medium / x-low) and ResBlock1 (high) stages: the f32 path's own gate against the
    oracle; and the generator tail inside the last stage against the separate conv_post_kernel: bit for bit."""
    cfg, w = voice(preset)
    ids, nw, nz = batch_inputs(cfg, lens, seed=81)
    res = {}
    for tail in (1, 0):
        env = {"PIPER_HIP_MATRIX": mode, "PIPER_HIP_MRF": 2, "PIPER_HIP_MRF_TAIL": tail}
        if ou:
            env["PIPER_HIP_MRF_OU"] = ou
        eng = make_engine(monkeypatch, cfg, w, env)
        names, worst = run_and_check(eng, cfg, w, ids, nw, nz, sample=sorted({0, len(lens) - 1}), cache_key="smrf-%s-%s" % (preset, "_".join(str(t) for t in lens)))
        assert any(n.startswith(f"mrf_split_kernel<{SPLIT_MODES[mode]},") for n in names), names
        assert not any(n.startswith("mrf_kernel<") for n in names), names
        assert ("conv_post_kernel" in names) == (tail == 0), names
        res[tail] = eng.synthesize_batch(ids, SCALES, noise_w=nw, noise_z=nz)
        eng.close()
        if tail:
            print(preset, mode, "fused split stage kernels:", sorted(n for n in names if n.startswith("mrf")), "worst |d audio| %.2e" % worst)
    for i in range(len(lens)):
        assert np.array_equal(res[1].audio[i], res[0].audio[i]), f"utterance {i}: waveform differs"
        assert np.array_equal(res[1].pcm[i], res[0].pcm[i]), f"utterance {i}: pcm differs"


@pytest.mark.parametrize("preset,lens,ou", [("medium", [128, 45, 3], 0), ("medium", [128], 4), ("medium", [60] * 9, 3),
                                            ("high", [50, 17], 0), ("x-low", [128, 64], 0)])
def test_generator_tail_inside_the_last_stage_kernel_is_bit_identical(monkeypatch, preset, lens, ou):
    """The last stage's mrf_kernel with the generator tail fused (leaky_relu(0.01) -> conv_post -> tanh, the utterance
    peak; windows overlapping by the 6 conv_post taps) against the same stage kernel followed by conv_post_kernel
    (PIPER_HIP_MRF_TAIL=0): float waveform and int16 PCM bit for bit (same channel-group partial sums, same order), on
    windows that start before the utterance, end behind it, and on utterances shorter than one window."""
    cfg, w = voice(preset)
    ids, nw, nz = batch_inputs(cfg, lens, seed=57)
    res = {}
    for tail in (1, 0):
        env = {"PIPER_HIP_MRF": 2, "PIPER_HIP_MRF_TAIL": tail}
        if ou:
            env["PIPER_HIP_MRF_OU"] = ou
        eng = make_engine(monkeypatch, cfg, w, env)
        names, worst = run_and_check(eng, cfg, w, ids, nw, nz, sample=[0, len(lens) - 1])
        assert any(n.startswith("mrf_kernel<32") for n in names), names
        assert ("conv_post_kernel" in names) == (tail == 0), names
        res[tail] = eng.synthesize_batch(ids, SCALES, noise_w=nw, noise_z=nz)
        eng.close()
    for i in range(len(lens)):
        assert np.array_equal(res[1].audio[i], res[0].audio[i]), f"utterance {i}: waveform differs"
        assert np.array_equal(res[1].pcm[i], res[0].pcm[i]), f"utterance {i}: pcm differs"


def test_speculative_stage_b_hits_and_misses(monkeypatch):
    """Stage B launched for a guessed frame bucket before the host has seen the frame counts (<= 4 utterances): a
    correct guess and a wrong one (length_scale jumps) must both end in the oracle's waveform."""
    from oracle import vits_oracle as O
    cfg, w = voice("medium")
    eng = make_engine(monkeypatch, cfg, w)
    wt = O.to_torch(w)
    rng = np.random.default_rng(93)
    for it, (T, ls) in enumerate([(100, 1.0), (100, 1.0), (120, 1.05), (120, 2.6), (60, 1.0), (150, 0.5), (128, 1.0)]):
        ids = W.synthetic_phoneme_ids(T, it, id_max=129)
        nw = rng.standard_normal((2, T)).astype(np.float32)
        r = eng.synthesize(ids, (0.0, ls, 0.8), noise_w=nw)          # no injected prior noise: graphs + speculation
        o = O.synthesize(wt, cfg, ids, (0.0, ls, 0.8), nw, None)
        assert np.array_equal(eng.durations(), o["durations"])
        assert r.audio[0].shape == o["audio"].shape, (it, r.audio[0].shape, o["audio"].shape)
        assert np.max(np.abs(r.audio[0] - o["audio"])) < TIGHT_AUDIO_TOL
    runs, misses = eng.speculation_stats
    assert runs >= 5 and 1 <= misses < runs, (runs, misses)     # the length_scale jump 1.05 -> 2.6 cannot be guessed
    eng.close()


def test_every_profiled_instantiation_is_parity_tested():
    """Closes the loop with profiles/: every templated pe:: kernel instantiation that appears in a committed
    rocprofv3 kernel-stats summary must have run inside one of the oracle comparisons above."""
    if not SEEN:
        pytest.skip("run the whole module: this test checks the union of the instantiations the others launched")
    profiled = set()
    # the summaries of the LATEST round describe the kernels of this source tree; earlier rounds' files are history
    # (kernel generations that no longer exist)
    allp = glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_*kernel_stats.csv"))
    latest = max((os.path.basename(q)[:3] for q in allp), default="")
    paths = [q for q in allp if os.path.basename(q).startswith(latest)]
    for path in paths:
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                n = row.get("Name", "")
                if "pe::" in n and "<" in n:
                    n = n.replace("void ", "").replace("pe::", "")
                    profiled.add(n[:n.index("(")].replace(" ", ""))
    assert profiled, "no committed kernel-stats summaries found"
    missing = sorted(profiled - SEEN)
    assert not missing, f"profiled but never compared with the oracle: {missing}; seen: {sorted(SEEN)}"



def test_rccl_load_path_at_world_size_one(tmp_path):
    """The multi-GPU load path on the single-GPU box: `PIPER_BENCH_DIST=1 python bench.py` initialises the "nccl" (RCCL)
    process group with ONE rank and goes through piper_amd.dist.load_sharded -- layout check, arena as a torch tensor,
    dist.broadcast of the packed weights -- before the timed loop. (More ranks need more GPUs: the driver's scaling run.)"""
    import subprocess
    import sys
    env = dict(os.environ, PIPER_BENCH_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-extra", "--no-cpu-baseline", "--no-roofline",
                          "--steps", "5", "--warmup", "2", "--min-seconds", "0"], capture_output=True, text=True, timeout=600,
                         env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["value"] > 1e6
    assert d["weight_broadcast"]["bytes"] > 50e6, d["weight_broadcast"]      # the medium voice's packed arena


@pytest.mark.parametrize("devices", [[0, 0], [0, 1]])
def test_engine_group_two_engines(monkeypatch, devices):
    """pe_group_* on hardware: two engines of one process -- on two GPUs when the box has them (hipMemcpyPeer over xGMI),
    else both on device 0 -- with the packed weights copied arena to arena, shards run by two host threads concurrently.
    Noise scales 0 make the result deterministic: every utterance must equal what a single engine returns, in the
    caller's order."""
    import torch
    from piper_amd import dist
    from piper_amd.group import EngineGroup
    if max(devices) >= torch.cuda.device_count():
        pytest.skip(f"needs {max(devices) + 1} GPUs")
    cfg, w = voice("medium")
    blob = W.pack_blob(cfg, w)
    lens = [128, 40, 77, 9, 101, 64]
    ids = [W.synthetic_phoneme_ids(T, 300 + i, id_max=129) for i, T in enumerate(lens)]
    scales = (0.0, 1.0, 0.0)
    monkeypatch.setenv("PIPER_HIP_GROUP_COALESCE", "0")      # every engine takes part (by default a device's small share goes to ONE of its engines)
    grp = EngineGroup(blob, devices)
    assert grp.broadcast_path == ("same-device" if devices[0] == devices[1] else "rccl"), grp.broadcast_path
    rg = grp.synthesize_batch(ids, scales)
    assign = grp.assignment(len(ids))
    table = dist.shard_indices(lens, 2)
    assert all(assign[i] == 0 for i in table[0]) and all(assign[i] == 1 for i in table[1])
    eng = make_engine(monkeypatch, cfg, w)
    rs = eng.synthesize_batch(ids, scales)
    assert list(rg.frames) == list(rs.frames)
    for i, (a, b) in enumerate(zip(rg.pcm, rs.pcm)):
        # (the shards are other batch shapes than the single call: other kernel routes, last-bit float differences)
        assert a.shape == b.shape and np.max(np.abs(a.astype(np.int32) - b.astype(np.int32))) <= 2, f"utterance {i}"
    for _ in range(3):                                   # repeated calls, other shapes
        r2 = grp.synthesize_batch(ids[::-1][:3], scales)
        assert np.max(np.abs(r2.pcm[0].astype(np.int32) - rs.pcm[5].astype(np.int32))) <= 2
    eng.close()
    grp.close()


def test_configs3_full_size_512_utterances_over_8_engines(monkeypatch):
    """BASELINE.json configs[3] at its STATED size: en_US-lessac-medium architecture, 512 utterances x 128 ids dealt to 8
    engines (the loop being sharded is the reference's per-phrase loop, src/cpp/piper.cpp:549-582). The box has one GPU, so
    the 8 engines of the group share device 0 -- the deal (LPT table, 64 per engine), the 8 concurrent 64-utterance
    pipelines and the gather in the caller's order are what 8 devices run. Every engine draws its own noise (seed + i);
    the oracle gets the very draws the pipeline used (pe_debug_tensor). Checked: the assignment equals
    dist.shard_indices; integer durations of ALL 512 utterances equal the oracle's; the int16 conversion of every
    utterance is 0 LSB from the oracle's restatement of piper.cpp:410-431 on the engine's float waveform; the float
    waveform of 2 utterances per shard against the full oracle; PCM comes back in the caller's order."""
    from oracle import vits_oracle as O
    from piper_amd import dist
    from piper_amd import _lib as L
    from piper_amd.group import EngineGroup
    import json as _json
    for k in [x["env"] for x in _json.loads(L.get_lib().pe_policy_describe().decode())] + ["PIPER_HIP_MATRIX"]:
        monkeypatch.delenv(k, raising=False)
    cfg, w = voice("medium")
    wt = O.to_torch(w)
    N, NE, T = 512, 8, 128
    ids = [W.synthetic_phoneme_ids(T, 5000 + i, id_max=129) for i in range(N)]
    grp = EngineGroup(W.pack_blob(cfg, w), [0] * NE)
    assert len(grp) == NE
    grp.set_seed(2024)
    r = grp.synthesize_batch(ids, SCALES)
    assign = grp.assignment(N)
    table = dist.shard_indices([T] * N, NE)
    assert [len(t) for t in table] == [N // NE] * NE
    for e, t in enumerate(table):
        assert all(assign[u] == e for u in t), f"engine {e}: assignment differs from dist.shard_indices"
    assert len(r.pcm) == N and len(r.frames) == N
    worst, checked, edge = 0.0, 0, []
    for e, t in enumerate(table):
        eng = grp.engine(e)
        durs = eng.durations().reshape(len(t), T)            # this engine's share of the call, in shard order
        res = eng.fetch(True, True)                          # float waveform + int16 of the share
        flipped = set()
        for k, u in enumerate(t):
            nw = eng.debug_tensor("noise_w", k)
            od, ow = O.durations_only(wt, cfg, ids[u], SCALES, nw, return_w=True)
            if not np.array_equal(durs[k], od):
                # 65 536 ids: a value in front of the ceil that sits within a few ulp of an integer may land on the other
                # side of it in the engine's summation order. Only THAT is tolerated: off by one, at a position whose
                # oracle value is within 2e-5 (relative) of the integer -- and on a handful of ids at most.
                bad = np.nonzero(durs[k] != od)[0]
                for i in bad:
                    near = abs(ow[i] - np.rint(ow[i])) <= 2e-5 * max(1.0, abs(ow[i]))
                    assert abs(int(durs[k][i]) - int(od[i])) == 1 and near, \
                        f"utterance {u} id {i}: duration {durs[k][i]} vs oracle {od[i]} (value in front of the ceil {ow[i]!r})"
                    edge.append((u, int(i), float(ow[i])))
                flipped.add(k)
            assert int(r.frames[u]) == max(int(durs[k].sum()), 1) == int(res.frames[k])
            assert np.array_equal(r.pcm[u], res.pcm[k]), f"utterance {u}: gathered out of order"
            assert np.array_equal(O.audio_float_to_int16(res.audio[k]), r.pcm[u]), f"utterance {u}: int16 not bit-exact"
        picks = [k for k in (0, len(t) - 1 - e, 1, 2) if k not in flipped][:2]
        for k in picks:                                      # two per shard, other slots in every shard
            u = t[k]
            o = O.synthesize(wt, cfg, ids[u], SCALES, eng.debug_tensor("noise_w", k), eng.debug_tensor("noise_z", k))
            assert np.array_equal(durs[k], o["durations"])
            assert res.audio[k].shape == o["audio"].shape
            d = float(np.max(np.abs(res.audio[k] - o["audio"])))
            worst = max(worst, d)
            assert d < TIGHT_AUDIO_TOL, f"utterance {u}: max |d audio| = {d}"
            assert pcm_rms(r.pcm[u], o["pcm"]) <= RMS_TOL
            checked += 1
        eng.close()
    assert checked == 2 * NE
    assert len(edge) <= 4, edge                              # (of 65 536 ids)
    if edge:
        print("ids whose value in front of the ceil sits on an integer (tolerated, off by one):", edge)
    # distinct utterances gave distinct audio (the gather did not duplicate a shard)
    assert len({(int(f), int(p[:2000].astype(np.int64).sum())) for f, p in zip(r.frames, r.pcm)}) > N * 0.95
    print("configs[3] full size: 512 utterances over 8 engines, worst |d audio| %.2e on %d oracle waveforms, call %.3f s"
          % (worst, checked, r.infer_seconds))
    grp.close()


CONFIGS3_WORKER = r'''
import os, sys, json
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from piper_amd import weights as W
from piper_amd.dist import ShardedSynthesizer, shard_indices
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
cfg = W.preset("medium")
blob = W.pack_blob(cfg, W.synthetic_weights(cfg, 1234)) if rank == 0 else None   # only rank 0 has the voice
syn = ShardedSynthesizer(blob=blob, device=0)
N, T = 512, 128
ids = [W.synthetic_phoneme_ids(T, 5000 + i, id_max=129) for i in range(N)]
out = syn.synthesize(ids, (0.0, 1.0, 0.0))
mine = shard_indices([T] * N, world)[rank]
durs = syn.engine.durations().reshape(len(mine), T)
all_d = [None] * world if rank == 0 else None
dist.gather_object((mine, durs), all_d, dst=0)
if rank == 0:
    D = np.zeros((N, T), np.int32)
    for m, d in all_d:
        D[m] = d
    np.savez(%(out)r, durations=D, broadcast_bytes=syn.broadcast_bytes, **{"pcm%%d" %% i: p for i, p in enumerate(out)})
dist.barrier()
dist.destroy_process_group()
'''


def test_configs3_full_size_through_sharded_synthesizer_8_ranks(tmp_path):
    """The same 512 x 128 deal through the one-process-per-GPU form (piper_amd.dist.ShardedSynthesizer): 8 ranks launched
    by torch.distributed.run, every rank on the box's one GPU, process group and the arena broadcast over gloo (the RCCL
    path needs one device per rank). Zero noise scales make it deterministic: durations of all 512 utterances equal the
    oracle's, PCM of 16 sampled utterances within the north-star RMS, all 512 come back in the caller's order."""
    import socket
    import subprocess
    import sys
    from oracle import vits_oracle as O
    out = str(tmp_path / "c3.npz")
    script = tmp_path / "worker.py"
    script.write_text(CONFIGS3_WORKER % {"root": ROOT, "out": out})
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                       capture_output=True, text=True, env=env, timeout=1500, cwd=str(tmp_path))
    assert p.returncode == 0, p.stderr[-3000:]
    got = np.load(out)
    assert int(got["broadcast_bytes"]) > 1e8
    cfg, w = voice("medium")
    wt = O.to_torch(w)
    scales = (0.0, 1.0, 0.0)
    edge = []
    for u in range(512):
        ids = W.synthetic_phoneme_ids(128, 5000 + u, id_max=129)
        d, wv = O.durations_only(wt, cfg, ids, scales, return_w=True)
        gd = got["durations"][u]
        for i in np.nonzero(gd != d)[0]:                    # (see the group test: integer boundaries within a few ulp)
            assert abs(int(gd[i]) - int(d[i])) == 1 and abs(wv[i] - np.rint(wv[i])) <= 2e-5 * max(1.0, abs(wv[i])), (u, i, wv[i])
            edge.append((u, int(i)))
        assert got["pcm%d" % u].size == max(int(gd.sum()), 1) * 256
        if u % 32 == 5 and np.array_equal(gd, d):
            o = O.synthesize(wt, cfg, ids, scales)
            assert pcm_rms(got["pcm%d" % u], o["pcm"]) <= RMS_TOL, f"utterance {u}"
    assert len(edge) <= 4, edge


def test_concurrent_requests_are_coalesced(monkeypatch):
    """several pending single-utterance requests run as ONE batched engine call instead of engines
    racing for the launch path. (a) pe_group_* with 8 engines on one GPU and 8 utterances: all 8 go to one engine
    (pe_group_assignment), PCM in caller order; (b) pe_coalescer_*: 8 caller threads on one engine -- batched calls, and
    every request gets what ITS OWN B=1 call computes: the int16 rule of piper.cpp:410-431 on its own float waveform
    (0 LSB), integer durations and waveform against the oracle with the engine's own noise draws."""
    import threading
    from oracle import vits_oracle as O
    from piper_amd.group import Coalescer, EngineGroup
    cfg, w = voice("medium")
    wt = O.to_torch(w)
    blob = W.pack_blob(cfg, w)
    lens = [128, 90, 128, 61, 128, 100, 77, 128]
    ids = [W.synthetic_phoneme_ids(T, 800 + i, id_max=129) for i, T in enumerate(lens)]
    grp = EngineGroup(blob, [0] * 8)
    grp.set_seed(11)
    r = grp.synthesize_batch(ids, SCALES)
    assert grp.assignment(8) == [0] * 8
    e0 = grp.engine(0)
    res = e0.fetch(True, True)
    durs = e0.durations()
    off = np.concatenate([[0], np.cumsum(lens)])
    for i in range(8):
        assert np.array_equal(r.pcm[i], res.pcm[i]) and np.array_equal(O.audio_float_to_int16(res.audio[i]), r.pcm[i])
        o = O.synthesize(wt, cfg, ids[i], SCALES, e0.debug_tensor("noise_w", i), e0.debug_tensor("noise_z", i))
        assert np.array_equal(durs[off[i]:off[i + 1]], o["durations"])
        assert np.max(np.abs(res.audio[i] - o["audio"])) < TIGHT_AUDIO_TOL and pcm_rms(r.pcm[i], o["pcm"]) <= RMS_TOL
    e0.close()
    grp.close()
    eng = make_engine(monkeypatch, cfg, w)
    zero = (0.0, 1.0, 0.0)
    want = [eng.synthesize(t, zero).pcm[0] for t in ids]
    co = Coalescer(eng, max_batch=8, max_wait_us=20000)
    out = [None] * 8

    def work(i):
        out[i] = co.synthesize(ids[i], zero)

    for rep in range(3):
        th = [threading.Thread(target=work, args=(i,)) for i in range(8)]
        [t.start() for t in th]
        [t.join() for t in th]
        for i in range(8):
            pcm, frames, secs, bs = out[i]
            assert pcm.shape == want[i].shape and np.max(np.abs(pcm.astype(np.int32) - want[i].astype(np.int32))) <= 2, i
            assert pcm_rms(pcm, O.synthesize(wt, cfg, ids[i], zero)["pcm"]) <= RMS_TOL
    calls, reqs = co.stats
    assert reqs == 24 and calls <= 12, (calls, reqs)
    print("coalescer: %d requests in %d engine calls" % (reqs, calls))
    co.close()
    eng.close()


def test_engine_group_weight_broadcast_through_rccl(tmp_path):
    """pe_group_create's collective path on a one-GPU box: PIPER_HIP_GROUP_BCAST=rccl takes the ncclBroadcast (librccl
    dlopen'ed, a communicator over the group's distinct devices -- here one -- and one broadcast of the packed arena from
    devices[0]) where two engines on one GPU would otherwise be filled by a device-to-device copy; =peer forces the copies.
    Either way the second engine, whose weights only ever arrived that way, gives the single engine's PCM. Runs in a process
    of its own WITHOUT PyTorch: torch ships its own ROCm runtime + RCCL, and a communicator created through that copy cannot
    take this library's device pointers (pe_group_create then reports "peer-copy (...)" and copies instead)."""
    import subprocess
    import sys
    script = tmp_path / "grp.py"
    script.write_text(f"""
import os, sys, json
sys.path.insert(0, {ROOT!r})
import numpy as np
from piper_amd import weights as W
from piper_amd.engine import Engine
from piper_amd.group import EngineGroup
assert "torch" not in sys.modules
cfg = W.preset("medium")
blob = W.pack_blob(cfg, W.synthetic_weights(cfg, 1234))
ids = [W.synthetic_phoneme_ids(T, 400 + i, id_max=129) for i, T in enumerate([96, 64])]
scales = (0.0, 1.0, 0.0)
eng = Engine(blob=blob, device=0)
ref = eng.synthesize_batch(ids, scales).pcm
eng.close()
out = {{}}
os.environ["PIPER_HIP_GROUP_COALESCE"] = "0"      # both engines take part
for mode in ("rccl", "peer"):
    os.environ["PIPER_HIP_GROUP_BCAST"] = mode
    grp = EngineGroup(blob, [0, 0])
    r = grp.synthesize_batch(ids, scales)
    out[mode] = dict(path=grp.broadcast_path, assign=sorted(grp.assignment(2)),
                     maxd=[int(np.max(np.abs(a.astype(np.int32) - b.astype(np.int32)))) if a.shape == b.shape else 99999 for a, b in zip(r.pcm, ref)])
    grp.close()
print(json.dumps(out))
""")
    p = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]       # (librccl may print at unload)
    assert lines, (p.stdout[-2000:], p.stderr[-2000:])
    out = json.loads(lines[-1])
    assert out["rccl"]["path"] == "rccl", out
    assert out["peer"]["path"] == "same-device", out
    for mode in ("rccl", "peer"):
        assert out[mode]["assign"] == [0, 1] and max(out[mode]["maxd"]) <= 2, out


@pytest.mark.parametrize("ranks", [2, 8])
def test_bench_n_ranks_on_one_gpu_prints_a_compact_line(ranks):
    """`bench.py --gpus N` end to end on ONE GPU (PIPER_BENCH_BACKEND=gloo: every rank shares device 0, the process group
    and load_sharded's broadcast run over gloo): the N > 1 plumbing -- self-launch, barriers, max-over-ranks timing, the
    per-rank gather -- must produce a driver-parsable last line before the first real multi-GPU run exists, and the
    record must be auditable: world size, backend, every rank's PCI bus id, the batched 1 -> N scaling
    next to `value`. At 8 ranks the line must still fit the driver's 4 KB."""
    import subprocess
    import sys
    full = os.path.join(ROOT, "gpurun_out", f"bench_full_{ranks}rank.json")
    env = dict(os.environ, PIPER_BENCH_BACKEND="gloo", PIPER_BENCH_FULL=full)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "3", "--warmup", "1",
                        "--min-seconds", "0", "--no-roofline"], capture_output=True, text=True, timeout=1500, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    line = p.stdout.strip().splitlines()[-1]
    assert len(line) < 4096, len(line)
    with open(os.path.join(ROOT, "gpurun_out", f"bench_{ranks}rank_gloo_1gpu.json"), "w") as f:
        f.write(line + "\n")
    d = json.loads(line)
    assert d["n_gpus"] == ranks and d["steps"] == 3 and d["value"] > 0 and d["scaling"] == "weak"
    assert len(d["per_rank_samples_per_s"]) == ranks and d["weight_broadcast"]["bytes"] > 1e8
    # the same per-GPU workload as the N = 1 headline (one weak-scaling curve), the batched share beside it
    assert "1 utterance(s) x 128" in d["config"]["workload"]
    bp = d["batched_per_gpu"]
    assert "64 utterance(s) x 128" in bp["workload"] and bp["value"] > 0 and bp["single_gpu_value"] > 0
    assert bp["speedup_over_single_gpu"] > 0
    rk = d["ranks"]
    assert rk["world_size"] == ranks and rk["backend"] == "gloo" and len(rk["pci_bus_ids"]) == ranks
    assert rk["distinct_devices"] == 1 and all(":" in x for x in rk["pci_bus_ids"])      # one GPU here: the record says so
    assert "batched_per_gpu" in d["headline_note"]


def test_bench_two_gpus_over_rccl_or_refuses(tmp_path):
    """Multi-GPU pre-flight. With >= 2 devices: `bench.py --gpus 2` over "nccl" (RCCL) must record two
    DISTINCT devices and a batched 1 -> 2 speed-up of at least 1.8x (utterances are independent; the only exchange is the
    weight broadcast at load). With one device the same command must FAIL (rc != 0) instead of printing a curve point --
    a run whose ranks share a GPU is not a multi-GPU measurement (bench.py: main, the distinct-device rule)."""
    import subprocess
    import sys
    import torch
    ndev = torch.cuda.device_count()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("PIPER_BENCH_BACKEND", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--min-seconds", "0",
           "--no-roofline"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env)
    if ndev < 2:
        assert p.returncode != 0, "two nccl ranks on one GPU must not produce a result line"
        assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
        return
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads(p.stdout.strip().splitlines()[-1])
    assert d["ranks"]["backend"] == "nccl" and d["ranks"]["distinct_devices"] == 2 == d["ranks"]["world_size"]
    assert d["batched_per_gpu"]["speedup_over_single_gpu"] >= 1.8, d["batched_per_gpu"]


def test_graph_cache_is_lru_and_warmup_stops_captures(monkeypatch):
    """The hipGraph cache: shape buckets bound the number of graphs a stream of texts needs, the cache evicts ONE graph
    (the least recently used) when full -- never all of them -- and after pe_warmup with a representative utterance a
    second pass over the same texts captures nothing. Results under eviction and replay equal a fresh engine's."""
    cfg, w = voice("medium")
    texts = [W.synthetic_phoneme_ids(T, 700 + i, id_max=129) for i, T in enumerate([40, 70, 100, 130, 170, 200, 90, 150])]
    scales = (0.0, 1.0, 0.0)
    ref_eng = make_engine(monkeypatch, cfg, w)
    want = [ref_eng.synthesize(t, scales).pcm[0] for t in texts]
    ref_eng.close()
    # ---- a 3-entry cache cycling through 8 buckets: every call evicts, every result is still right
    eng = make_engine(monkeypatch, cfg, w, {"PIPER_HIP_GRAPHS": 3})
    for rep in range(2):
        for t, wpcm in zip(texts, want):
            got = eng.synthesize(t, scales).pcm[0]
            assert got.shape == wpcm.shape and np.max(np.abs(got.astype(np.int32) - wpcm.astype(np.int32))) <= 2
            assert eng.graph_stats[0] <= 3
    assert eng.graph_stats[1] >= 8
    eng.close()
    # ---- the default cache after a warm-up: the second pass over the texts captures nothing
    eng = make_engine(monkeypatch, cfg, w)
    eng.warmup(max_batch=1, max_ids=224, frames_per_id=6.0, scales=scales, sample_ids=texts[3])
    c0 = eng.graph_stats[1]
    assert c0 >= 7                                      # seven id buckets up to 224
    for t in texts:
        eng.synthesize(t, scales)
    c1 = eng.graph_stats[1]
    for t, wpcm in zip(texts, want):
        got = eng.synthesize(t, scales).pcm[0]
        assert got.shape == wpcm.shape and np.max(np.abs(got.astype(np.int32) - wpcm.astype(np.int32))) <= 2
    assert eng.graph_stats[1] == c1, (c0, c1, eng.graph_stats)
    assert c1 - c0 <= len(texts)                        # at most one frame bucket the warm-up did not meet, per text
    print("graph captures: warm-up", c0, "first pass", c1 - c0, "second pass 0; cached", eng.graph_stats[0])
    eng.close()
