"""GPU parity tests (run with -m gpu on an MI355X). They call the HIP path through the C ABI
(libpiper_hip.so via piper_amd.engine.Engine) and compare it with
  * the committed golden vectors (outputs of the reference's own PyTorch graph), and
  * the CPU oracle on the same seeded inputs,
on integer durations first, then on the int16 PCM within the north-star tolerance:
RMS((pcm_hip - pcm_ref) / 32767) <= 1e-3 (fp32 arithmetic; in practice ~1e-6)."""
import glob
import os

import numpy as np
import pytest

from piper_amd import weights as W

pytestmark = pytest.mark.gpu

RMS_TOL = 1e-3           # BASELINE.json north_star
TIGHT_AUDIO_TOL = 2e-4   # what the fp32 MFMA path actually achieves on the float waveform (max abs)
GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))

_engines = {}


def engine_for(preset, seed=1234, family="gauss"):
    from piper_amd.engine import Engine
    key = (preset, seed, family)
    if key not in _engines:
        cfg = W.preset(preset)
        w = W.synthetic_weights(cfg, seed, family)
        _engines[key] = (cfg, w, Engine(blob=W.pack_blob(cfg, w), device=0))
    return _engines[key]


def noise_for(cfg, T, seed=7):
    rng = np.random.default_rng(seed)
    nw = rng.standard_normal((2, T)).astype(np.float32)
    nz = rng.standard_normal((cfg.inter, 32 * T + 64)).astype(np.float32)
    return nw, nz


def pcm_rms(a, b):
    d = (a.astype(np.float64) - b.astype(np.float64)) / 32767.0
    return float(np.sqrt(np.mean(d * d))) if d.size else 0.0


def test_native_library_is_loaded():
    from piper_amd import _lib
    lib = _lib.get_lib()
    assert os.path.basename(_lib.LIB_PATH) == "libpiper_hip.so"
    for s in _lib.SYMBOLS:
        assert hasattr(lib, s)


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_hip_matches_reference_golden(path):
    from oracle import vits_oracle as O
    g = np.load(path)
    cfg, w, eng = engine_for(str(g["preset"]), int(g["weight_seed"]))
    T = len(g["ids"])
    nw, nz = noise_for(cfg, T, int(g["noise_seed"]))
    sid = int(g["sid"])
    r = eng.synthesize(g["ids"], tuple(g["scales"]), sid=None if sid < 0 else sid, noise_w=nw, noise_z=nz)
    assert np.array_equal(eng.durations(), g["durations"])
    assert int(r.frames[0]) == int(g["frames"])
    assert r.audio[0].shape == g["audio"].shape
    assert np.max(np.abs(eng.debug_tensor("z") - g["z"])) < 1e-3
    assert np.max(np.abs(r.audio[0] - g["audio"])) < TIGHT_AUDIO_TOL
    # integer work is bit-exact: the int16 conversion of the engine's OWN float waveform (piper.cpp:410-431) -- 0 LSB
    assert np.array_equal(O.audio_float_to_int16(r.audio[0]), r.pcm[0])
    # against the reference's waveform the conversion inherits the float error: d_pcm <= (32767 / peak) * (d_audio +
    # |audio| * d_peak / peak) + 1 (truncation), evaluated with the observed float differences
    ref_pcm = O.audio_float_to_int16(g["audio"])
    assert pcm_rms(r.pcm[0], ref_pcm) <= RMS_TOL
    peak = max(0.01, float(np.max(np.abs(g["audio"]))))
    d_audio = float(np.max(np.abs(r.audio[0] - g["audio"])))
    d_peak = abs(max(0.01, float(np.max(np.abs(r.audio[0])))) - peak)
    bound = 32767.0 / peak * (d_audio + d_peak) + 1.0
    assert np.max(np.abs(r.pcm[0].astype(np.int32) - ref_pcm.astype(np.int32))) <= bound, bound


@pytest.mark.parametrize("preset,T,family", [("medium", 128, "gauss"), ("high", 64, "gauss"), ("x-low", 64, "gauss"),
                                             ("medium", 128, "heavy"), ("high", 64, "heavy")])
def test_hip_matches_oracle_full_size(preset, T, family):
    """BASELINE configs' architectures at their synthetic-input sizes vs the CPU oracle, for two weight families: i.i.d.
    Gaussian layers and a heavy-tailed one (Student-t weights, per-channel gains a factor ~4 apart, 4x biases:
    piper_amd/weights.py) -- real trained voices are not available offline."""
    from oracle import vits_oracle as O
    cfg, w, eng = engine_for(preset, 1234 if family == "gauss" else 4321, family)
    ids = W.synthetic_phoneme_ids(T, 3, id_max=min(cfg.n_vocab - 1, 129))
    nw, nz = noise_for(cfg, T, 11)
    scales = (0.667, 1.0, 0.8)
    o = O.synthesize(w, cfg, ids, scales, nw, nz, keep=True)
    r = eng.synthesize(ids, scales, noise_w=nw, noise_z=nz)
    assert np.array_equal(eng.durations(), o["durations"])
    for name, ref in (("x_enc", o["x_enc"]), ("z", o["z"])):
        got = eng.debug_tensor(name)
        assert got.shape == ref.shape
        assert np.max(np.abs(got - ref)) < 1e-3 * max(1.0, float(np.abs(ref).max())), name
    assert np.max(np.abs(r.audio[0] - o["audio"])) < TIGHT_AUDIO_TOL
    assert pcm_rms(r.pcm[0], o["pcm"]) <= RMS_TOL


def test_ragged_batch_equals_single_utterance_calls():
    """pe_synthesize_batch: every utterance of a ragged batch is computed exactly as its own B=1 call
    (no leakage through padding) and equals the oracle."""
    from oracle import vits_oracle as O
    cfg, w, eng = engine_for("tiny")
    Ts = [5, 33, 17, 64, 9]
    id_lists = [W.synthetic_phoneme_ids(T, i, id_max=cfg.n_vocab - 1) for i, T in enumerate(Ts)]
    Tm, Zs = max(Ts), 32 * max(Ts) + 64
    rng = np.random.default_rng(3)
    nw = rng.standard_normal((len(Ts), 2, Tm)).astype(np.float32)
    nz = rng.standard_normal((len(Ts), cfg.inter, Zs)).astype(np.float32)
    scales = (0.5, 1.2, 0.9)
    rb = eng.synthesize_batch(id_lists, scales, noise_w=nw, noise_z=nz)
    durs = eng.durations()
    off = 0
    for i, ids in enumerate(id_lists):
        o = O.synthesize(w, cfg, ids, scales, nw[i], nz[i])
        assert np.array_equal(durs[off:off + len(ids)], o["durations"])
        off += len(ids)
        assert rb.audio[i].shape == o["audio"].shape
        assert np.max(np.abs(rb.audio[i] - o["audio"])) < TIGHT_AUDIO_TOL
        assert pcm_rms(rb.pcm[i], o["pcm"]) <= RMS_TOL
        # the same utterance alone: identical up to float summation order (the launcher may pick a
        # different tile shape / split-K variant for a different batch size)
        r1 = eng.synthesize(ids, scales, noise_w=nw[i], noise_z=nz[i])
        assert r1.pcm[0].shape == rb.pcm[i].shape
        assert np.max(np.abs(r1.audio[0] - rb.audio[i])) < 2e-5
        for rr, k in ((r1, 0), (rb, i)):       # each int16 stream is the exact conversion of its own float waveform
            assert np.array_equal(O.audio_float_to_int16(rr.audio[k]), rr.pcm[k])


def test_zero_noise_is_deterministic_and_matches_oracle():
    from oracle import vits_oracle as O
    cfg, w, eng = engine_for("tiny")
    ids = W.synthetic_phoneme_ids(40, 9, id_max=cfg.n_vocab - 1)
    a = eng.synthesize(ids, (0.0, 1.0, 0.0))
    b = eng.synthesize(ids, (0.0, 1.0, 0.0))
    assert np.array_equal(a.pcm[0], b.pcm[0])
    o = O.synthesize(w, cfg, ids, (0.0, 1.0, 0.0))
    assert pcm_rms(a.pcm[0], o["pcm"]) <= RMS_TOL


def test_internal_rng_seeded():
    """Without injected noise the engine draws its own N(0,1): same seed + same call index -> same
    audio; a different seed changes it; the waveform stays inside tanh's range."""
    from piper_amd.engine import Engine
    cfg = W.preset("tiny")
    blob = W.pack_blob(cfg, W.synthetic_weights(cfg, 1234))
    ids = W.synthetic_phoneme_ids(30, 2, id_max=cfg.n_vocab - 1)
    outs = []
    for seed in (5, 5, 6):
        e = Engine(blob=blob, device=0)
        e.set_seed(seed)
        outs.append(e.synthesize(ids).audio[0])
        e.close()
    assert outs[0].shape == outs[1].shape and np.array_equal(outs[0], outs[1])
    assert outs[2].shape != outs[0].shape or not np.array_equal(outs[0], outs[2])
    assert all(np.all(np.abs(o) <= 1.0) for o in outs)


def test_length_scale_scales_output_length():
    cfg, w, eng = engine_for("tiny")
    ids = W.synthetic_phoneme_ids(48, 4, id_max=cfg.n_vocab - 1)
    f1 = int(eng.synthesize(ids, (0.0, 1.0, 0.0)).frames[0])
    f2 = int(eng.synthesize(ids, (0.0, 2.0, 0.0)).frames[0])
    assert f1 < f2 <= 2 * f1 + len(ids)
    assert eng.synthesize(ids, (0.0, 2.0, 0.0)).pcm[0].size == f2 * eng.hop


def test_errors_are_reported_not_crashes():
    from piper_amd.engine import EngineError
    cfg, w, eng = engine_for("tiny")
    with pytest.raises(EngineError):
        eng.synthesize([1, 0, cfg.n_vocab, 2])          # id out of range (ORT Gather would throw)
    with pytest.raises(EngineError):
        eng.synthesize_batch([[1, 2], []])                # empty utterance
    r = eng.synthesize([1, 0, 5, 0, 2])                   # still usable afterwards
    assert r.pcm[0].size == int(r.frames[0]) * eng.hop


def test_pcm_peak_normalised_like_reference():
    """piper.cpp:410-431: the loudest sample maps to +-32767 (peak >= 0.01)."""
    cfg, w, eng = engine_for("tiny")
    r = eng.synthesize(W.synthetic_phoneme_ids(32, 1, id_max=cfg.n_vocab - 1))
    assert np.max(np.abs(r.pcm[0].astype(np.int32))) == 32767
    from oracle import vits_oracle as O
    assert np.array_equal(O.audio_float_to_int16(r.audio[0]), r.pcm[0])      # integer work: 0 LSB


def test_piper_voice_loads_reference_export_and_matches_oracle(tmp_path):
    """The drop-in path end to end: a .onnx written by the reference's export code + its .onnx.json ->
    PiperVoice.load (pe_create parses the file on the host, packs, uploads) -> synthesize_ids_to_raw."""
    import json
    import wave
    from oracle import vits_oracle as O
    from piper_amd.voice import PiperVoice
    gold = os.path.join(os.path.dirname(__file__), "golden")
    voice = PiperVoice.load(os.path.join(gold, "tinyhms_voice.onnx"))
    assert voice.config.num_speakers == 4 and voice.config.sample_rate == 16000
    cfg = W.preset("tiny-high-ms")
    w = W.synthetic_weights(cfg, 1234)
    ids = W.synthetic_phoneme_ids(20, 1, id_max=cfg.n_vocab - 1).tolist()
    raw = voice.synthesize_ids_to_raw(ids, speaker_id=3, noise_scale=0.0, noise_w=0.0)
    pcm = np.frombuffer(raw, np.int16)
    o = O.synthesize(w, cfg, ids, (0.0, 1.0, 0.0), sid=3)
    assert pcm.shape == o["pcm"].shape and pcm_rms(pcm, o["pcm"]) <= RMS_TOL
    # mirror of the reference's only test (src/cpp/test.cpp:15-60): text -> WAV, file must not be tiny
    tv = PiperVoice.load(os.path.join(gold, "tiny_voice.onnx"))
    path = str(tmp_path / "test.wav")
    with wave.open(path, "wb") as wf:
        tv.synthesize("This is a test.", wf, sentence_silence=0.1)
    assert os.path.getsize(path) >= 10000
    with wave.open(path, "rb") as wf:
        assert (wf.getframerate(), wf.getsampwidth(), wf.getnchannels()) == (16000, 2, 1)
    batch = tv.synthesize_ids_batch_to_raw([ids, ids[:9] + [2]], noise_scale=0.0, noise_w=0.0)
    assert len(batch) == 2 and len(batch[0]) > len(batch[1]) > 0


def test_cpp_piper_api_on_gpu(tmp_path):
    """tests/cpp/test_piper.cpp (mirror of the reference's src/cpp/test.cpp) against libpiper_hip.so; the PCM that
    piper::synthesize and piper::textToAudio hand back is compared with the ORACLE (not only with itself): equal sample
    counts (= equal integer durations) and int16 RMS <= 1e-3."""
    import ctypes as C
    import json
    import subprocess
    from oracle import vits_oracle as O
    from piper_amd import _lib as L
    from piper_amd.voice import phonemes_to_ids_cpp
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", root, "tests/cpp/test_piper"], stdout=subprocess.DEVNULL)
    wav, dump = str(tmp_path / "t.wav"), str(tmp_path / "dump")
    onnx = os.path.join(root, "tests", "golden", "tiny_voice.onnx")
    out = subprocess.run([os.path.join(root, "tests", "cpp", "test_piper"), onnx, wav, dump],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    assert out.stdout.startswith("OK ") and os.path.getsize(wav) >= 10000
    # the voice's own weights, as the loader reads them from the .onnx
    lib = L.get_lib()
    blob, n = C.c_void_p(), C.c_size_t()
    assert lib.pe_onnx_to_blob(onnx.encode(), C.byref(blob), C.byref(n)) == 0
    cfg, w = W.unpack_blob(C.string_at(blob, n.value))
    lib.pe_free(blob)
    conf = json.load(open(onnx + ".json", encoding="utf-8"))
    ls = float(conf["inference"]["length_scale"])
    # piper::synthesize on explicit ids, noise switched off (test_piper.cpp)
    got = np.fromfile(dump + ".synth.pcm", dtype=np.int16)
    o = O.synthesize(w, cfg, np.array([1, 0, 10, 0, 11, 0, 12, 0, 2], np.int64), (0.0, ls, 0.0))
    assert got.size == o["pcm"].size, (got.size, o["pcm"].size)
    assert pcm_rms(got, o["pcm"]) <= RMS_TOL
    # piper::textToAudio("hello there"): code points -> ids (BOS, PAD, id + PAD ..., EOS), one sentence + its silence
    ids = phonemes_to_ids_cpp(list("hello there"), conf["phoneme_id_map"])
    o2 = O.synthesize(w, cfg, np.array(ids, np.int64), (0.0, ls, 0.0))
    got2 = np.fromfile(dump + ".text.pcm", dtype=np.int16)
    sil = int(0.2 * conf["audio"]["sample_rate"])    # SynthesisConfig::sentenceSilenceSeconds default (piper.hpp:62)
    assert got2.size == o2["pcm"].size + sil, (got2.size, o2["pcm"].size, sil)
    assert not got2[o2["pcm"].size:].any()
    assert pcm_rms(got2[:o2["pcm"].size], o2["pcm"]) <= RMS_TOL


@pytest.mark.parametrize("name,preset", [("medium_voice.onnx", "medium"), ("high_voice.onnx", "high"), ("high_stream", "high")])
def test_full_size_onnx_voices_load_on_gpu(name, preset):
    """FULL-SIZE voices written by the reference's exporter (voices/, made by __graft_entry__.build() where the reference
    is available: en_US-lessac-medium / -high shapes and the high voice's streaming pair) through pe_create on the GPU
    box: the engine built from the .onnx must equal, bit for bit, the engine built from the weight blob of the same
    file, carry the seeded weights (2e-6: the export folds the flow's weight norm), and match the oracle."""
    import ctypes as C
    from oracle import vits_oracle as O
    from piper_amd import _lib as L
    from piper_amd.engine import Engine
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "voices", name)
    if not os.path.exists(path):
        pytest.skip("voices/ not present: run __graft_entry__.build() in the build container (needs /root/reference)")
    eng = Engine(onnx_path=path, device=0)                 # pe_create: parse the .onnx, pack, upload
    lib = L.get_lib()
    blob, n = C.c_void_p(), C.c_size_t()
    assert lib.pe_onnx_to_blob(path.encode(), C.byref(blob), C.byref(n)) == 0, lib.pe_last_error()
    data = C.string_at(blob, n.value)
    lib.pe_free(blob)
    cfg, w = W.unpack_blob(data)
    ref_cfg = W.preset(preset)
    seeded = W.synthetic_weights(ref_cfg, 1234)
    assert set(w) == set(seeded)
    for k in seeded:
        assert w[k].shape == seeded[k].shape and np.max(np.abs(w[k] - seeded[k])) <= 2e-6, k
    eng2 = Engine(blob=data, device=0)
    T = 64
    ids = W.synthetic_phoneme_ids(T, 3, id_max=129)
    nw, nz = noise_for(ref_cfg, T, seed=41)
    scales = (0.667, 1.0, 0.8)
    r1 = eng.synthesize(ids, scales, noise_w=nw, noise_z=nz)
    d1 = eng.durations()
    r2 = eng2.synthesize(ids, scales, noise_w=nw, noise_z=nz)
    assert np.array_equal(d1, eng2.durations()) and np.array_equal(r1.audio[0], r2.audio[0])
    assert np.array_equal(r1.pcm[0], r2.pcm[0])
    o = O.synthesize(w, ref_cfg, ids, scales, nw, nz)
    assert np.array_equal(d1, o["durations"])
    assert np.max(np.abs(r1.audio[0] - o["audio"])) < TIGHT_AUDIO_TOL
    assert pcm_rms(r1.pcm[0], o["pcm"]) <= RMS_TOL
    eng.close()
    eng2.close()


@pytest.mark.parametrize("preset,T,chunk", [("medium", 96, 45), ("high", 40, 45), ("tiny", 50, 7)])
def test_streaming_chunks_equal_unchunked(preset, T, chunk):
    """BASELINE configs[4]: chunked HiFiGAN decode. With the exact receptive-field halo the concatenated
    chunks are the unchunked waveform (same noise), which the reference's heuristic padding is not."""
    cfg, w, eng = engine_for(preset)
    ids = W.synthetic_phoneme_ids(T, 5, id_max=min(cfg.n_vocab - 1, 129))
    nw, nz = noise_for(cfg, T, 13)
    scales = (0.667, 1.0, 0.8)
    full = eng.synthesize(ids, scales, noise_w=nw, noise_z=nz)
    chunks = list(eng.stream(ids, scales, chunk_frames=chunk, noise_w=nw, noise_z=nz))
    assert eng.stream_frames == int(full.frames[0])
    assert len(chunks) == -(-eng.stream_frames // chunk)
    cat = np.concatenate([c[0] for c in chunks])
    assert cat.shape == full.audio[0].shape
    assert np.max(np.abs(cat - full.audio[0])) < 2e-5
    assert all(c[0].size == chunk * eng.hop for c in chunks[:-1])


@pytest.mark.parametrize("preset,T,chunk", [("medium", 96, 45), ("high", 40, 45), ("tiny", 50, 7)])
def test_streaming_chunks_match_the_oracle_chunked_decode(preset, T, chunk):
    """pe_stream_next against the ORACLE's restatement of the reference's chunked decode (infer_onnx_streaming.py:76-124,
    oracle.stream_chunks) on the oracle's own latent z, padded by the engine's exact halo: float chunks within the f32
    tolerance, and the per-chunk int16 -- peak-normalised per chunk like the reference (:122) -- bit-exact on the chunk's
    own float samples and within 1e-3 RMS of the oracle's chunk."""
    from oracle import vits_oracle as O
    cfg, w, eng = engine_for(preset)
    ids = W.synthetic_phoneme_ids(T, 5, id_max=min(cfg.n_vocab - 1, 129))
    nw, nz = noise_for(cfg, T, 13)
    scales = (0.667, 1.0, 0.8)
    o = O.synthesize(w, cfg, ids, scales, nw, nz, keep=True)
    chunks = list(eng.stream(ids, scales, chunk_frames=chunk, noise_w=nw, noise_z=nz))
    assert eng.stream_frames == o["frames"]
    ref = O.stream_chunks(w, cfg, o["z"], chunk, eng.stream_halo)
    assert len(ref) == len(chunks)
    for (a, p), (ra, rp) in zip(chunks, ref):
        assert a.shape == ra.shape and p.shape == rp.shape
        assert np.max(np.abs(a - ra)) < TIGHT_AUDIO_TOL
        assert np.array_equal(O.audio_float_to_int16(a), p)          # integer work: 0 LSB
        assert pcm_rms(p, rp) <= RMS_TOL
    # the exact halo makes the chunked decode equal the unchunked one -- in the oracle too
    assert np.max(np.abs(np.concatenate([c[0] for c in ref]) - o["audio"])) < 1e-5


@pytest.mark.parametrize("T", [1, 2, 700, 1500])
def test_extreme_lengths_match_oracle(T):
    """Shortest possible inputs, one far longer than any test sentence (attention score slab, many column tiles) and
    one whose 32 x T score slab no longer fits LDS (attn_long_kernel: the slab in global memory; the reference has no
    length limit, attentions.py:225-272 builds the full T x T matrix)."""
    from oracle import vits_oracle as O
    cfg, w, eng = engine_for("tiny")
    ids = W.synthetic_phoneme_ids(T, 6, id_max=cfg.n_vocab - 1) if T > 2 else np.array([1, 2][:T], np.int64)
    nw, nz = noise_for(cfg, T, 17)
    o = O.synthesize(w, cfg, ids, (0.667, 1.0, 0.8), nw, nz)
    r = eng.synthesize(ids, (0.667, 1.0, 0.8), noise_w=nw, noise_z=nz)
    assert np.array_equal(eng.durations(), o["durations"])
    assert r.audio[0].shape == o["audio"].shape
    assert np.max(np.abs(r.audio[0] - o["audio"])) < TIGHT_AUDIO_TOL
    assert pcm_rms(r.pcm[0], o["pcm"]) <= RMS_TOL


def test_long_utterance_on_the_medium_voice_matches_oracle():
    """1000 ids through the medium architecture: the attention score slabs of a 192-channel voice (two heads of 96) leave
    LDS at ~830 ids, the text encoder runs attn_long_kernel<96> and the general kernels beyond the small-call limits; equal
    integer durations, encoder output and waveform against the oracle."""
    from oracle import vits_oracle as O
    cfg, w, eng = engine_for("medium")
    T = 1000
    ids = W.synthetic_phoneme_ids(T, 9, id_max=min(cfg.n_vocab - 1, 129))
    nw, nz = noise_for(cfg, T, 23)
    scales = (0.667, 1.0, 0.8)
    o = O.synthesize(w, cfg, ids, scales, nw, nz, keep=True)
    eng.profile_enable(2)
    r = eng.synthesize(ids, scales, noise_w=nw, noise_z=nz)
    names = {row["name"] for row in eng.profile()[5:] if row["launches"]}
    eng.profile_enable(0)
    assert "attn_long_kernel<96>" in names and "attn_kernel<96>" not in names, names
    assert np.array_equal(eng.durations(), o["durations"])
    got = eng.debug_tensor("x_enc")
    assert got.shape == o["x_enc"].shape and np.max(np.abs(got - o["x_enc"])) < 1e-3 * max(1.0, float(np.abs(o["x_enc"]).max()))
    assert r.audio[0].shape == o["audio"].shape
    assert np.max(np.abs(r.audio[0] - o["audio"])) < TIGHT_AUDIO_TOL
    assert pcm_rms(r.pcm[0], o["pcm"]) <= RMS_TOL


def test_all_zero_durations_give_one_frame():
    """length_scale 0 -> every ceil(w) is 0 -> the reference clamps the frame count to 1 and the path matrix
    is empty (models.py:702-716): z_p is then pure prior noise."""
    from oracle import vits_oracle as O
    cfg, w, eng = engine_for("tiny")
    ids = W.synthetic_phoneme_ids(6, 0, id_max=cfg.n_vocab - 1)
    nz = np.ones((cfg.inter, 8), np.float32)
    o = O.synthesize(w, cfg, ids, (0.3, 0.0, 0.0), noise_z=nz)
    r = eng.synthesize(ids, (0.3, 0.0, 0.0), noise_z=nz)
    assert int(r.frames[0]) == 1 == o["frames"] and not eng.durations().any()
    assert np.max(np.abs(r.audio[0] - o["audio"])) < TIGHT_AUDIO_TOL


def test_streaming_export_directory_is_the_same_voice():
    """encoder.onnx + decoder.onnx (reference export_onnx_streaming.py) loaded as a directory give the waveform
    of the single-file export of the same weights, also through the chunked streaming API."""
    from piper_amd.engine import Engine
    gold = os.path.join(os.path.dirname(__file__), "golden")
    cfg = W.preset("tiny-high-ms")
    ids = W.synthetic_phoneme_ids(14, 2, id_max=cfg.n_vocab - 1)
    nw, nz = noise_for(cfg, 14, seed=5)
    one = Engine(onnx_path=os.path.join(gold, "tinyhms_voice.onnx"), device=0)
    two = Engine(onnx_path=os.path.join(gold, "tinyhms_stream"), device=0)
    a = one.synthesize(ids, (0.5, 1.1, 0.7), sid=2, noise_w=nw, noise_z=nz)
    b = two.synthesize(ids, (0.5, 1.1, 0.7), sid=2, noise_w=nw, noise_z=nz)
    assert np.array_equal(a.audio[0], b.audio[0]) and np.array_equal(a.pcm[0], b.pcm[0])
    chunks = [c[0] for c in two.stream(ids, (0.5, 1.1, 0.7), sid=2, noise_w=nw, noise_z=nz, chunk_frames=9)]
    assert np.max(np.abs(np.concatenate(chunks) - a.audio[0])) < 1e-6
    one.close()
    two.close()


def test_jsonl_drivers_on_gpu(tmp_path):
    """piper_amd.infer / piper_amd.benchmark (the reference's infer_onnx.py / benchmark_onnx.py command lines: JSONL of
    phoneme ids on stdin) against libpiper_hip.so on the GPU: the WAVs written with noise off are the oracle's PCM."""
    import io
    import json
    import wave
    from oracle import vits_oracle as O
    from piper_amd import benchmark, infer
    gold = os.path.join(os.path.dirname(__file__), "golden")
    model = os.path.join(gold, "tinyhms_voice.onnx")
    cfg = W.preset("tiny-high-ms")
    w = W.synthetic_weights(cfg, 1234)
    utts = [([1, 5, 7, 9, 11, 2], 1), ([1, 9, 3, 3, 4, 8, 20, 2], None), ([1, 4, 2], 3)]
    lines = [json.dumps({"phoneme_ids": ids, **({"speaker_id": sid} if sid is not None else {})}) for ids, sid in utts]
    lines.insert(1, "")                                     # a blank line keeps its index (reference behaviour)
    out = tmp_path / "wavs"
    assert infer.main(["--model", model, "--output-dir", str(out), "--sample-rate", "16000", "--batch", "2",
                       "--noise-scale", "0", "--noise-scale-w", "0"], stdin=io.StringIO("\n".join(lines))) == 0
    assert sorted(p.name for p in out.iterdir()) == ["0.wav", "2.wav", "3.wav"]
    for name, (ids, sid) in zip(("0.wav", "2.wav", "3.wav"), utts):
        with wave.open(str(out / name), "rb") as wf:
            assert (wf.getframerate(), wf.getnchannels(), wf.getsampwidth()) == (16000, 1, 2)
            pcm = np.frombuffer(wf.readframes(wf.getnframes()), np.int16)
        o = O.synthesize(w, cfg, np.array(ids, np.int64), (0.0, 1.0, 0.0), sid=sid or 0)
        assert pcm.shape == o["pcm"].shape and pcm_rms(pcm, o["pcm"]) <= RMS_TOL, name
    buf = io.StringIO()
    assert benchmark.main(["-m", model], stdin=io.StringIO("\n".join(lines)), stdout=buf) == 0
    rep = json.loads(buf.getvalue())
    assert set(rep) == {"load_sec", "rtf_mean", "rtf_stdev", "rtfs"} and len(rep["rtfs"]) == 3
    assert all(r > 0 for r in rep["rtfs"]) and rep["rtfs"][-1] < 1 and rep["load_sec"] > 0
