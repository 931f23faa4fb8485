"""CPU tests of the boundary: the C-ABI library loads and exports every symbol of
include/piper_hip.h, the .onnx loader (host code, no GPU) recovers the canonical weights from files
written by the reference's own export path, the voice config and phoneme-id logic match the
reference's known-answer data. No compute calls."""
import ctypes as C
import dataclasses
import json
import os
import re

import numpy as np
import pytest

from piper_amd import _lib as L
from piper_amd import weights as W
from piper_amd.config import PhonemeType, PiperConfig
from piper_amd.voice import BOS, EOS, PAD, PiperVoice, phonemes_to_ids_cpp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(L.LIB_PATH):
        import subprocess
        subprocess.check_call(["make", "-C", ROOT, "all"])
    return L.get_lib()


def test_library_exports_every_declared_symbol(lib):
    header = open(os.path.join(ROOT, "include", "piper_hip.h")).read()
    declared = set(re.findall(r"\b(pe_[a-z_0-9]+)\s*\(", header))
    assert declared == set(L.SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s


def _onnx_to_blob(lib, path):
    blob, n = C.c_void_p(), C.c_size_t()
    rc = lib.pe_onnx_to_blob(path.encode(), C.byref(blob), C.byref(n))
    if rc:
        raise RuntimeError(lib.pe_last_error().decode())
    data = C.string_at(blob, n.value)
    lib.pe_free(blob)
    return data


@pytest.mark.parametrize("stem,preset", [("tiny_voice", "tiny"), ("tinyhms_voice", "tiny-high-ms")])
def test_onnx_loader_recovers_canonical_weights(lib, stem, preset):
    """Files produced by the reference's export path (oracle/make_voice.py): text embedding named
    'sid', anonymous weight-norm-folded flow weights, folded exp(-logs), ResBlock1/2, speaker cond."""
    cfg2, w2 = W.unpack_blob(_onnx_to_blob(lib, os.path.join(GOLD, stem + ".onnx")))
    cfg = W.preset(preset)
    assert dataclasses.replace(cfg2, sample_rate=cfg.sample_rate) == cfg      # rate lives in the .json
    w = W.synthetic_weights(cfg, 1234)
    assert set(w2) == set(w)
    for k in w:
        assert w2[k].shape == w[k].shape, k
        assert np.max(np.abs(w2[k] - w[k])) <= 1e-6, k      # weight_norm fold g*v/|v| rounding only


@pytest.mark.parametrize("stem,preset", [("tiny_stream", "tiny"), ("tinyhms_stream", "tiny-high-ms")])
def test_onnx_loader_reads_streaming_export(lib, stem, preset):
    """encoder.onnx + decoder.onnx written by the reference's export_onnx_streaming.py (fixtures made by
    oracle/make_voice.py --streaming through the reference's own VitsEncoder / VitsDecoder wrappers) load as one
    voice: by directory or by either file's path."""
    cfg = W.preset(preset)
    w = W.synthetic_weights(cfg, 1234)
    d = os.path.join(GOLD, stem)
    for path in (d, os.path.join(d, "encoder.onnx"), os.path.join(d, "decoder.onnx")):
        blob, n = C.c_void_p(), C.c_size_t()
        assert lib.pe_onnx_to_blob(path.encode(), C.byref(blob), C.byref(n)) == 0, lib.pe_last_error()
        arch, tensors = W.unpack_blob(C.string_at(blob, n.value))
        lib.pe_free(blob)
        assert set(tensors) == set(w)
        for k, v in w.items():
            assert tensors[k].shape == v.shape and np.max(np.abs(tensors[k] - v)) < 1e-6, k


def test_onnx_loader_errors(lib, tmp_path):
    with pytest.raises(RuntimeError, match="cannot open"):
        _onnx_to_blob(lib, str(tmp_path / "missing.onnx"))
    junk = tmp_path / "junk.onnx"
    junk.write_bytes(b"\x00\x01garbage" * 100)
    with pytest.raises(RuntimeError):
        _onnx_to_blob(lib, str(junk))
    good = open(os.path.join(GOLD, "tiny_voice.onnx"), "rb").read()
    cut = tmp_path / "cut.onnx"
    cut.write_bytes(good[: len(good) // 2])
    with pytest.raises(RuntimeError):
        _onnx_to_blob(lib, str(cut))


def test_blob_parser_rejects_bad_input(lib):
    h = C.c_void_p()
    bad = b"PEBLOB01" + b"\x00" * 64
    assert lib.pe_create_from_blob(bad, len(bad), 0, C.byref(h)) != 0
    assert b"blob" in lib.pe_last_error() or b"truncated" in lib.pe_last_error()


def test_config_parse_matches_reference_rules():
    conf = json.load(open(os.path.join(GOLD, "tiny_voice.onnx.json")))
    c = PiperConfig.from_dict(conf)
    assert (c.sample_rate, c.num_speakers, c.num_symbols) == (16000, 1, 40)
    assert (c.noise_scale, c.length_scale, c.noise_w) == (0.667, 1, 0.8)
    assert c.phoneme_type == PhonemeType.TEXT
    minimal = {"num_symbols": 5, "num_speakers": 1, "audio": {"sample_rate": 22050}, "espeak": {"voice": "en-us"},
               "phoneme_id_map": {"_": [0], "^": [1], "$": [2]}}
    d = PiperConfig.from_dict(minimal)         # defaults of config.py:41-52
    assert (d.noise_scale, d.length_scale, d.noise_w, d.phoneme_type) == (0.667, 1.0, 0.8, PhonemeType.ESPEAK)
    with pytest.raises(KeyError):
        PiperConfig.from_dict({"num_symbols": 5})


def test_phonemes_to_ids_against_reference_fixture():
    """etc/test_sentences/test_en-us.jsonl pins the C++ rule [^, _, (id, _)*, $]."""
    g = json.load(open(os.path.join(GOLD, "phoneme_ids_en-us.json"), encoding="utf-8"))
    assert len(g["rows"]) >= 3
    for row in g["rows"]:
        assert phonemes_to_ids_cpp(row["phonemes"], g["phoneme_id_map"]) == row["phoneme_ids"]
    # the Python runtime's variant (voice.py:72-87) drops the PAD after BOS
    v = PiperVoice(session=None, config=PiperConfig(0, 1, 16000, "en-us", 1.0, 0.667, 0.8, g["phoneme_id_map"],
                                                    PhonemeType.ESPEAK))
    row = g["rows"][0]
    py = v.phonemes_to_ids(row["phonemes"])
    assert py == [row["phoneme_ids"][0]] + row["phoneme_ids"][2:]
    assert v.phonemes_to_ids(["☃"]) == g["phoneme_id_map"][BOS] + g["phoneme_id_map"][EOS]   # unknown skipped
    assert PAD in g["phoneme_id_map"] and EOS in g["phoneme_id_map"]


def test_cpp_piper_api_on_emulator(tmp_path):
    """The reference-compatible C++ API (include/piper.hpp: loadVoice / textToWavFile / synthesize /
    phonemes_to_ids) compiled against the emulator build: host logic only (JSON config parsing, UTF-8,
    phrase/silence handling, WAV header, error propagation). The same program runs on the GPU in
    tests/test_gpu_parity.py."""
    import subprocess
    import wave
    subprocess.check_call(["make", "-C", ROOT, "emu", "tests/cpp/test_piper_emu"], stdout=subprocess.DEVNULL)
    wav = str(tmp_path / "t.wav")
    out = subprocess.run([os.path.join(ROOT, "tests", "cpp", "test_piper_emu"),
                          os.path.join(GOLD, "tiny_voice.onnx"), wav], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, PIPER_TEST_SHORT="1"))      # the GPU run of the same program uses the long text
    assert out.returncode == 0, out.stderr
    assert out.stdout.startswith("OK ") and "rate=16000" in out.stdout and "speakers=1" in out.stdout
    assert "pid=7 missing=1" in out.stdout          # a, b mapped (+PADs), the snowman counted as missing
    with wave.open(wav, "rb") as wf:
        assert (wf.getframerate(), wf.getsampwidth(), wf.getnchannels()) == (16000, 2, 1)
        assert wf.getnframes() * 2 + 44 == os.path.getsize(wav) >= 10000


def test_reference_voice_config_through_load_voice():
    """The reference's own voice config, etc/test_voice.onnx.json (a fixture: it is data, copied byte for byte to
    tests/golden/ref_test_voice.onnx.json), through piper::loadVoice's JSON reader in the C++ program: every field the
    reference's parsers fill (src/cpp/piper.cpp:47-132 phonemize, :135-195 synthesis, :197-214 model) -- eSpeak voice,
    phoneme type default, an EMPTY phoneme_map that still creates the map (:117-131), all 130 phoneme-id entries, scales,
    sample rate, num_speakers, an empty speaker_id_map, no speaker selected for a single-speaker voice (:326-331)."""
    import subprocess
    ref = os.path.join(GOLD, "ref_test_voice.onnx.json")
    if os.path.exists("/root/reference/etc/test_voice.onnx.json"):          # the fixture IS the reference's file
        assert open(ref, "rb").read() == open("/root/reference/etc/test_voice.onnx.json", "rb").read()
    subprocess.check_call(["make", "-C", ROOT, "emu", "tests/cpp/test_piper_emu"], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(ROOT, "tests", "cpp", "test_piper_emu"), "--config", os.path.join(GOLD, "tiny_voice.onnx"), ref],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    kv, ids = {}, {}
    for line in out.stdout.splitlines():
        if line.startswith("id U+"):
            parts = line.split()
            ids[chr(int(parts[1][2:], 16))] = [int(x) for x in parts[2:]]
        else:
            for tok in line.split():
                k, _, v = tok.partition("=")
                kv[k] = v
    cfg = json.load(open(ref, encoding="utf-8"))
    assert kv["phoneme_type"] == "espeak" and kv["espeak_voice"] == cfg["espeak"]["voice"] == "en-us"
    assert kv["phoneme_map"] == "0"                       # present and empty: the map exists, with no entries
    assert int(kv["phoneme_id_map"]) == len(cfg["phoneme_id_map"]) == 130
    assert ids == {k: v for k, v in cfg["phoneme_id_map"].items()}
    assert (kv["id_pad"], kv["id_bos"], kv["id_eos"]) == ("0", "1", "2")          # piper.hpp:44-47
    assert kv["interspersePad"] == "1"
    assert kv["sample_rate"] == "16000" and kv["sample_width"] == "2" and kv["channels"] == "1"
    assert abs(float(kv["noise_scale"]) - 0.667) < 1e-6 and float(kv["length_scale"]) == 1.0
    assert abs(float(kv["noise_w"]) - 0.8) < 1e-6 and abs(float(kv["sentence_silence"]) - 0.2) < 1e-6
    assert kv["phoneme_silence"] == "none" and kv["speaker_id"] == "none"
    assert kv["num_speakers"] == "1" and kv["speaker_id_map"] == "0"
    assert int(kv["config_text_bytes"]) == os.path.getsize(ref)


def test_jsonl_drivers_on_emulator(tmp_path):
    """piper_amd.infer / piper_amd.benchmark mirror the reference's infer_onnx.py / benchmark_onnx.py command
    lines (JSONL of phoneme ids on stdin); run here against the emulator build of the engine."""
    import io
    import wave
    from piper_amd import _lib as L, benchmark, infer
    emu = os.path.join(ROOT, "tests", "emu", "libpiper_hip_emu.so")
    if not os.path.exists(emu):
        subprocess.check_call(["make", "-C", ROOT, "emu"])
    elib = L.bind(emu)
    model = os.path.join(GOLD, "tinyhms_voice.onnx")
    lines = ['{"phoneme_ids": [1, 5, 2], "speaker_id": 1}', "", '{"phoneme_ids": [1, 9]}']
    out = tmp_path / "wavs"
    assert infer.main(["--model", model, "--output-dir", str(out), "--sample-rate", "16000", "--batch", "2",
                       "--seed", "3"], stdin=io.StringIO("\n".join(lines)), lib=elib) == 0
    assert sorted(p.name for p in out.iterdir()) == ["0.wav", "2.wav"]       # blank line keeps its index
    with wave.open(str(out / "0.wav"), "rb") as w:
        assert (w.getframerate(), w.getnchannels(), w.getsampwidth()) == (16000, 1, 2)
        assert w.getnframes() > 0 and w.getnframes() % 256 == 0
    buf = io.StringIO()
    assert benchmark.main(["-m", model], stdin=io.StringIO("\n".join(lines)), stdout=buf, lib=elib) == 0
    rep = json.loads(buf.getvalue())
    assert set(rep) == {"load_sec", "rtf_mean", "rtf_stdev", "rtfs"} and len(rep["rtfs"]) == 2
    assert all(r > 0 for r in rep["rtfs"]) and rep["load_sec"] > 0


def test_blob_parser_survives_hostile_offsets(lib):
    """a record whose offset + 4*numel wraps uint64 must be rejected, not memcpy'd."""
    cfg = W.preset("tiny")
    blob = bytearray(W.pack_blob(cfg, W.synthetic_weights(cfg, 1)))
    head = 8 + 4 * W.ARCH_INTS + 8
    rec = head                                    # first record: name[96] ndim dims[4] pad offset numel
    import struct
    for off, numel in ((2 ** 64 - 64, 32), (len(blob) - 8, 2 ** 62), (2 ** 63, 2 ** 62)):
        bad = bytearray(blob)
        struct.pack_into("<QQ", bad, rec + W.NAME_BYTES + 24, off, numel)
        h = C.c_void_p()
        assert lib.pe_create_from_blob(bytes(bad), len(bad), 0, C.byref(h)) != 0
        assert b"truncated" in lib.pe_last_error() or b"mismatch" in lib.pe_last_error()


def test_null_handles_are_errors_not_crashes(lib):
    name, ms, fl, n, by = C.c_char_p(), C.c_double(), C.c_double(), C.c_int64(), C.c_double()
    assert lib.pe_profile_enable(None, 1) != 0
    assert lib.pe_profile_reset(None) != 0
    assert lib.pe_profile_rows(None) == 0
    assert lib.pe_profile_get(None, 0, C.byref(name), C.byref(ms), C.byref(fl), C.byref(n)) != 0
    assert lib.pe_profile_bytes(None, 0, C.byref(by)) != 0
    r, c = C.c_int32(), C.c_int32()
    buf = (C.c_float * 4)()
    assert lib.pe_debug_tensor(None, b"z", 0, buf, 4, C.byref(r), C.byref(c)) != 0
    assert lib.pe_debug_randn(None, 0, 1, 0, 4, buf) != 0
    assert lib.pe_speculation_stats(None, None, None) != 0
    assert lib.pe_xcc_pattern(None, None, None) != 0
    assert lib.pe_rng_calls(None) == 0 and lib.pe_run_launches(None) == 0
    lib.pe_destroy(None)


def test_text_front_end_casefold_nfd_matches_cpp():
    """phoneme_type "text": piper-phonemize's phonemize_codepoints = full case folding, then NFD. The
    Python mirror must agree with the C++ shim (tests/cpp/test_piper.cpp checks the same strings there)."""
    g = json.load(open(os.path.join(GOLD, "tiny_voice.onnx.json"), encoding="utf-8"))
    v = PiperVoice(session=None, config=PiperConfig.from_dict(g))
    assert v.phonemize("This IS a TÉST Å ß 각") == [list("this is a tést å ss 각")]
    assert v.phonemize("ạ́") == [list("ạ́")]
    assert v.phonemes_to_ids(v.phonemize("HELLO")[0]) == v.phonemes_to_ids(v.phonemize("hello")[0])
    # the generated C++ tables are exactly unicodedata's
    hdr = open(os.path.join(ROOT, "piper_amd", "csrc", "unicode_tables.h")).read()
    assert f"uni_fold_count = {sum(1 for c in map(chr, range(0x110000)) if not 0xD800 <= ord(c) <= 0xDFFF and c.casefold() != c)};" in hdr


def test_launch_policy_table_is_exported_and_documented(lib):
    """pe_policy_describe: the knob table of piper_amd/csrc/policy.h as JSON. Every knob has a default inside its range,
    and DESIGN.md section 4.1 documents exactly the knobs the code reads (plus the string-valued PIPER_HIP_MATRIX and the
    group-level PIPER_HIP_GROUP_BCAST / PIPER_HIP_GROUP_COALESCE, which are not launch-policy integers)."""
    knobs = json.loads(lib.pe_policy_describe().decode())
    envs = [k["env"] for k in knobs]
    assert len(envs) == len(set(envs)) >= 20 and all(e.startswith("PIPER_HIP_") for e in envs)
    for k in knobs:
        assert k["lo"] <= k["default"] <= k["hi"] and k["doc"], k
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    sec = design[design.index("### 4.1"):design.index("## 5.")]
    documented = set(re.findall(r"`(PIPER_HIP_[A-Z0-9_]+)`", sec))
    assert documented - {"PIPER_HIP_MATRIX", "PIPER_HIP_GROUP_BCAST", "PIPER_HIP_GROUP_COALESCE"} == set(envs)
    # no other translation unit of the engine reads a PIPER_HIP_* integer on its own
    for fn in ("engine.cpp", "engine_pack.cpp", "engine_launch.cpp", "engine_issue.cpp", "pe_api.cpp"):
        path = os.path.join(ROOT, "piper_amd", "csrc", fn)
        if os.path.exists(path):
            reads = set(re.findall(r'getenv\("(PIPER_HIP_[A-Z0-9_]+)"\)', open(path).read()))
            assert reads <= {"PIPER_HIP_GROUP_BCAST", "PIPER_HIP_GROUP_COALESCE"}, (fn, reads)
