"""CPU tests: the structural .onnx loader (onnx_reader.cpp) against files rewritten the way post-export tooling
rewrites real voices (SURVEY.md section 7 hard part A; TRAINING.md:234 recommends onnx-simplifier):
numeral initialiser names, stripped node names, weights in Constant nodes, de-duplicated tensors, float_data
encoding -- and, when the reference is available (build container), a freshly exported FULL-SIZE medium voice."""
import ctypes as C
import dataclasses
import os

import numpy as np
import pytest

from oracle import onnx_mutate as M
from piper_amd import _lib as L
from piper_amd import weights as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def lib():
    return L.get_lib()


def load(lib, path):
    blob, n = C.c_void_p(), C.c_size_t()
    if lib.pe_onnx_to_blob(str(path).encode(), C.byref(blob), C.byref(n)):
        raise RuntimeError(lib.pe_last_error().decode())
    data = C.string_at(blob, n.value)
    lib.pe_free(blob)
    return W.unpack_blob(data)


def check(cfg2, w2, cfg, w, tol=1e-6):
    assert dataclasses.replace(cfg2, sample_rate=cfg.sample_rate) == cfg
    assert set(w2) == set(w)
    for k in w:
        assert w2[k].shape == w[k].shape, k
        assert np.max(np.abs(w2[k] - w[k])) <= tol, k


VARIANTS = {
    "numeral_names": lambda m: (m.rename_initializers_to_numerals(), m.strip_node_names()),
    "all_constants": lambda m: m.initializers_to_constants(1),
    "half_constants_numerals": lambda m: (m.rename_initializers_to_numerals(7), m.initializers_to_constants(2),
                                          m.strip_node_names()),
    "float_data": lambda m: m.float_data(),
    "everything": lambda m: (m.float_data(), m.dedup(), m.rename_initializers_to_numerals(), m.initializers_to_constants(3),
                             m.strip_node_names()),
}


@pytest.mark.parametrize("stem,preset", [("tiny_voice", "tiny"), ("tinyhms_voice", "tiny-high-ms")])
@pytest.mark.parametrize("variant", sorted(VARIANTS))
def test_rewritten_exports_load_identically(lib, tmp_path, stem, preset, variant):
    cfg = W.preset(preset)
    w = W.synthetic_weights(cfg, 1234)
    m = M.Model(open(os.path.join(GOLD, stem + ".onnx"), "rb").read())
    n_before = len(m.inits)
    VARIANTS[variant](m)
    out = tmp_path / f"{stem}_{variant}.onnx"
    out.write_bytes(m.save())
    if variant == "all_constants":
        assert len(m.inits) == 0 and sum(1 for n in m.nodes if (4, 2, b"Constant") in n) >= n_before
    cfg2, w2 = load(lib, out)
    check(cfg2, w2, cfg, w)


def test_deduplicated_layernorm_parameters(lib, tmp_path):
    """An untrained-looking voice: every LayerNorm gain 1 / offset 0. A simplifier folds the ~60 identical vectors into
    two initialisers that many nodes share; the loader assigns them by graph position, not by name or identity."""
    cfg = W.preset("tiny")
    w = dict(W.synthetic_weights(cfg, 1234))
    m = M.Model(open(os.path.join(GOLD, "tiny_voice.onnx"), "rb").read())
    touched = 0
    for t in m.inits:
        if t.name.endswith(".gamma"):
            t.set_constant(1.0); touched += 1
        elif t.name.endswith(".beta"):
            t.set_constant(0.0); touched += 1
    assert touched >= 20
    for k in w:
        if k.endswith(".gamma"):
            w[k] = np.ones_like(w[k])
        elif k.endswith(".beta"):
            w[k] = np.zeros_like(w[k])
    merged = m.dedup()
    assert merged >= touched - 2
    m.rename_initializers_to_numerals()
    m.strip_node_names()
    out = tmp_path / "dedup.onnx"
    out.write_bytes(m.save())
    cfg2, w2 = load(lib, out)
    check(cfg2, w2, cfg, w)


def test_resblock1_detected_without_names(lib, tmp_path):
    """ResBlock1 vs ResBlock2 must be told apart from the dilation pattern once node / tensor names are gone."""
    m = M.Model(open(os.path.join(GOLD, "tinyhms_voice.onnx"), "rb").read())
    m.rename_initializers_to_numerals()
    m.strip_node_names()
    out = tmp_path / "anon.onnx"
    out.write_bytes(m.save())
    cfg2, _ = load(lib, out)
    assert cfg2.resblock == 1 and cfg2.rb_kernel_sizes == (3, 7, 11)
    m2 = M.Model(open(os.path.join(GOLD, "tiny_voice.onnx"), "rb").read())
    m2.rename_initializers_to_numerals()
    m2.strip_node_names()
    out2 = tmp_path / "anon2.onnx"
    out2.write_bytes(m2.save())
    assert load(lib, out2)[0].resblock == 2


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/python/piper_train"),
                    reason="needs the reference's exporter (build container only)")
def test_full_size_medium_export_loads(lib, tmp_path):
    """A full-size en_US-lessac-medium-shaped voice written by the reference's own export path at test time (63 MB, not
    committed), also after the rewrites above."""
    import subprocess
    import sys
    prefix = str(tmp_path / "medium_voice")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "oracle", "make_voice.py"), "medium", prefix],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=1200)
    size = os.path.getsize(prefix + ".onnx")
    assert abs(size - 63201294) / 63201294 < 0.01          # voices.json: en_US-lessac-medium is 63 201 294 bytes
    cfg = W.preset("medium")
    w = W.synthetic_weights(cfg, 1234)
    cfg2, w2 = load(lib, prefix + ".onnx")
    check(cfg2, w2, cfg, w, tol=2e-6)
    m = M.Model(open(prefix + ".onnx", "rb").read())
    VARIANTS["everything"](m)
    out = tmp_path / "medium_everything.onnx"
    out.write_bytes(m.save())
    cfg3, w3 = load(lib, out)
    check(cfg3, w3, cfg, w, tol=2e-6)


# ---- round 5: what onnx-simplifier / torch-1.x exports leave behind --------------------------------
R5_VARIANTS = {
    "identity_shared": lambda m: m.identity_shared(2),
    "unsqueeze_weights_axes_input": lambda m: m.unsqueeze_k1_weights(True),
    "unsqueeze_weights_axes_attribute": lambda m: m.unsqueeze_k1_weights(False),
    "transposed_weights": lambda m: m.transpose_conv_weights(1),
    "squeezed_biases": lambda m: m.reshape_biases(),
    "opset_lt13_attribute_forms": lambda m: m.axes_inputs_to_attributes(),
    "all_of_them": lambda m: (m.axes_inputs_to_attributes(), m.transpose_conv_weights(2), m.unsqueeze_k1_weights(True),
                              m.reshape_biases(), m.identity_shared(3), m.rename_initializers_to_numerals(), m.strip_node_names()),
    "conv_bias_as_add": lambda m: m.conv_bias_to_add(2),
    "every_conv_bias_as_add": lambda m: m.conv_bias_to_add(1),
    "gelu_div_folded_pow_as_mul": lambda m: m.gelu_div_to_mul(),
    "bias_add_gelu_everything": lambda m: (m.gelu_div_to_mul(), m.conv_bias_to_add(1), m.identity_shared(3),
                                           m.rename_initializers_to_numerals(), m.strip_node_names()),
    "glue_and_qkv_reordered": lambda m: m.shuffle_nodes(7, keep_conv_order=True),
    "glue_and_qkv_reordered_anonymous": lambda m: (m.shuffle_nodes(11, keep_conv_order=True), m.rename_initializers_to_numerals(),
                                                   m.strip_node_names()),
}


@pytest.mark.parametrize("stem,preset", [("tiny_voice", "tiny"), ("tinyhms_voice", "tiny-high-ms")])
@pytest.mark.parametrize("variant", sorted(R5_VARIANTS))
def test_post_export_rewrites_load_identically(lib, tmp_path, stem, preset, variant):
    """Identity-shared initialisers, conv weights behind Unsqueeze (both axes forms) / Transpose, biases behind Squeeze,
    Squeeze / Unsqueeze / Split in the pre-opset-13 attribute form, and another topological order of everything but the
    sequential convolutions (q / k / v re-ordered: the loader tells them apart by their place in the attention products):
    each rewrite must give the unmutated file's weights, bit for bit."""
    cfg, w = load(lib, os.path.join(GOLD, stem + ".onnx"))          # the unmutated file as the loader reads it
    check(cfg, w, W.preset(preset), W.synthetic_weights(W.preset(preset), 1234))
    m = M.Model(open(os.path.join(GOLD, stem + ".onnx"), "rb").read())
    n = R5_VARIANTS[variant](m)
    if isinstance(n, int):
        assert n > 0, "the mutation did not apply to this file"
    out = tmp_path / f"{stem}_{variant}.onnx"
    out.write_bytes(m.save())
    cfg2, w2 = load(lib, out)
    check(cfg2, w2, cfg, w, tol=0.0)


def test_external_data_is_a_named_error(lib, tmp_path):
    """A tensor whose payload lives in another file (data_location = EXTERNAL): not supported, and said so with the
    tensor's name -- never a silent load of zeros."""
    m = M.Model(open(os.path.join(GOLD, "tiny_voice.onnx"), "rb").read())
    name = m.make_external(5)
    out = tmp_path / "external.onnx"
    out.write_bytes(m.save())
    with pytest.raises(RuntimeError) as ei:
        load(lib, out)
    assert "external tensor data is not supported" in str(ei.value) and name in str(ei.value)


@pytest.mark.parametrize("seed", range(6))
def test_any_topological_order_loads_identically_or_fails_by_name(lib, tmp_path, seed):
    """ONNX demands only A topological order. The loader walks the convolutions in file order (the exporter's execution
    order); after the walk every producer -> consumer adjacency it assumed is verified in the graph, so a file whose
    parallel branches (resblocks of an MRF stage, speaker-conditioning convs) come in another order either loads to exactly
    the unmutated weights or fails with an error that says so -- it never loads a tensor under the wrong name."""
    stem, preset = ("tiny_voice", "tiny") if seed % 2 == 0 else ("tinyhms_voice", "tiny-high-ms")
    cfg, w = load(lib, os.path.join(GOLD, stem + ".onnx"))
    m = M.Model(open(os.path.join(GOLD, stem + ".onnx"), "rb").read())
    m.shuffle_nodes(100 + seed)
    out = tmp_path / f"shuffled_{seed}.onnx"
    out.write_bytes(m.save())
    try:
        cfg2, w2 = load(lib, out)
    except RuntimeError as ex:
        assert "onnx: voice graph does not match the Piper VITS export" in str(ex), str(ex)
        return
    check(cfg2, w2, cfg, w, tol=0.0)


@pytest.mark.parametrize("pair", [(0, 1), (1, 3), (0, 3)])
def test_swapped_cond_layers_fail_by_name(lib, tmp_path, pair):
    """the four flow cond_layer convs share one shape and read only g, so a file that lists two of
    them in exchanged positions is a valid graph the positional walk would load with their weights exchanged. Every
    cond conv is now tied to a conv that consumes the sum it feeds (onnx_reader.cpp: the links of dp.cond, cond_layer,
    dec.cond) and verified like every other adjacency: the swapped file fails by name, it never loads."""
    m = M.Model(open(os.path.join(GOLD, "tinyhms_voice.onnx"), "rb").read())
    na, nb = m.swap_cond_layers(*pair)
    assert na != nb
    out = tmp_path / "swapped_cond.onnx"
    out.write_bytes(m.save())
    with pytest.raises(RuntimeError) as ei:
        load(lib, out)
    msg = str(ei.value)
    assert "node order is not the exporter's execution order" in msg and "cond_layer" in msg, msg


@pytest.mark.parametrize("as_input", [True, False])
def test_crafted_unsqueeze_axes_fail_cleanly(lib, tmp_path, as_input):
    """an Unsqueeze whose axes repeat (a crafted file) used to mark fewer slots than it counted and read
    past the source dims. The folder now refuses it, so the conv weight behind it is simply not a constant: a named error."""
    m = M.Model(open(os.path.join(GOLD, "tiny_voice.onnx"), "rb").read())
    assert m.unsqueeze_k1_weights(as_input, axes=(2, 2)) > 0
    out = tmp_path / "dup_axes.onnx"
    out.write_bytes(m.save())
    with pytest.raises(RuntimeError) as ei:
        load(lib, out)
    assert "is not a rank-3 constant" in str(ei.value), str(ei.value)
