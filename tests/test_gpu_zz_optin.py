"""GPU parity of OPT-IN modes that have not been measured yet. Kept in a module of its own that sorts last: the driver runs
the suite with -x, and a mode nobody has seen on the hardware must not be able to cut the parity tests of the default path
short."""
import pytest

from test_gpu_batched import batch_inputs, make_engine, run_and_check, voice

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("lens", [[128, 40, 7], [128]])
def test_fused_wn_layers_match_oracle(monkeypatch, lens):
    """PIPER_HIP_WN=1: every WN layer of the coupling flow as one launch (kernels/wn.h: gated channels dealt to the
    workgroups, partial res / skip products summed by the next layer's launch and by the post conv). Verified on the
    emulator (tests/test_emu_engine.py); this is its first run on the hardware."""
    cfg, w = voice("medium")
    eng = make_engine(monkeypatch, cfg, w, {"PIPER_HIP_WN": 1})
    ids, nw, nz = batch_inputs(cfg, lens, seed=97 + len(lens))
    names, worst = run_and_check(eng, cfg, w, ids, nw, nz, sample=sorted({0, len(lens) - 1}))
    eng.close()
    assert {"wn_kernel", "colchain4_kernel"} <= names, sorted(names)
    print("fused WN layers, kernels:", sorted(names), "worst |d audio| %.2e" % worst)


@pytest.mark.parametrize("preset,lens", [("medium", [128]), ("high", [64, 17])])
def test_preloaded_small_k_upconvs_match_oracle(monkeypatch, preset, lens):
    """PIPER_HIP_UPPRE=1: the late polyphase up-convs of a small call through conv_small_kernel (kernels/conv_small.h:
    every weight fragment and x slab requested before the first MFMA). Bit-identical to the tiled kernel on the emulator;
    this is its first run on the hardware."""
    cfg, w = voice(preset)
    eng = make_engine(monkeypatch, cfg, w, {"PIPER_HIP_UPPRE": 1})
    ids, nw, nz = batch_inputs(cfg, lens, seed=131 + len(lens))
    names, worst = run_and_check(eng, cfg, w, ids, nw, nz, sample=sorted({0, len(lens) - 1}))
    eng.close()
    assert "conv_small_kernel" in names, sorted(names)
    print("preloaded up-convs, kernels:", sorted(names), "worst |d audio| %.2e" % worst)


def test_deep_ring_sum_kernel_matches_oracle(monkeypatch):
    """PIPER_HIP_SUMD=16: the K-concatenated last convs of the 128-channel stage's sibling resblocks with a wave's whole K
    range in flight at kernel entry (conv_splitk_sum_kernel<4,16>); bit-identical to the 2-deep ring on the emulator."""
    cfg, w = voice("medium")
    eng = make_engine(monkeypatch, cfg, w, {"PIPER_HIP_SUMD": 16})
    ids, nw, nz = batch_inputs(cfg, [117], seed=151)
    names, worst = run_and_check(eng, cfg, w, ids, nw, nz, sample=[0])
    eng.close()
    assert "conv_splitk_sum_kernel<4,16>" in names, sorted(names)
