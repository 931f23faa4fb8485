"""Helper of tests/test_emu_engine.py::test_small_call_kernels_do_not_depend_on_wave_order: one ragged batch of a
192-channel voice through the emulator build under the fiber order EMU_ORDER names (hip_emu.cpp), compared with the
oracle; prints one JSON line. Run in a process of its own because the emulator reads EMU_ORDER once."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import vits_oracle as O                      # noqa: E402
from piper_amd import _lib as L, weights as W            # noqa: E402
from piper_amd.engine import Engine                      # noqa: E402


def main():
    elib = L.bind(os.path.join(ROOT, "tests", "emu", "libpiper_hip_emu.so"))
    case = sys.argv[1] if len(sys.argv) > 1 else "small192"
    if case == "small192":        # the 4-column small-call kernels of the 192-channel voices
        cfg = W.preset("tiny-ms", hidden=192, inter=192, filter=96, n_layers=2)
        lens, sids = [9, 21], [1, 3]
    elif case.endswith("+wide"):  # a first generator stage of 128 channels: the grouped / K-concatenated sibling launches of a
        cfg = W.preset(case[:-5], up_initial=256)      # one-utterance call (conv_splitk_group_kernel, conv_splitk_sum_kernel)
        lens, sids = [13], None
    else:                         # a tiny preset as it is: the general kernels (16-column DDSConv layers with the fused
        cfg = W.preset(case)      # pre / proj / spline, attention, split-K convs, the fused stage kernels)
        lens, sids = [17, 6], ([2, 0] if cfg.n_speakers > 1 else None)
    w = W.synthetic_weights(cfg, 77)
    ids = [W.synthetic_phoneme_ids(T, i, id_max=cfg.n_vocab - 1) for i, T in enumerate(lens)]
    nw = np.random.default_rng(5).standard_normal((len(lens), 2, max(lens))).astype(np.float32)
    eng = Engine(blob=W.pack_blob(cfg, w), lib=elib)
    eng.profile_enable(2)
    r = eng.synthesize_batch(ids, (0.0, 1.0, 0.8), noise_w=nw, sids=sids)
    names = sorted(row["name"] for row in eng.profile()[5:] if row["launches"])
    durs = eng.durations()
    off = np.concatenate([[0], np.cumsum(lens)])
    worst, same = 0.0, True
    for i in range(len(lens)):
        o = O.synthesize(w, cfg, ids[i], (0.0, 1.0, 0.8), nw[i][:, :lens[i]], sid=None if sids is None else sids[i])
        same = same and bool(np.array_equal(durs[off[i]:off[i + 1]], o["durations"])) and r.audio[i].shape == o["audio"].shape
        if r.audio[i].shape == o["audio"].shape:
            worst = max(worst, float(np.max(np.abs(r.audio[i] - o["audio"]))))
    print(json.dumps({"order": os.environ.get("EMU_ORDER", ""), "kernels": names, "durations_equal": same, "worst": worst,
                      "checksum": float(sum(float(np.sum(a.astype(np.float64))) for a in r.audio))}))


if __name__ == "__main__":
    main()
