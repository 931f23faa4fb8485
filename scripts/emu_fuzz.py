"""Randomised shape sweep of the whole pipeline on the kernel emulator (tests/emu) against the oracle: random valid
architectures around the tiny presets (channel widths, layer counts, heads, kernel sizes, resblock type, speakers), ragged
batches, both launch routes (split-K / tiled), a random wave order per case. Not part of the test suite (minutes per
case); run by hand: python scripts/emu_fuzz.py [cases] [seed]. Prints one line per case, exits non-zero on a mismatch."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
from oracle import vits_oracle as O
from piper_amd import _lib as L, weights as W
from piper_amd.engine import Engine
c = json.loads(sys.argv[1])
cfg = W.preset(c["preset"], **c["over"])
w = W.synthetic_weights(cfg, c["wseed"])
lens = c["lens"]
sids = c["sids"]
ids = [W.synthetic_phoneme_ids(T, i, id_max=cfg.n_vocab - 1) for i, T in enumerate(lens)]
nw = np.random.default_rng(c["wseed"]).standard_normal((len(lens), 2, max(lens))).astype(np.float32)
eng = Engine(blob=W.pack_blob(cfg, w), lib=L.bind(os.path.join(%(root)r, "tests", "emu", "libpiper_hip_emu.so")))
scales = tuple(c["scales"])
r = eng.synthesize_batch(ids, scales, noise_w=nw, sids=sids)
durs = eng.durations()
off = np.concatenate([[0], np.cumsum(lens)])
worst, same = 0.0, True
for i in range(len(lens)):
    o = O.synthesize(w, cfg, ids[i], scales, nw[i][:, :lens[i]], sid=None if sids is None else sids[i])
    same = same and bool(np.array_equal(durs[off[i]:off[i + 1]], o["durations"])) and r.audio[i].shape == o["audio"].shape
    if r.audio[i].shape == o["audio"].shape:
        worst = max(worst, float(np.max(np.abs(r.audio[i] - o["audio"]))))
print(json.dumps({"durations_equal": same, "worst": worst}))
'''


def main():
    import numpy as np
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = 0
    for k in range(n):
        hidden = int(rng.choice([32, 64, 96, 192], p=[0.3, 0.2, 0.2, 0.3]))      # 192: the small-call 4-column kernels
        heads = 2 if hidden == 192 else (int(rng.choice([1, 2])) if hidden % 64 == 0 or hidden == 96 else 2)
        over = dict(hidden=hidden, inter=(192 if hidden == 192 else int(rng.choice([32, 64]))), filter=int(rng.choice([48, 96]) if hidden == 192 else rng.choice([32, 64, 96])),
                    n_layers=int(rng.integers(1, 3)), n_heads=heads, window=int(rng.choice([2, 4])),
                    up_initial=int(rng.choice([32, 64, 128, 256])))      # 256: a 128-channel first stage (grouped sibling launches)
        preset = str(rng.choice(["tiny", "tiny-high", "tiny-ms", "tiny-high-ms"]))
        B = int(rng.integers(1, 4))
        lens = [int(rng.integers(1, 40)) for _ in range(B)]
        if hidden == 192 and rng.random() < 0.25: lens[0] = int(rng.integers(129, 200))      # attn4_kernel's double-buffered form
        ms = "ms" in preset
        case = {"preset": preset, "over": over, "lens": lens, "sids": [int(rng.integers(0, 4)) for _ in lens] if ms else None,
                "wseed": int(rng.integers(1, 1 << 30)), "scales": [0.0, float(rng.choice([0.8, 1.0, 1.3])), 0.8]}
        env = dict(os.environ, EMU_ORDER=str(rng.choice(["", "reverse", "shuffle"])))
        for knob, vals in (("PIPER_HIP_SPLITK_MAX", ["", "0"]), ("PIPER_HIP_MRF", ["", "0", "2"]), ("PIPER_HIP_FUSE_DP", ["", "0"]),
                           ("PIPER_HIP_COLCHAIN", ["", "0"]), ("PIPER_HIP_SPLITK16", ["", "3"]), ("PIPER_HIP_COL4", ["", "0", "2"]),
                           ("PIPER_HIP_ATTNO", ["", "0"]), ("PIPER_HIP_FFN", ["", "0"]), ("PIPER_HIP_GATE_HALF", ["", "0"]),
                           ("PIPER_HIP_CONV1X1", ["", "0"]), ("PIPER_HIP_CHAIN_RS", ["", "0"]), ("PIPER_HIP_STACK_PRE", ["", "0"]),
                           ("PIPER_HIP_ATTN4", ["", "0", "2"]), ("PIPER_HIP_GATE4", ["", "0", "2"]), ("PIPER_HIP_GROUP_TILED", ["", "0"]), ("PIPER_HIP_GROUP_MAXB", ["", "1", "2"]),
                           ("PIPER_HIP_IDS_ZC", ["", "0"]), ("PIPER_HIP_MRF_SPLIT", ["", "0"]),
                           ("PIPER_HIP_MATRIX", ["", "", "f16x3", "bf16x6", "bf16x3"]),          # split-operand matrix modes (same gate: 2e-5 here)
                           ("PIPER_HIP_DEBUG_POISON", ["", "1"])):
            v = str(rng.choice(vals))
            if v:
                env[knob] = v
        p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}, json.dumps(case)], capture_output=True, text=True, env=env,
                           timeout=3600)
        knobs = {k2: v for k2, v in env.items() if k2.startswith("PIPER_HIP_") or k2 == "EMU_ORDER"}
        knobs.pop("PIPER_HIP_GROUP_BCAST", None)
        if p.returncode != 0:
            msg = (p.stderr.strip().splitlines() or ["?"])[-1]
            # a configuration the loader rejects by design is not a finding
            ok = any(s in msg for s in ("not supported", "must be", "multiple of", "unsupported"))
            print(("REJECTED " if ok else "ERROR    ") + json.dumps(case) + " " + json.dumps(knobs) + " :: " + msg[:200], flush=True)
            bad += 0 if ok else 1
            continue
        o = json.loads(p.stdout.strip().splitlines()[-1])
        good = o["durations_equal"] and o["worst"] < (2e-4 if env.get("PIPER_HIP_MATRIX") == "bf16x3" else 2e-5)
        print(("ok       " if good else "MISMATCH ") + json.dumps(case) + " " + json.dumps(knobs) + " -> " + json.dumps(o), flush=True)
        bad += 0 if good else 1
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
