"""TEST INFRASTRUCTURE (bench.py's cpu_baseline leg only): one worker of the all-cores CPU baseline.

Runs the oracle (the torch-CPU restatement of the reference graph, oracle/vits_oracle.py) on the same synthetic voice and
utterance as the parent, with a fixed thread count, for a fixed number of seconds, and prints one JSON line
{"samples": .., "t0": .., "t1": .., "n": ..} with wall-clock stamps, so that the parent can add the workers' samples
over the window they share. The reference's own loop is sequential (piper.cpp:549-582, one session.Run per phrase);
P of these side by side are P such programs sharing the host, which is how a CPU server would use 256 cores.

usage: cpu_worker.py <preset> <ids> <threads> <seconds> <start_at_unix_time>
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    preset, T, threads, seconds, start_at = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), float(sys.argv[5])
    os.environ["OMP_NUM_THREADS"] = str(threads)
    import numpy as np
    import torch
    torch.set_num_threads(threads)
    from oracle import vits_oracle as O
    from piper_amd import weights as W
    cfg = W.preset(preset)
    wt = O.to_torch(W.synthetic_weights(cfg, 1234))
    ids = W.synthetic_phoneme_ids(T, 0, id_max=min(cfg.n_vocab - 1, 129))
    rng = np.random.default_rng(1234)
    nw = rng.standard_normal((2, T)).astype(np.float32)
    nz = np.random.default_rng(99).standard_normal((cfg.inter, 16 * T + 64)).astype(np.float32)
    scales = (0.667, 1.0, 0.8)
    O.synthesize(wt, cfg, ids, scales, nw, nz)                     # warm-up
    while time.time() < start_at:                                  # all workers start their window together
        time.sleep(0.01)
    t0 = time.time()
    n = samples = 0
    while time.time() - t0 < seconds:
        samples += O.synthesize(wt, cfg, ids, scales, nw, nz)["audio"].size
        n += 1
    print(json.dumps({"samples": samples, "t0": t0, "t1": time.time(), "n": n}), flush=True)


if __name__ == "__main__":
    main()
