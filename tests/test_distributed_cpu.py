"""The N>1 path on CPU: world_size-2 gloo group, weight blob broadcast from rank 0, utterances
sharded across ranks, results gathered in order. The per-rank engines here are the test-only
emulator build (no GPU in this container); on the GPU box the same code runs with the nccl (RCCL)
backend and the real library (bench.py --gpus N)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from piper_amd.dist import shard_indices

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu", "libpiper_hip_emu.so")

WORKER = r'''
import os, sys, json
sys.path.insert(0, %(root)r)
import numpy as np, torch.distributed as dist
from piper_amd import weights as W, _lib as L
from piper_amd.dist import ShardedSynthesizer
dist.init_process_group("gloo")
rank = dist.get_rank()
lib = L.bind(%(emu)r)
cfg = W.preset("tiny")
blob = W.pack_blob(cfg, W.synthetic_weights(cfg, 1234)) if rank == 0 else None   # only rank 0 has the voice
syn = ShardedSynthesizer(blob=blob, lib=lib)
Ts = [9, 4, 14, 6, 3]
ids = [W.synthetic_phoneme_ids(T, i, id_max=cfg.n_vocab - 1) for i, T in enumerate(Ts)]
out = syn.synthesize(ids, (0.0, 1.0, 0.0))
if rank == 0:
    np.savez(%(out)r, *out)
dist.barrier()
dist.destroy_process_group()
'''


def test_shard_indices_partition_and_balance():
    costs = [128, 17, 381, 64, 64, 200, 5, 90]
    for world in (1, 2, 3, 8):
        sh = shard_indices(costs, world)
        assert sorted(i for s in sh for i in s) == list(range(len(costs)))
        loads = [sum(costs[i] for i in s) for s in sh]
        assert max(loads) - min(loads) <= max(costs)
    assert shard_indices(costs, 2) == shard_indices(costs, 2)      # deterministic on every rank


def test_shard_indices_respect_the_engine_batch_limit():
    """many short utterances beside a few long ones must not put more than 4096 on one rank (an engine call
    takes at most 4096); the deal spills to the next-least-loaded rank, and a batch that cannot fit raises."""
    from piper_amd import dist
    costs = [4000] * 3 + [1] * 8000            # by load alone rank 1 would get ~all of the short ones
    table = dist.shard_indices(costs, 2)
    assert sorted(i for t in table for i in t) == list(range(len(costs)))
    assert max(len(t) for t in table) <= dist.MAX_PER_RANK
    with pytest.raises(ValueError):
        dist.shard_indices([1] * (2 * dist.MAX_PER_RANK + 1), 2)


@pytest.mark.parametrize("matrix", ["f32", "f16x3"])
def test_two_rank_gloo_matches_single_process(tmp_path, matrix):
    """(matrix = f16x3: the skeleton rank lays out the split-term fragments, the stage kernels' split weight streams and the
    per-conv un-scale factors from tensor SHAPES alone and receives them with the one broadcast.)"""
    if not os.path.exists(EMU):
        subprocess.check_call(["make", "-C", ROOT, "emu"])
    out = str(tmp_path / "dist.npz")
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "emu": EMU, "out": out})
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1", PIPER_HIP_MATRIX=matrix)
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                    "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                   check=True, env=env, timeout=600, cwd=str(tmp_path))
    got = np.load(out)
    from oracle import vits_oracle as O
    from piper_amd import weights as W
    cfg = W.preset("tiny")
    w = W.synthetic_weights(cfg, 1234)
    for i, T in enumerate([9, 4, 14, 6, 3]):
        ref = O.synthesize(w, cfg, W.synthetic_phoneme_ids(T, i, id_max=cfg.n_vocab - 1), (0.0, 1.0, 0.0))
        pcm = got[f"arr_{i}"]
        assert pcm.shape == ref["pcm"].shape
        assert np.max(np.abs(pcm.astype(np.int32) - ref["pcm"].astype(np.int32))) <= 4
