"""Real-time-factor benchmark over a JSONL of utterances, the counterpart of the reference's
``src/benchmark/benchmark_onnx.py`` (reference lines 19-122): loads ``-m voice.onnx`` (+ ``-c`` config, default
``<model>.json``, for the sample rate), synthesises every stdin line (``phoneme_ids``, optional ``speaker_id``)
with the reference's fixed scales 0.667 / 1.0 / 0.8 and prints one JSON object
``{"load_sec", "rtf_mean", "rtf_stdev", "rtfs"}`` where rtf = inference seconds / audio seconds per utterance.

    python -m piper_amd.benchmark -m voice.onnx < utterances.jsonl
"""
from __future__ import annotations

import argparse
import json
import statistics
import sys
import time

from .engine import Engine
from .infer import read_utterances

_NOISE_SCALE, _LENGTH_SCALE, _NOISE_W = 0.667, 1.0, 0.8


def main(argv=None, *, stdin=None, stdout=None, lib=None) -> int:
    parser = argparse.ArgumentParser(prog="piper_amd.benchmark")
    parser.add_argument("-m", "--model", required=True, help="Path to Onnx model file (.onnx)")
    parser.add_argument("-c", "--config", help="Path to model config file (.json)")
    parser.add_argument("--device", type=int, default=0)
    args = parser.parse_args(argv)
    config_path = args.config or f"{args.model}.json"
    with open(config_path, "r", encoding="utf-8") as f:
        sample_rate = json.load(f)["audio"]["sample_rate"]
    utts = read_utterances(stdin if stdin is not None else sys.stdin)

    t0 = time.monotonic_ns()
    engine = Engine(onnx_path=str(args.model), device=args.device, lib=lib)
    load_sec = (time.monotonic_ns() - t0) / 1e9

    rtfs = []
    scales = (_NOISE_SCALE, _LENGTH_SCALE, _NOISE_W)
    for _, ids, sid in utts:
        t1 = time.monotonic_ns()
        res = engine.synthesize(ids, scales, sid=sid)
        infer_sec = (time.monotonic_ns() - t1) / 1e9
        audio_sec = res.pcm[0].shape[-1] / sample_rate
        rtfs.append(infer_sec / audio_sec if audio_sec > 0 else 0.0)
    engine.close()
    json.dump({"load_sec": load_sec, "rtf_mean": statistics.mean(rtfs) if rtfs else 0.0,
               "rtf_stdev": statistics.stdev(rtfs) if len(rtfs) > 1 else 0.0, "rtfs": rtfs},
              stdout if stdout is not None else sys.stdout)
    return 0


if __name__ == "__main__":
    sys.exit(main())
