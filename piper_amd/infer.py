"""JSONL -> WAV driver for a voice .onnx, the counterpart of the reference's
``python -m piper_train.infer_onnx`` (reference src/python/piper_train/infer_onnx.py:19-102): one JSON object
per stdin line with ``phoneme_ids`` (and optionally ``speaker_id``), one ``<line index>.wav`` per utterance in
``--output-dir``, same scale options and defaults. The ONNX Runtime session is replaced by the HIP engine;
``--batch N`` (new) groups N consecutive lines into one batched GPU call.

    python -m piper_amd.infer --model voice.onnx --output-dir out/ < utterances.jsonl
"""
from __future__ import annotations

import argparse
import json
import logging
import sys
import time
import wave
from pathlib import Path
from typing import Iterable, List, Optional, Tuple

from .engine import Engine

_LOGGER = logging.getLogger("piper_amd.infer")


def read_utterances(lines: Iterable[str]) -> List[Tuple[int, List[int], Optional[int]]]:
    """(line index, phoneme ids, speaker id) for every non-empty line; the index names the WAV file, as in
    the reference (blank lines keep their number)."""
    utts = []
    for i, line in enumerate(lines):
        line = line.strip()
        if not line:
            continue
        obj = json.loads(line)
        utts.append((i, [int(p) for p in obj["phoneme_ids"]], obj.get("speaker_id")))
    return utts


def write_wav(path: Path, sample_rate: int, pcm) -> None:
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(sample_rate)
        w.writeframes(pcm.tobytes())


def main(argv=None, *, stdin=None, lib=None) -> int:
    parser = argparse.ArgumentParser(prog="piper_amd.infer")
    parser.add_argument("--model", required=True, help="Path to model (.onnx)")
    parser.add_argument("--output-dir", required=True, help="Path to write WAV files")
    parser.add_argument("--sample-rate", type=int, default=22050)
    parser.add_argument("--noise-scale", type=float, default=0.667)
    parser.add_argument("--noise-scale-w", type=float, default=0.8)
    parser.add_argument("--length-scale", type=float, default=1.0)
    parser.add_argument("--batch", type=int, default=1, help="utterances per GPU call")
    parser.add_argument("--device", type=int, default=0)
    parser.add_argument("--seed", type=int, default=None, help="seed of the engine's noise generator")
    args = parser.parse_args(argv)
    logging.basicConfig(level=logging.DEBUG)

    out_dir = Path(args.output_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    engine = Engine(onnx_path=str(args.model), device=args.device, lib=lib)
    _LOGGER.info("Loaded model from %s", args.model)
    if args.seed is not None:
        engine.set_seed(args.seed)
    scales = (args.noise_scale, args.length_scale, args.noise_scale_w)
    utts = read_utterances(stdin if stdin is not None else sys.stdin)
    step = max(1, args.batch)
    for k in range(0, len(utts), step):
        group = utts[k:k + step]
        sids = [u[2] for u in group]
        t0 = time.perf_counter()
        res = engine.synthesize_batch([u[1] for u in group], scales,
                                      sids=None if all(s is None for s in sids) else [s or 0 for s in sids])
        infer_sec = time.perf_counter() - t0
        audio_sec = sum(p.shape[-1] for p in res.pcm) / args.sample_rate
        _LOGGER.debug("Real-time factor for %s..%s: %0.4f (infer=%0.4f sec, audio=%0.2f sec)", group[0][0] + 1,
                      group[-1][0] + 1, infer_sec / audio_sec if audio_sec > 0 else 0.0, infer_sec, audio_sec)
        for (idx, _, _), pcm in zip(group, res.pcm):
            write_wav(out_dir / f"{idx}.wav", args.sample_rate, pcm)
    engine.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
