"""Voice configuration: the `<voice>.onnx.json` file, parsed exactly like the reference's
``PiperConfig.from_dict`` (reference src/python_run/piper/config.py:39-53; C++ twin
src/cpp/piper.cpp:47-214). Keys, defaults and the required/optional split are the reference's."""
from __future__ import annotations

from dataclasses import dataclass
from enum import Enum
from typing import Any, Dict, Mapping, Optional, Sequence


class PhonemeType(str, Enum):
    ESPEAK = "espeak"
    TEXT = "text"


@dataclass
class PiperConfig:
    num_symbols: int
    num_speakers: int
    sample_rate: int
    espeak_voice: str
    length_scale: float
    noise_scale: float
    noise_w: float
    phoneme_id_map: Mapping[str, Sequence[int]]
    phoneme_type: PhonemeType
    # read by the C++ reference only (piper.cpp:170-189, 199-212); kept so nothing in the file is lost
    phoneme_silence: Optional[Mapping[str, float]] = None
    speaker_id_map: Optional[Mapping[str, int]] = None

    @staticmethod
    def from_dict(config: Dict[str, Any]) -> "PiperConfig":
        inference = config.get("inference", {})
        return PiperConfig(
            num_symbols=config["num_symbols"],
            num_speakers=config["num_speakers"],
            sample_rate=config["audio"]["sample_rate"],
            noise_scale=inference.get("noise_scale", 0.667),
            length_scale=inference.get("length_scale", 1.0),
            noise_w=inference.get("noise_w", 0.8),
            espeak_voice=config["espeak"]["voice"],
            phoneme_id_map=config["phoneme_id_map"],
            phoneme_type=PhonemeType(config.get("phoneme_type", PhonemeType.ESPEAK)),
            phoneme_silence=inference.get("phoneme_silence"),
            speaker_id_map=config.get("speaker_id_map") or None,
        )
