"""``PiperVoice`` -- host-side mirror of the reference's Python runtime class
(reference src/python_run/piper/voice.py:20-185): same constructor surface (``load``), same
``phonemes_to_ids`` / ``synthesize_ids_to_raw`` / ``synthesize_stream_raw`` / ``synthesize``
signatures, argument meaning and defaults. The one difference is what sits behind it: the
``onnxruntime.InferenceSession`` is replaced by the MI355X HIP engine (libpiper_hip.so), which reads
the same ``.onnx`` file; the float->int16 conversion (util.py:5-12) runs on the GPU as well.

Phonemisation stays on the host and is out of scope (BASELINE.json north_star): ``phonemize`` handles
``phoneme_type: text`` voices natively (unicode code points after NFD normalisation, as
piper-phonemize's ``phonemize_codepoints`` does) and defers to the optional ``piper_phonemize``
package for espeak voices."""
from __future__ import annotations

import json
import logging
import unicodedata
import wave
from dataclasses import dataclass
from pathlib import Path
from typing import Iterable, List, Optional, Union

from .config import PhonemeType, PiperConfig
from .engine import Engine

PAD = "_"  # padding (0)        -- reference const.py
BOS = "^"  # beginning of sentence
EOS = "$"  # end of sentence


def phonemes_to_ids_cpp(phonemes, id_map, intersperse_pad: bool = True) -> List[int]:
    """The C++ front end's rule (piper-phonemize ``phonemes_to_ids`` as called at reference
    src/cpp/piper.cpp:555 with the defaults of piper.hpp:44-47): BOS, PAD, then id(s)+PAD per phoneme,
    EOS -- i.e. one more PAD after BOS than the Python runtime emits. This is the sequence the shipped
    voices were trained on; the reference's etc/test_sentences fixtures pin it."""
    ids: List[int] = list(id_map[BOS])
    if intersperse_pad:
        ids.extend(id_map[PAD])
    for phoneme in phonemes:
        if phoneme not in id_map:
            continue
        ids.extend(id_map[phoneme])
        if intersperse_pad:
            ids.extend(id_map[PAD])
    ids.extend(id_map[EOS])
    return ids


_LOGGER = logging.getLogger(__name__)


@dataclass
class PiperVoice:
    session: Engine
    config: PiperConfig

    @staticmethod
    def load(model_path: Union[str, Path], config_path: Optional[Union[str, Path]] = None,
             use_cuda: bool = True, device: int = 0) -> "PiperVoice":
        """Load an ONNX voice and its config. ``use_cuda`` is accepted for signature compatibility;
        the engine always runs on the GPU (``device``)."""
        if config_path is None:
            config_path = f"{model_path}.json"
        with open(config_path, "r", encoding="utf-8") as config_file:
            config_dict = json.load(config_file)
        return PiperVoice(config=PiperConfig.from_dict(config_dict),
                          session=Engine(onnx_path=str(model_path), device=device))

    def phonemize(self, text: str) -> List[List[str]]:
        """Text to phonemes grouped by sentence."""
        if self.config.phoneme_type == PhonemeType.TEXT:
            # piper_phonemize.phonemize_codepoints with its default casing: full case folding, then NFD; one sentence
            return [list(unicodedata.normalize("NFD", text.casefold()))]
        if self.config.phoneme_type == PhonemeType.ESPEAK:
            try:
                from piper_phonemize import phonemize_espeak, tashkeel_run  # type: ignore
            except ImportError as e:
                raise RuntimeError(
                    "espeak phonemisation needs the piper_phonemize package on the host "
                    "(out of scope for the GPU engine); pass phoneme ids to synthesize_ids_to_raw") from e
            if self.config.espeak_voice == "ar":
                text = tashkeel_run(text)
            return phonemize_espeak(text, self.config.espeak_voice)
        raise ValueError(f"Unexpected phoneme type: {self.config.phoneme_type}")

    def phonemes_to_ids(self, phonemes: List[str]) -> List[int]:
        """Phonemes to ids (voice.py:72-87: BOS, then id(s)+PAD per phoneme, then EOS)."""
        id_map = self.config.phoneme_id_map
        ids: List[int] = list(id_map[BOS])
        for phoneme in phonemes:
            if phoneme not in id_map:
                _LOGGER.warning("Missing phoneme from id map: %s", phoneme)
                continue
            ids.extend(id_map[phoneme])
            ids.extend(id_map[PAD])
        ids.extend(id_map[EOS])
        return ids

    def synthesize(self, text: str, wav_file: wave.Wave_write, speaker_id: Optional[int] = None,
                   length_scale: Optional[float] = None, noise_scale: Optional[float] = None,
                   noise_w: Optional[float] = None, sentence_silence: float = 0.0):
        """Synthesize WAV audio from text."""
        wav_file.setframerate(self.config.sample_rate)
        wav_file.setsampwidth(2)
        wav_file.setnchannels(1)
        for audio_bytes in self.synthesize_stream_raw(text, speaker_id=speaker_id, length_scale=length_scale,
                                                      noise_scale=noise_scale, noise_w=noise_w,
                                                      sentence_silence=sentence_silence):
            wav_file.writeframes(audio_bytes)

    def synthesize_stream_raw(self, text: str, speaker_id: Optional[int] = None,
                              length_scale: Optional[float] = None, noise_scale: Optional[float] = None,
                              noise_w: Optional[float] = None, sentence_silence: float = 0.0) -> Iterable[bytes]:
        """Synthesize raw audio per sentence from text."""
        sentence_phonemes = self.phonemize(text)
        num_silence_samples = int(sentence_silence * self.config.sample_rate)
        silence_bytes = bytes(num_silence_samples * 2)
        for phonemes in sentence_phonemes:
            phoneme_ids = self.phonemes_to_ids(phonemes)
            yield self.synthesize_ids_to_raw(phoneme_ids, speaker_id=speaker_id, length_scale=length_scale,
                                             noise_scale=noise_scale, noise_w=noise_w) + silence_bytes

    def _scales(self, length_scale, noise_scale, noise_w):
        if length_scale is None:
            length_scale = self.config.length_scale
        if noise_scale is None:
            noise_scale = self.config.noise_scale
        if noise_w is None:
            noise_w = self.config.noise_w
        return (noise_scale, length_scale, noise_w)

    def _speaker(self, speaker_id):
        if self.config.num_speakers <= 1:
            return None
        return 0 if speaker_id is None else speaker_id

    def synthesize_ids_to_raw(self, phoneme_ids: List[int], speaker_id: Optional[int] = None,
                              length_scale: Optional[float] = None, noise_scale: Optional[float] = None,
                              noise_w: Optional[float] = None) -> bytes:
        """Synthesize raw 16-bit mono audio from phoneme ids (voice.py:140-185)."""
        r = self.session.synthesize(phoneme_ids, self._scales(length_scale, noise_scale, noise_w),
                                    sid=self._speaker(speaker_id))
        return r.pcm[0].tobytes()

    def synthesize_ids_batch_to_raw(self, phoneme_id_lists: List[List[int]], speaker_ids=None,
                                    length_scale: Optional[float] = None, noise_scale: Optional[float] = None,
                                    noise_w: Optional[float] = None) -> List[bytes]:
        """Batched extension: several utterances in one GPU call, each identical to its own
        synthesize_ids_to_raw() (same noise stream aside)."""
        sids = None
        if self.config.num_speakers > 1:
            sids = [0 if s is None else s for s in (speaker_ids or [None] * len(phoneme_id_lists))]
        r = self.session.synthesize_batch(phoneme_id_lists, self._scales(length_scale, noise_scale, noise_w), sids=sids)
        return [p.tobytes() for p in r.pcm]
