// Mirror of the reference's only test (src/cpp/test.cpp:15-60): load a voice, synthesise
// "This is a test." to a WAV stream through the piper:: API, require a non-trivial file. Extended with
// the direct phoneme-id path and the config values, printed for the pytest wrapper to check.
//   usage: test_piper <voice.onnx> <out.wav>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>

#include "piper.hpp"

int main(int argc, char** argv) {
  if (argc < 3) {
    std::cerr << "usage: test_piper voice.onnx out.wav\n";
    return 2;
  }
  try {
    piper::PiperConfig config;
    piper::Voice voice;
    std::optional<piper::SpeakerId> speaker;
    piper::loadVoice(config, argv[1], std::string(argv[1]) + ".json", voice, speaker, true);
    piper::initialize(config);

    std::ofstream wav(argv[2], std::ios::binary);
    piper::SynthesisResult result;
    piper::textToWavFile(config, voice, "This is a test.", wav, result);
    const long size = (long)wav.tellp();
    wav.close();
    if (size < 10000) {
      std::cerr << "ERROR: Output file is smaller than expected!\n";
      return 1;
    }

    // direct ids: append semantics + timing contract
    std::vector<piper::PhonemeId> ids = {1, 0, 10, 0, 11, 0, 12, 0, 2};
    std::vector<int16_t> audio(7, 123);
    piper::SynthesisResult r2;
    voice.synthesisConfig.noiseScale = 0.0f;
    voice.synthesisConfig.noiseW = 0.0f;
    piper::synthesize(ids, voice.synthesisConfig, voice.session, audio, r2);
    if (audio.size() <= 7 || audio[0] != 123 || r2.audioSeconds <= 0 || r2.inferSeconds <= 0) {
      std::cerr << "ERROR: synthesize() contract violated\n";
      return 1;
    }
    std::vector<piper::PhonemeId> pid;
    std::map<piper::Phoneme, std::size_t> missing;
    piper::phonemes_to_ids({U'a', U'☃', U'b'}, voice.phonemizeConfig, pid, missing);
    std::printf("OK wav_bytes=%ld rate=%d speakers=%d ids_samples=%zu rtf=%.5f pid=%zu missing=%zu sum=%ld\n", size,
                voice.synthesisConfig.sampleRate, voice.modelConfig.numSpeakers, audio.size() - 7, r2.realTimeFactor,
                pid.size(), missing.size(), [&] { long s = 0; for (size_t i = 7; i < audio.size(); ++i) s += audio[i]; return s; }());
    // phrases of a sentence (phoneme_silence) go through the engine as one batch: with the noise switched off the
    // result must be the per-phrase synthesize() outputs joined by the configured silences (piper.cpp:548-575)
    {
      voice.synthesisConfig.phonemeSilenceSeconds.emplace();
      (*voice.synthesisConfig.phonemeSilenceSeconds)[U','] = 0.01f;
      std::vector<int16_t> joined;
      piper::SynthesisResult r3;
      piper::textToAudio(config, voice, "abc, de, f", joined, r3, nullptr);
      std::vector<int16_t> expect;
      const std::size_t sil = (std::size_t)(0.01f * voice.synthesisConfig.sampleRate * voice.synthesisConfig.channels);
      for (const std::u32string phrase : {U"abc,", U" de,", U" f"}) {
        std::vector<piper::PhonemeId> ids3;
        std::map<piper::Phoneme, std::size_t> miss3;
        piper::phonemes_to_ids(std::vector<piper::Phoneme>(phrase.begin(), phrase.end()), voice.phonemizeConfig, ids3, miss3);
        piper::SynthesisResult r4;
        piper::synthesize(ids3, voice.synthesisConfig, voice.session, expect, r4);
        if (phrase.back() == U',') expect.insert(expect.end(), sil, (int16_t)0);
      }
      if (voice.synthesisConfig.sentenceSilenceSeconds > 0)
        expect.insert(expect.end(), (std::size_t)(voice.synthesisConfig.sentenceSilenceSeconds * voice.synthesisConfig.sampleRate *
                                                  voice.synthesisConfig.channels), (int16_t)0);
      long maxd = joined.size() == expect.size() ? 0 : 1 << 20;
      for (std::size_t i = 0; i < joined.size() && i < expect.size(); ++i) {
        const long d = std::labs((long)joined[i] - (long)expect[i]);
        if (d > maxd) maxd = d;
      }
      if (maxd > 2 || r3.audioSeconds <= 0) {
        std::cerr << "ERROR: batched phrases differ from per-phrase synthesis (max |d| = " << maxd << ", sizes "
                  << joined.size() << " vs " << expect.size() << ")\n";
        return 1;
      }
      voice.synthesisConfig.phonemeSilenceSeconds.reset();
    }
    piper::terminate(config);
    // errors surface as std::runtime_error, like the reference
    try {
      std::vector<piper::PhonemeId> bad = {1, 99999, 2};
      piper::synthesize(bad, voice.synthesisConfig, voice.session, audio, r2);
      std::cerr << "ERROR: out-of-range id accepted\n";
      return 1;
    } catch (const std::runtime_error&) {
    }
    return 0;
  } catch (const std::exception& e) {
    std::cerr << "EXCEPTION: " << e.what() << "\n";
    return 1;
  }
}
