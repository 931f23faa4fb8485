// Mirror of the reference's only test (src/cpp/test.cpp:15-60): load a voice, synthesise
// "This is a test." to a WAV stream through the piper:: API, require a non-trivial file. Extended with
// the direct phoneme-id path and the config values, printed for the pytest wrapper to check.
//   usage: test_piper <voice.onnx> <out.wav> [dump-prefix]
// With a dump prefix the int16 PCM of piper::synthesize (ids 1 0 10 0 11 0 12 0 2) and of piper::textToAudio
// ("hello there"), both with the noise switched off, are written to <prefix>.synth.pcm / <prefix>.text.pcm for the pytest
// wrapper, which compares them with the CPU oracle.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>

#include "piper.hpp"

// test_piper --config <voice.onnx> <voice.onnx.json>: piper::loadVoice with an explicit config file; every field the
// reference's parsers fill (src/cpp/piper.cpp:47-214) is printed, one `key=value` per line, for the pytest wrapper.
static int dump_config(const char* onnx, const char* json) {
  try {
    piper::PiperConfig config;
    piper::Voice voice;
    std::optional<piper::SpeakerId> speaker;
    piper::loadVoice(config, onnx, json, voice, speaker, false);
    const piper::PhonemizeConfig& pc = voice.phonemizeConfig;
    std::printf("phoneme_type=%s\n", pc.phonemeType == piper::TextPhonemes ? "text" : "espeak");
    std::printf("espeak_voice=%s\n", pc.eSpeak.voice.c_str());
    std::printf("phoneme_map=%s\n", pc.phonemeMap ? std::to_string(pc.phonemeMap->size()).c_str() : "none");
    std::printf("phoneme_id_map=%zu\n", pc.phonemeIdMap.size());
    for (auto& kv : pc.phonemeIdMap) {
      std::printf("id U+%04X", (unsigned)kv.first);
      for (auto v : kv.second) std::printf(" %lld", (long long)v);
      std::printf("\n");
    }
    std::printf("id_pad=%lld id_bos=%lld id_eos=%lld interspersePad=%d\n", (long long)pc.idPad, (long long)pc.idBos,
                (long long)pc.idEos, (int)pc.interspersePad);
    const piper::SynthesisConfig& sc = voice.synthesisConfig;
    std::printf("sample_rate=%d sample_width=%d channels=%d\n", sc.sampleRate, sc.sampleWidth, sc.channels);
    std::printf("noise_scale=%.9g length_scale=%.9g noise_w=%.9g sentence_silence=%.9g\n", sc.noiseScale, sc.lengthScale, sc.noiseW,
                sc.sentenceSilenceSeconds);
    std::printf("phoneme_silence=%s\n", sc.phonemeSilenceSeconds ? std::to_string(sc.phonemeSilenceSeconds->size()).c_str() : "none");
    std::printf("speaker_id=%s\n", sc.speakerId ? std::to_string(*sc.speakerId).c_str() : "none");
    std::printf("num_speakers=%d\n", voice.modelConfig.numSpeakers);
    std::printf("speaker_id_map=%s\n", voice.modelConfig.speakerIdMap ? std::to_string(voice.modelConfig.speakerIdMap->size()).c_str() : "none");
    std::printf("config_text_bytes=%zu\n", voice.configText.size());
    return 0;
  } catch (const std::exception& e) {
    std::cerr << "EXCEPTION: " << e.what() << "\n";
    return 1;
  }
}

int main(int argc, char** argv) {
  if (argc == 4 && std::string(argv[1]) == "--config") return dump_config(argv[2], argv[3]);
  if (argc < 3) {
    std::cerr << "usage: test_piper voice.onnx out.wav\n";
    return 2;
  }
  try {
    piper::PiperConfig config;
    piper::Voice voice;
    std::optional<piper::SpeakerId> speaker;
    // useCuda = false is what the reference's own test.cpp:41-42 and main.cpp's default pass: it must load too
    piper::loadVoice(config, argv[1], std::string(argv[1]) + ".json", voice, speaker, false);
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
    auto dump = [&](const char* what, const int16_t* p, std::size_t n) {
      if (argc < 4) return;
      std::ofstream f(std::string(argv[3]) + "." + what + ".pcm", std::ios::binary);
      f.write(reinterpret_cast<const char*>(p), (std::streamsize)(n * sizeof(int16_t)));
    };
    dump("synth", audio.data() + 7, audio.size() - 7);
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
    // ---- text front end == piper-phonemize's phonemize_codepoints: full case folding, then NFD
    {
      std::vector<std::vector<piper::Phoneme>> a, b, c;
      piper::phonemize_codepoints("This IS a T\xC3\x89ST \xC3\x85 \xC3\x9F \xEA\xB0\x81", a);   // É, Å, ß (folds to ss), Hangul U+AC01
      piper::phonemize_codepoints("this is a te\xCC\x81st a\xCC\x8A ss \xEA\xB0\x81", b);
      const std::u32string want = U"this is a te\u0301st a\u030A ss \u1100\u1161\u11A8";
      if (a.size() != 1 || a[0] != std::vector<piper::Phoneme>(want.begin(), want.end()) || a[0] != b[0]) {
        std::cerr << "ERROR: phonemize_codepoints is not casefold + NFD\n";
        return 1;
      }
      // canonical ordering of combining marks: dot below (ccc 220) sorts before acute (ccc 230)
      piper::phonemize_codepoints("a\xCC\x81\xCC\xA3", c);
      const std::u32string want2 = U"a\u0323\u0301";
      if (c[0] != std::vector<piper::Phoneme>(want2.begin(), want2.end())) {
        std::cerr << "ERROR: NFD canonical ordering\n";
        return 1;
      }
      // upper-case / precomposed input synthesises exactly what its folded + decomposed spelling does
      voice.synthesisConfig.noiseScale = 0.0f;
      voice.synthesisConfig.noiseW = 0.0f;
      std::vector<int16_t> u, l;
      piper::SynthesisResult ru, rl;
      std::vector<std::string> warned;
      config.warn = [&](const std::string& m) { warned.push_back(m); };
      piper::textToAudio(config, voice, "HELLO THERE", u, ru, nullptr);
      piper::textToAudio(config, voice, "hello there", l, rl, nullptr);
      if (u != l || u.empty() || !warned.empty()) {
        std::cerr << "ERROR: upper-case text differs from its case-folded form (" << u.size() << " vs " << l.size() << ")\n";
        return 1;
      }
      dump("text", l.data(), l.size());
      // a phoneme without an id is dropped AND reported (piper.cpp:600-610)
      std::vector<int16_t> m;
      piper::textToAudio(config, voice, "ab\xE2\x98\x83\xE2\x98\x83", m, rl, nullptr);
      if (warned.size() != 2 || warned[0].find("Missing 1 phoneme") == std::string::npos ||
          warned[1].find("2 time(s)") == std::string::npos || warned[1].find("2603") == std::string::npos) {
        std::cerr << "ERROR: missing phonemes not reported (" << warned.size() << " message(s))\n";
        return 1;
      }
      config.warn = nullptr;
    }
    // ---- eSpeak voices: the host's phonemizer slot (espeak-ng stays on the host)
    {
      const std::string cfgPath = std::string(argv[2]) + ".espeak.json";
      {
        std::ifstream in(std::string(argv[1]) + ".json");
        std::stringstream ss;
        ss << in.rdbuf();
        std::string js = ss.str();
        const std::string key = "\"phoneme_type\": \"text\"";
        const size_t at = js.find(key);
        if (at == std::string::npos) { std::cerr << "ERROR: fixture config has no phoneme_type text\n"; return 1; }
        js.replace(at, key.size(), "\"phoneme_type\": \"espeak\"");
        std::ofstream(cfgPath) << js;
      }
      piper::PiperConfig ec;
      piper::Voice ev;
      std::optional<piper::SpeakerId> none;
      piper::loadVoice(ec, argv[1], cfgPath, ev, none, true);
      ev.synthesisConfig.noiseScale = 0.0f;
      ev.synthesisConfig.noiseW = 0.0f;
      std::vector<int16_t> got, want;
      piper::SynthesisResult r5;
      bool threw = false;
      try { piper::textToAudio(ec, ev, "two. sentences", got, r5, nullptr); } catch (const std::runtime_error&) { threw = true; }
      if (!threw) { std::cerr << "ERROR: eSpeak voice without a phonemizer must throw\n"; return 1; }
      std::string seenVoice;
      ec.phonemizer = [&](const std::string& text, const std::string& v, std::vector<std::vector<piper::Phoneme>>& out) {
        seenVoice = v;                       // a stand-in for phonemize_eSpeak: one "sentence" per '.'-separated part
        out.emplace_back();
        for (char ch : text) {
          if (ch == '.') { out.emplace_back(); continue; }
          out.back().push_back((piper::Phoneme)(unsigned char)ch);
        }
      };
      int callbacks = 0;
      std::vector<int16_t> all;
      piper::textToAudio(ec, ev, "two. sentences", got, r5, [&] { ++callbacks; all.insert(all.end(), got.begin(), got.end()); });
      if (callbacks != 2 || !got.empty() || all.empty() || seenVoice.empty() || r5.inferSeconds <= 0 || r5.audioSeconds <= 0) {
        std::cerr << "ERROR: phonemizer slot / per-sentence callback contract (" << callbacks << " callbacks)\n";
        return 1;
      }
      // ---- several sentences: without a callback the whole text is ONE engine call, with a callback the sentences go
      // in groups of 1, 2, 4, ... (the next group runs while the callbacks consume the previous one). Both must give
      // what the reference's sequential loop gives: per sentence synthesize() + sentence silence (piper.cpp:548-598).
      {
        // 8 sentences, one of them ~empty (PIPER_TEST_SHORT: 5, for the emulator run of the CPU suite -- groups of 1, 2, 2)
        const std::string text = std::getenv("PIPER_TEST_SHORT") ? "ab cd. ef gh ab. a. bcd efg. hi"
                                                                 : "ab cd. ef gh ab. a. bcd efg. hi. abc abc abc. de. fgh";
        std::vector<int16_t> whole, parts, expect, buf;
        piper::SynthesisResult ra, rb;
        piper::textToAudio(ec, ev, text, whole, ra, nullptr);
        int ncb = 0;
        piper::textToAudio(ec, ev, text, buf, rb, [&] { ++ncb; parts.insert(parts.end(), buf.begin(), buf.end()); });
        std::vector<std::vector<piper::Phoneme>> sents;
        ec.phonemizer(text, "", sents);
        const std::size_t ssil = (std::size_t)(ev.synthesisConfig.sentenceSilenceSeconds * ev.synthesisConfig.sampleRate *
                                               ev.synthesisConfig.channels);
        for (auto& sp : sents) {
          std::vector<piper::PhonemeId> sid3;
          std::map<piper::Phoneme, std::size_t> miss3;
          piper::phonemes_to_ids(sp, ev.phonemizeConfig, sid3, miss3);
          piper::SynthesisResult r6;
          if (!sp.empty()) piper::synthesize(sid3, ev.synthesisConfig, ev.session, expect, r6);
          expect.insert(expect.end(), ssil, (int16_t)0);
        }
        auto maxdiff = [](const std::vector<int16_t>& a, const std::vector<int16_t>& b) {
          long m = a.size() == b.size() ? 0 : 1 << 20;
          for (std::size_t i = 0; i < a.size() && i < b.size(); ++i) m = std::max(m, std::labs((long)a[i] - (long)b[i]));
          return m;
        };
        if (ncb != (int)sents.size() || !buf.empty() || maxdiff(whole, expect) > 2 || maxdiff(parts, expect) > 2 ||
            ra.audioSeconds <= 0 || rb.inferSeconds <= 0) {
          std::cerr << "ERROR: multi-sentence text: " << ncb << " callbacks for " << sents.size() << " sentences, |whole - seq| = "
                    << maxdiff(whole, expect) << ", |callback parts - seq| = " << maxdiff(parts, expect) << " (sizes "
                    << whole.size() << " / " << parts.size() << " / " << expect.size() << ")\n";
          return 1;
        }
      }
      // ---- the padded-size budget of a batched call (PiperConfig::maxBatchIds): a text with many short sentences and one
      // long one, cut into several engine calls -- also in the middle of the text and with a phrase longer than the
      // budget -- gives exactly what the unbounded grouping gives
      {
        const std::string text = "ab. cd. ef. abcdefgh abcdefgh abcdefgh abcdefgh. gh. ab. cd. a. b";
        std::vector<int16_t> one, cut, tiny, cb_parts, buf2;
        piper::SynthesisResult ra, rb, rc, rd;
        ec.maxBatchIds = 1 << 20;
        piper::textToAudio(ec, ev, text, one, ra, nullptr);
        ec.maxBatchIds = 40;                       // 2-3 short sentences per call; the long one (> 40 padded ids) alone
        piper::textToAudio(ec, ev, text, cut, rb, nullptr);
        int ncb = 0;
        piper::textToAudio(ec, ev, text, buf2, rd, [&] { ++ncb; cb_parts.insert(cb_parts.end(), buf2.begin(), buf2.end()); });
        ec.maxBatchIds = 1;                        // every phrase alone: the reference's own schedule
        piper::textToAudio(ec, ev, text, tiny, rc, nullptr);
        ec.maxBatchIds = 8192;
        auto maxdiff2 = [](const std::vector<int16_t>& a, const std::vector<int16_t>& b) {
          long m = a.size() == b.size() ? 0 : 1 << 20;
          for (std::size_t i = 0; i < a.size() && i < b.size(); ++i) m = std::max(m, std::labs((long)a[i] - (long)b[i]));
          return m;
        };
        if (one.empty() || maxdiff2(one, cut) > 2 || maxdiff2(one, tiny) > 2 || maxdiff2(one, cb_parts) > 2 || ncb != 9) {
          std::cerr << "ERROR: maxBatchIds changes the audio: |unbounded - 40| = " << maxdiff2(one, cut) << ", |unbounded - 1| = "
                    << maxdiff2(one, tiny) << ", |unbounded - callbacks| = " << maxdiff2(one, cb_parts) << " (" << ncb << " callbacks)\n";
          return 1;
        }
      }
      // ---- Arabic diacritization slot (piper.cpp:457-464): useTashkeel without a function throws like the reference does
      // without a tashkeelState; with one the text passes through it before phonemization
      {
        std::vector<int16_t> a1, a2;
        piper::SynthesisResult r7;
        ec.useTashkeel = true;
        bool threw2 = false;
        try { piper::textToAudio(ec, ev, "ab", a1, r7, nullptr); }
        catch (const std::runtime_error& e) { threw2 = std::string(e.what()) == "Tashkeel model is not loaded"; }
        if (!threw2) { std::cerr << "ERROR: useTashkeel without a tashkeel function must throw\n"; return 1; }
        ec.tashkeel = [](const std::string& t) { return t + "cd"; };
        piper::textToAudio(ec, ev, "ab", a1, r7, nullptr);
        ec.useTashkeel = false;
        piper::textToAudio(ec, ev, "abcd", a2, r7, nullptr);
        if (a1.empty() || a1 != a2) { std::cerr << "ERROR: tashkeel slot not applied before phonemization\n"; return 1; }
      }
      // single-speaker voice: no speaker id is fed (reference omits the "sid" input)
      if (ev.synthesisConfig.speakerId) { std::cerr << "ERROR: speakerId set on a single-speaker voice\n"; return 1; }
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
