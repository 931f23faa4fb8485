// C++ host side of the drop-in boundary: the reference's namespace piper API (include/piper.hpp)
// implemented on the C ABI of the HIP engine. Behaviour follows the reference's src/cpp/piper.cpp:
// config parsing :47-214, loadVoice :309-334, synthesize :337-441 (timing of the inference call only,
// append-only audio buffer), textToAudio :446-616 (phrase splitting on phoneme_silence, sentence
// silence, missing-phoneme accounting, per-sentence callback), textToWavFile :619-634.
#include "../../include/piper.hpp"

#include <chrono>
#include <cmath>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

#include <algorithm>
#include <cstdio>

#include "../../include/piper_hip.h"
#include "unicode_tables.h"

namespace piper {

// ------------------------------------------------------------------------------------------------
// minimal JSON (objects, arrays, strings with \u escapes, numbers, true/false/null)
// ------------------------------------------------------------------------------------------------
namespace {

struct JVal {
  enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
  bool b = false;
  double num = 0;
  std::string str;
  std::vector<JVal> arr;
  std::vector<std::pair<std::string, JVal>> obj;
  const JVal* get(const std::string& k) const {
    for (auto& kv : obj)
      if (kv.first == k) return &kv.second;
    return nullptr;
  }
  bool contains(const std::string& k) const { return kind == Obj && get(k) != nullptr; }
};

static void append_utf8(uint32_t cp, std::string& out) {
  if (cp < 0x80) out.push_back((char)cp);
  else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
  else if (cp < 0x10000) {
    out.push_back((char)(0xE0 | (cp >> 12)));
    out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
    out.push_back((char)(0x80 | (cp & 0x3F)));
  } else {
    out.push_back((char)(0xF0 | (cp >> 18)));
    out.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
    out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
    out.push_back((char)(0x80 | (cp & 0x3F)));
  }
}

struct JParser {
  const std::string& s;
  size_t p = 0;
  explicit JParser(const std::string& t) : s(t) {}
  [[noreturn]] void fail(const char* m) { throw std::runtime_error(std::string("voice config JSON: ") + m); }
  void ws() { while (p < s.size() && (s[p] == ' ' || s[p] == '\n' || s[p] == '\t' || s[p] == '\r')) ++p; }
  uint32_t hex4() {
    if (p + 4 > s.size()) fail("bad \\u escape");
    uint32_t v = 0;
    for (int i = 0; i < 4; ++i) {
      char c = s[p++];
      v <<= 4;
      if (c >= '0' && c <= '9') v |= c - '0';
      else if (c >= 'a' && c <= 'f') v |= c - 'a' + 10;
      else if (c >= 'A' && c <= 'F') v |= c - 'A' + 10;
      else fail("bad \\u escape");
    }
    return v;
  }
  std::string string() {
    if (s[p] != '"') fail("expected string");
    ++p;
    std::string out;
    while (p < s.size() && s[p] != '"') {
      char c = s[p++];
      if (c != '\\') { out.push_back(c); continue; }
      if (p >= s.size()) fail("bad escape");
      char e = s[p++];
      switch (e) {
        case 'n': out.push_back('\n'); break;
        case 't': out.push_back('\t'); break;
        case 'r': out.push_back('\r'); break;
        case 'b': out.push_back('\b'); break;
        case 'f': out.push_back('\f'); break;
        case 'u': {
          uint32_t cp = hex4();
          if (cp >= 0xD800 && cp < 0xDC00 && p + 1 < s.size() && s[p] == '\\' && s[p + 1] == 'u') {
            p += 2;
            uint32_t lo = hex4();
            cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
          }
          append_utf8(cp, out);
        } break;
        default: out.push_back(e);
      }
    }
    if (p >= s.size()) fail("unterminated string");
    ++p;
    return out;
  }
  JVal value() {
    ws();
    if (p >= s.size()) fail("unexpected end");
    JVal v;
    char c = s[p];
    if (c == '{') {
      v.kind = JVal::Obj;
      ++p; ws();
      if (p < s.size() && s[p] == '}') { ++p; return v; }
      while (true) {
        ws();
        std::string k = string();
        ws();
        if (p >= s.size() || s[p] != ':') fail("expected ':'");
        ++p;
        v.obj.emplace_back(k, value());
        ws();
        if (p < s.size() && s[p] == ',') { ++p; continue; }
        if (p < s.size() && s[p] == '}') { ++p; break; }
        fail("expected ',' or '}'");
      }
    } else if (c == '[') {
      v.kind = JVal::Arr;
      ++p; ws();
      if (p < s.size() && s[p] == ']') { ++p; return v; }
      while (true) {
        v.arr.push_back(value());
        ws();
        if (p < s.size() && s[p] == ',') { ++p; continue; }
        if (p < s.size() && s[p] == ']') { ++p; break; }
        fail("expected ',' or ']'");
      }
    } else if (c == '"') {
      v.kind = JVal::Str;
      v.str = string();
    } else if (s.compare(p, 4, "true") == 0) { v.kind = JVal::Bool; v.b = true; p += 4; }
    else if (s.compare(p, 5, "false") == 0) { v.kind = JVal::Bool; p += 5; }
    else if (s.compare(p, 4, "null") == 0) { p += 4; }
    else {
      size_t q = p;
      while (q < s.size() && (strchr("+-0123456789.eE", s[q]) != nullptr)) ++q;
      if (q == p) fail("unexpected character");
      v.kind = JVal::Num;
      v.num = std::stod(s.substr(p, q - p));
      p = q;
    }
    return v;
  }
};

// UTF-8 -> code points (invalid bytes become U+FFFD)
static std::vector<Phoneme> decode_utf8(const std::string& s) {
  std::vector<Phoneme> out;
  size_t i = 0;
  while (i < s.size()) {
    unsigned char c = (unsigned char)s[i];
    uint32_t cp;
    int n;
    if (c < 0x80) { cp = c; n = 1; }
    else if ((c >> 5) == 6) { cp = c & 0x1F; n = 2; }
    else if ((c >> 4) == 14) { cp = c & 0x0F; n = 3; }
    else if ((c >> 3) == 30) { cp = c & 0x07; n = 4; }
    else { out.push_back(0xFFFD); ++i; continue; }
    if (i + n > s.size()) { out.push_back(0xFFFD); break; }
    bool ok = true;
    for (int k = 1; k < n; ++k) {
      unsigned char cc = (unsigned char)s[i + k];
      if ((cc >> 6) != 2) { ok = false; break; }
      cp = (cp << 6) | (cc & 0x3F);
    }
    out.push_back(ok ? (Phoneme)cp : (Phoneme)0xFFFD);
    i += ok ? n : 1;
  }
  return out;
}

// ---- Unicode: full case folding and NFD (tables generated from unicodedata, scripts/gen_unicode_tables.py)
static const UniMapEntry* uni_find(const UniMapEntry* idx, int n, char32_t cp) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (idx[mid].cp < cp) lo = mid + 1; else hi = mid;
  }
  return (lo < n && idx[lo].cp == cp) ? &idx[lo] : nullptr;
}
static int uni_combining(char32_t cp) {
  int lo = 0, hi = uni_ccc_count;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (uni_ccc[mid].cp < cp) lo = mid + 1; else hi = mid;
  }
  return (lo < uni_ccc_count && uni_ccc[lo].cp == cp) ? uni_ccc[lo].ccc : 0;
}
static std::vector<Phoneme> casefold(const std::vector<Phoneme>& in) {
  std::vector<Phoneme> out;
  out.reserve(in.size());
  for (Phoneme c : in) {
    if (const UniMapEntry* e = uni_find(uni_fold_index, uni_fold_count, c))
      out.insert(out.end(), uni_fold_pool + e->off, uni_fold_pool + e->off + e->n);
    else out.push_back(c);
  }
  return out;
}
static std::vector<Phoneme> nfd(const std::vector<Phoneme>& in) {
  std::vector<Phoneme> out;
  out.reserve(in.size() + 8);
  for (Phoneme c : in) {
    if (c >= 0xAC00 && c <= 0xD7A3) {                       // Hangul syllable: algorithmic L V (T)
      const uint32_t s = c - 0xAC00;
      out.push_back(0x1100 + s / 588);
      out.push_back(0x1161 + (s % 588) / 28);
      if (s % 28) out.push_back(0x11A7 + s % 28);
    } else if (const UniMapEntry* e = uni_find(uni_nfd_index, uni_nfd_count, c)) {
      out.insert(out.end(), uni_nfd_pool + e->off, uni_nfd_pool + e->off + e->n);
    } else {
      out.push_back(c);
    }
  }
  // canonical ordering: stable sort of every run of non-starters by combining class
  size_t i = 0;
  while (i < out.size()) {
    if (uni_combining(out[i]) == 0) { ++i; continue; }
    size_t j = i;
    while (j < out.size() && uni_combining(out[j]) != 0) ++j;
    std::stable_sort(out.begin() + i, out.begin() + j,
                     [](Phoneme a, Phoneme b) { return uni_combining(a) < uni_combining(b); });
    i = j;
  }
  return out;
}

static void check(int rc) {
  if (rc != 0) throw std::runtime_error(pe_last_error());
}

static void write_le(std::ostream& o, uint32_t v, int bytes) {
  for (int i = 0; i < bytes; ++i) o.put((char)((v >> (8 * i)) & 0xFF));
}

}  // namespace

// ------------------------------------------------------------------------------------------------

const std::string VERSION = "piper-hip 0.1 (MI355X)";
std::string getVersion() { return VERSION; }

bool isSingleCodepoint(std::string s) { return decode_utf8(s).size() == 1; }

Phoneme getCodepoint(std::string s) {
  auto v = decode_utf8(s);
  if (v.empty()) throw std::runtime_error("empty string has no codepoint");
  return v[0];
}

ModelSession::~ModelSession() {
  if (engine) pe_destroy(engine);
}

void initialize(PiperConfig& config) {
  // The reference initialises espeak-ng / libtashkeel here (piper.cpp:216-249). Phonemisation is host
  // work outside this library: nothing to start for text voices.
  (void)config;
}

void terminate(PiperConfig& config) { (void)config; }

static void parseConfigs(const JVal& root, Voice& voice) {
  PhonemizeConfig& pc = voice.phonemizeConfig;
  if (const JVal* e = root.get("espeak"))
    if (const JVal* v = e->get("voice")) pc.eSpeak.voice = v->str;
  if (const JVal* t = root.get("phoneme_type"))
    if (t->str == "text") pc.phonemeType = TextPhonemes;
  if (const JVal* m = root.get("phoneme_id_map")) {
    for (auto& kv : m->obj) {
      if (!isSingleCodepoint(kv.first)) throw std::runtime_error("Phonemes must be one codepoint (phoneme id map)");
      Phoneme from = getCodepoint(kv.first);
      for (auto& idv : kv.second.arr) pc.phonemeIdMap[from].push_back((PhonemeId)idv.num);
    }
  }
  if (const JVal* m = root.get("phoneme_map")) {
    if (!pc.phonemeMap) pc.phonemeMap.emplace();
    for (auto& kv : m->obj) {
      if (!isSingleCodepoint(kv.first)) throw std::runtime_error("Phonemes must be one codepoint (phoneme map)");
      Phoneme from = getCodepoint(kv.first);
      for (auto& to : kv.second.arr) {
        if (!isSingleCodepoint(to.str)) throw std::runtime_error("Phonemes must be one codepoint (phoneme map)");
        (*pc.phonemeMap)[from].push_back(getCodepoint(to.str));
      }
    }
  }
  SynthesisConfig& sc = voice.synthesisConfig;
  if (const JVal* a = root.get("audio"))
    if (const JVal* r = a->get("sample_rate")) sc.sampleRate = (int)r->num;
  if (const JVal* inf = root.get("inference")) {
    if (const JVal* v = inf->get("noise_scale")) sc.noiseScale = (float)v->num;
    if (const JVal* v = inf->get("length_scale")) sc.lengthScale = (float)v->num;
    if (const JVal* v = inf->get("noise_w")) sc.noiseW = (float)v->num;
    if (const JVal* ps = inf->get("phoneme_silence")) {
      sc.phonemeSilenceSeconds.emplace();
      for (auto& kv : ps->obj) {
        if (!isSingleCodepoint(kv.first)) throw std::runtime_error("Phonemes must be one codepoint (phoneme silence)");
        (*sc.phonemeSilenceSeconds)[getCodepoint(kv.first)] = (float)kv.second.num;
      }
    }
  }
  const JVal* ns = root.get("num_speakers");
  if (!ns) throw std::runtime_error("voice config: missing num_speakers");
  voice.modelConfig.numSpeakers = (int)ns->num;
  if (const JVal* sm = root.get("speaker_id_map")) {
    if (!voice.modelConfig.speakerIdMap) voice.modelConfig.speakerIdMap.emplace();
    for (auto& kv : sm->obj) (*voice.modelConfig.speakerIdMap)[kv.first] = (SpeakerId)kv.second.num;
  }
}

void loadVoice(PiperConfig& config, std::string modelPath, std::string modelConfigPath, Voice& voice,
               std::optional<SpeakerId>& speakerId, bool useCuda) {
  (void)config;
  std::ifstream f(modelConfigPath);
  if (!f) throw std::runtime_error("cannot open voice config " + modelConfigPath);
  std::stringstream ss;
  ss << f.rdbuf();
  voice.configText = ss.str();
  JParser jp(voice.configText);
  JVal root = jp.value();
  if (root.kind != JVal::Obj) throw std::runtime_error("voice config: top level is not an object");
  parseConfigs(root, voice);
  if (voice.modelConfig.numSpeakers > 1) voice.synthesisConfig.speakerId = speakerId ? speakerId : std::optional<SpeakerId>(0);
  // The reference's flag picks the ORT execution provider (piper.cpp:266-274). There is one execution path here,
  // the GPU voice.session.device, so both values load the same engine (the reference's test.cpp and main.cpp default
  // pass false and must keep working against this header).
  (void)useCuda;
  if (voice.session.engine) { pe_destroy(voice.session.engine); voice.session.engine = nullptr; }
  check(pe_create(modelPath.c_str(), voice.session.device, &voice.session.engine));
}

void synthesize(std::vector<PhonemeId>& phonemeIds, SynthesisConfig& synthesisConfig, ModelSession& session,
                std::vector<int16_t>& audioBuffer, SynthesisResult& result) {
  if (!session.engine) throw std::runtime_error("voice model is not loaded");
  const float scales[3] = {synthesisConfig.noiseScale, synthesisConfig.lengthScale, synthesisConfig.noiseW};
  const int64_t offsets[2] = {0, (int64_t)phonemeIds.size()};
  // "sid" is only fed for multi-speaker voices (piper.cpp:367-377); single-speaker graphs have no such input
  const int64_t sid = synthesisConfig.speakerId.value_or(0);
  const int64_t* sids = synthesisConfig.speakerId ? &sid : nullptr;
  // inferSeconds spans what session.Run() spans in the reference (piper.cpp:385-395): inputs handed over in host
  // memory, result available in host memory = upload + device pipeline + fetch. The copy into the caller's vector
  // is outside, like the reference's conversion loops (:410-431; the int16 conversion itself runs on the GPU).
  const auto t0 = std::chrono::steady_clock::now();
  check(pe_upload(session.engine, phonemeIds.data(), offsets, 1, scales, sids, nullptr));
  check(pe_run(session.engine));
  pe_result r;
  check(pe_fetch(session.engine, 0, 1, &r));
  result.inferSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  const int64_t n = r.sample_offsets[1];
  result.audioSeconds = (double)n / (double)synthesisConfig.sampleRate;
  result.realTimeFactor = result.audioSeconds > 0 ? result.inferSeconds / result.audioSeconds : 0.0;
  audioBuffer.insert(audioBuffer.end(), r.pcm, r.pcm + n);
}

void synthesizeBatch(std::vector<std::vector<PhonemeId>>& phonemeIdLists, SynthesisConfig& synthesisConfig,
                     ModelSession& session, std::vector<std::vector<int16_t>>& audioBuffers,
                     SynthesisResult& result) {
  if (!session.engine) throw std::runtime_error("voice model is not loaded");
  const int32_t nb = (int32_t)phonemeIdLists.size();
  audioBuffers.assign(nb, {});
  if (nb == 0) return;
  const float scales[3] = {synthesisConfig.noiseScale, synthesisConfig.lengthScale, synthesisConfig.noiseW};
  std::vector<PhonemeId> flat;
  std::vector<int64_t> offsets(1, 0), sids(nb, synthesisConfig.speakerId.value_or(0));
  for (auto& ids : phonemeIdLists) {
    flat.insert(flat.end(), ids.begin(), ids.end());
    offsets.push_back((int64_t)flat.size());
  }
  const auto t0 = std::chrono::steady_clock::now();     // same span as synthesize(): upload + pipeline + fetch
  check(pe_upload(session.engine, flat.data(), offsets.data(), nb, scales,
                  synthesisConfig.speakerId ? sids.data() : nullptr, nullptr));
  check(pe_run(session.engine));
  pe_result r;
  check(pe_fetch(session.engine, 0, 1, &r));
  result.inferSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  for (int32_t i = 0; i < nb; ++i)
    audioBuffers[i].assign(r.pcm + r.sample_offsets[i], r.pcm + r.sample_offsets[i + 1]);
  result.audioSeconds = (double)r.sample_offsets[nb] / (double)synthesisConfig.sampleRate;
  result.realTimeFactor = result.audioSeconds > 0 ? result.inferSeconds / result.audioSeconds : 0.0;
}

void phonemize_codepoints(const std::string& text, std::vector<std::vector<Phoneme>>& sentencePhonemes) {
  sentencePhonemes.push_back(nfd(casefold(decode_utf8(text))));
}

void phonemes_to_ids(const std::vector<Phoneme>& phonemes, const PhonemizeConfig& config,
                     std::vector<PhonemeId>& phonemeIds, std::map<Phoneme, std::size_t>& missingPhonemes) {
  phonemeIds.push_back(config.idBos);
  if (config.interspersePad) phonemeIds.push_back(config.idPad);
  for (Phoneme ph : phonemes) {
    auto it = config.phonemeIdMap.find(ph);
    if (it == config.phonemeIdMap.end()) {
      missingPhonemes[ph] += 1;
      continue;
    }
    for (PhonemeId id : it->second) {
      phonemeIds.push_back(id);
      if (config.interspersePad) phonemeIds.push_back(config.idPad);
    }
  }
  phonemeIds.push_back(config.idEos);
}

void textToAudio(PiperConfig& config, Voice& voice, std::string text, std::vector<int16_t>& audioBuffer,
                 SynthesisResult& result, const std::function<void()>& audioCallback) {
  const SynthesisConfig& sc = voice.synthesisConfig;
  std::size_t sentenceSilenceSamples = 0;
  if (sc.sentenceSilenceSeconds > 0)
    sentenceSilenceSamples = (std::size_t)(sc.sentenceSilenceSeconds * sc.sampleRate * sc.channels);

  std::vector<std::vector<Phoneme>> sentences;
  if (voice.phonemizeConfig.phonemeType == eSpeakPhonemes) {
    // piper.cpp:470-479: phonemize_eSpeak(text, {voice}, phonemes) -- espeak-ng lives on the host, behind the slot
    if (!config.phonemizer)
      throw std::runtime_error(
          "this voice uses eSpeak phonemes: set PiperConfig::phonemizer (e.g. to piper-phonemize's phonemize_eSpeak) "
          "or pass phoneme ids to piper::synthesize()");
    config.phonemizer(text, voice.phonemizeConfig.eSpeak.voice, sentences);
  } else {
    // piper.cpp:480-484: UTF-8 code points as phonemes, default CodepointsPhonemeConfig (case folding + NFD; the
    // voice's phoneme_map is NOT applied on this path in the reference)
    phonemize_codepoints(text, sentences);
  }

  std::vector<PhonemeId> phonemeIds;
  std::map<Phoneme, std::size_t> missingPhonemes;
  for (auto& sentence : sentences) {
    std::vector<std::vector<Phoneme>> phrases;
    std::vector<std::size_t> phraseSilence;
    if (sc.phonemeSilenceSeconds) {
      phrases.emplace_back();
      for (Phoneme p : sentence) {
        phrases.back().push_back(p);
        auto it = sc.phonemeSilenceSeconds->find(p);
        if (it != sc.phonemeSilenceSeconds->end()) {
          phraseSilence.push_back((std::size_t)(it->second * sc.sampleRate * sc.channels));
          phrases.emplace_back();
        }
      }
    } else {
      phrases.push_back(sentence);
    }
    phraseSilence.resize(phrases.size(), 0);
    // The reference runs one session.Run() per phrase (piper.cpp:548-575); the phrases of a sentence are
    // independent, so here they go through the engine as one batch and are appended in order.
    std::vector<std::vector<PhonemeId>> idLists;
    std::vector<std::size_t> owner;
    for (std::size_t i = 0; i < phrases.size(); ++i) {
      if (phrases[i].empty()) continue;
      phonemes_to_ids(phrases[i], voice.phonemizeConfig, phonemeIds, missingPhonemes);
      idLists.push_back(phonemeIds);
      owner.push_back(i);
      phonemeIds.clear();
    }
    if (idLists.size() == 1) {
      SynthesisResult pr;
      synthesize(idLists[0], voice.synthesisConfig, voice.session, audioBuffer, pr);
      audioBuffer.insert(audioBuffer.end(), phraseSilence[owner[0]], (int16_t)0);
      result.audioSeconds += pr.audioSeconds;
      result.inferSeconds += pr.inferSeconds;
    } else if (!idLists.empty()) {
      SynthesisResult pr;
      std::vector<std::vector<int16_t>> parts;
      synthesizeBatch(idLists, voice.synthesisConfig, voice.session, parts, pr);
      for (std::size_t k = 0; k < parts.size(); ++k) {
        audioBuffer.insert(audioBuffer.end(), parts[k].begin(), parts[k].end());
        audioBuffer.insert(audioBuffer.end(), phraseSilence[owner[k]], (int16_t)0);
      }
      result.audioSeconds += pr.audioSeconds;
      result.inferSeconds += pr.inferSeconds;
    }
    if (sentenceSilenceSamples > 0) audioBuffer.insert(audioBuffer.end(), sentenceSilenceSamples, (int16_t)0);
    if (audioCallback) {
      audioCallback();      // the callback must copy: the buffer is cleared afterwards (piper.cpp:591-595)
      audioBuffer.clear();
    }
  }
  if (!missingPhonemes.empty()) {     // piper.cpp:600-610
    auto warn = [&](const std::string& m) {
      if (config.warn) config.warn(m);
      else fprintf(stderr, "[piper] warning: %s\n", m.c_str());
    };
    warn("Missing " + std::to_string(missingPhonemes.size()) + " phoneme(s) from phoneme/id map!");
    for (auto& pc : missingPhonemes) {
      std::string ph;
      append_utf8((uint32_t)pc.first, ph);
      char buf[64];
      snprintf(buf, sizeof(buf), "\" (\\u%04X): %zu time(s)", (unsigned)pc.first, pc.second);
      warn("Missing \"" + ph + buf);
    }
  }
  if (result.audioSeconds > 0) result.realTimeFactor = result.inferSeconds / result.audioSeconds;
}

void textToWavFile(PiperConfig& config, Voice& voice, std::string text, std::ostream& audioFile,
                   SynthesisResult& result) {
  std::vector<int16_t> audio;
  textToAudio(config, voice, text, audio, result, nullptr);
  const SynthesisConfig& sc = voice.synthesisConfig;
  // 44-byte RIFF/WAVE PCM header, fields as the reference's wavfile.hpp:6-38
  const uint32_t dataSize = (uint32_t)audio.size() * sc.sampleWidth * sc.channels;
  audioFile.write("RIFF", 4);
  write_le(audioFile, dataSize + 44 - 8, 4);
  audioFile.write("WAVE", 4);
  audioFile.write("fmt ", 4);
  write_le(audioFile, 16, 4);
  write_le(audioFile, 1, 2);
  write_le(audioFile, (uint32_t)sc.channels, 2);
  write_le(audioFile, (uint32_t)sc.sampleRate, 4);
  write_le(audioFile, (uint32_t)(sc.sampleRate * sc.sampleWidth * sc.channels), 4);
  write_le(audioFile, (uint32_t)(sc.sampleWidth * sc.channels), 2);
  write_le(audioFile, 16, 2);
  audioFile.write("data", 4);
  write_le(audioFile, dataSize, 4);
  audioFile.write((const char*)audio.data(), sizeof(int16_t) * audio.size());
}

}  // namespace piper
