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

  if (config.useTashkeel) {     // piper.cpp:457-464: diacritize first; the model lives on the host, behind the slot
    if (!config.tashkeel) throw std::runtime_error("Tashkeel model is not loaded");
    text = config.tashkeel(text);
  }

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

  // ---- host side first: every sentence -> phrases (split at the phoneme_silence phonemes, piper.cpp:497-546) -> ids
  struct Phrase {
    std::vector<PhonemeId> ids;
    std::size_t silenceAfter = 0;       // samples of phoneme silence appended behind the phrase
    std::size_t sentence = 0;
  };
  std::vector<Phrase> phrases;
  std::map<Phoneme, std::size_t> missingPhonemes;
  for (std::size_t si = 0; si < sentences.size(); ++si) {
    std::vector<std::vector<Phoneme>> parts;
    std::vector<std::size_t> partSilence;
    if (sc.phonemeSilenceSeconds) {
      parts.emplace_back();
      for (Phoneme p : sentences[si]) {
        parts.back().push_back(p);
        auto it = sc.phonemeSilenceSeconds->find(p);
        if (it != sc.phonemeSilenceSeconds->end()) {
          partSilence.push_back((std::size_t)(it->second * sc.sampleRate * sc.channels));
          parts.emplace_back();
        }
      }
    } else {
      parts.push_back(sentences[si]);
    }
    partSilence.resize(parts.size(), 0);
    for (std::size_t i = 0; i < parts.size(); ++i) {
      if (parts[i].empty()) continue;
      Phrase ph;
      phonemes_to_ids(parts[i], voice.phonemizeConfig, ph.ids, missingPhonemes);
      ph.silenceAfter = partSilence[i];
      ph.sentence = si;
      phrases.push_back(std::move(ph));
    }
  }

  // ---- device side. The reference runs one session.Run() per phrase, one after the other (piper.cpp:548-575, the
  // call at :570). Phrases are independent utterances, so they go through the engine in batches:
  //   * no audioCallback (textToWavFile, piper.cpp:619-634): every phrase of every sentence in ONE call (a 16-sentence
  //     text costs about one batch-16 call instead of 16 latency-bound ones);
  //   * with a callback (audio is consumed sentence by sentence): sentence groups of 1, 2, 4, ... -- the first sentence
  //     alone, for the time to first audio -- and while the caller's callbacks consume group g the engine already runs
  //     group g + 1 (its upload + launch are enqueued first; the PCM of g was copied out of the engine's buffer before).
  // Either way the caller sees the reference's sequence: per sentence its phrases + silences appended to audioBuffer,
  // then audioCallback(), then the buffer cleared (piper.cpp:577-595).
  // A group is bounded by a PADDED-size budget (config.maxBatchIds: utterances x the longest one -- what the engine's
  // workspaces are sized by), so that a long document with one long sentence costs a bounded amount of device and pinned
  // host memory; a phrase longer than the budget goes alone, as in the reference.
  struct Group { std::size_t p0 = 0, p1 = 0; };       // phrases [p0, p1): whole sentences, except where a limit cuts in
  std::vector<Group> groups;
  {
    const std::size_t budget = std::max<std::size_t>(config.maxBatchIds, 1);
    std::size_t p = 0, s = 0, want = audioCallback ? 1 : sentences.size();
    while (s < sentences.size()) {
      Group g;
      g.p0 = p;
      const std::size_t s1 = std::min(sentences.size(), s + std::max<std::size_t>(want, 1));
      std::size_t longest = 0;
      while (p < phrases.size() && phrases[p].sentence < s1 && p - g.p0 < 4096) {      // 4096 = the engine's batch limit
        const std::size_t l = std::max(longest, phrases[p].ids.size());
        if (p > g.p0 && l * (p - g.p0 + 1) > budget) break;
        longest = l;
        ++p;
      }
      g.p1 = p;
      groups.push_back(g);
      s = (p < phrases.size() && phrases[p].sentence < s1) ? phrases[p].sentence : s1;   // limit hit: go on from there
      if (audioCallback) want = std::min<std::size_t>(want * 2, 64);
    }
  }
  const float scales[3] = {sc.noiseScale, sc.lengthScale, sc.noiseW};
  auto enqueue = [&](const Group& g) -> double {          // upload + launch; returns the host seconds spent
    if (g.p1 == g.p0) return 0.0;
    std::vector<PhonemeId> flat;
    std::vector<int64_t> offsets(1, 0), sids(g.p1 - g.p0, sc.speakerId.value_or(0));
    for (std::size_t k = g.p0; k < g.p1; ++k) {
      flat.insert(flat.end(), phrases[k].ids.begin(), phrases[k].ids.end());
      offsets.push_back((int64_t)flat.size());
    }
    const auto t0 = std::chrono::steady_clock::now();
    check(pe_upload(voice.session.engine, flat.data(), offsets.data(), (int32_t)(g.p1 - g.p0), scales,
                    sc.speakerId ? sids.data() : nullptr, nullptr));
    check(pe_run(voice.session.engine));
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  };
  // the finished group's PCM, copied out of the engine's (reused) host buffer: per phrase
  auto collect = [&](const Group& g, std::vector<std::vector<int16_t>>& out) -> double {
    out.clear();
    if (g.p1 == g.p0) return 0.0;
    const auto t0 = std::chrono::steady_clock::now();
    pe_result r;
    check(pe_fetch(voice.session.engine, 0, 1, &r));
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    out.resize(g.p1 - g.p0);
    for (std::size_t k = 0; k < out.size(); ++k) out[k].assign(r.pcm + r.sample_offsets[k], r.pcm + r.sample_offsets[k + 1]);
    result.audioSeconds += (double)r.sample_offsets[out.size()] / (double)sc.sampleRate;
    return dt;
  };
  if (!voice.session.engine) throw std::runtime_error("voice model is not loaded");
  std::size_t open_sentence = (std::size_t)-1;     // sentence whose audio is being assembled in audioBuffer
  auto close_sentence = [&]() {
    if (open_sentence == (std::size_t)-1) return;
    if (sentenceSilenceSamples > 0) audioBuffer.insert(audioBuffer.end(), sentenceSilenceSamples, (int16_t)0);
    if (audioCallback) {
      audioCallback();      // the callback must copy: the buffer is cleared afterwards (piper.cpp:591-595)
      audioBuffer.clear();
    }
    open_sentence = (std::size_t)-1;
  };
  // sentences without a single phrase (empty) still get their silence + callback, in order
  std::size_t next_sentence = 0;
  auto emit = [&](const Group& g, const std::vector<std::vector<int16_t>>& pcm) {
    for (std::size_t k = g.p0; k < g.p1; ++k) {
      const std::size_t si = phrases[k].sentence;
      if (si != open_sentence) {
        close_sentence();
        for (; next_sentence < si; ++next_sentence) { open_sentence = next_sentence; close_sentence(); }
        open_sentence = si;
        next_sentence = si + 1;
      }
      audioBuffer.insert(audioBuffer.end(), pcm[k - g.p0].begin(), pcm[k - g.p0].end());
      audioBuffer.insert(audioBuffer.end(), phrases[k].silenceAfter, (int16_t)0);
    }
  };
  std::vector<std::vector<int16_t>> cur, prev;
  if (!groups.empty()) result.inferSeconds += enqueue(groups[0]);
  for (std::size_t gi = 0; gi < groups.size(); ++gi) {
    result.inferSeconds += collect(groups[gi], cur);                        // waits for group gi
    if (gi + 1 < groups.size()) result.inferSeconds += enqueue(groups[gi + 1]);   // the engine starts on gi + 1 ...
    emit(groups[gi], cur);                                                  // ... while the caller consumes gi
    // a sentence is closed (silence, callback) as soon as no later group continues it
    if (gi + 1 == groups.size() || groups[gi + 1].p0 >= phrases.size() ||
        phrases[groups[gi + 1].p0].sentence != open_sentence)
      close_sentence();
  }
  close_sentence();
  for (; next_sentence < sentences.size(); ++next_sentence) { open_sentence = next_sentence; close_sentence(); }
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
