// piper.hpp -- the reference's C++ API surface for the synthesis path, re-hosted on the MI355X engine.
//
// Same namespace, type names, function names, argument meaning and error behaviour (std::runtime_error)
// as the reference's src/cpp/piper.hpp:20-128 + piper.cpp:337, so existing callers (the reference's
// main.cpp loop, test.cpp) compile against it unchanged in spirit. What changed:
//   * ModelSession no longer wraps Ort::Session/Env/Options; it holds an opaque pe_engine* (piper_hip.h);
//   * the vendored nlohmann json / utf8 / piper-phonemize headers are not needed by this header
//     (config parsing uses a small built-in JSON reader; Phoneme/PhonemeId are defined here with the
//     reference's types: char32_t / int64_t);
//   * `useCuda` is accepted with either value (the reference's own test.cpp / main.cpp default pass false): this
//     library has exactly one execution path, the GPU selected by ModelSession::device;
//   * espeak-ng / libtashkeel phonemisation stays host-side and is not linked: voices with
//     "phoneme_type": "text" are phonemised natively (casefold + NFD code points, like piper-phonemize's
//     phonemize_codepoints); eSpeak voices are phonemised through PiperConfig::phonemizer -- the slot a host that
//     links piper-phonemize fills with phonemize_eSpeak -- or synthesised from ids (synthesize()).
#ifndef PIPER_H_
#define PIPER_H_

#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <optional>
#include <ostream>
#include <string>
#include <vector>

struct pe_engine;

namespace piper {

struct eSpeakConfig {
  std::string voice = "en-us";
};

typedef char32_t Phoneme;      // piper-phonemize/phoneme_ids.hpp
typedef int64_t PhonemeId;
typedef int64_t SpeakerId;

// text + espeak voice name -> phonemes of each sentence: the signature of piper-phonemize's phonemize_eSpeak as
// piper.cpp:470-479 calls it (text, eSpeakPhonemeConfig{voice}, phonemes)
typedef std::function<void(const std::string &text, const std::string &espeakVoice,
                           std::vector<std::vector<Phoneme>> &sentencePhonemes)> PhonemizeFn;
// text -> diacritized text: the signature of libtashkeel's tashkeel_run minus its state argument (piper.cpp:463)
typedef std::function<std::string(const std::string &text)> TashkeelFn;
// where missing-phoneme warnings go (the reference logs them with spdlog::warn, piper.cpp:600-610); default: stderr
typedef std::function<void(const std::string &message)> WarnFn;

struct PiperConfig {
  std::string eSpeakDataPath;
  bool useESpeak = true;
  bool useTashkeel = false;
  std::optional<std::string> tashkeelModelPath;
  // Host-side phonemizer for eSpeak voices (espeak-ng stays on the host: BASELINE.json north_star). Unset: textToAudio
  // throws for eSpeak voices, exactly where the reference would call phonemize_eSpeak.
  PhonemizeFn phonemizer;
  WarnFn warn;
  // Host-side diacritizer for Arabic voices: the slot for libtashkeel's tashkeel_run (piper.cpp:457-464), which stays on
  // the host like espeak-ng. With useTashkeel set and no function here textToAudio throws "Tashkeel model is not
  // loaded", as the reference does without a tashkeelState.
  TashkeelFn tashkeel;
  // textToAudio runs the phrases of a text as batched engine calls; a call takes at most this many PADDED phoneme ids
  // (utterances x the longest one), so that device and pinned host memory stay bounded by a budget instead of growing
  // with sentences x the longest sentence (the reference's memory is bounded by one phrase: one session.Run per phrase,
  // piper.cpp:549-582). 8192 = the 64 x 128-id configuration the engine is tuned for.
  std::size_t maxBatchIds = 8192;
};

enum PhonemeType { eSpeakPhonemes, TextPhonemes };

struct PhonemizeConfig {
  PhonemeType phonemeType = eSpeakPhonemes;
  std::optional<std::map<Phoneme, std::vector<Phoneme>>> phonemeMap;
  std::map<Phoneme, std::vector<PhonemeId>> phonemeIdMap;

  PhonemeId idPad = 0;  // padding (optionally interspersed)
  PhonemeId idBos = 1;  // beginning of sentence
  PhonemeId idEos = 2;  // end of sentence
  bool interspersePad = true;

  eSpeakConfig eSpeak;
};

struct SynthesisConfig {
  // VITS inference settings
  float noiseScale = 0.667f;
  float lengthScale = 1.0f;
  float noiseW = 0.8f;

  // Audio settings
  int sampleRate = 22050;
  int sampleWidth = 2;  // 16-bit
  int channels = 1;     // mono

  // Speaker id from 0 to numSpeakers - 1
  std::optional<SpeakerId> speakerId;

  // Extra silence
  float sentenceSilenceSeconds = 0.2f;
  std::optional<std::map<piper::Phoneme, float>> phonemeSilenceSeconds;
};

struct ModelConfig {
  int numSpeakers = 1;
  std::optional<std::map<std::string, SpeakerId>> speakerIdMap;   // speaker name -> id
};

// Was: Ort::Session + allocator + options + env (reference piper.hpp:78-85).
struct ModelSession {
  pe_engine* engine = nullptr;
  int device = 0;
  ModelSession() = default;
  ModelSession(const ModelSession&) = delete;
  ModelSession& operator=(const ModelSession&) = delete;
  ~ModelSession();
};

struct SynthesisResult {
  double inferSeconds = 0;
  double audioSeconds = 0;
  double realTimeFactor = 0;
};

struct Voice {
  std::string configText;        // raw .onnx.json (the reference keeps the parsed json root)
  PhonemizeConfig phonemizeConfig;
  SynthesisConfig synthesisConfig;
  ModelConfig modelConfig;
  ModelSession session;
};

// True if the string is a single UTF-8 codepoint
bool isSingleCodepoint(std::string s);

// Get the first UTF-8 codepoint of a string
Phoneme getCodepoint(std::string s);

// Get version of Piper
std::string getVersion();

// Must be called before using textTo* functions
void initialize(PiperConfig &config);

// Clean up
void terminate(PiperConfig &config);

// Load Onnx model and JSON config file
void loadVoice(PiperConfig &config, std::string modelPath, std::string modelConfigPath, Voice &voice,
               std::optional<SpeakerId> &speakerId, bool useCuda);

// Phoneme ids to WAV audio: appends to audioBuffer (never clears it), fills result like piper.cpp:385-406
void synthesize(std::vector<PhonemeId> &phonemeIds, SynthesisConfig &synthesisConfig, ModelSession &session,
                std::vector<int16_t> &audioBuffer, SynthesisResult &result);

// Extension (not in the reference): several id sequences in ONE device call (pe_synthesize_batch). Every sequence
// is computed, and peak-normalised to int16, exactly as its own synthesize() call would; audioBuffers[i] receives
// sequence i. textToAudio uses it for the phrases of a sentence.
void synthesizeBatch(std::vector<std::vector<PhonemeId>> &phonemeIdLists, SynthesisConfig &synthesisConfig,
                     ModelSession &session, std::vector<std::vector<int16_t>> &audioBuffers,
                     SynthesisResult &result);

// piper-phonemize's phonemize_codepoints with its default config (casing = fold, no phoneme map), as piper.cpp:480-484
// calls it: Unicode full case folding, then NFD; one "sentence" holding every code point.
void phonemize_codepoints(const std::string &text, std::vector<std::vector<Phoneme>> &sentencePhonemes);

// Phonemes -> ids with the piper-phonemize rule used at piper.cpp:555 (BOS, PAD, (id.., PAD)*, EOS)
void phonemes_to_ids(const std::vector<Phoneme> &phonemes, const PhonemizeConfig &config,
                     std::vector<PhonemeId> &phonemeIds, std::map<Phoneme, std::size_t> &missingPhonemes);

// Phonemize text and synthesize audio
void textToAudio(PiperConfig &config, Voice &voice, std::string text, std::vector<int16_t> &audioBuffer,
                 SynthesisResult &result, const std::function<void()> &audioCallback);

// Phonemize text and synthesize audio to WAV file
void textToWavFile(PiperConfig &config, Voice &voice, std::string text, std::ostream &audioFile,
                   SynthesisResult &result);

}  // namespace piper

#endif  // PIPER_H_
