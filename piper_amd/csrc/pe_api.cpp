// C ABI (include/piper_hip.h) over pe::Engine.
#include "../../include/piper_hip.h"

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>

#include "engine.h"

struct pe_engine {
  pe::Engine* eng;
  std::vector<int64_t> one_off;
};

static thread_local std::string g_err;

template <class F>
static int guard(F&& f) {
  try {
    f();
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return 1;
  } catch (...) {
    g_err = "unknown error";
    return 1;
  }
}

static void fill_result(pe_engine* e, pe_result* r, double secs) {
  if (!r) return;
  r->batch = e->eng->batch();
  r->sample_offsets = e->eng->sample_offsets().data();
  r->audio = e->eng->audio_host();
  r->pcm = e->eng->pcm_host();
  r->frames = e->eng->frames_host().data();
  r->infer_seconds = secs;
}

extern "C" {

const char* pe_last_error(void) { return g_err.c_str(); }

int pe_create_from_blob(const void* blob, size_t nbytes, int device, pe_engine** out) {
  return guard([&] {
    if (!blob || !out) throw std::runtime_error("null argument");
    pe::WeightSet ws = pe::parse_blob(blob, nbytes);
    auto* h = new pe_engine{nullptr, {}};
    try {
      h->eng = new pe::Engine(ws, device);
    } catch (...) {
      delete h;
      throw;
    }
    *out = h;
  });
}

int pe_weights_bound(const void* blob, size_t nbytes, size_t* bound) {
  return guard([&] {
    if (!blob || !bound) throw std::runtime_error("null argument");
    pe::WeightSet ws = pe::parse_blob(blob, nbytes, true);
    *bound = pe::Engine::arena_bound(ws);
  });
}

int pe_create_in_arena(const void* blob, size_t nbytes, int device, void* arena, size_t arena_bytes, int skeleton,
                       pe_engine** out) {
  return guard([&] {
    if (!blob || !out || !arena) throw std::runtime_error("null argument");
    pe::WeightSet ws = pe::parse_blob(blob, nbytes, skeleton != 0);
    auto* h = new pe_engine{nullptr, {}};
    try {
      h->eng = new pe::Engine(ws, device, pe::ArenaSpec{arena, arena_bytes, skeleton != 0});
    } catch (...) {
      delete h;
      throw;
    }
    *out = h;
  });
}

int pe_weights_used(pe_engine* e, size_t* used) {
  return guard([&] {
    if (!e || !used) throw std::runtime_error("null argument");
    *used = e->eng->arena_used();
  });
}

int pe_arena_ready(pe_engine* e) {
  return guard([&] {
    if (!e) throw std::runtime_error("null engine");
    e->eng->arena_ready();
  });
}

int pe_create(const char* onnx_path, int device, pe_engine** out) {
  return guard([&] {
    if (!onnx_path || !out) throw std::runtime_error("null argument");
    pe::WeightSet ws = pe::load_onnx(onnx_path);
    auto* h = new pe_engine{nullptr, {}};
    try {
      h->eng = new pe::Engine(ws, device);
    } catch (...) {
      delete h;
      throw;
    }
    *out = h;
  });
}

int pe_onnx_to_blob(const char* onnx_path, void** blob, size_t* nbytes) {
  return guard([&] {
    if (!onnx_path || !blob || !nbytes) throw std::runtime_error("null argument");
    pe::WeightSet ws = pe::load_onnx(onnx_path);
    std::vector<uint8_t> b = pe::serialize_blob(ws);
    void* p = malloc(b.size());
    if (!p) throw std::runtime_error("out of memory");
    memcpy(p, b.data(), b.size());
    *blob = p;
    *nbytes = b.size();
  });
}

void pe_free(void* p) { free(p); }

void pe_destroy(pe_engine* e) {
  if (!e) return;
  delete e->eng;
  delete e;
}

int pe_upload(pe_engine* e, const int64_t* ids, const int64_t* offsets, int32_t batch, const float scales[3],
              const int64_t* sids, const pe_noise* noise) {
  return guard([&] {
    if (!e || !ids || !offsets || !scales) throw std::runtime_error("null argument");
    pe::NoiseIn n;
    if (noise) {
      n.noise_w = noise->noise_w; n.w_stride = noise->w_stride;
      n.noise_z = noise->noise_z; n.z_stride = noise->z_stride;
    }
    e->eng->upload(ids, offsets, batch, scales, sids, noise ? &n : nullptr);
  });
}

int pe_run(pe_engine* e) {
  return guard([&] {
    if (!e) throw std::runtime_error("null engine");
    e->eng->run();
  });
}

int pe_fetch(pe_engine* e, int want_audio, int want_pcm, pe_result* result) {
  return guard([&] {
    if (!e) throw std::runtime_error("null engine");
    e->eng->download(want_audio != 0, want_pcm != 0);
    fill_result(e, result, 0.0);
  });
}

int pe_synthesize_batch(pe_engine* e, const int64_t* ids, const int64_t* offsets, int32_t batch,
                        const float scales[3], const int64_t* sids, const pe_noise* noise, pe_result* result) {
  return guard([&] {
    if (!e || !ids || !offsets || !scales) throw std::runtime_error("null argument");
    pe::NoiseIn n;
    if (noise) {
      n.noise_w = noise->noise_w; n.w_stride = noise->w_stride;
      n.noise_z = noise->noise_z; n.z_stride = noise->z_stride;
    }
    const auto t0 = std::chrono::steady_clock::now();
    e->eng->upload(ids, offsets, batch, scales, sids, noise ? &n : nullptr);
    e->eng->run();
    e->eng->download(true, true);
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    fill_result(e, result, secs);
  });
}

int pe_synthesize(pe_engine* e, const int64_t* ids, int64_t n_ids, const float scales[3], int64_t sid,
                  const pe_noise* noise, pe_result* result) {
  if (!e) {
    g_err = "null engine";
    return 1;
  }
  e->one_off = {0, n_ids};
  const int64_t sids[1] = {sid < 0 ? 0 : sid};
  return pe_synthesize_batch(e, ids, e->one_off.data(), 1, scales, sids, noise, result);
}

int pe_stream_begin(pe_engine* e, const int64_t* ids, int64_t n_ids, const float scales[3], int64_t sid,
                    const pe_noise* noise, int32_t* total_frames, int32_t* halo_frames) {
  return guard([&] {
    if (!e || !ids || !scales) throw std::runtime_error("null argument");
    pe::NoiseIn n;
    if (noise) {
      n.noise_w = noise->noise_w; n.w_stride = noise->w_stride;
      n.noise_z = noise->noise_z; n.z_stride = noise->z_stride;
    }
    const int f = e->eng->stream_begin(ids, n_ids, scales, sid, noise ? &n : nullptr);
    if (total_frames) *total_frames = f;
    if (halo_frames) *halo_frames = e->eng->decoder_halo_frames();
  });
}

int pe_stream_next(pe_engine* e, int32_t chunk_frames, const float** audio, const int16_t** pcm, int64_t* n_samples) {
  return guard([&] {
    if (!e || !n_samples) throw std::runtime_error("null argument");
    e->eng->stream_next(chunk_frames, audio, pcm, n_samples);
  });
}

int pe_get_durations(pe_engine* e, int32_t* out, int64_t capacity, int64_t* n) {
  return guard([&] {
    if (!e || !n) throw std::runtime_error("null argument");
    const std::vector<int32_t>& d = e->eng->durations_host();
    *n = (int64_t)d.size();
    if (out) {
      if (capacity < (int64_t)d.size()) throw std::runtime_error("durations buffer too small");
      memcpy(out, d.data(), d.size() * sizeof(int32_t));
    }
  });
}

int pe_get_info(pe_engine* e, int32_t* sample_rate, int32_t* hop, int32_t* n_speakers, int32_t* n_symbols,
                int64_t* weight_bytes) {
  return guard([&] {
    if (!e) throw std::runtime_error("null engine");
    if (sample_rate) *sample_rate = e->eng->sample_rate();
    if (hop) *hop = e->eng->hop();
    if (n_speakers) *n_speakers = e->eng->arch()[pe::A_NSPK];
    if (n_symbols) *n_symbols = e->eng->arch()[pe::A_NVOCAB];
    if (weight_bytes) *weight_bytes = (int64_t)e->eng->weight_bytes();
  });
}

void pe_set_seed(pe_engine* e, uint64_t seed) {
  if (e) e->eng->set_seed(seed);
}

int pe_profile_enable(pe_engine* e, int on) {
  return guard([&] {
    if (!e) throw std::runtime_error("null engine");
    e->eng->set_profile(on);
  });
}
int pe_profile_reset(pe_engine* e) {
  return guard([&] {
    if (!e) throw std::runtime_error("null engine");
    e->eng->reset_profile();
  });
}
int pe_profile_rows(pe_engine* e) { return e ? (int)e->eng->profile().size() : 0; }
int pe_profile_get(pe_engine* e, int row, const char** name, double* ms, double* flops, int64_t* launches) {
  return guard([&] {
    if (!e) throw std::runtime_error("null engine");
    const auto& p = e->eng->profile();
    if (row < 0 || row >= (int)p.size()) throw std::runtime_error("profile row out of range");
    if (name) *name = p[row].name;
    if (ms) *ms = p[row].ms;
    if (flops) *flops = p[row].flops;
    if (launches) *launches = p[row].launches;
  });
}

int pe_profile_bytes(pe_engine* e, int row, double* bytes) {
  return guard([&] {
    if (!e) throw std::runtime_error("null engine");
    const auto& p = e->eng->profile();
    if (row < 0 || row >= (int)p.size() || !bytes) throw std::runtime_error("profile row out of range");
    *bytes = p[row].bytes;
  });
}

void* pe_stream(pe_engine* e) { return e ? (void*)e->eng->stream() : nullptr; }

int pe_debug_tensor(pe_engine* e, const char* name, int32_t b, float* out, int64_t capacity, int32_t* rows,
                    int32_t* cols) {
  return guard([&] {
    if (!e || !name || !out || !rows || !cols) throw std::runtime_error("null argument");
    std::vector<float> v;
    int r = 0, c = 0;
    e->eng->debug_tensor(name, b, v, &r, &c);
    if ((int64_t)v.size() > capacity) throw std::runtime_error("debug buffer too small");
    memcpy(out, v.data(), v.size() * sizeof(float));
    *rows = r;
    *cols = c;
  });
}

int pe_debug_randn(pe_engine* e, int32_t site, uint64_t call, int64_t n, float* out) {
  return guard([&] {
    if (!e || !out) throw std::runtime_error("null argument");
    e->eng->debug_randn(site, call, n, out);
  });
}

uint64_t pe_rng_calls(pe_engine* e) { return e ? e->eng->rng_call() : 0; }

int64_t pe_run_launches(pe_engine* e) { return e ? (int64_t)e->eng->run_launches() : 0; }

}  // extern "C"
