// C ABI (include/piper_hip.h) over pe::Engine.
#include "../../include/piper_hip.h"

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <algorithm>
#include <numeric>
#include <string>
#include <thread>

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

int pe_debug_randn(pe_engine* e, int32_t site, uint64_t call, int64_t row, int64_t n, float* out) {
  return guard([&] {
    if (!e || !out) throw std::runtime_error("null argument");
    e->eng->debug_randn(site, call, row, n, out);
  });
}

uint64_t pe_rng_calls(pe_engine* e) { return e ? e->eng->rng_call() : 0; }

int64_t pe_run_launches(pe_engine* e) { return e ? (int64_t)e->eng->run_launches() : 0; }

int pe_speculation_stats(pe_engine* e, int64_t* runs, int64_t* misses) {
  return guard([&] {
    if (!e) throw std::runtime_error("null engine");
    if (runs) *runs = (int64_t)e->eng->speculation_runs();
    if (misses) *misses = (int64_t)e->eng->speculation_misses();
  });
}

int pe_warmup(pe_engine* e, int32_t max_batch, int32_t max_ids, float frames_per_id, const float scales[3],
              const int64_t* sample_ids, int64_t n_sample) {
  return guard([&] {
    if (!e) throw std::runtime_error("null engine");
    e->eng->warmup(max_batch, max_ids, frames_per_id, scales, sample_ids, n_sample);
  });
}

int pe_graph_stats(pe_engine* e, int64_t* cached, int64_t* captures) {
  return guard([&] {
    if (!e) throw std::runtime_error("null engine");
    if (cached) *cached = (int64_t)e->eng->graphs_cached();
    if (captures) *captures = (int64_t)e->eng->graph_captures();
  });
}

const char* pe_policy_describe(void) { return pe::LaunchPolicy::describe(); }

int pe_xcc_pattern(pe_engine* e, int32_t xcc[64], int32_t* period) {
  return guard([&] {
    if (!e) throw std::runtime_error("null engine");
    int P = 0;
    const int* x = e->eng->xcc_pattern(&P);
    if (xcc) for (int i = 0; i < 64; ++i) xcc[i] = x[i];
    if (period) *period = P;
  });
}

int pe_device_pci_bus_id(int device, char* out, int32_t capacity) {
  return guard([&] {
    if (!out || capacity < 16) throw std::runtime_error("pci bus id buffer too small (16 bytes at least)");
    out[0] = 0;
#ifdef PE_EMU
    snprintf(out, (size_t)capacity, "emu:%02d:00.0", device);
#else
    PE_HIP(hipDeviceGetPCIBusId(out, capacity, device));
#endif
  });
}

// ---------------------------------------------------------------------------------------------------------------------
// pe_group_*: one engine / stream / worker thread per device in ONE process (include/piper_hip.h)
// ---------------------------------------------------------------------------------------------------------------------
}  // extern "C"

struct pe_group {
  std::vector<pe_engine*> eng;
  std::vector<int> device;
  std::vector<void*> arena;               // one packed-weight arena per engine, on its device
  // last call, caller order
  std::vector<int32_t> assign;
  std::vector<int64_t> sample_off;
  std::vector<int16_t> pcm;
  std::vector<int32_t> frames;
  bool coalesce = true;                   // PIPER_HIP_GROUP_COALESCE (read at pe_group_create): 0 = every engine takes part in every call
};

namespace {
void group_free(pe_group* g) {
  if (!g) return;
  for (pe_engine* e : g->eng) pe_destroy(e);
  for (size_t i = 0; i < g->arena.size(); ++i)
    if (g->arena[i]) {
      hipSetDevice(g->device[i]);
      hipFree(g->arena[i]);
    }
  delete g;
}

// Longest-first onto the least-loaded engine, ties to the lower index: the same deterministic table as
// piper_amd/dist.py::shard_indices (load = phoneme ids; frames per id vary little within a voice).
std::vector<std::vector<int>> lpt(const int64_t* offsets, int batch, int n) {
  std::vector<int> order(batch);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
    return offsets[a + 1] - offsets[a] > offsets[b + 1] - offsets[b];
  });
  std::vector<std::vector<int>> shard(n);
  std::vector<int64_t> load(n, 0);
  for (int u : order) {
    // least-loaded engine that still has room: an engine call takes at most 4096 utterances (Engine::upload), so many
    // short utterances beside a few long ones spill to the next-least-loaded engine instead of overfilling one
    int best = -1;
    for (int i = 0; i < n; ++i)
      if (shard[i].size() < 4096 && (best < 0 || load[i] < load[best])) best = i;
    if (best < 0) throw std::runtime_error("batch size must be in [1, 4096 per engine]");
    shard[best].push_back(u);
    load[best] += offsets[u + 1] - offsets[u];
  }
  for (auto& s : shard) std::sort(s.begin(), s.end());      // caller order inside a shard
  return shard;
}
}  // namespace

// ---- the weight broadcast of pe_group_create over RCCL (north_star: "RCCL broadcast of the shared voice weights over xGMI").
// librccl is loaded on first use (a process that drives one GPU never pays for it); one communicator over the DISTINCT
// devices of the group, one ncclBroadcast of the packed arena's used prefix from devices[0]. Returns false (with the
// reason in `why`) when the library or a call is unavailable: the caller then copies peer to peer.
#ifndef PE_EMU
#include <dlfcn.h>
namespace {
struct Rccl {
  typedef int (*InitAll)(void**, int, const int*);
  typedef int (*Bcast)(const void*, void*, size_t, int, int, void*, hipStream_t);
  typedef int (*Void0)();
  typedef int (*Destroy)(void*);
  typedef const char* (*ErrStr)(int);
  InitAll init_all = nullptr; Bcast bcast = nullptr; Void0 group_start = nullptr, group_end = nullptr;
  Destroy destroy = nullptr; ErrStr err = nullptr;
  bool ok = false;
  Rccl() {
    // the RCCL that belongs to THIS library's HIP runtime: next to the libamdhip64 this library is linked against (a host
    // process may carry another ROCm copy -- PyTorch bundles one -- whose RCCL cannot take this runtime's device pointers)
    void* h = nullptr;
    Dl_info di;
    if (dladdr((void*)&hipGetDeviceCount, &di) && di.dli_fname) {
      std::string dir(di.dli_fname);
      const size_t slash = dir.rfind('/');
      if (slash != std::string::npos) h = dlopen((dir.substr(0, slash) + "/librccl.so.1").c_str(), RTLD_NOW | RTLD_LOCAL);
    }
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
    init_all = (InitAll)dlsym(h, "ncclCommInitAll");
    bcast = (Bcast)dlsym(h, "ncclBroadcast");
    group_start = (Void0)dlsym(h, "ncclGroupStart");
    group_end = (Void0)dlsym(h, "ncclGroupEnd");
    destroy = (Destroy)dlsym(h, "ncclCommDestroy");
    err = (ErrStr)dlsym(h, "ncclGetErrorString");
    ok = init_all && bcast && group_start && group_end && destroy;
  }
};
// arenas[i] lives on devs[i] (distinct devices, devs[0] = the packing device); `used` bytes travel
bool rccl_broadcast(const std::vector<int>& devs, const std::vector<void*>& arenas, size_t used, std::string& why) {
  static Rccl r;
  if (!r.ok) { why = "librccl not loadable"; return false; }
  const int n = (int)devs.size();
  std::vector<void*> comms(n, nullptr);
  std::vector<hipStream_t> streams(n, nullptr);
  auto fail = [&](const char* what, int rc) {
    why = std::string(what) + ": " + (r.err ? r.err(rc) : "error") + " (" + std::to_string(rc) + ")";
    for (int i = 0; i < n; ++i) {
      if (streams[i]) { hipSetDevice(devs[i]); hipStreamDestroy(streams[i]); }
      if (comms[i]) r.destroy(comms[i]);
    }
    return false;
  };
  int rc = r.init_all(comms.data(), n, devs.data());
  if (rc) return fail("ncclCommInitAll", rc);
  for (int i = 0; i < n; ++i) {
    if (hipSetDevice(devs[i]) != hipSuccess || hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking) != hipSuccess)
      return fail("stream creation", 1);
  }
  if ((rc = r.group_start())) return fail("ncclGroupStart", rc);
  for (int i = 0; i < n; ++i) {
    hipSetDevice(devs[i]);
    if ((rc = r.bcast(arenas[0], arenas[i], used, /*ncclUint8*/ 1, /*root*/ 0, comms[i], streams[i]))) {
      r.group_end();
      return fail("ncclBroadcast", rc);
    }
  }
  if ((rc = r.group_end())) return fail("ncclGroupEnd", rc);
  for (int i = 0; i < n; ++i) {
    hipSetDevice(devs[i]);
    if (hipStreamSynchronize(streams[i]) != hipSuccess) return fail("stream synchronisation", 1);
  }
  for (int i = 0; i < n; ++i) {
    hipSetDevice(devs[i]);
    hipStreamDestroy(streams[i]);
    r.destroy(comms[i]);
  }
  return true;
}
}  // namespace
#endif

static thread_local std::string g_group_bcast = "none";

extern "C" {

// how the last pe_group_create on this thread moved the packed weights between devices: "rccl", "peer-copy (<reason>)",
// "same-device" or "none" (one engine)
const char* pe_group_broadcast_path(void) { return g_group_bcast.c_str(); }

int pe_group_create(const void* blob, size_t nbytes, const int32_t* devices, int32_t n_devices, pe_group** out) {
  pe_group* g = nullptr;
  const int rc = guard([&] {
    if (!blob || !devices || !out || n_devices < 1) throw std::runtime_error("null argument");
    size_t bound = 0;
    {
      pe::WeightSet ws = pe::parse_blob(blob, nbytes, true);
      bound = pe::Engine::arena_bound(ws);
    }
    const size_t header = pe::blob_header_bytes(blob, nbytes);
    g = new pe_group();
    if (const char* t = getenv("PIPER_HIP_GROUP_COALESCE")) g->coalesce = !(t[0] == '0' && !t[1]);
    g->device.assign(devices, devices + n_devices);
    g->arena.assign(n_devices, nullptr);
    for (int i = 0; i < n_devices; ++i) {
      PE_HIP(hipSetDevice(devices[i]));
      PE_HIP(hipMalloc(&g->arena[i], bound));
      pe_engine* e = nullptr;
      // engine 0 parses, packs and uploads; the others only lay their arena out (no weight data is read)
      if (pe_create_in_arena(blob, i == 0 ? nbytes : header, devices[i], g->arena[i], bound, i == 0 ? 0 : 1, &e))
        throw std::runtime_error(g_err);
      g->eng.push_back(e);
    }
    size_t used = 0;
    if (pe_weights_used(g->eng[0], &used)) throw std::runtime_error(g_err);
    for (int i = 1; i < n_devices; ++i) {
      size_t u = 0;
      if (pe_weights_used(g->eng[i], &u)) throw std::runtime_error(g_err);
      if (u != used) throw std::runtime_error("internal: arena layouts differ between devices");
    }
    g_group_bcast = n_devices > 1 ? "same-device" : "none";
    // ---- one engine per DISTINCT device receives the packed arena from devices[0]: ONE RCCL broadcast over xGMI on a
    // communicator made of the distinct devices (PIPER_HIP_GROUP_BCAST=peer skips it; =rccl also takes it for a group on
    // ONE device -- the only way to exercise the collective on a single-GPU box); if RCCL is unavailable: peer copies
    std::vector<int> first;                       // index of the first engine on each distinct device, devices[0] first
    for (int i = 0; i < n_devices; ++i) {
      bool seen = false;
      for (int j : first) seen = seen || devices[j] == devices[i];
      if (!seen) first.push_back(i);
    }
    std::vector<bool> have(n_devices, false);
    have[0] = true;
    const char* mode = getenv("PIPER_HIP_GROUP_BCAST");
    const bool want_rccl = !(mode && std::string(mode) == "peer") && (first.size() > 1 || (mode && std::string(mode) == "rccl"));
#ifndef PE_EMU
    if (want_rccl) {
      std::vector<int> devs;
      std::vector<void*> arenas;
      for (int j : first) { devs.push_back(devices[j]); arenas.push_back(g->arena[j]); }
      std::string why;
      PE_HIP(hipSetDevice(devices[0]));
      PE_HIP(hipDeviceSynchronize());             // engine 0's uploads into its arena are complete
      if (rccl_broadcast(devs, arenas, used, why)) {
        for (int j : first) have[j] = true;
        g_group_bcast = "rccl";
      } else {
        g_group_bcast = "peer-copy (" + why + ")";
      }
    }
#else
    (void)want_rccl;
#endif
    for (int i = 1; i < n_devices; ++i) {
      if (!have[i]) {
        // source: the first engine on the same device if it already has the weights, else engine 0 (peer copy)
        int src = 0;
        for (int j : first)
          if (devices[j] == devices[i] && have[j]) src = j;
        if (devices[i] != devices[src]) {
          int can = 0;
          PE_HIP(hipDeviceCanAccessPeer(&can, devices[i], devices[src]));
          if (can) {
            PE_HIP(hipSetDevice(devices[i]));
            hipDeviceEnablePeerAccess(devices[src], 0);       // "already enabled" is fine
            (void)hipGetLastError();
          }
          PE_HIP(hipMemcpyPeer(g->arena[i], devices[i], g->arena[src], devices[src], used));   // staged by the runtime without P2P
          if (g_group_bcast == "same-device") g_group_bcast = "peer-copy (PIPER_HIP_GROUP_BCAST=peer)";
        } else {
          PE_HIP(hipSetDevice(devices[i]));
          PE_HIP(hipMemcpy(g->arena[i], g->arena[src], used, hipMemcpyDeviceToDevice));
        }
        PE_HIP(hipSetDevice(devices[i]));
        PE_HIP(hipDeviceSynchronize());
        have[i] = true;
      }
      if (pe_arena_ready(g->eng[i])) throw std::runtime_error(g_err);
    }
    *out = g;
  });
  if (rc) {
    const std::string keep = g_err;
    group_free(g);
    g_err = keep;
  }
  return rc;
}

int32_t pe_group_size(pe_group* g) { return g ? (int32_t)g->eng.size() : 0; }

pe_engine* pe_group_engine(pe_group* g, int32_t i) {
  return (g && i >= 0 && i < (int32_t)g->eng.size()) ? g->eng[i] : nullptr;
}

int pe_group_synthesize_batch(pe_group* g, const int64_t* ids, const int64_t* offsets, int32_t batch,
                              const float scales[3], const int64_t* sids, pe_result* result) {
  return guard([&] {
    if (!g || !ids || !offsets || !scales) throw std::runtime_error("null argument");
    const int n = (int)g->eng.size();
    // the offsets are the caller's: check them BEFORE they size a copy (the same rules and messages as Engine::upload,
    // which would only see them after the deal)
    if (batch < 1 || (int64_t)batch > (int64_t)4096 * n)
      throw std::runtime_error("batch size must be in [1, 4096 per engine]");
    if (offsets[0] < 0) throw std::runtime_error("negative phoneme id offset");
    for (int u = 0; u < batch; ++u) {
      const int64_t T = offsets[u + 1] - offsets[u];
      if (T <= 0) throw std::runtime_error("empty phoneme id sequence");
      if (T > 8192) throw std::runtime_error("phoneme id sequence longer than 8192");
    }
    const auto t0 = std::chrono::steady_clock::now();
    // Coalesce instead of stream: engines that share a GPU do not add throughput for small shares -- N B=1 pipelines racing
    // for the HIP runtime's launch path reach 1.5x of one (profiles/r04_notes.md) while ONE call of N utterances runs the
    // batched kernels (8 x 128 ids: 2.1x) -- so a device's utterances go to as few of its engines as 64-utterance shares
    // need; engines on distinct devices always all take part. `active` = the engines the deal runs over.
    std::vector<int> active;
    {
      std::vector<int> seen_dev;
      for (int i = 0; i < n; ++i) {
        const int d = g->device[i];
        if (std::find(seen_dev.begin(), seen_dev.end(), d) != seen_dev.end()) continue;
        seen_dev.push_back(d);
        std::vector<int> on_d;
        for (int k = 0; k < n; ++k)
          if (g->device[k] == d) on_d.push_back(k);
        const long share = ((long)batch * (long)on_d.size() + n - 1) / n;        // utterances this device will see
        const int want = g->coalesce ? (int)std::min<long>((long)on_d.size(), std::max<long>(1, (share + 63) / 64)) : (int)on_d.size();
        for (int k = 0; k < want; ++k) active.push_back(on_d[k]);
      }
      std::sort(active.begin(), active.end());
    }
    const int na = (int)active.size();
    if ((int64_t)batch > (int64_t)4096 * na) throw std::runtime_error("batch size must be in [1, 4096 per engine]");
    std::vector<std::vector<int>> shard(n);
    {
      std::vector<std::vector<int>> sa = lpt(offsets, batch, na);
      for (int k = 0; k < na; ++k) shard[active[k]] = std::move(sa[k]);
    }
    g->assign.assign(batch, 0);
    struct Work {
      std::vector<int64_t> ids, off, sids;
      pe_result res{};
      int rc = 0;
      std::string err;
    };
    std::vector<Work> work(n);
    for (int i = 0; i < n; ++i) {
      Work& w = work[i];
      w.off.push_back(0);
      for (int u : shard[i]) {
        g->assign[u] = i;
        w.ids.insert(w.ids.end(), ids + offsets[u], ids + offsets[u + 1]);
        w.off.push_back((int64_t)w.ids.size());
        if (sids) w.sids.push_back(sids[u]);
      }
    }
    // a shard = upload + device pipeline + int16 PCM to the host; the float waveform stays on the device (the group
    // result carries no `audio`)
    auto run = [&](int i) {
      Work& w = work[i];
      if (shard[i].empty()) return;
      try {
        pe::Engine* e = g->eng[i]->eng;
        e->upload(w.ids.data(), w.off.data(), (int)shard[i].size(), scales, sids ? w.sids.data() : nullptr, nullptr);
        e->run();
        e->download(false, true);
        fill_result(g->eng[i], &w.res, 0.0);
      } catch (const std::exception& ex) {
        w.rc = 1;
        w.err = ex.what();
      } catch (...) {
        w.rc = 1;
        w.err = "unknown error";
      }
    };
#ifdef PE_EMU
    for (int i = 0; i < n; ++i) run(i);      // the emulator is single-threaded
#else
    {
      // joins whatever was started, also when starting a later thread throws (a joinable std::thread that is destroyed
      // terminates the process)
      struct Joiner {
        std::vector<std::thread> th;
        ~Joiner() {
          for (auto& t : th)
            if (t.joinable()) t.join();
        }
      } joiner;
      joiner.th.reserve(n);
      for (int i = 1; i < n; ++i) joiner.th.emplace_back(run, i);
      run(0);
    }
#endif
    for (int i = 0; i < n; ++i)
      if (work[i].rc) throw std::runtime_error("device " + std::to_string(g->device[i]) + ": " + work[i].err);
    // gather in the caller's order
    g->frames.assign(batch, 0);
    g->sample_off.assign(batch + 1, 0);
    for (int i = 0; i < n; ++i)
      for (size_t k = 0; k < shard[i].size(); ++k) g->frames[shard[i][k]] = work[i].res.frames[k];
    std::vector<int64_t> len(batch, 0);
    for (int i = 0; i < n; ++i)
      for (size_t k = 0; k < shard[i].size(); ++k)
        len[shard[i][k]] = work[i].res.sample_offsets[k + 1] - work[i].res.sample_offsets[k];
    for (int u = 0; u < batch; ++u) g->sample_off[u + 1] = g->sample_off[u] + len[u];
    g->pcm.resize((size_t)g->sample_off[batch]);
    for (int i = 0; i < n; ++i)
      for (size_t k = 0; k < shard[i].size(); ++k) {
        const int u = shard[i][k];
        memcpy(g->pcm.data() + g->sample_off[u], work[i].res.pcm + work[i].res.sample_offsets[k], len[u] * sizeof(int16_t));
      }
    if (result) {
      result->batch = batch;
      result->sample_offsets = g->sample_off.data();
      result->audio = nullptr;
      result->pcm = g->pcm.data();
      result->frames = g->frames.data();
      result->infer_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
  });
}

// ---------------------------------------------------------------------------------------------------------------------
// pe_coalescer_*: concurrent single-utterance requests of many caller threads as batched engine calls (include/piper_hip.h)
// ---------------------------------------------------------------------------------------------------------------------
}  // extern "C"

#include <condition_variable>
#include <deque>
#include <mutex>

struct pe_coalescer {
  struct Req {
    const int64_t* ids; int64_t n; float scales[3]; int64_t sid;
    int16_t* pcm = nullptr; int64_t samples = 0; int32_t frames = 0; double secs = 0; int32_t batch = 0;
    int state = 0;            // 0 queued, 1 taken by a leader, 2 done, 3 failed
    std::string err;
  };
  pe_engine* eng;
  int max_batch, max_wait_us;
  std::mutex m;
  std::condition_variable cv;
  std::deque<Req*> q;
  bool busy = false;          // a leader is inside the engine
  int inside = 0;             // callers inside pe_coalescer_synthesize (pe_coalescer_destroy waits for them)
  int64_t calls = 0, requests = 0;
};

extern "C" {

int pe_coalescer_create(pe_engine* e, int32_t max_batch, int32_t max_wait_us, pe_coalescer** out) {
  return guard([&] {
    if (!e || !out) throw std::runtime_error("null argument");
    if (max_batch < 1 || max_batch > 4096) throw std::runtime_error("batch size must be in [1, 4096]");
    auto* c = new pe_coalescer();
    c->eng = e;
    c->max_batch = max_batch;
    c->max_wait_us = max_wait_us < 0 ? 0 : max_wait_us;
    *out = c;
  });
}

void pe_coalescer_destroy(pe_coalescer* c) {
  if (!c) return;
  {
    // requests still queued or a leader inside the engine: wait for them (their stack frames hold pointers into *c)
    std::unique_lock<std::mutex> lk(c->m);
    c->cv.wait(lk, [&] { return !c->busy && c->q.empty() && c->inside == 0; });
  }
  delete c;
}

int pe_coalescer_stats(pe_coalescer* c, int64_t* engine_calls, int64_t* requests) {
  return guard([&] {
    if (!c) throw std::runtime_error("null argument");
    std::lock_guard<std::mutex> lk(c->m);
    if (engine_calls) *engine_calls = c->calls;
    if (requests) *requests = c->requests;
  });
}

int pe_coalescer_synthesize(pe_coalescer* c, const int64_t* ids, int64_t n_ids, const float scales[3], int64_t sid,
                            int16_t** pcm, int64_t* n_samples, int32_t* frames, double* infer_seconds, int32_t* batch_size) {
  return guard([&] {
    if (!c || !ids || !scales || !pcm || !n_samples) throw std::runtime_error("null argument");
    if (n_ids <= 0) throw std::runtime_error("empty phoneme id sequence");
    if (n_ids > 8192) throw std::runtime_error("phoneme id sequence longer than 8192");
    pe_coalescer::Req me;
    me.ids = ids; me.n = n_ids; me.sid = sid;
    memcpy(me.scales, scales, sizeof(me.scales));
    std::unique_lock<std::mutex> lk(c->m);
    // Whatever ends this call -- also an exception between taking the lead and the engine call (std::bad_alloc of the
    // batch vector) -- `me` must leave the queue, a lead must be given up, and requests this thread had taken must fail
    // rather than wait forever: a scope guard, run with the lock held.
    std::vector<pe_coalescer::Req*> take;
    bool leading = false, finished = false;
    struct Cleanup {
      pe_coalescer* c; pe_coalescer::Req* me; std::unique_lock<std::mutex>& lk; std::vector<pe_coalescer::Req*>& take;
      bool& leading; bool& finished;
      ~Cleanup() {
        if (!lk.owns_lock()) lk.lock();
        if (!finished) {
          for (auto it = c->q.begin(); it != c->q.end();) it = (*it == me) ? c->q.erase(it) : it + 1;
          if (leading) {
            for (auto* r : take)
              if (r != me && r->state == 1) { r->err = "the request's batch leader failed"; r->state = 3; }
            c->busy = false;
          }
        }
        --c->inside;
        c->cv.notify_all();
      }
    } cleanup{c, &me, lk, take, leading, finished};
    ++c->inside;
    take.reserve((size_t)c->max_batch);
    c->q.push_back(&me);
    ++c->requests;
    c->cv.notify_all();                         // (a leader collecting its batch counts the queue)
    // follower: wait until a leader has served this request, or until nobody leads and it can lead itself
    while (me.state < 2) {
      if (me.state == 0 && !c->busy) {
        // ---- leader: optionally give concurrent callers max_wait_us to arrive, then take what is queued (requests with
        // the leader's scales, up to max_batch) and run it as ONE engine call
        c->busy = true;
        leading = true;
        take.clear();
        if (c->max_wait_us > 0) {
          const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(c->max_wait_us);
          c->cv.wait_until(lk, until, [&] { return (int)c->q.size() >= c->max_batch; });
        }
        for (auto it = c->q.begin(); it != c->q.end() && (int)take.size() < c->max_batch;) {
          pe_coalescer::Req* r = *it;
          if (r == &me || !memcmp(r->scales, me.scales, sizeof(me.scales))) {
            r->state = 1;
            take.push_back(r);
            it = c->q.erase(it);
          } else {
            ++it;
          }
        }
        ++c->calls;
        lk.unlock();
        std::string err;
        try {
          std::vector<int64_t> cat, off{0}, sids;
          for (auto* r : take) {
            cat.insert(cat.end(), r->ids, r->ids + r->n);
            off.push_back((int64_t)cat.size());
            sids.push_back(r->sid < 0 ? 0 : r->sid);
          }
          const auto t0 = std::chrono::steady_clock::now();
          pe::Engine* e = c->eng->eng;
          e->upload(cat.data(), off.data(), (int)take.size(), me.scales, sids.data(), nullptr);
          e->run();
          e->download(false, true);
          const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
          const auto& so = e->sample_offsets();
          const int16_t* all = e->pcm_host();
          const auto& fr = e->frames_host();
          for (size_t k = 0; k < take.size(); ++k) {
            pe_coalescer::Req* r = take[k];
            r->samples = so[k + 1] - so[k];
            r->pcm = static_cast<int16_t*>(malloc(std::max<size_t>(1, (size_t)r->samples) * sizeof(int16_t)));
            if (!r->pcm) throw std::runtime_error("out of memory");
            memcpy(r->pcm, all + so[k], (size_t)r->samples * sizeof(int16_t));
            r->frames = fr[k];
            r->secs = secs;
            r->batch = (int32_t)take.size();
          }
        } catch (const std::exception& ex) {
          err = ex.what();
        } catch (...) {
          err = "unknown error";
        }
        lk.lock();
        for (auto* r : take) {
          if (!err.empty()) {
            free(r->pcm);
            r->pcm = nullptr;
            r->err = err;
            r->state = 3;
          } else {
            r->state = 2;
          }
        }
        c->busy = false;
        leading = false;
        c->cv.notify_all();                     // followers pick their results up; one of the queued becomes the next leader
        continue;
      }
      c->cv.wait(lk);
    }
    finished = true;                            // (served: `me` left the queue when a leader took it)
    lk.unlock();
    if (me.state == 3) throw std::runtime_error(me.err);
    *pcm = me.pcm;
    *n_samples = me.samples;
    if (frames) *frames = me.frames;
    if (infer_seconds) *infer_seconds = me.secs;
    if (batch_size) *batch_size = me.batch;
  });
}

int pe_group_assignment(pe_group* g, int32_t* engine_index, int64_t capacity) {
  return guard([&] {
    if (!g || !engine_index) throw std::runtime_error("null argument");
    if (capacity < (int64_t)g->assign.size()) throw std::runtime_error("assignment buffer too small");
    memcpy(engine_index, g->assign.data(), g->assign.size() * sizeof(int32_t));
  });
}

void pe_group_destroy(pe_group* g) { group_free(g); }

}  // extern "C"
