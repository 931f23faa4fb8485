// Host engine, life cycle and the synthesis call: construction / destruction, the XCD probe, workspaces, upload, the
// hipGraph cache (capture per shape bucket, LRU), run / finish_run with speculative stage-B sizing, download, streaming, debug
// hooks. Weight packing: engine_pack.cpp; launchers: engine_launch.cpp; the kernel sequences: engine_issue.cpp; which
// kernel form a launch takes: policy.h.
#include "engine_internal.h"

namespace pe {

thread_local long g_launches = 0;
std::shared_mutex g_capture_mu;
thread_local int g_entry_depth = 0;

// PIPER_HIP_DEBUG_POISON: a freshly allocated workspace is filled with NaN bit patterns so that a kernel relying on memory
// it never wrote shows up in the result. The fill runs on the null stream and is asynchronous to the host, while the
// engine's stream does not synchronise with the null stream: without the wait below the fill could land AFTER the first
// copies / launches into the new buffer (seen as NaNs in the first injected noise row, profiles/r05_notes.md, call 31-35).
static void poison(void* p, size_t bytes) {
  PE_HIP(hipMemset(p, 0xFF, bytes));
  PE_HIP(hipDeviceSynchronize());
}

Engine::Engine(const WeightSet& ws, int device, ArenaSpec arena) : device_(device) {
  EntryLock entry_lock;       // allocations / synchronising copies must not overlap another engine's graph capture
  // a constructor that throws does not run the destructor: release what was acquired so far
  try {
    PE_HIP(hipSetDevice(device_));
    skeleton_ = arena.skeleton;
    if (arena.base) {
      if (((uintptr_t)arena.base & 255) != 0) throw std::runtime_error("weight arena must be 256-byte aligned");
      arena_ = static_cast<char*>(arena.base);
      arena_bytes_ = arena.bytes;
    } else {
      if (skeleton_) throw std::runtime_error("a skeleton engine needs a caller-provided arena");
      arena_bytes_ = arena_bound(ws);
      PE_HIP(hipMalloc((void**)&arena_, arena_bytes_));
      arena_owned_ = true;
    }
    init(ws);
  } catch (...) {
    free_all();
    throw;
  }
}

Engine::~Engine() { free_all(); }

void Engine::arena_ready() {
  EntryLock entry_lock;
  PE_HIP(hipSetDevice(device_));
  float m = 0.f, lg = 0.f;
  PE_HIP(hipMemcpy(&m, ea_dev_m_, sizeof(float), hipMemcpyDeviceToHost));
  PE_HIP(hipMemcpy(&lg, ea_dev_logs_, sizeof(float), hipMemcpyDeviceToHost));
  ea_m0_ = m;
  ea_es0_ = std::exp(-lg);
  skeleton_ = false;
}

void Engine::free_all() {
  EntryLock entry_lock;
  if (stream_) hipStreamSynchronize(stream_);
  drop_graphs();
  for (void* p : owned_) hipFree(p);
  owned_.clear();
  if (arena_owned_ && arena_) hipFree(arena_);
  arena_ = nullptr;
  if (wsA_) hipFree(wsA_);
  if (wsB_) hipFree(wsB_);
  if (h_audio_) hipHostFree(h_audio_);
  if (h_pcm_) hipHostFree(h_pcm_);
  if (h_pcm_zc_) hipHostFree(h_pcm_zc_);
  if (h_frames_) hipHostFree(h_frames_);
  if (h_in_) hipHostFree(h_in_);
  h_in_ = nullptr; h_in_cap_ = 0;
  if (ev0_) hipEventDestroy(ev0_);
  if (ev1_) hipEventDestroy(ev1_);
  for (auto& k : kev_) { hipEventDestroy(k.a); hipEventDestroy(k.b); }
  kev_.clear();
  for (hipEvent_t e : ev_pool_) hipEventDestroy(e);
  ev_pool_.clear();
  for (float*& p : side_) { if (p) hipFree(p); p = nullptr; }
  if (ffn_parts_) { hipFree(ffn_parts_); ffn_parts_ = nullptr; }
  if (stream_) hipStreamDestroy(stream_);
  wsA_ = wsB_ = nullptr; h_audio_ = nullptr; h_pcm_ = nullptr; h_pcm_zc_ = nullptr; h_frames_ = nullptr;
  ev0_ = ev1_ = nullptr; stream_ = nullptr;
}

// Which XCD runs which workgroup of a small 1-D launch (kernels/glue.h xcc_probe_kernel). The 4-column kernels hand out
// column tiles so that one XCD owns a contiguous run of them (col4.h c4_tile); that needs the dispatch to be a
// round-robin over P XCDs -- workgroup i on XCD pattern[i mod P], the first P all different -- which is what this checks.
// Anything else (PIPER_HIP_XCD=0 forces it): period 0, tiles in workgroup order.
void Engine::probe_xcds() {
  int* d = nullptr;
  PE_HIP(hipMalloc((void**)&d, 64 * sizeof(int)));
  PE_HIP(hipMemsetAsync(d, 0xff, 64 * sizeof(int), stream_));
  launch::xcc_probe(stream_, d);
  PE_HIP(hipMemcpyAsync(xcc_of_, d, 64 * sizeof(int), hipMemcpyDeviceToHost, stream_));
  PE_HIP(hipStreamSynchronize(stream_));
  PE_HIP(hipFree(d));
  int P = 0;
  for (int c = 1; c <= 32 && !P; ++c) {           // smallest period with pairwise different ids inside it
    bool ok = true;
    for (int i = 0; i < c && ok; ++i)
      for (int j = 0; j < i && ok; ++j) ok = xcc_of_[i] != xcc_of_[j];
    for (int i = c; i < 64 && ok; ++i) ok = xcc_of_[i] == xcc_of_[i - c];
    if (ok && c > 1 && xcc_of_[c] == xcc_of_[0]) P = c;
  }
  xcd_period_ = P;
  if (pol_.xcd >= 0) xcd_period_ = (int)pol_.xcd;       // PIPER_HIP_XCD (A/B, tests): 0 = tiles in workgroup order
}

// ------------------------------------------------------------------------------------------------
// workspaces
// ------------------------------------------------------------------------------------------------

struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(char* b) : base(b) {}
  template <class T> T* take(size_t n) {
    off = (off + 255) / 256 * 256;
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
};

// Workspace policy. Capacities only grow while the grown block fits the budget (a third of the device's memory per
// stage): utterance count and padded length are separate capacities, so a huge batch of short texts followed by one long
// text would otherwise ask for their product. A call whose grown capacities do not fit gets a block sized for exactly
// that call (capacities may shrink); a call that does not fit by itself is an error, raised before anything changes.
size_t Engine::ws_budget() {
  if (!ws_budget_) {
    if (pol_.ws_budget_mb > 0) {
      ws_budget_ = (size_t)pol_.ws_budget_mb << 20;
    } else {
      size_t fr = 0, tot = 0;
      PE_HIP(hipMemGetInfo(&fr, &tot));
      ws_budget_ = std::max<size_t>(tot / 3, (size_t)1 << 30);
    }
  }
  return ws_budget_;
}

void Engine::ensure_stage_a(int B, int Tmax) {
  if (!ffn_parts_ && H_ == 192 && FC_ % 48 == 0 && FC_ / 48 <= 16 && !enc_.empty() && enc_[0].f1p) {
    // partial outputs of the fused small-call FFN (kernels/ffn.h): [utterance][slice][192][columns], once
    PE_HIP(hipStreamSynchronize(stream_));
    const size_t fbytes = (size_t)(FC_ / 48) * H_ * LaunchPolicy::ffn_max_cols * sizeof(float);
    PE_HIP(hipMalloc((void**)&ffn_parts_, fbytes));
    if (pol_.debug_poison) poison(ffn_parts_, fbytes);
  }
  const int Ts = rup(Tmax, 128);    // row strides are multiples of 128 columns (conv epilogue relies on it)
  auto carve = [&](char* base, size_t Bc, size_t T) -> size_t {
    Carver c(base);
    // input block: [rng 4 x u64 | lengths Bc | speaker ids Bc | ids Bc x T] (contiguous, copied as one piece by upload())
    d_in_ = c.take<char>(32 + (2 * Bc + Bc * T) * sizeof(int));
    d_rng_ = reinterpret_cast<unsigned long long*>(d_in_);
    d_tlens_ = reinterpret_cast<int*>(d_in_ + 32);
    d_sids_ = d_tlens_ + Bc;
    d_ids_ = d_sids_ + Bc;
    in_bytes_ = 32 + (2 * Bc + Bc * T) * sizeof(int);
    d_dur_ = c.take<int>(Bc * T);
    d_cum_ = c.take<int>(Bc * T);
    d_frames_ = c.take<int>(Bc);
    d_framesc_ = c.take<int>(Bc);
    absmax_ = c.take<unsigned>(Bc);
    x_ = c.take<float>(Bc * H_ * T);
    y_ = c.take<float>(Bc * H_ * T);
    qkv_ = c.take<float>(Bc * 3 * H_ * T);
    att_ = c.take<float>(Bc * H_ * T);
    kT_ = c.take<float>(Bc * H_ * T);            // K as [utterance][channel quad][column][4] and
    vQ_ = c.take<float>(Bc * H_ * T);            // V as [utterance][column quad][H][4]: attn4_kernel's operands (kernels/attn4.h)
    ffh_ = c.take<float>(Bc * FC_ * T);
    stats_ = c.take<float>(Bc * 2 * C_ * T);
    xg_ = c.take<float>(Bc * H_ * T);
    dh_ = c.take<float>(Bc * H_ * T);
    dy_ = c.take<float>(Bc * H_ * T);
    dy2_ = c.take<float>(Bc * H_ * T);
    hproj_ = c.take<float>(Bc * 32 * T);
    z2_ = c.take<float>(Bc * 2 * T);
    logw_ = c.take<float>(Bc * T);
    noise_w_ = c.take<float>(Bc * 2 * T);
    cond_ = c.take<float>(Bc * (size_t)std::max(cond_bs_, 1));
    // utterances whose attention score slab does not fit LDS: [utterance][head][query block][32][SP] in global memory
    att_s_ = attn_scores_global((int)T) ? c.take<float>(Bc * nh_ * (size_t)rup((int)T, ATT_QB) * (rup((int)T, 64) + 1)) : nullptr;
    return c.off + 256;
  };
  // (growth re-creates every graph: grow the id capacity by at least half, so that texts of slowly increasing length
  // cost a few re-creations, not one per 128 ids)
  size_t nB = std::max<size_t>(B, capA_B_), nT = capA_T_;
  if ((size_t)Ts > capA_T_) nT = std::max<size_t>(Ts, capA_T_ ? rup((int)(capA_T_ + capA_T_ / 2), 128) : 0);
  if (nB != capA_B_ || nT != capA_T_ || !wsA_) {
    const size_t budget = ws_budget();
    if (carve(nullptr, nB, nT) > budget) { nB = B; nT = Ts; }          // exactly this call
    const size_t exact = carve(nullptr, B, Ts);
    if (exact > budget) {
      carve(wsA_, capA_B_, capA_T_);                                      // (the probes above moved the pointers)
      throw std::runtime_error("call too large: " + std::to_string(B) + " utterances padded to " + std::to_string(Ts) +
                               " ids need " + std::to_string(exact >> 20) + " MiB of text-encoder workspace (budget " +
                               std::to_string(budget >> 20) + " MiB)");
    }
    PE_HIP(hipStreamSynchronize(stream_));
    drop_graphs();
    if (wsA_) { PE_HIP(hipFree(wsA_)); wsA_ = nullptr; }
    if (wsB_) { PE_HIP(hipFree(wsB_)); wsB_ = nullptr; }                  // stage-B pointers sit in the dropped graphs
    capA_B_ = capA_T_ = 0; capB_B_ = capB_F_ = 0;
    void* blk = nullptr;
    size_t bytes = carve(nullptr, nB, nT);
    if (hipMalloc(&blk, bytes) != hipSuccess) {
      (void)hipGetLastError();
      nB = B; nT = Ts; bytes = exact; blk = nullptr;
      if (hipMalloc(&blk, bytes) != hipSuccess) {
        (void)hipGetLastError();
        carve(nullptr, 0, 0);
        throw std::runtime_error("out of device memory: " + std::to_string(bytes >> 20) + " MiB of text-encoder workspace");
      }
    }
    wsA_ = static_cast<char*>(blk); wsA_bytes_ = bytes; capA_B_ = nB; capA_T_ = nT;
    if (pol_.debug_poison) poison(wsA_, bytes);
    carve(wsA_, capA_B_, capA_T_);
    PE_HIP(hipMemsetAsync(d_in_, 0, 32, stream_));        // generator state: "no upload ingested yet" (embed_kernel)
    if (h_in_cap_ < in_bytes_) {
      if (h_in_) PE_HIP(hipHostFree(h_in_));
      h_in_cap_ = in_bytes_;
      PE_HIP(hipHostMalloc((void**)&h_in_, h_in_cap_));
    }
  }
  Ts_ = (int)capA_T_;
  carve(wsA_, capA_B_, capA_T_);
}

void Engine::ensure_stage_b(int Fmax, int batch) {
  const int Fs = rup(Fmax, 128);
  const size_t Bnow = (size_t)std::max(batch > 0 ? batch : B_, 1);      // (warmup sizes for ITS batch, not the last call's)
  auto hmax_of = [&](size_t F) {        // largest [channels x length] activation of the generator
    size_t hm = (size_t)U_ * F, L = F;
    for (auto& st : ups_) { L *= st.rate; hm = std::max(hm, (size_t)st.ch * L); }
    return hm;
  };
  // per-utterance activations are addressed with 32-bit byte offsets (buffer descriptors)
  if (hmax_of(Fs) * sizeof(float) >= (size_t)1 << 31 || (size_t)3 * H_ * Fs * sizeof(float) >= (size_t)1 << 31)
    throw std::runtime_error("utterance too long: a per-utterance activation would exceed 2 GiB");
  auto carve = [&](char* base, size_t Bc, size_t F) -> size_t {
    Carver c(base);
    const size_t hmax = hmax_of(F), S = F * (size_t)hop_;
    zp_ = c.take<float>(Bc * C_ * F);
    fh_ = c.take<float>(Bc * H_ * F);
    facts_ = c.take<float>(Bc * H_ * F);
    fskip_ = c.take<float>(Bc * H_ * F);
    noise_z_ = c.take<float>(Bc * C_ * F);
    for (int i = 0; i < 5; ++i) hb_[i] = c.take<float>(Bc * hmax);
    zwin_ = c.take<float>((size_t)C_ * F);
    zp_keep_ = pol_.debug_keep ? c.take<float>(Bc * C_ * F) : nullptr;
    d_win_ = c.take<int>(4);
    audio_ = c.take<float>(Bc * S);
    pcm_ = c.take<int16_t>(Bc * S);
    return c.off + 256;
  };
  // the batch capacity of this stage follows stage A's while that fits; under memory pressure it is this call's batch
  size_t nB = std::max(capB_B_, std::max(Bnow, capA_B_)), nF = capB_F_;
  if ((size_t)Fs > capB_F_) nF = std::max<size_t>(Fs, capB_F_ ? rup((int)(capB_F_ + capB_F_ / 2), 128) : 0);
  // (a grown frame capacity past the 2 GiB descriptor range falls back to the exact one)
  if (hmax_of(nF) * sizeof(float) >= (size_t)1 << 31 || (size_t)3 * H_ * nF * sizeof(float) >= (size_t)1 << 31) nF = Fs;
  // (after an exact-size fallback under memory pressure capB_B_ is below stage A's capacity: a block that already holds
  // this call stays -- re-entering here would free and re-allocate it, and drop every graph, on each call)
  const bool fits = capB_exact_ && wsB_ && Bnow <= capB_B_ && (size_t)Fs <= capB_F_;
  if (!fits && (nB != capB_B_ || nF != capB_F_ || !wsB_)) {
    const size_t budget = ws_budget();
    capB_exact_ = false;
    if (carve(nullptr, nB, nF) > budget) { nB = Bnow; nF = Fs; capB_exact_ = true; }
    const size_t exact = carve(nullptr, Bnow, Fs);
    if (exact > budget) {
      carve(wsB_, capB_B_, capB_F_);
      throw std::runtime_error("call too large: " + std::to_string(Bnow) + " utterances of up to " + std::to_string(Fs) +
                               " frames need " + std::to_string(exact >> 20) + " MiB of vocoder workspace (budget " +
                               std::to_string(budget >> 20) + " MiB)");
    }
    PE_HIP(hipStreamSynchronize(stream_));
    drop_graphs();
    if (wsB_) { PE_HIP(hipFree(wsB_)); wsB_ = nullptr; }
    capB_B_ = capB_F_ = 0;
    void* blk = nullptr;
    size_t bytes = carve(nullptr, nB, nF);
    if (hipMalloc(&blk, bytes) != hipSuccess) {
      (void)hipGetLastError();
      nB = Bnow; nF = Fs; bytes = exact; blk = nullptr; capB_exact_ = true;
      if (hipMalloc(&blk, bytes) != hipSuccess) {
        (void)hipGetLastError();
        carve(nullptr, 0, 0);
        throw std::runtime_error("out of device memory: " + std::to_string(bytes >> 20) + " MiB of vocoder workspace");
      }
    }
    wsB_ = static_cast<char*>(blk); wsB_bytes_ = bytes; capB_B_ = nB; capB_F_ = nF;
    if (pol_.debug_poison) poison(wsB_, bytes);
  }
  Fs_ = (int)capB_F_;
  Ss_ = (long)capB_F_ * hop_;
  carve(wsB_, capB_B_, capB_F_);
  const size_t Bc = capB_B_, hmax = hmax_of(capB_F_);
  // zero-copy PCM: room for every utterance of the batch capacity, up to 256 MiB of pinned memory (beyond: copies)
  const size_t zc_want = Bc * (size_t)Ss_;
  if (pol_.pcm_zc && zc_want * sizeof(int16_t) <= ((size_t)256 << 20) && h_pcm_zc_cap_ < zc_want) {
    PE_HIP(hipStreamSynchronize(stream_));
    drop_graphs();                                     // the pointer is a kernel argument inside the graphs
    if (h_pcm_zc_) PE_HIP(hipHostFree(h_pcm_zc_));
    h_pcm_zc_cap_ = zc_want;
    PE_HIP(hipHostMalloc((void**)&h_pcm_zc_, h_pcm_zc_cap_ * sizeof(int16_t)));
  }
  // per-resblock buffers of the grouped sibling schedule (one-utterance calls, first generator stage): allocated
  // here, outside any graph capture; the schedule only applies below 700 64x64 blocks per stage
  const size_t want = std::min<size_t>(Bc * hmax, (size_t)(pol_.group_tiled ? LaunchPolicy::group_tiled_max_blocks : LaunchPolicy::group_max_blocks64) * 4096);
  if ((pol_.group_mrf || pol_.group_tiled) && side_floats_ < want) {
    PE_HIP(hipStreamSynchronize(stream_));
    drop_graphs();
    for (float*& sp : side_) { if (sp) PE_HIP(hipFree(sp)); sp = nullptr; }
    side_floats_ = 0;                  // (the capacity is published only once EVERY buffer exists: a failed allocation must not
                                       // leave null pointers behind a capacity that says they are there)
    for (float*& sp : side_) {
      if (hipMalloc((void**)&sp, want * sizeof(float)) != hipSuccess) {
        (void)hipGetLastError();
        for (float*& q : side_) { if (q) hipFree(q); q = nullptr; }
        throw std::runtime_error("out of device memory: " + std::to_string((want * sizeof(float) * 9) >> 20) +
                                 " MiB of per-resblock buffers");
      }
      if (pol_.debug_poison) poison(sp, want * sizeof(float));
    }
    side_floats_ = want;
  }
}

// ------------------------------------------------------------------------------------------------
// the synthesis call
// ------------------------------------------------------------------------------------------------

void Engine::upload(const int64_t* ids, const int64_t* offsets, int B, const float scales[3],
                    const int64_t* sids, const NoiseIn* noise) {
  EntryLock entry_lock;
  if (B <= 0 || B > 4096) throw std::runtime_error("batch size must be in [1, 4096]");
  PE_HIP(hipSetDevice(device_));
  B_ = B;
  id_off_.assign(offsets, offsets + B + 1);
  tlens_h_.resize(B);
  int Tmax = 0;
  for (int b = 0; b < B; ++b) {
    const int64_t T = offsets[b + 1] - offsets[b];
    if (T <= 0) throw std::runtime_error("empty phoneme id sequence");
    if (T > 8192) throw std::runtime_error("phoneme id sequence longer than 8192");
    tlens_h_[b] = (int)T;
    Tmax = std::max(Tmax, (int)T);
  }
  Tmax_ = Tmax;
  // a speculative run whose results were never fetched may still be reading the pinned input block (zero-copy ids)
  if (spec_pending_) { PE_HIP(hipStreamSynchronize(stream_)); spec_pending_ = false; }
  ensure_stage_a(B, Tmax);
  const int Ts = Ts_;
  // the pinned mirror of the input block; a copy of the previous call that might still read it ended with that call's
  // final synchronisation (an abandoned upload is simply overwritten)
  const size_t Bc = capA_B_;
  unsigned long long* hr = reinterpret_cast<unsigned long long*>(h_in_);
  int* htl = reinterpret_cast<int*>(h_in_ + 32);
  int* hsid = htl + Bc;
  int* hid = hsid + Bc;
  for (int b = 0; b < B; ++b) {
    int* row = hid + (size_t)b * Ts;
    const int64_t* src = ids + offsets[b];
    const int T = tlens_h_[b];
    for (int t = 0; t < T; ++t) {
      const int64_t id = src[t];
      if (id < 0 || id >= arch_[A_NVOCAB])
        throw std::runtime_error("phoneme id " + std::to_string(id) + " outside [0, num_symbols)");
      row[t] = (int)id;
    }
    memset(row + T, 0, (size_t)(Ts - T) * sizeof(int));
    htl[b] = T;
    hsid[b] = 0;
  }
  if (nspk_ > 1)
    for (int b = 0; b < B; ++b) {
      const int64_t sp = sids ? sids[b] : 0;
      if (sp < 0 || sp >= nspk_) throw std::runtime_error("speaker id outside [0, num_speakers)");
      hsid[b] = (int)sp;
    }
  // {seed, runs so far, serial of this upload}: the first kernel of every run() advances the counter on the device
  // (embed_kernel). Short calls enqueue no copy at all: embed_kernel reads the pinned block in place and publishes the
  // lengths / speaker ids / generator state to device memory for the kernels behind it (the serial tells it a replay
  // without a new upload from a fresh one).
  hr[0] = seed_; hr[1] = call_; hr[2] = ++upload_serial_; hr[3] = 0;
  ids_zc_ = pol_.ids_from_host((long)B * Ts);
  if (!ids_zc_)
    PE_HIP(hipMemcpyAsync(d_in_, h_in_, 32 + (2 * Bc + (size_t)B * Ts) * sizeof(int), hipMemcpyHostToDevice, stream_));
  scales_[0] = scales[0]; scales_[1] = scales[1]; scales_[2] = scales[2];
  have_noise_w_ = noise && noise->noise_w;
  have_noise_z_ = noise && noise->noise_z;
  h_noise_z_ = have_noise_z_ ? noise->noise_z : nullptr;
  h_noise_z_stride_ = have_noise_z_ ? noise->z_stride : 0;
  if (have_noise_w_) {      // injected duration noise (parity tests): pageable staging, so wait for the copy
    if (noise->w_stride < Tmax) throw std::runtime_error("noise_w stride shorter than the longest utterance");
    std::vector<float> nb((size_t)B * 2 * Ts, 0.f);
    for (int b = 0; b < B; ++b)
      for (int c = 0; c < 2; ++c)
        memcpy(&nb[((size_t)b * 2 + c) * Ts], noise->noise_w + ((size_t)b * 2 + c) * noise->w_stride,
               tlens_h_[b] * sizeof(float));
    PE_HIP(hipMemcpyAsync(noise_w_, nb.data(), nb.size() * sizeof(float), hipMemcpyHostToDevice, stream_));
    PE_HIP(hipStreamSynchronize(stream_));
  }
}

// hipGraph cache: the kernel sequence of a stage is captured once per shape bucket and replayed;
// one utterance is ~160 short launches, which would otherwise be bound by host launch rate.
void Engine::run_stage(char which, const std::string& key) {
  const long l0 = g_launches;
#ifndef PE_EMU
  if (use_graphs_ && !prof_on_) {
    auto hit = graph_of_.find(key);
    if (hit == graph_of_.end()) {
      hipGraph_t g = nullptr;
      hipGraphExec_t ex = nullptr;
      // exclusive: no other engine of this process is inside a HIP call while this one captures (see g_capture_mu)
      g_capture_mu.unlock_shared();
      g_capture_mu.lock();
      try {
        PE_HIP(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal));
        try {
          dispatch_stage(which);
        } catch (...) {
          hipStreamEndCapture(stream_, &g);
          if (g) hipGraphDestroy(g);
          throw;
        }
        PE_HIP(hipStreamEndCapture(stream_, &g));
        PE_HIP(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        PE_HIP(hipGraphDestroy(g));
      } catch (...) {
        g_capture_mu.unlock();
        g_capture_mu.lock_shared();
        throw;
      }
      g_capture_mu.unlock();
      g_capture_mu.lock_shared();
      ++graph_captures_;
      if (graphs_.size() >= (size_t)pol_.graphs) {          // evict the least recently used graph only
        // (it may still be executing: destroying the exec object of a launched graph is deferred by the runtime until
        // the launch completes; the stream is in order, so nothing of this engine runs concurrently with it anyway)
        PE_HIP(hipStreamSynchronize(stream_));
        hipGraphExecDestroy((hipGraphExec_t)graphs_.front().exec);
        graph_of_.erase(graphs_.front().key);
        graphs_.pop_front();
      }
      graphs_.push_back(GraphEntry{key, (void*)ex, g_launches - l0});
      hit = graph_of_.emplace(key, std::prev(graphs_.end())).first;
    } else if (std::next(hit->second) != graphs_.end()) {
      graphs_.splice(graphs_.end(), graphs_, hit->second);      // most recently used last; iterators stay valid
    }
    PE_HIP(hipGraphLaunch((hipGraphExec_t)hit->second->exec, stream_));
    run_launches_ += hit->second->launches;
    return;
  }
#endif
  (void)key;
  dispatch_stage(which);
  run_launches_ += g_launches - l0;
}

void Engine::dispatch_stage(char which) {
  switch (which) {
    case 'A': issue_stage_a(); break;
    case 'B': issue_stage_b(); break;
    case 'C': issue_stage_a(); issue_stage_b(); break;
    case 'F': issue_flow(); break;
    default: issue_window(); break;
  }
}

void Engine::drop_graphs() {
#ifndef PE_EMU
  for (auto& e : graphs_) hipGraphExecDestroy((hipGraphExec_t)e.exec);
#endif
  graphs_.clear();
  graph_of_.clear();
}

// Shape buckets (engine.h). Steps of 32 ids up to 512, then 8 per octave; steps of 64 frames up to 1024, then 16 per octave.
int Engine::id_bucket(int T) {
  int g = rup(std::max(T, 1), 32);
  if (g > 512) {
    int step = 64;
    while (step * 16 <= g) step *= 2;             // step = (largest power of two <= g) / 8
    g = rup(T, step);
  }
  return g;
}
int Engine::frame_bucket(int F) {
  int g = rup(std::max(F, 1), 64);
  if (g > 1024) {
    int step = 64;
    while (step * 32 <= g) step *= 2;             // step = (largest power of two <= g) / 16
    g = rup(F, step);
  }
  return g;
}

void Engine::warmup(int max_batch, int max_ids, float frames_per_id, const float* scales_in, const int64_t* sample, int64_t n_sample) {
  EntryLock entry_lock;
  if (max_batch < 1 || max_batch > 4096) throw std::runtime_error("batch size must be in [1, 4096]");
  if (max_ids < 1 || max_ids > 8192) throw std::runtime_error("phoneme id sequence longer than 8192");
  PE_HIP(hipSetDevice(device_));
  if (!(frames_per_id > 0.f)) frames_per_id = 8.f;
  const long fmax = std::min<long>(MAX_FRAMES, (long)std::ceil((double)frames_per_id * max_ids) + 1);
  ensure_stage_a(max_batch, max_ids);
  ensure_stage_b(frame_bucket((int)fmax), max_batch);
  if (!sample || n_sample < 1) return;
  // the single-utterance graphs of every id bucket up to max_ids: the sample cut / tiled to the bucket length, twice --
  // the first call of a bucket runs as two graphs around the frame-count read-back, the second as the one speculative
  // graph later calls replay (engine.h)
  const float scales[3] = {scales_in ? scales_in[0] : scales_[0], scales_in ? scales_in[1] : scales_[1], scales_in ? scales_in[2] : scales_[2]};
  std::vector<int64_t> ids;
  int prev = 0;
  for (int T = 32; prev < max_ids; T = id_bucket(T + 1)) {
    const int len = std::min(T, max_ids);
    ids.resize(len);
    for (int t = 0; t < len; ++t) ids[t] = sample[t % n_sample];
    const int64_t off[2] = {0, len};
    for (int rep = 0; rep < 2; ++rep) {
      upload(ids.data(), off, 1, scales, nullptr, nullptr);
      run();
      finish_run();
    }
    // ... and the frame buckets next to the one the sample landed in: other texts of this length differ by a few per cent
    // in frames per id (each call below replays or captures the whole-utterance graph for a forced guess; a guess that
    // is too small for the sample costs one re-run of the second half, like any miss -- not counted)
    const int fg = frame_bucket(Fmax_);
    const long runs0 = spec_runs_, miss0 = spec_misses_;
    const float margin0 = spec_margin_;
    const int streak0 = spec_hit_streak_, rm0 = spec_recent_misses_, rr0 = spec_recent_runs_, cd0 = spec_cooldown_;
    for (int d = -1; d <= 1; ++d) {
      spec_fg_force_ = frame_bucket(std::max(1, fg + d * 64));
      upload(ids.data(), off, 1, scales, nullptr, nullptr);
      run();
      finish_run();
    }
    spec_fg_force_ = 0;
    spec_runs_ = runs0; spec_misses_ = miss0; spec_margin_ = margin0;
    spec_hit_streak_ = streak0; spec_recent_misses_ = rm0; spec_recent_runs_ = rr0; spec_cooldown_ = cd0;
    prev = len;
  }
  PE_HIP(hipStreamSynchronize(stream_));
}

void Engine::run() {
  EntryLock entry_lock;
  PE_HIP(hipSetDevice(device_));
  const int B = B_;
  spec_pending_ = false;
  Tg_ = std::min(id_bucket(Tmax_), Ts_);
  run_launches_ = 0;
  // speculative sizing of stage B from the previous run's frames-per-id ratio (see engine.h)
  bool spec = pol_.speculate(B) && last_ratio_ > 0.f && !have_noise_z_ && use_graphs_ && !prof_on_;
  if (spec && spec_cooldown_ > 0) { --spec_cooldown_; spec = false; }
  int fguess = 0;
  if (spec) {
    fguess = spec_fg_force_ ? spec_fg_force_ : frame_bucket((int)std::ceil(last_ratio_ * spec_margin_ * (float)Tmax_) + 1);
    if (fguess > MAX_FRAMES) spec = false;
  }
  if (spec) {
    // before stage A is enqueued: growing the workspace drops every graph. The guess carries a margin and a bucket
    // rounding: when IT does not fit the workspace budget / the 2 GiB descriptor range the real frame count still may, so
    // the call falls back to the two-graph form and only ensure_stage_b on the real counts can fail it
    try {
      ensure_stage_b(fguess);
    } catch (const std::runtime_error& ex) {
      // only the sizing conditions fall back; a HIP error or an allocation failure is the call's error
      const std::string what = ex.what();
      if (what.compare(0, 14, "call too large") != 0 && what.compare(0, 18, "utterance too long") != 0) throw;
      spec = false;
    }
  }
  char key[200];
  if (spec) {
    // the whole utterance -- text encoder to int16 -- as ONE graph: stage B is issued right behind stage A for the
    // guessed frame bucket (the kernels read the real frame counts from device memory, clamped to the bucket)
    Fg_ = std::min(fguess, Fs_);
    // What the cost models see while the graph is issued (window geometry of the stage kernels, column thresholds): the
    // EXPECTED frame counts -- ratio x ids, without the safety margin and the bucket rounding that size the grids. With
    // the bucket capacity here a 417-frame utterance in the 512-frame bucket got the last stage's two-round geometry
    // (mrf_kernel<32,3,1>: 125.6 us per replay) instead of the one-round one its real length takes (<32,4,1>: 85.7 us;
    // profiles/r04_notes.md). Grids and clamps are sized by Fg_; the real counts arrive in finish_run().
    frames_h_.resize(B);
    for (int b = 0; b < B; ++b)
      frames_h_[b] = pol_.spec_expect ? std::min(Fg_, std::max(1, (int)std::ceil(last_ratio_ * (float)tlens_h_[b]))) : Fg_;
    lens_b_ = d_framesc_;
    snprintf(key, sizeof(key), "C|%d|%d|%d|%a|%a|%d|%d|%d|%a", B, Tg_, Ts_, scales_[1], scales_[2], (int)have_noise_w_,
             Fs_, Fg_, scales_[0]);
    fold_dur_ = Tg_ <= REG_MAXT;              // (part of what graph 'C' is: a fixed function of its key)
    try {
      run_stage('C', key);
    } catch (...) {
      fold_dur_ = false;
      throw;
    }
    fold_dur_ = false;
    ++call_;                                  // mirrors the device-side counter bump of this run
    spec_pending_ = true;
    spec_fg_ = Fg_;
    ++spec_runs_;
    return;
  }
  snprintf(key, sizeof(key), "A|%d|%d|%d|%a|%a|%d|%d", B, Tg_, Ts_, scales_[1], scales_[2], (int)have_noise_w_, Fs_);
  run_stage('A', key);
  ++call_;                                    // mirrors the device-side counter bump of this run
  PE_HIP(hipStreamSynchronize(stream_));      // the only data-dependent shape: F
  finish_stage_b_sizes();
  ensure_stage_b(frame_bucket(Fmax_));
  Fg_ = std::min(frame_bucket(Fmax_), Fs_);
  lens_b_ = d_frames_;
  if (have_noise_z_) {
    const long l0 = g_launches;
    issue_stage_b();                           // host-injected noise (tests): not graph-captured
    run_launches_ += g_launches - l0;
  } else {
    snprintf(key, sizeof(key), "B|%d|%d|%d|%d|%a", B, Fg_, Fs_, Ts_, scales_[0]);
    run_stage('B', key);
  }
}

// Host view of the frame counts stage A produced (the stream is synchronised): frames, sample offsets, the ratio the
// next run's guess is made from.
void Engine::finish_stage_b_sizes() {
  const int B = B_;
#ifdef PE_EMU
  if (const char* pf = getenv("EMU_PLAN_FRAMES"))      // emulator plan-only mode (tests/emu): frames are not computed
    for (int b = 0; b < B; ++b) h_frames_[b] = atoi(pf);
#endif
  frames_h_.assign(h_frames_, h_frames_ + B);
  int Fmax = 1;
  float ratio = 0.f;
  for (int b = 0; b < B; ++b) {
    Fmax = std::max(Fmax, frames_h_[b]);
    ratio = std::max(ratio, (float)frames_h_[b] / (float)tlens_h_[b]);
  }
  if (Fmax > MAX_FRAMES)
    throw std::runtime_error("utterance too long: more than " + std::to_string(MAX_FRAMES) + " spectrogram frames "
                             "(check length_scale)");
  Fmax_ = Fmax;
  // decaying maximum: one long-winded utterance keeps the estimate up for a while, a lasting change of voice / scales
  // is followed within ~50 calls
  last_ratio_ = std::max(ratio, last_ratio_ * 0.98f + ratio * 0.02f);
  sample_off_.assign(B + 1, 0);
  for (int b = 0; b < B; ++b) sample_off_[b + 1] = sample_off_[b] + (int64_t)frames_h_[b] * hop_;
}

bool Engine::finish_run() {
  EntryLock entry_lock;
  if (!spec_pending_) return true;
  spec_pending_ = false;
  PE_HIP(hipStreamSynchronize(stream_));
  finish_stage_b_sizes();
  if (++spec_recent_runs_ >= 16) { spec_recent_runs_ = 0; spec_recent_misses_ = 0; }
  if (Fmax_ <= spec_fg_) {                     // the guessed bucket covered every utterance: the results stand
    if (++spec_hit_streak_ >= 32) { spec_hit_streak_ = 0; spec_margin_ = std::max(1.10f, spec_margin_ / 1.05f); }
    return true;
  }
  ++spec_misses_;
  spec_hit_streak_ = 0;
  spec_margin_ = std::min(1.5f, spec_margin_ * 1.15f);
  if (++spec_recent_misses_ >= 4) { spec_recent_misses_ = 0; spec_recent_runs_ = 0; spec_cooldown_ = 64; }
  ensure_stage_b(frame_bucket(Fmax_));
  Fg_ = std::min(frame_bucket(Fmax_), Fs_);
  lens_b_ = d_frames_;
  char key[160];
  snprintf(key, sizeof(key), "B|%d|%d|%d|%d|%a", B_, Fg_, Fs_, Ts_, scales_[0]);
  run_stage('B', key);
  return false;
}

void Engine::download(bool want_audio, bool want_pcm) {
  EntryLock entry_lock;
  auto grow = [&](size_t total) {
    if (want_audio && total > h_audio_cap_) {
      if (h_audio_) PE_HIP(hipHostFree(h_audio_));
      h_audio_cap_ = total + total / 2;
      PE_HIP(hipHostMalloc((void**)&h_audio_, h_audio_cap_ * sizeof(float)));
    }
    if (want_pcm && total > h_pcm_cap_) {
      if (h_pcm_) PE_HIP(hipHostFree(h_pcm_));
      h_pcm_cap_ = total + total / 2;
      PE_HIP(hipHostMalloc((void**)&h_pcm_, h_pcm_cap_ * sizeof(int16_t)));
    }
  };
  // pcm16_kernel already wrote the samples into pinned host memory (zero-copy), packed back to back: nothing to enqueue
  const bool zc = pol_.pcm_zc && h_pcm_zc_ != nullptr && h_pcm_zc_cap_ >= (size_t)B_ * (size_t)Ss_;
  pcm_zc_live_ = false;
  if (spec_pending_ && B_ == 1 && (want_audio || want_pcm)) {
    // one utterance, speculative run: the copies are enqueued for the guessed length (>= the real one when the guess
    // holds) behind stage B, so that one synchronisation ends the whole call; the host view is trimmed afterwards
    const size_t n = (size_t)spec_fg_ * hop_;
    grow(n);
    if (want_audio) PE_HIP(hipMemcpyAsync(h_audio_, audio_, n * sizeof(float), hipMemcpyDeviceToHost, stream_));
    if (want_pcm && !zc) PE_HIP(hipMemcpyAsync(h_pcm_, pcm_, n * sizeof(int16_t), hipMemcpyDeviceToHost, stream_));
    if (finish_run()) {                        // synchronises; sample_off_ now holds the real length
      pcm_zc_live_ = zc;
      return;
    }
    // the guess missed: stage B was re-issued (its pcm16_kernel writes the host buffer again); copies below
  } else {
    finish_run();
  }
  const size_t total = (size_t)sample_off_[B_];
  grow(total);
  if (want_audio)
    for (int b = 0; b < B_; ++b)
      PE_HIP(hipMemcpyAsync(h_audio_ + sample_off_[b], audio_ + (size_t)b * Ss_,
                            (size_t)(sample_off_[b + 1] - sample_off_[b]) * sizeof(float), hipMemcpyDeviceToHost,
                            stream_));
  if (want_pcm && !zc)
    for (int b = 0; b < B_; ++b)
      PE_HIP(hipMemcpyAsync(h_pcm_ + sample_off_[b], pcm_ + (size_t)b * Ss_,
                            (size_t)(sample_off_[b + 1] - sample_off_[b]) * sizeof(int16_t), hipMemcpyDeviceToHost,
                            stream_));
  PE_HIP(hipStreamSynchronize(stream_));
  pcm_zc_live_ = zc;
}

int Engine::stream_begin(const int64_t* ids, int64_t n, const float scales[3], int64_t sid, const NoiseIn* noise) {
  EntryLock entry_lock;
  const int64_t offs[2] = {0, n};
  const int64_t sids[1] = {sid < 0 ? 0 : sid};
  upload(ids, offs, 1, scales, sids, noise);
  PE_HIP(hipSetDevice(device_));
  spec_pending_ = false;
  Tg_ = std::min(id_bucket(Tmax_), Ts_);
  char key[160];
  snprintf(key, sizeof(key), "A|%d|%d|%d|%a|%a|%d|%d", 1, Tg_, Ts_, scales_[1], scales_[2], (int)have_noise_w_, Fs_);
  run_stage('A', key);
  ++call_;
  PE_HIP(hipStreamSynchronize(stream_));
  finish_stage_b_sizes();
  ensure_stage_b(frame_bucket(Fmax_));
  Fg_ = std::min(frame_bucket(Fmax_), Fs_);
  lens_b_ = d_frames_;
  if (have_noise_z_) {
    issue_flow();
  } else {
    snprintf(key, sizeof(key), "F|%d|%d|%d|%d|%a", 1, Fg_, Fs_, Ts_, scales_[0]);
    run_stage('F', key);
  }
  s_frames_ = Fmax_;
  s_pos_ = 0;
  s_active_ = true;
  sample_off_.assign(2, 0);
  return s_frames_;
}

bool Engine::stream_next(int chunk_frames, const float** audio, const int16_t** pcm, int64_t* nsamples) {
  EntryLock entry_lock;
  if (!s_active_ || s_pos_ >= s_frames_) {
    s_active_ = false;
    if (nsamples) *nsamples = 0;
    return false;
  }
  if (chunk_frames < 1) throw std::runtime_error("chunk_frames must be >= 1");
  PE_HIP(hipSetDevice(device_));
  const int f0 = s_pos_, f1 = std::min(s_frames_, s_pos_ + chunk_frames);
  const int a = std::max(0, f0 - halo_frames_), b = std::min(s_frames_, f1 + halo_frames_);
  const int win[2] = {a, b - a};
  PE_HIP(hipMemcpyAsync(d_win_, win, sizeof(win), hipMemcpyHostToDevice, stream_));
  s_wg_ = std::min(rup(chunk_frames + 2 * halo_frames_, 32), Fs_);
  if (s_wg_ < b - a) s_wg_ = std::min(rup(b - a, 32), Fs_);
  char key[96];
  snprintf(key, sizeof(key), "W|%d|%d", s_wg_, Fs_);
  run_stage('W', key);
  const size_t n = (size_t)(f1 - f0) * hop_;
  if (n > h_audio_cap_) {
    if (h_audio_) PE_HIP(hipHostFree(h_audio_));
    h_audio_cap_ = n + n / 2;
    PE_HIP(hipHostMalloc((void**)&h_audio_, h_audio_cap_ * sizeof(float)));
  }
  PE_HIP(hipMemcpyAsync(h_audio_, audio_ + (size_t)(f0 - a) * hop_, n * sizeof(float), hipMemcpyDeviceToHost, stream_));
  PE_HIP(hipStreamSynchronize(stream_));
  // per-chunk peak normalisation, as the reference's streaming script does (infer_onnx_streaming.py:122)
  float peak = 0.01f;
  for (size_t i = 0; i < n; ++i) peak = std::max(peak, std::fabs(h_audio_[i]));
  const float sc = 32767.0f / peak;
  s_pcm_.resize(n);
  for (size_t i = 0; i < n; ++i) {
    float v = h_audio_[i] * sc;
    v = std::min(std::max(v, -32768.0f), 32767.0f);
    s_pcm_[i] = (int16_t)v;
  }
  s_pos_ = f1;
  if (audio) *audio = h_audio_;
  if (pcm) *pcm = s_pcm_.data();
  if (nsamples) *nsamples = (int64_t)n;
  return true;
}

const std::vector<int32_t>& Engine::durations_host() {
  EntryLock entry_lock;
  finish_run();
  std::vector<int> tmp((size_t)B_ * Ts_);
  PE_HIP(hipMemcpy(tmp.data(), d_dur_, tmp.size() * sizeof(int), hipMemcpyDeviceToHost));
  dur_h_.clear();
  for (int b = 0; b < B_; ++b)
    for (int t = 0; t < tlens_h_[b]; ++t) dur_h_.push_back(tmp[(size_t)b * Ts_ + t]);
  return dur_h_;
}

// Test hook: what randn_kernel draws for (seed_, call, site) -- the generator of the product path when the caller
// injects no noise: draws [row * RNG_PITCH, row * RNG_PITCH + n) of the site's stream (kernels.h: the pipeline's noise
// for column f of logical row r = utterance * channels + channel is draw r * RNG_PITCH + f).
void Engine::debug_randn(int site, uint64_t call, int64_t row, int64_t n, float* out) {
  if (n <= 0 || row < 0 || !out || site < 0 || site > 1) throw std::runtime_error("debug_randn: bad arguments");
  EntryLock entry_lock;
  PE_HIP(hipSetDevice(device_));
  float* d = nullptr;
  unsigned long long* st = nullptr;
  const long rows = (long)((n + RNG_PITCH - 1) / RNG_PITCH);
  const int cols = rows > 1 ? RNG_PITCH : (int)n;
  PE_HIP(hipMalloc((void**)&d, (size_t)rows * cols * sizeof(float)));
  if (hipMalloc((void**)&st, 16) != hipSuccess) { hipFree(d); throw std::runtime_error("debug_randn: out of memory"); }
  const unsigned long long hst[2] = {seed_, call};
  hipMemcpy(st, hst, sizeof(hst), hipMemcpyHostToDevice);
  launch::randn(stream_, d, rows, cols, (long)cols, (long)row, st, site);
  hipStreamSynchronize(stream_);
  hipMemcpy(out, d, (size_t)n * sizeof(float), hipMemcpyDeviceToHost);
  hipFree(d);
  hipFree(st);
}

// Per-stage tensors for parity debugging (tests only): name in {x_enc, stats (m_p | logs_p), xg, logw, z_p, z, noise_w,
// noise_z, audio}.
void Engine::debug_tensor(const std::string& name, int b, std::vector<float>& out, int* rows, int* cols) {
  EntryLock entry_lock;
  finish_run();
  PE_HIP(hipStreamSynchronize(stream_));
  const float* src = nullptr;
  int R = 0, Cn = 0;
  long stride = 0;
  if (name == "x_enc") { src = stage_a_enc_out() + (size_t)b * H_ * Ts_; R = H_; Cn = tlens_h_[b]; stride = Ts_; }
  else if (name == "stats") { src = stats_ + (size_t)b * 2 * C_ * Ts_; R = 2 * C_; Cn = tlens_h_[b]; stride = Ts_; }
  else if (name == "xg") { src = xg_ + (size_t)b * H_ * Ts_; R = H_; Cn = tlens_h_[b]; stride = Ts_; }
  else if (name == "logw") { src = logw_ + (size_t)b * Ts_; R = 1; Cn = tlens_h_[b]; stride = Ts_; }
  else if (name == "z") { src = zp_ + (size_t)b * C_ * Fs_; R = C_; Cn = frames_h_[b]; stride = Fs_; }
  else if (name == "z_p") {
    if (!zp_keep_) throw std::runtime_error("z_p is only kept with PIPER_HIP_DEBUG_KEEP=1");
    src = zp_keep_ + (size_t)b * C_ * Fs_; R = C_; Cn = frames_h_[b]; stride = Fs_;
  }
  else if (name == "noise_w") { src = noise_w_ + (size_t)b * 2 * Ts_; R = 2; Cn = tlens_h_[b]; stride = Ts_; }
  else if (name == "noise_z") { src = noise_z_ + (size_t)b * C_ * Fs_; R = C_; Cn = frames_h_[b]; stride = Fs_; }
  else if (name == "audio") { src = audio_ + (size_t)b * Ss_; R = 1; Cn = frames_h_[b] * hop_; stride = Ss_; }
  else throw std::runtime_error("unknown debug tensor " + name);
  out.resize((size_t)R * Cn);
  for (int r = 0; r < R; ++r)
    PE_HIP(hipMemcpy(out.data() + (size_t)r * Cn, src + (size_t)r * stride, Cn * sizeof(float), hipMemcpyDeviceToHost));
  *rows = R;
  *cols = Cn;
}

}  // namespace pe

#ifdef PE_STAMPS
// tuning build only (`make stamps`): the kernels live in four launch translation units, each with its own copy of the
// stamp / trace arrays (pe_rt.h PE_TRACE_FETCHER); these two entry points merge them.
extern "C" int pe_trace_fetch_conv(long long*, long long*, unsigned*);
extern "C" int pe_trace_fetch_bf3(long long*, long long*, unsigned*);
extern "C" int pe_trace_fetch_front(long long*, long long*, unsigned*);
extern "C" int pe_trace_fetch_tail(long long*, long long*, unsigned*);
static int pe_trace_merge(long long* stamps, long long* trace, unsigned* count) {
  int (*const fetch[4])(long long*, long long*, unsigned*) = {pe_trace_fetch_conv, pe_trace_fetch_bf3, pe_trace_fetch_front,
                                                               pe_trace_fetch_tail};
  std::vector<long long> st(PE_NSTAMP_K * PE_NSTAMP_I), tr((size_t)PE_NTRACE * 5);
  unsigned total = 0;
  for (auto f : fetch) {
    unsigned n = 0;
    const int rc = f(st.data(), tr.data(), &n);
    if (rc) return rc;
    if (stamps)
      for (size_t i = 0; i < st.size(); ++i)
        if (st[i]) stamps[i] = st[i];
    if (trace)
      for (unsigned r = 0; r < (n < PE_NTRACE ? n : PE_NTRACE) && total < PE_NTRACE; ++r, ++total)
        memcpy(trace + (size_t)total * 5, tr.data() + (size_t)r * 5, 5 * sizeof(long long));
  }
  if (count) *count = total;
  return 0;
}
// phase timestamps of pe_rt.h's PE_STAMP, [PE_NSTAMP_K][PE_NSTAMP_I] 100 MHz ticks
extern "C" int pe_debug_stamps(long long* out) {
  memset(out, 0, sizeof(long long) * PE_NSTAMP_K * PE_NSTAMP_I);
  return pe_trace_merge(out, nullptr, nullptr);
}
// per-launch trace: up to PE_NTRACE records (id, wall in, wall out, clock in, clock out) and their count; resets the counters
extern "C" int pe_debug_trace(long long* out, unsigned* count) { return pe_trace_merge(nullptr, out, count); }
#endif
