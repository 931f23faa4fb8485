#include "engine.h"
#include "kernels/launch.h"

#include <algorithm>
#include <cmath>
#include <shared_mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace pe {

thread_local long g_launches = 0;

// a launch with a level-2 profile row of its own (the element-wise / integer glue kernels; the conv / attention /
// fused-stage launchers bracket themselves and also carry FLOP and byte counts)
#define PE_LAUNCH_KB(kname, bytes, call)                                         \
  do {                                                                           \
    const int kh_ = kbegin(prof_level_ >= 2 ? krow(kname) : 0, 0.0, (bytes));    \
    call;                                                                        \
    kend(kh_);                                                                   \
  } while (0)
#define PE_LAUNCH_K(kname, call) PE_LAUNCH_KB(kname, 0.0, call)

// Engines that share a process (pe_group_*: one per device, each on its own thread) must not be inside a HIP call while
// another one CAPTURES a graph: allocations / synchronising copies on a second thread invalidate a capture in progress on
// this runtime, whatever the capture mode. Every public entry holds this lock shared; a capture takes it exclusively.
// A single engine per process never contends.
static std::shared_mutex g_capture_mu;
static thread_local int g_entry_depth = 0;
struct EntryLock {
  EntryLock() { if (g_entry_depth++ == 0) g_capture_mu.lock_shared(); }
  ~EntryLock() { if (--g_entry_depth == 0) g_capture_mu.unlock_shared(); }
};

static inline int rup(int v, int m) { return (v + m - 1) / m * m; }

// tile configurations of conv_mfma_kernel: {WM, WN, MT, NT}
enum { CFG_A = 0, CFG_B = 1, CFG_C = 2, CFG_S = 3, CFG_G = 4, CFG_C2 = 5, CFG_B2 = 6 };
static const int CFG_BM[] = {128, 64, 32, 64, 128, 32, 64};
static const int CFG_BN[] = {128, 128, 128, 64, 64, 256, 256};

// ------------------------------------------------------------------------------------------------
// setup
// ------------------------------------------------------------------------------------------------

// Every packed weight tensor is carved from ONE arena in a deterministic order, so that the arena of the rank that
// parsed and packed the voice can be broadcast device-to-device into the identically laid-out arenas of the other ranks.
float* Engine::dev_alloc(size_t nfloats, const float* src) {
  arena_off_ = (arena_off_ + 255) / 256 * 256;
  const size_t bytes = std::max<size_t>(nfloats, 1) * sizeof(float);
  if (arena_off_ + bytes > arena_bytes_) throw std::runtime_error("internal: packed weights exceed the arena bound");
  float* d = reinterpret_cast<float*>(arena_ + arena_off_);
  arena_off_ += bytes;
  if (src && nfloats && !skeleton_) PE_HIP(hipMemcpy(d, src, nfloats * sizeof(float), hipMemcpyHostToDevice));
  weight_bytes_ += nfloats * sizeof(float);
  return d;
}
float* Engine::dev_copy(const std::vector<float>& v) { return dev_alloc(v.size(), v.data()); }

float* Engine::dev_tensor(const WeightSet& ws, const std::string& name) {
  const HostTensor& t = ws.get(name);
  return dev_alloc((size_t)t.numel(), t.data.empty() ? nullptr : t.data.data());
}

// Packed copies: conv weights once in 32x32x2 fragment order (rows padded to the block tile), long-K convs once more in
// 16x16x4 order, DDSConv / proj matrices in 16x16x4 order, the <= 64-channel resblock convs as mrf_kernel weight streams, plus the raw
// small tensors. 3.5x the raw floats + slack covers every architecture the loader accepts; checked while carving.
size_t Engine::arena_bound(const WeightSet& ws) {
  size_t n = 0;
  for (auto& kv : ws.t) n += (size_t)kv.second.numel() + 64;
  // matrix mode bf16x3: the flow / generator conv weights once more as split bf16 fragments (same size as the f32 packing)
  return (n * (env_bf3() ? 10 : 7) / 2 + (4u << 20)) * sizeof(float);
}
bool Engine::env_bf3() { return LaunchPolicy::matrix_bf3_env(); }

static inline uint16_t bf16_rne(float v) {
  uint32_t u;
  memcpy(&u, &v, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float bf16_to_f32(uint16_t h) {
  const uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// Packs a dense [rows][Cin][ntaps] matrix into the A-operand order of conv_mfma_kernel:
//   [mtile][chunk][tap][q = 0..3][lane = 0..63][j = 0..3] with kk = 4q + j, lane -> row = mtile*32 + (lane&31),
//   ci = chunk*32 + 2*kk + (lane>>5): the 16 fragments of a step are four 16-byte loads per lane. With gate=true the 32-row tiles alternate between the tanh
//   half (rows [0,split)) and the sigmoid half (rows [split,2*split)) so that one wave owns both.
// `bias`: nbias values or null (none). In skeleton mode W / bias are not read (only sizes matter).
PackedConv Engine::pack_matrix(const std::vector<float>& W, int rows, int Cin, int ntaps,
                               const std::vector<float>* bias, int nbias, int dil, int padl, bool gate, int split) {
  PackedConv pc;
  pc.rows = rows;
  pc.Cin = Cin;
  pc.ntaps = ntaps;
  pc.dil = dil;
  pc.padl = padl;
  pc.nchunks = (Cin + KC - 1) / KC;
  pc.gate = gate;
  pc.split = split;
  int vt = gate ? 2 * ((split + 31) / 32) : (rows + 31) / 32;
  if (gate) pc.cfg = (vt % 4 == 0) ? CFG_A : CFG_B;
  else pc.cfg = (vt % 4 == 0) ? CFG_A : (vt % 2 == 0 ? CFG_B : CFG_C);
  const int tiles_per_block = CFG_BM[pc.cfg] / 32;
  pc.mtiles = rup(vt, tiles_per_block);
  const size_t np = (size_t)pc.mtiles * pc.nchunks * ntaps * (KC / 2) * 64;
  std::vector<float> P(skeleton_ ? 0 : np, 0.f);
  for (int mt = 0; mt < (skeleton_ ? 0 : pc.mtiles); ++mt)
    for (int c = 0; c < pc.nchunks; ++c)
      for (int tap = 0; tap < ntaps; ++tap)
        for (int kk = 0; kk < KC / 2; ++kk)
          for (int lane = 0; lane < 64; ++lane) {
            int r = lane & 31, row;
            if (gate) {
              int q = mt >> 1, ch = q * 32 + r;
              row = (ch < split) ? ((mt & 1) ? split + ch : ch) : -1;
            } else {
              row = mt * 32 + r;
              if (row >= rows) row = -1;
            }
            int ci = c * KC + 2 * kk + (lane >> 5);
            float v = 0.f;
            if (row >= 0 && ci < Cin) v = W[((size_t)row * Cin + ci) * ntaps + tap];
            // within a (tile, chunk, tap) step a lane's 16 values are four float4 (kk = 4q + j)
            P[(((size_t)mt * pc.nchunks + c) * ntaps + tap) * (KC / 2) * 64 + (kk >> 2) * 256 + lane * 4 + (kk & 3)] = v;
          }
  pc.wp = dev_alloc(np, skeleton_ ? nullptr : P.data());
  if (pc.nchunks * ntaps >= 24 || pol_.splitk16 >= 3) {          // 3 = every conv (tests)
    // long-K convs may run through conv_splitk16_kernel: [16-row sub-tile][chunk][tap][q][lane][4], lane ->
    // (row = lane & 15, k = lane >> 4), float4 element j of group q = k-step 4q + j = input channel chunk*32 + 4s + k
    const size_t nq = (size_t)pc.mtiles * 2 * pc.nchunks * ntaps * (KC / 4) * 64;
    std::vector<float> Q(skeleton_ ? 0 : nq, 0.f);
    for (int st = 0; st < (skeleton_ ? 0 : pc.mtiles * 2); ++st)
      for (int c = 0; c < pc.nchunks; ++c)
        for (int tap = 0; tap < ntaps; ++tap)
          for (int q = 0; q < KC / 16; ++q)
            for (int lane = 0; lane < 64; ++lane)
              for (int j = 0; j < 4; ++j) {
                const int mt = st >> 1, r = (st & 1) * 16 + (lane & 15);
                int row;
                if (gate) {
                  const int ch = (mt >> 1) * 32 + r;
                  row = (ch < split) ? ((mt & 1) ? split + ch : ch) : -1;
                } else {
                  row = mt * 32 + r;
                  if (row >= rows) row = -1;
                }
                const int ci = c * KC + 4 * (4 * q + j) + (lane >> 4);
                if (row >= 0 && ci < Cin)
                  Q[((((size_t)st * pc.nchunks + c) * ntaps + tap) * (KC / 16) + q) * 256 + lane * 4 + j] =
                      W[((size_t)row * Cin + ci) * ntaps + tap];
              }
    pc.wp16 = dev_alloc(nq, skeleton_ ? nullptr : Q.data());
  }
  if (pack_bf3_now_) {
    // conv_bf3_kernel (kernels/conv_bf3.h): every weight as hi = bf16(w), lo = bf16(w - hi), in the A-operand order of
    // v_mfma_f32_32x32x16_bf16: [m tile][chunk][tap][part hi|lo][k-step][lane][8], lane -> row = lane & 31, input channel
    // chunk*32 + 8*(2*kstep + (lane >> 5)) + e. One (tile, chunk, tap) step = 1024 floats, like the f32 packing.
    std::vector<uint16_t> R(skeleton_ ? 0 : np * 2, 0);
    for (int mt = 0; mt < (skeleton_ ? 0 : pc.mtiles); ++mt)
      for (int c = 0; c < pc.nchunks; ++c)
        for (int tap = 0; tap < ntaps; ++tap) {
          const size_t step = (((size_t)mt * pc.nchunks + c) * ntaps + tap) * 2048;      // in bf16 elements
          for (int ks = 0; ks < 2; ++ks)
            for (int lane = 0; lane < 64; ++lane) {
              int r = lane & 31, row;
              if (gate) {
                int q = mt >> 1, ch = q * 32 + r;
                row = (ch < split) ? ((mt & 1) ? split + ch : ch) : -1;
              } else {
                row = mt * 32 + r;
                if (row >= rows) row = -1;
              }
              for (int e = 0; e < 8; ++e) {
                const int ci = c * KC + 8 * (2 * ks + (lane >> 5)) + e;
                if (row < 0 || ci >= Cin) continue;
                const float v = W[((size_t)row * Cin + ci) * ntaps + tap];
                const uint16_t hi = bf16_rne(v), lo = bf16_rne(v - bf16_to_f32(hi));
                R[step + ((size_t)(0 * 2 + ks) * 64 + lane) * 8 + e] = hi;
                R[step + ((size_t)(1 * 2 + ks) * 64 + lane) * 8 + e] = lo;
              }
            }
        }
    pc.wpb = dev_alloc(np, skeleton_ ? nullptr : reinterpret_cast<const float*>(R.data()));
  }
  pc.bias = bias ? dev_alloc((size_t)nbias, skeleton_ ? nullptr : bias->data()) : nullptr;
  pc.macs_per_col = (double)rows * Cin * ntaps;
  return pc;
}

// Conv1d weight [Cout][Cin][K] -> packed. in_rev / out_rev fold a channel Flip (modules.py:385-391)
// into the weights: in_rev reverses the input-channel order, out_rev the output rows (and bias).
PackedConv Engine::pack_conv(const WeightSet& ws, const std::string& wname, const std::string& bname, int dil,
                             int padl_override, bool gate, int in_rev, int out_rev) {
  const HostTensor& w = ws.get(wname);
  if (w.dims.size() != 3) throw std::runtime_error(wname + ": expected a rank-3 conv weight");
  const int Co = (int)w.dims[0], Ci = (int)w.dims[1], K = (int)w.dims[2];
  std::vector<float> W(skeleton_ ? 0 : (size_t)Co * Ci * K);
  if (!skeleton_) {
    if (w.data.size() != W.size()) throw std::runtime_error(wname + ": data size mismatch");
    for (int o = 0; o < Co; ++o)
      for (int i = 0; i < Ci; ++i)
        for (int k = 0; k < K; ++k) {
          const int so = out_rev ? Co - 1 - o : o, si = in_rev ? Ci - 1 - i : i;
          W[((size_t)o * Ci + i) * K + k] = w.data[((size_t)so * Ci + si) * K + k];
        }
  }
  std::vector<float> bias;
  bool has_b = !bname.empty() && ws.has(bname);
  if (has_b) {
    if (ws.get(bname).numel() != Co) throw std::runtime_error(bname + ": bias size mismatch");
    if (!skeleton_) {
      bias = ws.get(bname).data;
      if (out_rev) std::reverse(bias.begin(), bias.end());
    }
  }
  // "same" padding: get_padding (commons.py:17-18) == (K-1)*dil/2 ; FFN._same_padding left pad (K-1)/2
  const int padl = padl_override >= 0 ? padl_override : (K - 1) * dil / 2;
  return pack_matrix(W, Co, Ci, K, has_b ? &bias : nullptr, Co, dil, padl, gate, gate ? Co / 2 : 0);
}

PackedConv Engine::pack_qkv(const WeightSet& ws, const std::string& prefix, float** out16) {
  // conv_q / conv_k / conv_v (attentions.py:216-218) share their input: one GEMM with 3H rows.
  std::vector<float> W, bias;
  int H = 0;
  for (const char* n : {"conv_q", "conv_k", "conv_v"}) {
    const HostTensor& w = ws.get(prefix + "." + n + ".weight");
    const HostTensor& b = ws.get(prefix + "." + n + ".bias");
    H = (int)w.dims[0];
    if (!skeleton_) {
      W.insert(W.end(), w.data.begin(), w.data.end());
      bias.insert(bias.end(), b.data.begin(), b.data.end());
    }
  }
  if (out16) *out16 = pack16(W, 3 * H, H);
  return pack_matrix(W, 3 * H, H, 1, &bias, 3 * H, 1, 0, false, 0);
}

// ConvTranspose1d weight [Cin][Cout][K] with K == 2*stride, padding (K-stride)/2 (models.py:321-332):
// polyphase GEMM rows (co*stride + phase), two taps: tap0 reads x[j-1] with W[ci][co][phase+stride],
// tap1 reads x[j] with W[ci][co][phase]; output t = j*stride + phase - pad.
PackedConv Engine::pack_convT(const WeightSet& ws, const std::string& prefix, int stride) {
  const HostTensor& w = ws.get(prefix + ".weight");
  const int Ci = (int)w.dims[0], Co = (int)w.dims[1], K = (int)w.dims[2];
  if (K != 2 * stride || ((K - stride) & 1))
    throw std::runtime_error(prefix + ": ConvTranspose1d with kernel != 2*stride is not supported");
  const int rows = Co * stride;
  std::vector<float> W(skeleton_ ? 0 : (size_t)rows * Ci * 2);
  for (int co = 0; co < (skeleton_ ? 0 : Co); ++co)
    for (int ph = 0; ph < stride; ++ph)
      for (int ci = 0; ci < Ci; ++ci) {
        const size_t row = (size_t)co * stride + ph;
        W[(row * Ci + ci) * 2 + 0] = w.data[((size_t)ci * Co + co) * K + ph + stride];
        W[(row * Ci + ci) * 2 + 1] = w.data[((size_t)ci * Co + co) * K + ph];
      }
  std::vector<float> bias = ws.get(prefix + ".bias").data;
  PackedConv pc = pack_matrix(W, rows, Ci, 2, &bias, Co, 1, 1, false, 0);
  pc.up = stride;
  pc.padT = (K - stride) / 2;
  return pc;
}

// A dense [rows][K] matrix in the A-operand order of the 16x16x4 MFMA used by dds_layer16_kernel:
// [16-row tile][q][lane][4], lane -> (row = lane & 15, k = lane >> 4), float4 element j of group q = k-step 4q + j, i.e.
// input channel 4 * (4q + j) + k. K is padded to a multiple of 32 (the kernel's Hp).
float* Engine::pack16(const std::vector<float>& W, int rows, int K) {
  const int Kp = rup(K, 32), nq = Kp / 16, ntile = (rows + 15) / 16;
  const size_t np = (size_t)ntile * nq * 256;
  std::vector<float> P(skeleton_ ? 0 : np, 0.f);
  for (int mt = 0; mt < (skeleton_ ? 0 : ntile); ++mt)
    for (int q = 0; q < nq; ++q)
      for (int lane = 0; lane < 64; ++lane)
        for (int jj = 0; jj < 4; ++jj) {
          const int row = mt * 16 + (lane & 15), ci = 4 * (4 * q + jj) + (lane >> 4);
          if (row < rows && ci < K) P[(((size_t)mt * nq + q) * 64 + lane) * 4 + jj] = W[(size_t)row * K + ci];
        }
  float* d16 = dev_alloc(np, skeleton_ ? nullptr : P.data());
  if (const float* d4 = pack4(W, rows, K)) w4_of_[d16] = d4;
  return d16;
}

// The same matrix in the A-operand order of the 4x4x1 MFMA used by dds_layer4_kernel (kernels/dds4.h):
// [64-row tile][k quad][lane][4], lane -> row 64 * tile + lane, float4 element j of quad q = input channel 4q + j. Only
// packed for the K = 192 / 96 shapes the 4-column kernels are compiled for (kernels/col4.h).
float* Engine::pack4(const std::vector<float>& W, int rows, int K) {
  if (K != 192 && K != 96) return nullptr;
  const int nq = K / 4, ntile = (rows + 63) / 64;
  const size_t np = (size_t)ntile * nq * 256;
  std::vector<float> P(skeleton_ ? 0 : np, 0.f);
  for (int mt = 0; mt < (skeleton_ ? 0 : ntile); ++mt)
    for (int q = 0; q < nq; ++q)
      for (int lane = 0; lane < 64; ++lane)
        for (int jj = 0; jj < 4; ++jj) {
          const int row = mt * 64 + lane;
          if (row < rows) P[(((size_t)mt * nq + q) * 64 + lane) * 4 + jj] = W[(size_t)row * K + 4 * q + jj];
        }
  return dev_alloc(np, skeleton_ ? nullptr : P.data());
}

// FFN weights in ffn_kernel's per-slice orders (kernels/ffn.h). conv_1 [FC][192][3] ->
// [slice][tile 3][wave 4][tap 3][quad 3][lane][4]: row 48 slice + 16 tile + (lane & 15), channel 48 wave + 4 (4 quad + j) + (lane >> 4).
const float* Engine::pack_ffn1(const WeightSet& ws, const std::string& wname) {
  const HostTensor& w = ws.get(wname);
  if (w.dims.size() != 3 || w.dims[1] != 192 || w.dims[2] != 3 || w.dims[0] % 48 || w.dims[0] / 48 > 16) return nullptr;
  const int FC = (int)w.dims[0], S = FC / 48;
  const size_t np = (size_t)FC * 192 * 3;
  std::vector<float> P(skeleton_ ? 0 : np, 0.f);
  for (int s = 0; s < (skeleton_ ? 0 : S); ++s)
    for (int m = 0; m < 3; ++m)
      for (int wv = 0; wv < 4; ++wv)
        for (int tp = 0; tp < 3; ++tp)
          for (int q = 0; q < 3; ++q)
            for (int lane = 0; lane < 64; ++lane)
              for (int j = 0; j < 4; ++j) {
                const int row = 48 * s + 16 * m + (lane & 15), ch = 48 * wv + 4 * (4 * q + j) + (lane >> 4);
                P[(((((size_t)(s * 3 + m) * 4 + wv) * 3 + tp) * 3 + q) * 64 + lane) * 4 + j] = w.data[((size_t)row * 192 + ch) * 3 + tp];
              }
  return dev_alloc(np, skeleton_ ? nullptr : P.data());
}
// conv_2 [192][FC][3] -> [slice][row tile 12][tap 3][quad 3][lane][4]: row 16 tile + (lane & 15), hidden channel
// 48 slice + 4 (4 quad + j) + (lane >> 4).
const float* Engine::pack_ffn2(const WeightSet& ws, const std::string& wname) {
  const HostTensor& w = ws.get(wname);
  if (w.dims.size() != 3 || w.dims[0] != 192 || w.dims[2] != 3 || w.dims[1] % 48 || w.dims[1] / 48 > 16) return nullptr;
  const int FC = (int)w.dims[1], S = FC / 48;
  const size_t np = (size_t)FC * 192 * 3;
  std::vector<float> P(skeleton_ ? 0 : np, 0.f);
  for (int s = 0; s < (skeleton_ ? 0 : S); ++s)
    for (int rt = 0; rt < 12; ++rt)
      for (int tp = 0; tp < 3; ++tp)
        for (int q = 0; q < 3; ++q)
          for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 4; ++j) {
              const int row = 16 * rt + (lane & 15), hid = 48 * s + 4 * (4 * q + j) + (lane >> 4);
              P[((((size_t)(s * 12 + rt) * 3 + tp) * 3 + q) * 64 + lane) * 4 + j] = w.data[((size_t)row * FC + hid) * 3 + tp];
            }
  return dev_alloc(np, skeleton_ ? nullptr : P.data());
}

// A 1x1 conv weight [Co][Ci][1] (optionally with reversed input / output channels: the Flip folded in) in pack16 order
float* Engine::pack16_conv(const WeightSet& ws, const std::string& wname, int in_rev, int out_rev) {
  const HostTensor& w = ws.get(wname);
  if (w.dims.size() != 3 || w.dims[2] != 1) throw std::runtime_error(wname + ": expected a 1x1 conv weight");
  const int Co = (int)w.dims[0], Ci = (int)w.dims[1];
  std::vector<float> W(skeleton_ ? 0 : (size_t)Co * Ci);
  if (!skeleton_)
    for (int o = 0; o < Co; ++o)
      for (int i = 0; i < Ci; ++i)
        W[(size_t)o * Ci + i] = w.data[(size_t)(out_rev ? Co - 1 - o : o) * Ci + (in_rev ? Ci - 1 - i : i)];
  return pack16(W, Co, Ci);
}

// A 1x1 conv weight [Co][Ci][1] with Ci < 192 in pack4 order with K zero-padded to 192, for colchain4_kernel mode 3 (whose
// input descriptor ends after the Ci real rows, so the padded channels read as zeros too).
const float* Engine::pack4_conv_pad192(const WeightSet& ws, const std::string& wname, int in_rev, int out_rev) {
  const HostTensor& w = ws.get(wname);
  if (w.dims.size() != 3 || w.dims[2] != 1 || w.dims[1] > 192) return nullptr;
  const int Co = (int)w.dims[0], Ci = (int)w.dims[1];
  std::vector<float> W(skeleton_ ? 0 : (size_t)Co * 192, 0.f);
  if (!skeleton_)
    for (int o = 0; o < Co; ++o)
      for (int i = 0; i < Ci; ++i)
        W[(size_t)o * 192 + i] = w.data[(size_t)(out_rev ? Co - 1 - o : o) * Ci + (in_rev ? Ci - 1 - i : i)];
  return pack4(W, Co, 192);
}

DdsW Engine::load_dds(const WeightSet& ws, const std::string& p) {
  DdsW d;
  for (int i = 0; i < arch_[A_DDSLAYERS]; ++i) {
    const std::string s = std::to_string(i);
    d.dw_w.push_back(dev_tensor(ws, p + ".convs_sep." + s + ".weight"));
    d.dw_b.push_back(dev_tensor(ws, p + ".convs_sep." + s + ".bias"));
    d.c1x1.push_back(pack_conv(ws, p + ".convs_1x1." + s + ".weight", p + ".convs_1x1." + s + ".bias", 1, -1,
                               false, 0, 0));
    {
      const HostTensor& w1 = ws.get(p + ".convs_1x1." + s + ".weight");
      d.w16.push_back(pack16(w1.data, (int)w1.dims[0], (int)w1.dims[1]));     // the same matrix for dds_layer16_kernel
    }
    d.g1.push_back(dev_tensor(ws, p + ".norms_1." + s + ".gamma"));
    d.b1.push_back(dev_tensor(ws, p + ".norms_1." + s + ".beta"));
    d.g2.push_back(dev_tensor(ws, p + ".norms_2." + s + ".gamma"));
    d.b2.push_back(dev_tensor(ws, p + ".norms_2." + s + ".beta"));
  }
  return d;
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

void Engine::init(const WeightSet& ws) {
  pol_.read_env();
  use_graphs_ = !pol_.no_graph;
  matrix_bf3_ = env_bf3();
  memcpy(arch_, ws.arch, sizeof(arch_));
  PE_HIP(hipSetDevice(device_));
  PE_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
  ls_ = stream_;
  H_ = arch_[A_HIDDEN]; C_ = arch_[A_INTER]; FC_ = arch_[A_FILTER]; nh_ = arch_[A_NHEADS];
  nlayers_ = arch_[A_NLAYERS]; ksz_ = arch_[A_KSIZE]; window_ = arch_[A_WINDOW]; U_ = arch_[A_UPINIT];
  gin_ = arch_[A_GIN]; nspk_ = arch_[A_NSPK];
  if (H_ <= 0 || C_ <= 0 || nh_ <= 0 || H_ % nh_ || (C_ & 1)) throw std::runtime_error("bad architecture header");
  dk_ = H_ / nh_;
  if (dk_ > 128 || (dk_ & 1)) throw std::runtime_error("head dimension must be even and <= 128");
  if ((2 * window_ + 1) * dk_ > 1280 || window_ > 4)
    throw std::runtime_error("relative-attention window too wide (window <= 4, (2*window+1) * head dim <= 1280)");
  if (H_ % 32 || H_ > 256) throw std::runtime_error("hidden_channels must be a multiple of 32 and <= 256");
  if (ksz_ > 3 || !(ksz_ & 1)) throw std::runtime_error("kernel_size must be 1 or 3");
  hop_ = 1;
  for (int i = 0; i < arch_[A_NUPS]; ++i) hop_ *= arch_[A_UPR0 + i];

  // ---- text encoder
  emb_ = dev_tensor(ws, "enc_p.emb.weight");
  const int padl_ffn = (ksz_ - 1) / 2;    // attentions.py:419-427
  for (int l = 0; l < nlayers_; ++l) {
    const std::string s = std::to_string(l), a = "enc_p.encoder.attn_layers." + s,
                      f = "enc_p.encoder.ffn_layers." + s;
    EncLayer e;
    e.qkv = pack_qkv(ws, a, &e.qkv16);
    e.o = pack_conv(ws, a + ".conv_o.weight", a + ".conv_o.bias", 1, -1, false, 0, 0);
    e.o16 = pack16_conv(ws, a + ".conv_o.weight", 0, 0);
    e.relk = dev_tensor(ws, a + ".emb_rel_k");
    e.relv = dev_tensor(ws, a + ".emb_rel_v");
    e.g1 = dev_tensor(ws, "enc_p.encoder.norm_layers_1." + s + ".gamma");
    e.b1 = dev_tensor(ws, "enc_p.encoder.norm_layers_1." + s + ".beta");
    e.f1 = pack_conv(ws, f + ".conv_1.weight", f + ".conv_1.bias", 1, padl_ffn, false, 0, 0);
    e.f2 = pack_conv(ws, f + ".conv_2.weight", f + ".conv_2.bias", 1, padl_ffn, false, 0, 0);
    if (H_ == 192 && ksz_ == 3 && padl_ffn == 1) {     // the fused small-call FFN (kernels/ffn.h)
      e.f1p = pack_ffn1(ws, f + ".conv_1.weight");
      e.f2p = pack_ffn2(ws, f + ".conv_2.weight");
    }
    e.g2 = dev_tensor(ws, "enc_p.encoder.norm_layers_2." + s + ".gamma");
    e.b2 = dev_tensor(ws, "enc_p.encoder.norm_layers_2." + s + ".beta");
    enc_.push_back(e);
  }
  enc_proj_ = pack_conv(ws, "enc_p.proj.weight", "enc_p.proj.bias", 1, -1, false, 0, 0);
  enc_proj16_ = pack16_conv(ws, "enc_p.proj.weight", 0, 0);

  // ---- duration predictor (reverse path)
  dp_pre_ = pack_conv(ws, "dp.pre.weight", "dp.pre.bias", 1, -1, false, 0, 0);
  dp_pre16_ = (H_ == 192) ? pack16_conv(ws, "dp.pre.weight", 0, 0) : nullptr;      // its pack4 twin: colchain4_kernel mode 3
  dp_dds_ = load_dds(ws, "dp.convs");
  dp_proj_ = pack_conv(ws, "dp.proj.weight", "dp.proj.bias", 1, -1, false, 0, 0);
  {
    const HostTensor& w = ws.get("dp.proj.weight");
    dp_proj16_ = pack16(w.data, (int)w.dims[0], (int)w.dims[1]);
  }
  for (int i = arch_[A_DPFLOWS] - 1; i >= 1; --i) {     // dp.flows.{7,5,3} (models.py:108-110)
    const std::string p = "dp.flows." + std::to_string(2 * i + 1);
    CFlow cf;
    cf.pre_w = dev_tensor(ws, p + ".pre.weight");
    cf.pre_b = dev_tensor(ws, p + ".pre.bias");
    cf.dds = load_dds(ws, p + ".convs");
    cf.proj = pack_conv(ws, p + ".proj.weight", p + ".proj.bias", 1, -1, false, 0, 0);
    if (cf.proj.rows != 3 * arch_[A_NBINS] - 1 || arch_[A_NBINS] != 10)
      throw std::runtime_error("spline with num_bins != 10 is not supported");
    {
      const HostTensor& w = ws.get(p + ".proj.weight");
      cf.proj16 = pack16(w.data, (int)w.dims[0], (int)w.dims[1]);
    }
    cflows_.push_back(cf);
  }
  {
    // After the (odd number of) Flip/ConvFlow pairs and the final Flip, logical channel 0 is ...
    // tracked in run(); here only the scalars of ElementwiseAffine channel 0 are needed. They are kernel arguments (host
    // values); a copy sits in the arena so that a skeleton engine can fetch them once the arena has arrived.
    ea_dev_m_ = dev_tensor(ws, "dp.flows.0.m");
    ea_dev_logs_ = dev_tensor(ws, "dp.flows.0.logs");
    if (!skeleton_) {
      ea_m0_ = ws.get("dp.flows.0.m").data[0];
      ea_es0_ = std::exp(-ws.get("dp.flows.0.logs").data[0]);
    }
  }

  // ---- coupling flow, execution order = reversed module order, Flip folded into weights
  pack_bf3_now_ = matrix_bf3_;           // from here on (flow + generator) the convs are also packed for conv_bf3_kernel
  {
    const int nf = arch_[A_FLOWN], half = C_ / 2, wnl = arch_[A_WNLAYERS], wnk = arch_[A_WNK];
    int flips = 0;
    for (int f = nf - 1; f >= 0; --f) {
      ++flips;                                   // the Flip that precedes this layer in reverse
      const bool odd = flips & 1;
      const std::string p = "flow.flows." + std::to_string(2 * f);
      Rcl r;
      // odd parity: x0 = reversed upper half of the physical tensor, x1 = reversed lower half
      r.in_off = odd ? half : 0;
      r.out_off = odd ? 0 : half;
      r.pre = pack_conv(ws, p + ".pre.weight", p + ".pre.bias", 1, -1, false, odd, 0);
      for (int i = 0; i < wnl; ++i) {
        const std::string s = std::to_string(i);
        r.in.push_back(pack_conv(ws, p + ".enc.in_layers." + s + ".weight", p + ".enc.in_layers." + s + ".bias",
                                 1, -1, true, 0, 0));
        r.rs.push_back(pack_conv(ws, p + ".enc.res_skip_layers." + s + ".weight",
                                 p + ".enc.res_skip_layers." + s + ".bias", 1, -1, false, 0, 0));
        {
          const HostTensor& wrs = ws.get(p + ".enc.res_skip_layers." + s + ".weight");
          r.rs4.push_back(wrs.dims.size() == 3 && wrs.dims[2] == 1 ? pack4(wrs.data, (int)wrs.dims[0], (int)wrs.dims[1]) : nullptr);
        }
        (void)wnk;
      }
      r.post = pack_conv(ws, p + ".post.weight", p + ".post.bias", 1, -1, false, 0, odd);
      r.pre16 = pack16_conv(ws, p + ".pre.weight", odd, 0);
      if (H_ == 192 && rcls_.empty()) r.pre4pad = pack4_conv_pad192(ws, p + ".pre.weight", odd, 0);   // first layer's pre: a launch of its own
      r.post16 = pack16_conv(ws, p + ".post.weight", 0, odd);
      rcls_.push_back(r);
      if (gin_) {
        const HostTensor& cw = ws.get(p + ".enc.cond_layer.weight");
        cond_wn_.push_back(CondW{dev_tensor(ws, p + ".enc.cond_layer.weight"), dev_tensor(ws, p + ".enc.cond_layer.bias"),
                                 (int)cw.dims[0]});
      }
    }
    if (flips & 1) throw std::runtime_error("odd number of flow layers is not supported");
  }

  // ---- HiFiGAN
  dec_pre_ = pack_conv(ws, "dec.conv_pre.weight", "dec.conv_pre.bias", 1, -1, false, 0, 0);
  {
    const int nk = arch_[A_NRB], nd = arch_[A_NDIL];
    int ch = U_;
    for (int i = 0; i < arch_[A_NUPS]; ++i) {
      UpStage st;
      st.rate = arch_[A_UPR0 + i];
      st.up = pack_convT(ws, "dec.ups." + std::to_string(i), st.rate);
      ch = U_ >> (i + 1);
      st.ch = ch;
      for (int j = 0; j < nk; ++j) {
        const std::string rb = "dec.resblocks." + std::to_string(i * nk + j);
        std::vector<PackedConv> cv;
        std::vector<UpStage::HostConv> hv;
        auto add = [&](const std::string& wn, const std::string& bn, int dil) {
          cv.push_back(pack_conv(ws, wn, bn, dil, -1, false, 0, 0));
          const HostTensor& w = ws.get(wn);
          UpStage::HostConv h;
          h.w = w.data; h.co = (int)w.dims[0]; h.ci = (int)w.dims[1]; h.k = (int)w.dims[2]; h.dil = dil;
          h.bias = cv.back().bias;
          hv.push_back(std::move(h));
        };
        for (int d = 0; d < nd; ++d) {
          const int dil = arch_[A_RBDIL0 + j * MAX_DIL + d];
          const std::string s = std::to_string(d);
          if (arch_[A_RESBLOCK] == 1) {
            add(rb + ".convs1." + s + ".weight", rb + ".convs1." + s + ".bias", dil);
            add(rb + ".convs2." + s + ".weight", rb + ".convs2." + s + ".bias", 1);
          } else {
            add(rb + ".convs." + s + ".weight", rb + ".convs." + s + ".bias", dil);
          }
        }
        st.rb.push_back(cv);
        st.rb_host.push_back(std::move(hv));
      }
      {
        // sum of the resblocks' last biases: the K-concatenated last step adds it once
        std::vector<float> bs(skeleton_ ? 0 : (size_t)ch, 0.f);
        for (int j = 0; j < nk; ++j) {
          const std::string rb = "dec.resblocks." + std::to_string(i * nk + j);
          const std::string bn = rb + (arch_[A_RESBLOCK] == 1 ? ".convs2." : ".convs.") + std::to_string(nd - 1) + ".bias";
          if (!ws.has(bn) || ws.get(bn).numel() != ch) throw std::runtime_error(bn + ": bias size mismatch");
          if (!skeleton_)
            for (int c = 0; c < ch; ++c) bs[c] += ws.get(bn).data[c];
        }
        st.last_bias_sum = dev_alloc((size_t)ch, skeleton_ ? nullptr : bs.data());
      }
      build_mrf(st);
      st.rb_host.clear();
      ups_.push_back(st);
    }
    const HostTensor& pw = ws.get("dec.conv_post.weight");
    post_w_ = dev_tensor(ws, "dec.conv_post.weight");
    post_cin_ = (int)pw.dims[1];
    if ((int)pw.dims[0] != 1 || post_cin_ != ch || (int)pw.dims[2] != POST_K)
      throw std::runtime_error("dec.conv_post shape mismatch");
  }

  pack_bf3_now_ = false;
  // ---- speaker conditioning
  if (nspk_ > 1) {
    if (!gin_) throw std::runtime_error("multi-speaker voice without gin_channels");
    emb_g_ = dev_tensor(ws, "emb_g.weight");
    const HostTensor& dw = ws.get("dp.cond.weight");
    cond_dp_ = CondW{dev_tensor(ws, "dp.cond.weight"), dev_tensor(ws, "dp.cond.bias"), (int)dw.dims[0]};
    const HostTensor& cw = ws.get("dec.cond.weight");
    cond_dec_ = CondW{dev_tensor(ws, "dec.cond.weight"), dev_tensor(ws, "dec.cond.bias"), (int)cw.dims[0]};
    cond_off_dp_ = 0;
    int off = cond_dp_.rows;
    for (auto& c : cond_wn_) { cond_off_wn_.push_back(off); off += c.rows; }
    cond_off_dec_ = off;
    off += cond_dec_.rows;
    cond_bs_ = off;
  }

  {
    // receptive half-width of the generator in frames (SURVEY.md section 7 hard part F), walking back from
    // the waveform: conv_post, then per stage the widest resblock and the transposed conv, then conv_pre
    long r = 3;
    const int nk = arch_[A_NRB], nd = arch_[A_NDIL];
    for (int i = (int)ups_.size() - 1; i >= 0; --i) {
      long widest = 0;
      for (int j = 0; j < nk; ++j) {
        const long hk = (arch_[A_RBK0 + j] - 1) / 2;
        long w = 0;
        for (int d = 0; d < nd; ++d) {
          w += hk * arch_[A_RBDIL0 + j * MAX_DIL + d];
          if (arch_[A_RESBLOCK] == 1) w += hk;
        }
        widest = std::max(widest, w);
      }
      r += widest;
      r = (r + ups_[i].rate - 1) / ups_[i].rate + 1;
    }
    halo_frames_ = (int)(r + 3);
  }
  launch::init_conv();
  launch::init_bf3();
  launch::init_front();
  launch::init_tail();
  probe_xcds();
  static const char* rows[] = {"text_encoder", "duration_predictor", "regulate+flow", "hifigan", "post+pcm"};
  for (auto n : rows) prof_.push_back(ProfileRow{n});
  PE_HIP(hipEventCreate(&ev0_));
  PE_HIP(hipEventCreate(&ev1_));
  PE_HIP(hipHostMalloc((void**)&h_frames_, 4096 * sizeof(int)));
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

void Engine::ensure_stage_a(int B, int Tmax) {
  if (!ffn_parts_ && H_ == 192 && FC_ % 48 == 0 && FC_ / 48 <= 16 && !enc_.empty() && enc_[0].f1p) {
    // partial outputs of the fused small-call FFN (kernels/ffn.h): [utterance][slice][192][columns], once
    PE_HIP(hipStreamSynchronize(stream_));
    PE_HIP(hipMalloc((void**)&ffn_parts_, (size_t)(FC_ / 48) * H_ * LaunchPolicy::ffn_max_cols * sizeof(float)));
  }
  const int Ts = rup(Tmax, 128);    // row strides are multiples of 128 columns (conv epilogue relies on it)
  bool grow = false;
  if ((size_t)B > capA_B_) { capA_B_ = B; grow = true; }
  // (growth re-creates every graph: grow the id capacity by at least half, so that texts of slowly increasing length
  // cost a few re-creations, not one per 128 ids)
  if ((size_t)Ts > capA_T_) { capA_T_ = std::max<size_t>(Ts, capA_T_ ? rup((int)(capA_T_ + capA_T_ / 2), 128) : 0); grow = true; }
  Ts_ = (int)capA_T_;
  const size_t Bc = capA_B_, T = capA_T_;
  auto carve = [&](char* base) -> size_t {
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
    return c.off + 256;
  };
  if (grow || !wsA_) {
    PE_HIP(hipStreamSynchronize(stream_));
    drop_graphs();
    if (wsA_) PE_HIP(hipFree(wsA_));
    if (wsB_) { PE_HIP(hipFree(wsB_)); wsB_ = nullptr; }   // stage-B sizes depend on the batch capacity
    capB_F_ = 0;
    wsA_bytes_ = carve(nullptr);
    PE_HIP(hipMalloc((void**)&wsA_, wsA_bytes_));
    carve(wsA_);
    if (h_in_cap_ < in_bytes_) {
      if (h_in_) PE_HIP(hipHostFree(h_in_));
      h_in_cap_ = in_bytes_;
      PE_HIP(hipHostMalloc((void**)&h_in_, h_in_cap_));
    }
  }
  carve(wsA_);
}

void Engine::ensure_stage_b(int Fmax) {
  const int Fs = rup(Fmax, 128);
  bool grow = false;
  if ((size_t)Fs > capB_F_) { capB_F_ = std::max<size_t>(Fs, capB_F_ ? rup((int)(capB_F_ + capB_F_ / 2), 128) : 0); grow = true; }
  Fs_ = (int)capB_F_;
  const size_t Bc = capA_B_, F = capB_F_;
  // largest [channels x length] activation of the generator
  size_t hmax = (size_t)U_ * F;
  {
    size_t L = F;
    for (auto& st : ups_) { L *= st.rate; hmax = std::max(hmax, (size_t)st.ch * L); }
  }
  Ss_ = (long)F * hop_;
  // per-utterance activations are addressed with 32-bit byte offsets (buffer descriptors)
  if (hmax * sizeof(float) >= (size_t)1 << 31 || (size_t)3 * H_ * F * sizeof(float) >= (size_t)1 << 31)
    throw std::runtime_error("utterance too long: a per-utterance activation would exceed 2 GiB");
  auto carve = [&](char* base) -> size_t {
    Carver c(base);
    zp_ = c.take<float>(Bc * C_ * F);
    fh_ = c.take<float>(Bc * H_ * F);
    facts_ = c.take<float>(Bc * H_ * F);
    fskip_ = c.take<float>(Bc * H_ * F);
    noise_z_ = c.take<float>(Bc * C_ * F);
    for (int i = 0; i < 5; ++i) hb_[i] = c.take<float>(Bc * hmax);
    zwin_ = c.take<float>((size_t)C_ * F);
    zp_keep_ = pol_.debug_keep ? c.take<float>(Bc * C_ * F) : nullptr;
    d_win_ = c.take<int>(4);
    audio_ = c.take<float>(Bc * (size_t)Ss_);
    pcm_ = c.take<int16_t>(Bc * (size_t)Ss_);
    return c.off + 256;
  };
  if (grow || !wsB_) {
    PE_HIP(hipStreamSynchronize(stream_));
    drop_graphs();
    if (wsB_) PE_HIP(hipFree(wsB_));
    wsB_bytes_ = carve(nullptr);
    PE_HIP(hipMalloc((void**)&wsB_, wsB_bytes_));
  }
  carve(wsB_);
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
  const size_t want = std::min<size_t>(Bc * hmax, (size_t)700 * 4096);
  if (pol_.group_mrf && side_floats_ < want) {
    PE_HIP(hipStreamSynchronize(stream_));
    drop_graphs();
    for (float*& sp : side_) { if (sp) PE_HIP(hipFree(sp)); sp = nullptr; }
    side_floats_ = want;
    for (float*& sp : side_) PE_HIP(hipMalloc((void**)&sp, side_floats_ * sizeof(float)));
  }
}

// ------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------

// A conv that may ride in a grouped split-K launch: few enough column tiles that the launch is latency- rather than
// throughput-bound, and a halo the 128-column slab covers.
bool Engine::can_group(const PackedConv& pc, int ncols) const {
  const long blocks = (long)((ncols + CFG_BN[pc.cfg] - 1) / CFG_BN[pc.cfg]) * (pc.mtiles * 32 / CFG_BM[pc.cfg]) * B_;
  return pol_.groupable(pc.gate, pc.up != 0, blocks, (pc.ntaps - 1) * pc.dil);
}
void Engine::group_begin() {
  grouping_ = true;
  group_.clear();
  group_flops_ = group_bytes_ = 0;
}
void Engine::group_end() {
  grouping_ = false;
  if (group_.empty()) return;
  // 4 waves per workgroup: 32 / 64 KB of slabs, so 4 / 2 workgroups share a CU and the <= 3 x ~420 workgroups of a
  // group run in one or two rounds (8 waves: 64 / 128 KB, five rounds, slower than one launch per conv). The convs that
  // need the 128-column slab go in a launch of their own: two resident workgroups per CU carry ~420 of them, not 1260.
  constexpr int NW = 4;
  // workgroups are dispatched in grid order (z slowest): the conv with the most taps goes first so that the longest
  // workgroups do not form the tail of the launch
  std::stable_sort(group_.begin(), group_.end(), [](const ConvP& a, const ConvP& b) { return a.ntaps > b.ntaps; });
  for (int wide = 0; wide < 2; ++wide) {
    ConvG g{};
    int n = 0, mt = 0;
    for (const ConvP& c : group_)
      if ((c.xhalo > 32) == (wide == 1)) {
        g.c[n++] = c;
        mt = std::max(mt, (c.rows + 31) / 32);
      }
    if (!n) continue;
    g.n = n;
    g.B = B_;
    const int XW = wide ? 128 : 64;
    const size_t smem = std::max<size_t>((size_t)NW * KC * XW, (size_t)NW * 16 * 64) * sizeof(float);
    const dim3 grid((group_ncols_ + 31) / 32, mt, n * B_);
    const double share = (double)n / (double)group_.size();
    const int kh = kbegin(prof_level_ >= 2 ? krow(wide ? "conv_splitk_group_kernel<4,2,128>" : "conv_splitk_group_kernel<4,2,64>") : 0,
                          group_flops_ * share, group_bytes_ * share);
    launch::conv_group(wide != 0, grid, smem, ls_, g);
    kend(kh);
  }
  group_.clear();
}

bool Engine::can_group_sum() const {
  if (group_.size() < 2 || group_.size() > 3) return false;
  const ConvP& a = group_[0];
  for (const ConvP& c : group_)
    if (c.rows != a.rows || c.Cin != a.Cin || c.nchunks != a.nchunks || c.x_bs != a.x_bs || c.x_cs != a.x_cs ||
        c.r_bs != a.r_bs || c.r_cs != a.r_cs || c.in_slope != a.in_slope || c.epi != EPI_RESADD || c.xhalo > 96)
      return false;
  return true;
}
void Engine::group_end_sum(View out, const float* bias_sum, float alpha) {
  grouping_ = false;
  if (!can_group_sum()) throw std::runtime_error("internal: convs do not fit a K-concatenated launch");
  ConvP q = group_[0];
  q.nseg = (int)group_.size();
  for (int i = 0; i < q.nseg; ++i) {
    q.seg_x[i] = group_[i].x; q.seg_wp[i] = group_[i].wp;
    q.seg_ntaps[i] = group_[i].ntaps; q.seg_dil[i] = group_[i].dil; q.seg_padl[i] = group_[i].padl;
  }
  q.res2 = group_[1].res;
  q.res3 = q.nseg > 2 ? group_[2].res : nullptr;
  q.bias = bias_sum;
  q.out = out.p; q.o_bs = out.bs; q.o_cs = out.cs;
  q.epi = EPI_ACCUM; q.mode = 3; q.alpha = alpha;      // (acc + bias + residuals) * alpha, nothing read back
  q.tgroups = 1;
  constexpr int NW = 4;
  const size_t smem = std::max<size_t>((size_t)NW * KC * 128, (size_t)NW * 16 * 64) * sizeof(float);
  const dim3 grid((group_ncols_ + 31) / 32, (q.rows + 31) / 32, B_);
  // (a 4-deep weight ring measured slower than 2: hifigan stage 0.345 vs 0.338 ms)
  // (a 16-deep ring -- a wave's whole K range in flight at entry, 256 registers, one workgroup per CU -- measured 59 us
  // against 30: profiles/r04_notes.md)
  const int kh = kbegin(prof_level_ >= 2 ? krow("conv_splitk_sum_kernel<4,2>") : 0, group_flops_, group_bytes_);
  launch::conv_group_sum(grid, smem, ls_, q);
  kend(kh);
  group_.clear();
}

// Which kernel family a conv launch goes to (the policy conv() applies).
int Engine::route(const PackedConv& pc, int ncols, int epi) const {
  const int cfg = pc.cfg;
  const long blocks = (long)((ncols + CFG_BN[cfg] - 1) / CFG_BN[cfg]) * (pc.mtiles * 32 / CFG_BM[cfg]) * B_;
  if (!pol_.splitk(blocks, (pc.ntaps - 1) * pc.dil)) return ROUTE_TILE;
  return pol_.splitk_16col(pc.wp16 != nullptr, epi == EPI_CONVT, pc.gate, pc.nchunks * pc.ntaps) ? ROUTE_SPLITK16 : ROUTE_SPLITK;
}
void Engine::conv(const PackedConv& pc, View x, View out, const int* lens, int len_mul, int Lmax, int epi,
                  float in_slope, int act, View res, View out2, int mode, float alpha, const float* bias2,
                  int bias2_bs) {
  ConvP p;
  p.x = x.p; p.x_bs = x.bs; p.x_cs = x.cs;
  p.wp = pc.wp; p.wp16 = pc.wp16; p.wpb = pc.wpb; p.bias = pc.bias;
  p.bias2 = bias2; p.bias2_bs = bias2_bs;
  p.out = out.p; p.o_bs = out.bs; p.o_cs = out.cs;
  p.res = res.p; p.r_bs = res.bs; p.r_cs = res.cs;
  p.out2 = out2.p; p.o2_bs = out2.bs; p.o2_cs = out2.cs;
  p.lens = lens; p.len_mul = len_mul;
  p.Cin = pc.Cin; p.rows = pc.rows; p.nchunks = pc.nchunks;
  p.ntaps = pc.ntaps; p.dil = pc.dil; p.padl = pc.padl;
  p.xhalo = (pc.ntaps - 1) * pc.dil;
  p.in_slope = in_slope;
  p.epi = epi; p.act = act;
  p.split = (epi == EPI_GATE) ? pc.split : (epi == EPI_WNRS ? (pc.rows > H_ ? H_ : 0) : 0);
  p.up = pc.up; p.padT = pc.padT;
  p.up_magic = pc.up ? (unsigned)((0x100000000ULL + pc.up - 1) / pc.up) : 0u;
  p.up_shift = -1;
  p.mode = mode; p.alpha = alpha;
  p.tpb = 1;
  p.tgroups = 1;
  if ((epi == EPI_GATE) != pc.gate) throw std::runtime_error("internal: gate epilogue/packing mismatch");

  const int ncols = (epi == EPI_CONVT) ? Lmax + 1 : Lmax;
  int cfg = pc.cfg;
  double kflops = 0, kbytes = 0;
  if (prof_level_ >= 2) {
    // algorithmic FLOPs of this launch: 2 * MACs per output column * valid columns over the batch
    const std::vector<int32_t>& lh = (lens == d_tlens_) ? tlens_h_ : frames_h_;
    double cols = 0;
    for (int b = 0; b < B_; ++b) cols += (double)lh[b] * len_mul;
    kflops = 2.0 * pc.macs_per_col * cols;
    // algorithmic bytes: every input channel and every output row once per column, residual / read-modify-write
    // operands once more each, the weights once per launch
    const bool rd_res = epi == EPI_RESADD || epi == EPI_ACCUM;
    const bool rd_old = epi == EPI_SUBFROM || epi == EPI_WNRS || (epi == EPI_ACCUM && (mode == 1 || mode == 2));
    const double out_rows = epi == EPI_GATE ? pc.split : pc.rows;
    kbytes = 4.0 * (cols * (pc.Cin + out_rows * (1 + (rd_res ? 1 : 0) + (rd_old ? 1 : 0))) +
                    (double)pc.rows * pc.Cin * pc.ntaps);
  }
  const long blocks = (long)((ncols + CFG_BN[cfg] - 1) / CFG_BN[cfg]) * (pc.mtiles * 32 / CFG_BM[cfg]) * B_;
  if (grouping_) {
    if (!can_group(pc, ncols) || epi == EPI_CONVT || epi == EPI_GATE || group_.size() >= 3 ||
        (!group_.empty() && group_ncols_ != ncols))
      throw std::runtime_error("internal: conv does not fit a grouped launch");
    p.tgroups = 1;
    group_.push_back(p);
    group_ncols_ = ncols;
    group_flops_ += kflops;
    group_bytes_ += kbytes;
    return;
  }
  if (pol_.splitk(blocks, p.xhalo)) {
    // few columns (one utterance through encoder / duration predictor / flow): split K across the waves
    const int MT = pc.gate ? 2 : 1;
    // waves per workgroup: 4 / 8 take whole chunks; the WN gate conv (6 chunks x 5 taps, two M tiles per wave)
    // goes to 12 waves whose two halves split the taps: 15.8 -> 14.2 us per launch at B=1. Measured and not used:
    // the same 12 waves for conv_pre (6 x 7) and FFN conv_2 (24 chunks) are slower than 8.
    const int units = pc.nchunks * pc.ntaps;
    int NW = pc.nchunks >= 5 ? 8 : 4;
    p.tgroups = 1;
    if (pol_.splitk_12wave(pc.gate, units, pc.nchunks, pc.ntaps)) {
      NW = 12;
      p.tgroups = pc.nchunks <= 6 ? 2 : 1;
    }
    dim3 grid((ncols + 31) / 32, pc.mtiles / MT, B_);
    const size_t smem = std::max<size_t>((size_t)NW * KC * 64, (size_t)NW * MT * 16 * 64) * sizeof(float);
    const bool k16 = pol_.splitk_16col(pc.wp16 != nullptr, epi == EPI_CONVT, pc.gate, units);
    // profile rows carry the instantiation exactly as rocprofv3 prints it (minus spaces)
    int kh = -1;
    if (prof_level_ >= 2) {
      char nm[96];
      if (k16) snprintf(nm, sizeof(nm), "conv_splitk16_kernel<%s>", pc.gate ? "true,12,2" : "false,8,4");
      else snprintf(nm, sizeof(nm), "conv_splitk_kernel<%d,%s,%d,%d>", MT, pc.gate ? "true" : "false", NW,
                    pc.gate ? (NW == 12 ? 2 : 3) : 4);
      kh = kbegin(krow(std::string(nm)), kflops, kbytes);
    }
    // MFMA-pipe bound inside the workgroup (>= 24 chunk-tap units) although most CUs idle: 16 output columns
    if (k16) {
      dim3 grid16((ncols + 15) / 16, pc.mtiles / MT, B_);
      const int nw = pc.gate ? 12 : 8;
      p.tgroups = pc.gate ? (pc.nchunks <= 6 ? 2 : 1) : 1;
      launch::conv_splitk16(pc.gate, grid16, (size_t)nw * KC * 64 * sizeof(float), ls_, p);
      kend(kh);
      return;
    }
    launch::conv_splitk(pc.gate, NW, grid, smem, ls_, p);
    kend(kh);
    return;
  }
  if (matrix_bf3_ && pc.wpb && p.xhalo <= 128) {
    // matrix mode bf16x3: 128 x 128 / 64 x 128 / 32 x 256 tiles (two 32x32 MFMA tiles per wave at least: the bf16 pipe
    // is fast enough that operand traffic per MFMA matters more than workgroup count)
    static const int BF3_BM[] = {128, 64, 32}, BF3_BN[] = {128, 128, 256};
    static const char* bnames[] = {"2,2,2,2", "2,2,1,2", "1,4,1,2"};
    const int bc = cfg == CFG_A ? 0 : (cfg == CFG_B ? 1 : 2);
    if (pc.gate && bc == 2) throw std::runtime_error("internal: gate conv packed for 32-row blocks");
    const int BM = BF3_BM[bc], BN = BF3_BN[bc];
    const int HALO = p.xhalo <= 64 ? 64 : 128;
    const int nbuf = pc.nchunks == 1 ? 1 : 2;
    const size_t smem = (size_t)nbuf * 128 * ((BN + HALO + 63) / 64 * 64);       // [2 parts][4 k groups][XS] x 16 B
    dim3 grid((ncols + BN - 1) / BN, pc.mtiles * 32 / BM, B_);
    int kh = -1;
    if (prof_level_ >= 2) {
      char nm[96];
      snprintf(nm, sizeof(nm), "conv_bf3_kernel<%s,%s,%d>", (pc.gate && bc == 1) ? "1,4,2,1" : bnames[bc],
               pc.gate ? "true" : "false", HALO);
      kh = kbegin(krow(std::string(nm)), kflops, kbytes);
    }
    launch::conv_bf3(bc, pc.gate, HALO, grid, smem, ls_, p);
    kend(kh);
    return;
  }
  // 32x32 wave tiles everywhere (64x64 / 32x128 workgroup tiles; the gate form pairs two row tiles per wave): measured in
  // rounds 1-3 against 128x128, 64x128 and 256-column tiles at every batch size -- latency here is hidden across
  // workgroups, occupancy beats register reuse (profiles/r01_ablation.txt, r02_notes.md); the larger instantiations are gone
  if (cfg == CFG_A) cfg = pc.gate ? CFG_G : CFG_S;
  else if (cfg == CFG_B && !pc.gate) cfg = CFG_S;
  const int BM = CFG_BM[cfg], BN = CFG_BN[cfg];
  const int ntile = (ncols + BN - 1) / BN, mblocks = pc.mtiles * 32 / BM;
  // Column tiles walked by one workgroup. Measured on MI355X (profiles/r01_tpb_sweep.txt): with 2-3
  // workgroups resident per CU, one tile per workgroup (latency hidden across workgroups) beats walking
  // several tiles with the in-kernel prefetch pipeline at every batch size, so the default is 1; the
  // multi-tile path stays available through PIPER_HIP_TPB.
  const int tpb = pol_.tiles_per_workgroup();
  p.tpb = tpb;
  dim3 grid((ntile + tpb - 1) / tpb, mblocks, B_);
  if (p.xhalo > 128) throw std::runtime_error("conv halo (kernel-1)*dilation > 128 is not supported");
  // one x slab when the workgroup only ever stages one (single chunk, single tile): more workgroups per CU
  const int nbuf = (tpb == 1 && pc.nchunks == 1) ? 1 : 2;
  const int HALO = p.xhalo <= 64 ? 64 : 128;
  const size_t smem = (size_t)nbuf * KC * ((BN + HALO + 63) / 64 * 64) * sizeof(float);
  // polyphase up-conv: the tile leaves through LDS as rows of consecutive output samples (conv_mfma.h) when the stride is
  // a power of two that divides the tile's rows, one tile per workgroup, and the slab area holds BM x BN + padding
  p.up_shift = -1;
  // (measured, profiles/r04_notes.md: stride 8 -6 % per launch at batch; strides 4 and 2 gain nothing or lose -- their
  // LDS writes are 4- / 2-way bank conflicts for a store pattern the L2 was already merging; PIPER_HIP_CONVT_LDS=2 forces it)
  if (epi == EPI_CONVT && pc.up >= 2 &&
      pol_.convt_through_lds(pc.up, tpb, BM, ((size_t)BM * BN + (size_t)(BM / pc.up) * 4) * sizeof(float), smem)) {
    int sh = 0;
    while ((1 << sh) < pc.up) ++sh;
    p.up_shift = sh;
  }
  static const char* knames[] = {"2,2,2,2,8", "1,4,2,1,16", "1,4,1,1,16", "2,2,1,1,16", "2,2,2,1,16", "1,4,1,2,16", "1,4,2,2,8"};
  int kh = -1;
  if (prof_level_ >= 2) {
    char nm[96];
    int n = snprintf(nm, sizeof(nm), "conv_mfma_kernel<%s,%s,%d>", knames[cfg], pc.gate ? "true" : "false", HALO);
    // tuning aid (PIPER_HIP_PROF_SITES=1): one profile row per conv SHAPE instead of per instantiation
    if (pol_.prof_sites) snprintf(nm + n, sizeof(nm) - n, "|%dx%dx%d d%d e%d L%d", pc.rows, pc.Cin, pc.ntaps, pc.dil, epi, len_mul);
    kh = kbegin(krow(std::string(nm)), kflops, kbytes);
  }
  launch::conv_tile(cfg, pc.gate, HALO, grid, smem, ls_, p);
  kend(kh);
}

// Fused MRF stage (kernels/mrf.h): flattens the resblocks of a <= 64-channel stage into phases (one per conv) and writes
// the weights as one stream in execution order: per phase its (chunk, tap) steps, chunk-major, each step =
// [16-row tile][q][lane][4] with lane -> (row = lane & 15, k = lane >> 4), float4 element jj of group q = k-step 4q + jj =
// input channel chunk*32 + 4*(4q + jj) + k. ResBlock2 (modules.py:355-364): x <- x + c_d(lrelu(x)); ResBlock1 (:301-314):
// x <- x + c2_d(lrelu(c1_d(lrelu(x)))).
void Engine::build_mrf(UpStage& st) {
  const int ch = st.ch;
  if (!pol_.mrf_build(ch) || st.rb_host.empty()) return;
  const int CP = ch <= 32 ? 32 : 64, MS = CP / 16, NCH = CP / KC, STEPF = MS * 512;
  const bool rb1 = arch_[A_RESBLOCK] == 1;
  int hx = 0;                              // halo of the stage = the widest resblock chain
  for (auto& hv : st.rb_host) {
    int e = 0;
    for (auto& h : hv) e += h.dil * (h.k - 1) / 2;
    hx = std::max(hx, e);
  }
  const int hxa = rup(hx, 16);
  std::vector<MrfPhase> phases;
  std::vector<float> wstream;
  for (size_t j = 0; j < st.rb_host.size(); ++j) {
    auto& hv = st.rb_host[j];
    const int n = (int)hv.size();
    if (n == 0 || (rb1 && (n & 1))) return;
    int e = 0;
    for (auto& h : hv) {
      if (!(h.k & 1) || h.ci != ch || h.co != ch) return;
      e += h.dil * (h.k - 1) / 2;
    }
    for (int i = 0; i < n; ++i) {
      const auto& h = hv[i];
      e -= h.dil * (h.k - 1) / 2;
      MrfPhase P{};
      P.bias = h.bias; P.ntaps = h.k; P.dil = h.dil; P.e = e;
      const bool last = i == n - 1;
      if (rb1) {
        if (!(i & 1)) { P.src = 0; P.dst = 1; P.flags = 0; }
        else { P.src = 1; P.dst = last ? -1 : 0; P.flags = MRF_RES | MRF_KEEP; }
      } else {
        P.src = i == 0 ? 0 : 1; P.dst = last ? -1 : 1; P.flags = MRF_RES | MRF_KEEP;
        if (n > 2) return;             // a longer ResBlock2 chain would need ping-pong chain buffers
      }
      if (last) P.flags |= MRF_FINAL;
      if (i == 0) P.flags |= MRF_INIT | ((rb1 && j > 0) ? MRF_RESTAGE : 0);
      const int nsteps = NCH * h.k;
      const size_t w0 = wstream.size();
      wstream.resize(w0 + (size_t)nsteps * STEPF, 0.f);
      for (int step = 0; step < (skeleton_ ? 0 : nsteps); ++step) {
        const int c = step / h.k, tap = step % h.k;
        for (int ms = 0; ms < MS; ++ms)
          for (int q = 0; q < 2; ++q)
            for (int lane = 0; lane < 64; ++lane)
              for (int jj = 0; jj < 4; ++jj) {
                const int row = ms * 16 + (lane & 15), ci = c * KC + 4 * (4 * q + jj) + (lane >> 4);
                if (row < ch && ci < ch)
                  wstream[w0 + ((size_t)(step * MS + ms) * 2 + q) * 256 + lane * 4 + jj] = h.w[((size_t)row * ch + ci) * h.k + tap];
              }
      }
      phases.push_back(P);
    }
  }
  if ((int)phases.size() > MRF_MAXPH) return;
  {
    // some N = 16 * NCG * OU must fit the kernel's fixed row stride and its halo-unit capacity
    const int NCG = CP == 32 ? 8 : 4, HU = CP == 32 ? 1 : 2, n1 = 16 * NCG;
    const int nh = (hxa + n1 + hx + 15) / 16 - (hxa - hx) / 16 - n1 / 16;
    if (hxa + n1 + hx > mrf_ws(CP) || nh > NCG * HU) return;
  }
  void* d = nullptr;
  PE_HIP(hipMalloc(&d, phases.size() * sizeof(MrfPhase)));
  PE_HIP(hipMemcpy(d, phases.data(), phases.size() * sizeof(MrfPhase), hipMemcpyHostToDevice));
  owned_.push_back(d);
  st.mrf_phases = d;
  st.mrf_w = dev_alloc(wstream.size(), wstream.data());      // weights: in the arena (travels with the broadcast)
  st.mrf_wfloats = (int)wstream.size();
  st.mrf_cp = CP;
  st.mrf_ph = phases;
  st.mrf_hx = hx;
  st.mrf_rb1 = rb1;
  st.mrf_ok = true;
}

// mrf_kernel launch: the window geometry -- output columns per workgroup N = 16 * NCG * OU -- is chosen here. Few
// utterances: the launch is one or two rounds of workgroups over the 256 CUs, so the workgroup count should sit just under
// a multiple of 256 and a workgroup should be short; batches: many rounds, so large N (less halo recompute, fewer
// prologues) wins. Cost model: rounds x (MFMA columns of one workgroup incl. recompute + a fixed prologue / epilogue).
bool Engine::mrf_geo(const UpStage& st, int len_mul, bool tail, MrfGeo& best) const {
  const int CP = st.mrf_cp, NCG = CP == 32 ? 8 : 4, HU = CP == 32 ? 1 : 2;
  const int OUMAX = CP == 32 ? 4 : 3;      // 32 channels: 4 units per wave (N = 512) fill the 160 KB of LDS
  const int hx = st.mrf_hx, hxa = rup(hx, 16);
  auto geo = [&](int ou, MrfGeo& g) {
    g.ou = ou; g.N = 16 * NCG * ou;
    if (hxa + g.N + hx > mrf_ws(CP, ou)) return false;
    g.cu_lo = (hxa - hx) / 16; g.cu_hi = (hxa + g.N + hx + 15) / 16;
    g.nleft = hxa / 16 - g.cu_lo; g.nhalo = g.cu_hi - g.cu_lo - g.N / 16;
    return g.nhalo <= NCG * HU;
  };
  best = MrfGeo{};
  double best_cost = 0;
  MrfGeo forced;
  const int force = (pol_.mrf_ou >= 1 && pol_.mrf_ou <= OUMAX && geo((int)pol_.mrf_ou, forced)) ? (int)pol_.mrf_ou : 0;     // (tests / A-B; ignored when it does not fit)
  for (int ou = 1; ou <= OUMAX; ++ou) {
    MrfGeo g;
    if ((force && ou != force) || !geo(ou, g)) continue;
    double wgs = 0;
    const int stride = tail ? g.N - (POST_K - 1) : g.N;
    for (int b = 0; b < B_; ++b) wgs += (double)(((long)frames_h_[b] * len_mul + stride - 1) / stride);
    double work = 0, taps = 0;
    for (auto& P : st.mrf_ph) {
      const int lo = (hxa - P.e) / 16, hi = (hxa + g.N + P.e + 15) / 16;
      work += (double)(hi - lo) * 16 * P.ntaps;
      taps += P.ntaps;
    }
    double cost = std::ceil(wgs / 256.0) * (work + 16.0 * taps);      // prologue + epilogue ~ 16 columns' worth
    // 4 units per wave run at the register limit (a few spilled VGPRs): measured 3-5 % slower per column than 3 units at
    // batch (B=16: 1108 vs 1082 us, B=64: 4.29 vs 4.25 ms), but one round instead of two for a single utterance's last
    // stage (B=1: 84.6 vs 93.4 us) -- profiles/r03_notes.md
    if (ou == 4) cost *= 1.06;
    if (!best.ou || cost < best_cost) { best = g; best_cost = cost; }
  }
  return best.ou != 0;
}

void Engine::mrf(const UpStage& st, View x, View out, const int* lens, int len_mul, int Lmax, bool tail) {
  const int CP = st.mrf_cp, HU = CP == 32 ? 1 : 2;
  const int hx = st.mrf_hx, hxa = rup(hx, 16);
  MrfGeo best;
  if (!mrf_geo(st, len_mul, tail, best)) throw std::runtime_error("internal: no mrf_kernel geometry for this stage");
  MrfP p{};
  p.x = x.p; p.x_bs = x.bs; p.x_cs = x.cs;
  p.out = out.p; p.o_bs = out.bs; p.o_cs = out.cs;
  p.lens = lens; p.len_mul = len_mul;
  p.phases = static_cast<const MrfPhase*>(st.mrf_phases); p.nphases = (int)st.mrf_ph.size();
  p.wstream = st.mrf_w; p.wfloats = st.mrf_wfloats;
  p.C = st.ch; p.N = best.N; p.wcols = hxa + best.N + hx; p.hxa = hxa; p.cu_lo = best.cu_lo; p.cu_hi = best.cu_hi;
  p.nleft = best.nleft; p.nhalo = best.nhalo;
  p.slope = 0.1f;                            // modules.py LRELU_SLOPE
  p.alpha = 1.0f / (float)st.rb.size();
  p.stride = best.N; p.n0off = 0;
  p.post_w = nullptr; p.audio = nullptr; p.a_bs = 0; p.absmax = nullptr; p.post_slope = 0.01f;
  if (tail) {       // generator tail inside the stage kernel: windows overlap by the conv_post taps
    p.stride = best.N - (POST_K - 1); p.n0off = (POST_K - 1) / 2;
    p.post_w = post_w_; p.audio = audio_; p.a_bs = Ss_; p.absmax = absmax_;
  }
  double kflops = 0, kbytes = 0;
  if (prof_level_ >= 2) {
    double cols = 0;
    for (int b = 0; b < B_; ++b) cols += (double)frames_h_[b] * len_mul;
    double macs = 0;
    for (auto& cv : st.rb)
      for (auto& c : cv) macs += c.macs_per_col;
    kflops = 2.0 * macs * cols;
    kbytes = 8.0 * st.ch * cols + 4.0 * st.mrf_wfloats;      // one read of x, one write of the mean, the weights once
    if (tail) {
      kflops += 2.0 * cols * st.ch * POST_K;
      kbytes = 4.0 * (st.ch + 1) * cols + 4.0 * st.mrf_wfloats;   // one read of x, one write of the waveform
    }
  }
  dim3 grid((Lmax + p.stride - 1) / p.stride, B_);
  char nm[64];
  snprintf(nm, sizeof(nm), "mrf_kernel<%d,%d,%d>", CP, best.ou, HU);
  const int kh = prof_level_ >= 2 ? kbegin(krow(std::string(nm)), kflops, kbytes) : -1;
  launch::mrf(CP, best.ou, grid, ls_, p);
  kend(kh);
}

void Engine::layer_norm(View in, View out, const float* g, const float* b, int C, const int* lens, int Lmax) {
  LnP p;
  p.in = in.p; p.i_bs = in.bs; p.i_cs = in.cs;
  p.out = out.p; p.o_bs = out.bs; p.o_cs = out.cs;
  p.gamma = g; p.beta = b;
  p.lens = lens; p.C = C;
  if (C > LN_COLS * 32) throw std::runtime_error("LayerNorm over more than 256 channels is not supported");
  dim3 grid((Lmax + LN_COLS - 1) / LN_COLS, B_);
  const int kh = kbegin(prof_level_ >= 2 ? krow("ln_kernel") : 0, 0.0, 4.0 * 2.0 * C * (lens == d_tlens_ ? cols_ids_ : cols_frames_));
  launch::layer_norm(grid, stream_, p);
  kend(kh);
}

// DDSConv.forward (modules.py:117-129): one fused launch per layer (dds_layer16_kernel), ping-ponging between
// `out` and `tmp` so that the last layer lands in `out`; `in` must not alias the first layer's target.
void Engine::dds_params(const DdsW& d, View in, View out, View tmp, const DdsOpt* opt, std::vector<DdsP>& list) {
  int dil = 1;
  const int n = (int)d.c1x1.size();
  View cur = in;
  for (int i = 0; i < n; ++i) {
    const View dst = ((n - 1 - i) & 1) ? tmp : out;
    if (dst.p == cur.p) throw std::runtime_error("internal: DDSConv buffer aliasing");
    DdsP p{};
    if (opt && i == 0 && opt->pre_z) {
      p.pre_z = opt->pre_z; p.pre_z_bs = opt->pre_z_bs; p.pre_w = opt->pre_w; p.pre_b = opt->pre_b;
    }
    p.z_scale = opt ? opt->z_scale : 1.f;
    if (opt && i == n - 1 && opt->post_w16) {
      p.post_w16 = opt->post_w16; p.post_w4 = w4_of(opt->post_w16); p.post_bias = opt->post_bias; p.post_rows = opt->post_rows;
      p.post_out = opt->post_out.p; p.po_bs = opt->post_out.bs; p.po_cs = opt->post_out.cs;
      p.zin = opt->zin; p.zin_bs = opt->zin_bs; p.z_cs = opt->z_cs; p.c0 = opt->c0; p.c1 = opt->c1;
      p.zout = opt->zout; p.zout_bs = opt->zout_bs;
      p.inv_sqrt_h = 1.0f / std::sqrt((float)H_);
    }
    p.x = cur.p; p.x_bs = cur.bs; p.x_cs = cur.cs;
    p.out = dst.p; p.o_bs = dst.bs; p.o_cs = dst.cs;
    p.dw_w = d.dw_w[i]; p.dw_b = d.dw_b[i]; p.dw_k = ksz_; p.dw_dil = dil;
    p.g1 = d.g1[i]; p.b1 = d.b1[i]; p.g2 = d.g2[i]; p.b2 = d.b2[i];
    p.bias = d.c1x1[i].bias;
    p.wp16 = d.w16[i];
    p.wp4 = w4_of(d.w16[i]);
    p.nchunks = d.c1x1[i].nchunks;
    p.lens = d_tlens_; p.H = H_;
    list.push_back(p);
    dil *= ksz_;
    cur = dst;
  }
}

// algorithmic bytes of one DDSConv layer launch: x in, out (+ the fused 1x1 conv's rows), the 1x1 matrix (+ the fused one)
double Engine::dds_bytes(const DdsP& p) const {
  return 4.0 * (cols_ids_ * (2.0 * H_ + p.post_rows) + (double)H_ * H_ + (double)p.post_rows * H_);
}

void Engine::dds(const DdsW& d, View in, View out, View tmp, const DdsOpt* opt) {
  std::vector<DdsP> list;
  dds_params(d, in, out, tmp, opt, list);
  // Small calls of the 192-channel voices: 4-column workgroups on 4x the CUs (kernels/dds4.h). Every layer of the chain
  // needs its matrices in the 4x4x1 order; the form reads 4x the weight bytes, hence the column limit.
  bool four = H_ == 192 && ksz_ <= 3 && pol_.chain4((long)B_ * Tg_);
  for (const DdsP& p : list) four = four && p.wp4 && (!p.post_w16 || p.post_w4);
  for (const DdsP& p : list) {
    if (four) {
      const int kh4 = kbegin(prof_level_ >= 2 ? krow("dds_layer4_kernel") : 0, 0.0, dds_bytes(p));
      DdsP p4 = p;
      p4.xcd = xcd_period_;
      launch::dds_layer4(dim3((Tg_ + 3) / 4, B_), col4_smem(), stream_, p4);
      kend(kh4);
      continue;
    }
    const int kh = kbegin(prof_level_ >= 2 ? krow(p.nchunks == 3 ? "dds_layer16_kernel<3>" : p.nchunks == 6 ? "dds_layer16_kernel<6>"
                                                                                       : "dds_layer16_kernel<8>") : 0, 0.0, dds_bytes(p));
    const dim3 grid16((Tg_ + 15) / 16, B_);
    const size_t smem16 = ((size_t)2 * p.nchunks * 32 * 16 + 16 * 16) * sizeof(float);
    // <3> / <6> are compiled for exactly 96 / 192 padded channels; <8> takes any width up to 256
    launch::dds_layer(p.nchunks, grid16, smem16, stream_, p);
    kend(kh);
  }
}

void Engine::set_profile(int level) {
  prof_level_ = level;
  prof_on_ = level > 0;
}
int Engine::krow(const char* name) {
  for (size_t i = 5; i < prof_.size(); ++i)
    if (!strcmp(prof_[i].name, name)) return (int)i;
  prof_.push_back(ProfileRow{name});
  return (int)prof_.size() - 1;
}
int Engine::krow(const std::string& name) {
  for (size_t i = 5; i < prof_.size(); ++i)
    if (name == prof_[i].name) return (int)i;
  names_.push_back(name);
  prof_.push_back(ProfileRow{names_.back().c_str()});
  return (int)prof_.size() - 1;
}
void Engine::lngemm(View y, const float* g, const float* b, View x, const float* w16, const float* bias, int rows,
                    View out, int T, double flops, const float* parts, int nparts, const float* pbias) {
  // algorithmic bytes: y in, LN(y) out, the conv's rows out, the FFN's partial outputs in; weights once
  const double kbytes = 4.0 * (cols_ids_ * (2.0 * H_ + rows + (parts ? (double)nparts * H_ : 0.0)) + (double)rows * H_);
  LnGemmP p{};
  p.in = y.p; p.in_bs = y.bs; p.in_cs = y.cs;
  p.gamma = g; p.beta = b;
  p.xout = x.p; p.x_bs = x.bs; p.x_cs = x.cs;
  p.w16 = w16; p.bias = bias; p.rows = rows;
  p.out = out.p; p.o_bs = out.bs; p.o_cs = out.cs;
  p.lens = d_tlens_;
  // small calls: 4-column workgroups on the 4x4x1 MFMA (kernels/col4.h), like Engine::dds
  if (parts) {        // y = (View y: the residual) + pbias + the fused FFN's partial outputs (ffn_kernel)
    const int Tp = rup(T, 4);
    p.parts = parts; p.nparts = nparts; p.pbias = pbias;
    p.p_bs = (long)nparts * H_ * Tp;
    if (!(pol_.chain4((long)B_ * T) && w4_of(w16))) throw std::runtime_error("internal: FFN partials without the 4-column consumer");
  }
  if (const float* w4 = pol_.chain4((long)B_ * T) ? w4_of(w16) : nullptr) {
    p.w16 = w4;
    p.xcd = xcd_period_;
    const int kh4 = kbegin(prof_level_ >= 2 ? krow("lngemm4_kernel") : 0, flops, kbytes);
    launch::lngemm4(dim3((T + 3) / 4, B_, (rows + 191) / 192), col4_smem(), stream_, p);
    kend(kh4);
    return;
  }
  const int kh = kbegin(prof_level_ >= 2 ? krow("lngemm_kernel<6>") : 0, flops, kbytes);
  const size_t smem = ((size_t)192 * 16 + 16 * 16) * sizeof(float);
  launch::lngemm(dim3((T + 15) / 16, B_, (rows + 191) / 192), smem, stream_, p);
  kend(kh);
}

// A plain 1x1 conv over 192 input channels of a small call on 4-column workgroups (colchain4_kernel mode 3); false: the
// caller launches the conv kernel instead. `w16`: the conv's pack16 matrix (its pack4 twin is looked up).
bool Engine::conv1x1_col4(const float* w16, const float* bias, int rows, View in, View out, const int* lens, int B, int Lmax,
                          double flops, const float* bias2, long bias2_bs, const float* w4direct, int kin, long max_cols) {
  const bool small = pol_.chain4((long)B * Lmax, max_cols);
  const float* w4 = (H_ == 192 && small) ? (w4direct ? w4direct : w4_of(w16)) : nullptr;
  if (!w4) return false;
  ColP cp{};
  cp.in1 = in.p; cp.in1_bs = in.bs; cp.in1_cs = in.cs; cp.K1 = kin;
  cp.w1 = w4; cp.b1 = bias; cp.rows1 = rows;
  cp.mode = 3;
  cp.xcd = xcd_period_;
  cp.res = bias2; cp.res_bs = bias2_bs;
  cp.out = out.p; cp.out_bs = out.bs; cp.out_cs = out.cs;
  cp.lens = lens;
  const double cols = lens == d_tlens_ ? cols_ids_ : cols_frames_;
  const int kh4 = kbegin(prof_level_ >= 2 ? krow("colchain4_kernel") : 0, flops, 4.0 * (cols * (kin + rows) + (double)rows * kin));
  launch::colchain4(dim3((Lmax + 3) / 4, B, (rows + 191) / 192), col4_smem(), stream_, cp);
  kend(kh4);
  return true;
}

void Engine::colchain(const ColP& p, int B, int Lmax, double flops) {
  // algorithmic bytes: GEMM input, the residual / x1 read and written, the second GEMM's output; weights once
  const double cols = p.lens == d_tlens_ ? cols_ids_ : cols_frames_;
  const double kbytes = 4.0 * (cols * (p.K1 + 2.0 * p.rows1 + (p.w2 ? p.rows2 : 0)) + (double)p.rows1 * p.K1 +
                               (p.w2 ? (double)p.rows2 * p.rows1 : 0.0));
  // mode 1 runs on frames (coupling post + pre), mode 0 on ids: separate column limits (profiles/r03_notes.md)
  if ((p.mode == 1 ? pol_.chain4_frames((long)B * Lmax) : pol_.chain4((long)B * Lmax)) && p.K1 == 192 && (p.mode == 0 ? p.rows1 == 192 : (p.rows1 == 96 && (!p.w2 || p.rows2 <= 192)))) {
    const float* w1 = w4_of(p.w1);
    const float* w2 = p.w2 ? w4_of(p.w2) : nullptr;
    if (w1 && (!p.w2 || w2)) {
      ColP q = p;
      q.w1 = w1; q.w2 = w2;
      q.xcd = xcd_period_;
      const int kh4 = kbegin(prof_level_ >= 2 ? krow("colchain4_kernel") : 0, flops, kbytes);
      launch::colchain4(dim3((Lmax + 3) / 4, B), col4_smem(), stream_, q);
      kend(kh4);
      return;
    }
  }
  const int kh = kbegin(prof_level_ >= 2 ? krow("colchain_kernel<6>") : 0, flops, kbytes);
  const size_t smem = ((size_t)2 * 6 * 32 * 16 + 16 * 16) * sizeof(float);
  launch::colchain(dim3((Lmax + 15) / 16, B), smem, stream_, p);
  kend(kh);
}

int Engine::kbegin(int row, double flops, double bytes) {
  if (prof_level_ < 2) return -1;
  hipEvent_t a, b;
  if (ev_pool_.size() >= 2) {
    a = ev_pool_.back(); ev_pool_.pop_back();
    b = ev_pool_.back(); ev_pool_.pop_back();
  } else {
    PE_HIP(hipEventCreate(&a));
    PE_HIP(hipEventCreate(&b));
  }
  PE_HIP(hipEventRecord(a, ls_));
  kev_.push_back(KEvent{row, flops, bytes, a, b});
  return (int)kev_.size() - 1;
}
void Engine::kend(int h) {
  if (h >= 0) PE_HIP(hipEventRecord(kev_[h].b, ls_));
}
const std::vector<ProfileRow>& Engine::profile() {
  if (!kev_.empty()) {
    PE_HIP(hipStreamSynchronize(stream_));
    for (auto& k : kev_) {
      float ms = 0;
      PE_HIP(hipEventElapsedTime(&ms, k.a, k.b));
      prof_[k.row].ms += ms;
      prof_[k.row].flops += k.flops;
      prof_[k.row].bytes += k.bytes;
      prof_[k.row].launches += 1;
      ev_pool_.push_back(k.a);
      ev_pool_.push_back(k.b);
    }
    kev_.clear();
  }
  return prof_;
}
void Engine::reset_profile() {
  profile();
  for (auto& r : prof_) { r.ms = 0; r.flops = 0; r.launches = 0; r.bytes = 0; }
}
void Engine::prof_begin() {
  if (prof_on_) PE_HIP(hipEventRecord(ev0_, stream_));
}
void Engine::prof_end(int row, double flops) {
  if (!prof_on_) return;
  PE_HIP(hipEventRecord(ev1_, stream_));
  PE_HIP(hipEventSynchronize(ev1_));
  float ms = 0;
  PE_HIP(hipEventElapsedTime(&ms, ev0_, ev1_));
  prof_[row].ms += ms;
  prof_[row].flops += flops;
  prof_[row].launches += 1;
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
  // {seed, runs so far}: the first kernel of every run() advances the counter on the device (embed_kernel)
  hr[0] = seed_; hr[1] = call_; hr[2] = hr[3] = 0;
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

// Whether stage A of the CURRENT call (B_, Tg_, tlens_h_) runs the encoder FFNs as ffn_kernel launches, and which of the
// two ping-pong buffers then holds the encoder output: functions of the call alone, so that a replayed graph and the code
// that captured it agree (the fused path swaps x / y once per layer; debug_tensor("x_enc") reads the result).
bool Engine::stage_a_ffn_fused() const {
  double tsum = 0;
  for (int b = 0; b < B_; ++b) tsum += tlens_h_[b];
  const bool chain_q = pol_.chain16(tsum, false, H_, 96);
  bool f = pol_.ffn && chain_q && pol_.chain4((long)B_ * Tg_) && ffn_parts_ && (long)B_ * rup(Tg_, 4) <= LaunchPolicy::ffn_max_cols &&
           FC_ % 48 == 0 && FC_ / 48 <= 16 && w4_of(enc_proj16_);
  for (auto& e : enc_) f = f && e.f1p && e.f2p && w4_of(e.qkv16);
  return f;
}
float* Engine::stage_a_enc_out() const { return (stage_a_ffn_fused() && (enc_.size() & 1)) ? y_ : x_; }

// Everything up to the frame counts: speaker vectors, text encoder, duration predictor, durations.
// Grids are sized by the bucketed maximum length Tg_; kernels bound themselves by the device-side
// per-utterance lengths, so the same captured graph serves every batch of that bucket.
void Engine::issue_stage_a() {
  const int B = B_, Ts = Ts_, T = Tg_;
  const long bsH = (long)H_ * Ts;
  auto V = [&](float* p, int ch) { return View{p, (long)ch * Ts, Ts}; };
  View x = V(x_, H_), y = V(y_, H_);
  const View qkv = V(qkv_, 3 * H_), att = V(att_, H_), ffh = V(ffh_, FC_),
             stats = V(stats_, 2 * C_), xg = V(xg_, H_), dh = V(dh_, H_), dy = V(dy_, H_), dy2 = V(dy2_, H_),
             hproj = V(hproj_, 32);
  const View none{nullptr, 0, 0};
  (void)bsH;
  double tsum = 0;
  for (int b = 0; b < B; ++b) tsum += tlens_h_[b];
  cols_ids_ = tsum;

  // ================= speaker conditioning vectors
  const float* cb_dp = nullptr;
  if (nspk_ > 1) {
    auto cond = [&](const CondW& c, int off) {
      PE_LAUNCH_K("cond_kernel", launch::cond(dim3((c.rows + 127) / 128, B), stream_, emb_g_, gin_, d_sids_, c.w, c.b, c.rows, cond_ + off, cond_bs_));
    };
    cond(cond_dp_, cond_off_dp_);
    for (size_t i = 0; i < cond_wn_.size(); ++i) cond(cond_wn_[i], cond_off_wn_[i]);
    cond(cond_dec_, cond_off_dec_);
    cb_dp = cond_ + cond_off_dp_;
  }

  // ================= text encoder (models.py:198-209, attentions.py:60-74)
  prof_begin();
  double fl = 0;
  PE_LAUNCH_KB("embed_kernel", 4.0 * tsum * (1.0 + H_), launch::embed(dim3((T + 63) / 64, (H_ + 15) / 16, B), stream_, d_ids_, Ts, d_tlens_, emb_, H_, std::sqrt((float)H_), x_, (long)H_ * Ts, Ts, d_rng_));
  // norm_layers_2 of a layer feeds only the next layer's q/k/v conv (or, after the last layer, proj) + the residual of
  // conv_o. Small batches with the 192-channel encoder run norm_layers_2 + that conv as one launch (lngemm_kernel), and
  // conv_o + residual + norm_layers_1 as another (colchain_kernel); otherwise conv, then ln_kernel.
  const float *pg = nullptr, *pb = nullptr;        // pending norm_layers_2 of the previous layer (input still in y)
  const bool chain_q = pol_.chain16(tsum, false, H_, 96);
  // Small calls: the FFN as ONE launch that leaves FC/48 partial outputs for lngemm4_kernel to sum (kernels/ffn.h). That
  // consumer then reads the residual from x and writes LN(y) to the other buffer (its parts read x concurrently): x / y
  // swap roles per layer.
  const bool ffn_fused = stage_a_ffn_fused();
  const int nsl = FC_ / 48;
  const float* pend_bias = nullptr;                // conv_2 bias of the layer whose partial outputs are pending
  for (auto& e : enc_) {
    if (pg && pend_bias) {
      lngemm(x, pg, pb, y, e.qkv16, e.qkv.bias, 3 * H_, qkv, T, 2.0 * tsum * e.qkv.macs_per_col, ffn_parts_, nsl, pend_bias);
      std::swap(x, y);
    } else if (pg) lngemm(y, pg, pb, x, e.qkv16, e.qkv.bias, 3 * H_, qkv, T, 2.0 * tsum * e.qkv.macs_per_col);
    else if (!(chain_q && conv1x1_col4(e.qkv16, e.qkv.bias, 3 * H_, x, qkv, d_tlens_, B, T, 2.0 * tsum * e.qkv.macs_per_col)))
      conv(e.qkv, x, qkv, d_tlens_, 1, T, EPI_STORE);
    pg = pb = nullptr;
    pend_bias = nullptr;
    // Small calls of the 192-channel voices: attention + conv_o + residual + norm_layers_1 as ONE launch (kernels/attno.h:
    // 16 queries of both heads per workgroup)
    const int ao_sp = rup(T, 64) + 2;
    const size_t ao_smem = ((size_t)2 * 16 * ao_sp + 2 * 64 * (dk_ + 1) + 2 * dk_ * 16 + (size_t)2 * (2 * window_ + 1) * dk_ + 8 * 256 + 256) * sizeof(float);
    const bool attno = pol_.attno && chain_q && pol_.chain4((long)B * T) && H_ == 192 && nh_ == 2 && dk_ == 96 && window_ <= 4 && e.o16 &&
                       ao_smem <= (size_t)160 * 1024;
    if (attno) {
      AttnOP ap{};
      ap.qkv = qkv_; ap.q_bs = (long)3 * H_ * Ts; ap.q_cs = Ts;
      ap.relk = e.relk; ap.relv = e.relv;
      ap.lens = d_tlens_; ap.window = window_; ap.SP = ao_sp;
      ap.qscale = 1.0f / std::sqrt((float)dk_);
      ap.wo16 = e.o16; ap.bo = e.o.bias; ap.gamma = e.g1; ap.beta = e.b1;
      ap.x = x.p; ap.x_bs = x.bs; ap.x_cs = x.cs;
      double afl = 0;
      for (int b = 0; b < B; ++b) afl += 4.0 * (double)tlens_h_[b] * tlens_h_[b] * H_;
      const int kh = kbegin(prof_level_ >= 2 ? krow("attno_kernel<96>") : 0, afl + 2.0 * tsum * e.o.macs_per_col,
                            4.0 * (tsum * 5.0 * H_ + e.o.macs_per_col));
      launch::attno(dim3((T + 15) / 16, B), ao_smem, stream_, ap);
      kend(kh);
    } else {
    AttnP ap;
    ap.qkv = qkv_; ap.q_bs = (long)3 * H_ * Ts; ap.q_cs = Ts;
    ap.relk = e.relk; ap.relv = e.relv;
    ap.out = att_; ap.o_bs = (long)H_ * Ts; ap.o_cs = Ts;
    ap.lens = d_tlens_; ap.H = H_; ap.dk = dk_; ap.window = window_;
    ap.SP = rup(T, 64) + 1;
    ap.qscale = 1.0f / std::sqrt((float)dk_);
    const int VS = dk_ + 1 + (dk_ & 1);
    const size_t smem = ((size_t)ATT_QB * ap.SP + (size_t)ATT_KCH * VS + (size_t)dk_ * ATT_QB +
                         (size_t)2 * (2 * window_ + 1) * dk_ + 4 * ATT_QB * 16) * sizeof(float);
    if (smem > 160 * 1024) throw std::runtime_error("utterance too long for the attention score tile");
    double afl = 0;
    for (int b = 0; b < B; ++b) afl += 4.0 * (double)tlens_h_[b] * tlens_h_[b] * H_;
    const int kh = kbegin(prof_level_ >= 2 ? krow(ap.dk == 96 ? "attn_kernel<96>" : ap.dk == 48 ? "attn_kernel<48>" : "attn_kernel<0>") : 0, afl, 4.0 * 4.0 * H_ * tsum);
    const dim3 agrid((T + ATT_QB - 1) / ATT_QB, nh_, B);
    launch::attention(ap.dk, agrid, smem, stream_, ap);
    kend(kh);
    const bool chain_o = chain_q;
    if (chain_o) {
      // conv_o + residual + norm_layers_1 in one launch (the 192 x 192 GEMM fits one workgroup per 16 columns)
      ColP cp{};
      cp.in1 = att.p; cp.in1_bs = att.bs; cp.in1_cs = att.cs; cp.K1 = H_;
      cp.w1 = e.o16; cp.b1 = e.o.bias; cp.rows1 = H_;
      cp.mode = 0;
      cp.res = x.p; cp.res_bs = x.bs; cp.res_cs = x.cs;
      cp.gamma = e.g1; cp.beta = e.b1;
      cp.out = x.p; cp.out_bs = x.bs; cp.out_cs = x.cs;
      cp.lens = d_tlens_;
      colchain(cp, B, T, 2.0 * tsum * e.o.macs_per_col);
    } else {
      conv(e.o, att, y, d_tlens_, 1, T, EPI_RESADD, 1.f, ACT_NONE, x);
    }
    if (!chain_o) layer_norm(y, x, e.g1, e.b1, H_, d_tlens_, T);
    }      // !attno
    if (ffn_fused) {
      FfnP fp{};
      fp.xcd = pol_.xcd_ffn ? xcd_period_ : 0;          // (column tile, slice) dealt to the XCDs slice-major
      const int Tp = rup(T, 4);
      fp.x = x.p; fp.x_bs = x.bs; fp.x_cs = x.cs;
      fp.w1p = e.f1p; fp.b1 = e.f1.bias; fp.w2p = e.f2p;
      fp.parts = ffn_parts_; fp.nslices = nsl; fp.p_bs = (long)nsl * H_ * Tp;
      fp.lens = d_tlens_;
      const int khf = kbegin(prof_level_ >= 2 ? krow("ffn_kernel") : 0, 2.0 * tsum * (e.f1.macs_per_col + e.f2.macs_per_col),
                             4.0 * (tsum * (1.0 + nsl) * H_ + e.f1.macs_per_col + e.f2.macs_per_col));
      const size_t smemf = ((size_t)192 * 48 + 4 * 48 * 16 + 48 * 48) * sizeof(float);
      launch::ffn(dim3((T + 11) / 12, nsl, B), smemf, stream_, fp);
      kend(khf);
      pend_bias = e.f2.bias;
    } else {
      conv(e.f1, x, ffh, d_tlens_, 1, T, EPI_STORE, 1.f, ACT_RELU);
      conv(e.f2, ffh, y, d_tlens_, 1, T, EPI_RESADD, 1.f, ACT_NONE, x);
    }
    if (chain_q) { pg = e.g2; pb = e.b2; }
    else layer_norm(y, x, e.g2, e.b2, H_, d_tlens_, T);
    fl += 2.0 * tsum * (e.qkv.macs_per_col + e.o.macs_per_col + e.f1.macs_per_col + e.f2.macs_per_col);
    for (int b = 0; b < B; ++b) fl += 2.0 * 2.0 * (double)tlens_h_[b] * tlens_h_[b] * H_;
  }
  if (pg && pend_bias) {
    lngemm(x, pg, pb, y, enc_proj16_, enc_proj_.bias, enc_proj_.rows, stats, T, 2.0 * tsum * enc_proj_.macs_per_col, ffn_parts_, nsl, pend_bias);
    std::swap(x, y);
  } else if (pg) lngemm(y, pg, pb, x, enc_proj16_, enc_proj_.bias, enc_proj_.rows, stats, T, 2.0 * tsum * enc_proj_.macs_per_col);
  else conv(enc_proj_, x, stats, d_tlens_, 1, T, EPI_STORE);
  if (x.p != stage_a_enc_out()) throw std::runtime_error("internal: encoder output buffer bookkeeping");
  fl += 2.0 * tsum * enc_proj_.macs_per_col;
  prof_end(0, fl);

  // ================= stochastic duration predictor, reverse (models.py:63-71,108-117)
  prof_begin();
  fl = 0;
  if (!(chain_q && dp_pre16_ && conv1x1_col4(dp_pre16_, dp_pre_.bias, dp_pre_.rows, x, dy, d_tlens_, B, T, 2.0 * tsum * dp_pre_.macs_per_col,
                                             cb_dp, cond_bs_)))
    conv(dp_pre_, x, dy, d_tlens_, 1, T, EPI_STORE, 1.f, ACT_NONE, none, none, 0, 1.f, cb_dp, cond_bs_);
  if (pol_.fuse_dp) {
    DdsOpt o;                      // dp.proj fused after the last DDSConv layer (models.py:65)
    o.post_w16 = dp_proj16_; o.post_bias = dp_proj_.bias; o.post_rows = dp_proj_.rows; o.post_out = xg;
    dds(dp_dds_, dy, dh, dy2, &o);
  } else {
    dds(dp_dds_, dy, dh, dy2);
    conv(dp_proj_, dh, xg, d_tlens_, 1, T, EPI_STORE);
  }
  fl += 2.0 * tsum * (2 + arch_[A_DDSLAYERS]) * dp_pre_.macs_per_col;
  // z = noise * noise_scale_w   [B][2][Ts]
  if (!have_noise_w_)
    PE_LAUNCH_KB("randn_kernel", 4.0 * 2.0 * tsum, launch::randn(stream_, noise_w_, (long)B * 2, T, (long)Ts, 0L, d_rng_, 0));
  if (!pol_.fuse_dp) {
    const long n = (long)B * 2 * Ts;
    PE_LAUNCH_K("scale_kernel", launch::scale(dim3((unsigned)((n + 255) / 256)), stream_, noise_w_, z2_, n, scales_[2]));
  }
  // Flip is folded into which physical channel is x0 (conditioning) and which is x1 (transformed):
  // logical = physical when an even number of flips has been applied.
  int flips = 0;
  for (size_t fi = 0; fi < cflows_.size(); ++fi) {
    auto& cf = cflows_[fi];
    ++flips;
    const int c0 = (flips & 1) ? 1 : 0;     // physical channel holding logical x0
    const int c1 = 1 - c0;
    if (pol_.fuse_dp) {
      // One launch per DDSConv layer and nothing else: ConvFlow.pre (+ g) is folded into the first layer's input,
      // proj and the spline run on the last layer's columns. The first flow reads the raw N(0,1) draw and applies
      // noise_scale_w itself; its spline epilogue also moves the pass-through channel into z2_.
      const float* zin = fi == 0 ? noise_w_ : z2_;
      DdsOpt o;
      o.pre_z = zin + (long)c0 * Ts; o.pre_z_bs = (long)2 * Ts; o.pre_w = cf.pre_w; o.pre_b = cf.pre_b;
      o.z_scale = fi == 0 ? scales_[2] : 1.f;
      o.post_w16 = cf.proj16; o.post_bias = cf.proj.bias; o.post_rows = cf.proj.rows;
      o.zin = zin; o.zin_bs = (long)2 * Ts; o.z_cs = Ts; o.c0 = c0; o.c1 = c1; o.zout = z2_; o.zout_bs = (long)2 * Ts;
      dds(cf.dds, xg, dh, dy2, &o);
    } else {
      PE_LAUNCH_K("cf_pre_kernel", launch::cf_pre(dim3((T + 63) / 64, H_, B), stream_, z2_ + (long)c0 * Ts, (long)2 * Ts, cf.pre_w, cf.pre_b, xg_, (long)H_ * Ts, Ts, dy_, (long)H_ * Ts, Ts, d_tlens_, H_));
      dds(cf.dds, dy, dh, dy2);
      conv(cf.proj, dh, hproj, d_tlens_, 1, T, EPI_STORE);
      PE_LAUNCH_K("spline_inverse_kernel", launch::spline_inverse(dim3((T + 63) / 64, B), stream_, hproj_, (long)32 * Ts, Ts, z2_ + (long)c1 * Ts, (long)2 * Ts, d_tlens_, 1.0f / std::sqrt((float)H_)));
    }
    fl += 2.0 * tsum * (arch_[A_DDSLAYERS] * dp_pre_.macs_per_col + cf.proj.macs_per_col);
  }
  ++flips;   // the Flip before ElementwiseAffine
  {
    const int c0 = (flips & 1) ? 1 : 0;     // physical channel holding logical channel 0 = logw
    DurP dp{};
    dp.z0 = z2_ + (long)c0 * Ts; dp.z_bs = (long)2 * Ts; dp.m0 = ea_m0_; dp.es0 = ea_es0_; dp.length_scale = scales_[1];
    dp.lens = d_tlens_; dp.dur = d_dur_; dp.cum = d_cum_; dp.d_bs = Ts; dp.frames = d_frames_; dp.logw_out = logw_;
    dp.frames_host = h_frames_; dp.frames_clamped = d_framesc_; dp.frame_cap = std::max(Fs_, 1);
    {
      PE_LAUNCH_KB("duration_kernel", 4.0 * 4.0 * tsum, launch::duration(dim3(B), stream_, dp));
    }
  }
  prof_end(1, fl);
}

// Length regulator, prior sample, coupling flow, HiFiGAN, int16 conversion -- sized by the bucketed
// maximum frame count Fg_.
void Engine::issue_flow() {
  const int B = B_, Ts = Ts_, Fmax = Fg_, Fs = Fs_;
  const View none{nullptr, 0, 0};
  double fsum = 0;
  for (int b = 0; b < B; ++b) fsum += frames_h_[b];
  cols_frames_ = fsum;
  double fl = 0;

  // ================= length regulator + prior noise + coupling flow (models.py:705-719)
  prof_begin();
  if (have_noise_z_) {
    // rows of the caller's [B][C][z_stride] buffer -> [B][C][Fs]
    if (h_noise_z_stride_ < Fmax_) throw std::runtime_error("noise_z stride shorter than the frame count");
    for (int b = 0; b < B; ++b)
      for (int c = 0; c < C_; ++c)
        PE_HIP(hipMemcpyAsync(noise_z_ + ((size_t)b * C_ + c) * Fs,
                              h_noise_z_ + ((size_t)b * C_ + c) * h_noise_z_stride_,
                              frames_h_[b] * sizeof(float), hipMemcpyHostToDevice, stream_));
  } else {
    // (drawing the noise inside regulate_kernel was tried: one launch fewer, but a Philox block + Box-Muller per element
    // in its 16-channels-per-thread loop cost 21 us against this launch's 5, profiles/r02_notes.md)
    PE_LAUNCH_KB("randn_kernel", 4.0 * C_ * fsum, launch::randn(stream_, noise_z_, (long)B * C_, Fmax, (long)Fs, 0L, d_rng_, 1));
  }
  {
    RegP rp;
    rp.stats = stats_; rp.s_bs = (long)2 * C_ * Ts; rp.s_cs = Ts;
    rp.cum = d_cum_; rp.d_bs = Ts; rp.tlens = d_tlens_; rp.frames = lens_b_;
    rp.noise = noise_z_; rp.n_bs = (long)C_ * Fs; rp.n_cs = Fs;
    rp.noise_scale = scales_[0];
    rp.out = zp_; rp.o_bs = (long)C_ * Fs; rp.o_cs = Fs; rp.C = C_;
    rp.absmax = absmax_;
    PE_LAUNCH_KB("regulate_kernel", 4.0 * (2.0 * C_ * cols_ids_ + 2.0 * C_ * fsum), launch::regulate(dim3((Fmax + 63) / 64, (C_ + 15) / 16, B), stream_, rp));
    if (zp_keep_)     // tests: z_p, the flow's input (the flow transforms zp_ in place)
      PE_HIP(hipMemcpyAsync(zp_keep_, zp_, (size_t)B * C_ * Fs * sizeof(float), hipMemcpyDeviceToDevice, stream_));
  }
  auto VF = [&](float* p, int ch) { return View{p, (long)ch * Fs, Fs}; };
  const View fh = VF(fh_, H_), facts = VF(facts_, H_), fskip = VF(fskip_, H_);
  const int half = C_ / 2;
  const bool chain = pol_.chain16(fsum, true, H_, half);
  for (size_t ri = 0; ri < rcls_.size(); ++ri) {
    Rcl& r = rcls_[ri];
    const View x0{zp_ + (long)r.in_off * Fs, (long)C_ * Fs, Fs};
    const View x1{zp_ + (long)r.out_off * Fs, (long)C_ * Fs, Fs};
    if (!(chain && ri > 0)) {                                                     // else: written by the previous layer's chain
      if (!(chain && r.pre4pad && conv1x1_col4(nullptr, r.pre.bias, r.pre.rows, x0, fh, lens_b_, B, Fmax, 2.0 * fsum * r.pre.macs_per_col,
                                               nullptr, 0, r.pre4pad, half, LaunchPolicy::col4_max_frames)))
        conv(r.pre, x0, fh, lens_b_, 1, Fmax, EPI_STORE);
    }
    const int nl = (int)r.in.size();
    for (int i = 0; i < nl; ++i) {
      const float* b2 = nspk_ > 1 ? cond_ + cond_off_wn_[ri] + (long)i * 2 * H_ : nullptr;
      conv(r.in[i], fh, facts, lens_b_, 1, Fmax, EPI_GATE, 1.f, ACT_NONE, none, none, 0, 1.f, b2, cond_bs_);
      if (r.rs4[i] && pol_.chain4_frames((long)B * Fmax) && H_ == 192 && r.rs[i].rows <= 2 * H_) {
        // small calls: the res/skip 1x1 conv on 4-column workgroups (colchain4_kernel mode 2), one part per 192 rows
        ColP cp{};
        cp.in1 = facts.p; cp.in1_bs = facts.bs; cp.in1_cs = facts.cs; cp.K1 = H_;
        cp.w1 = r.rs4[i]; cp.b1 = r.rs[i].bias; cp.rows1 = r.rs[i].rows;
        cp.mode = 2; cp.first = i == 0 ? 1 : 0;
        cp.xcd = xcd_period_;
        cp.x1 = fh.p; cp.x1_bs = fh.bs; cp.x1_cs = fh.cs;
        cp.out = fskip.p; cp.out_bs = fskip.bs; cp.out_cs = fskip.cs;
        cp.lens = lens_b_;
        const int kh4 = kbegin(prof_level_ >= 2 ? krow("colchain4_kernel") : 0, 2.0 * fsum * r.rs[i].macs_per_col,
                               4.0 * (fsum * (H_ + 2.0 * cp.rows1) + (double)cp.rows1 * H_));
        launch::colchain4(dim3((Fmax + 3) / 4, B, (cp.rows1 + 191) / 192), col4_smem(), stream_, cp);
        kend(kh4);
      } else {
        conv(r.rs[i], facts, fh, lens_b_, 1, Fmax, EPI_WNRS, 1.f, ACT_NONE, none, fskip, i == 0 ? 1 : 0);
      }
      fl += 2.0 * fsum * (r.in[i].macs_per_col + r.rs[i].macs_per_col);
    }
    if (chain) {
      // post + "x1 -= m" + the next coupling layer's pre over the updated half, one launch
      ColP cp{};
      cp.in1 = fskip.p; cp.in1_bs = fskip.bs; cp.in1_cs = fskip.cs; cp.K1 = H_;
      cp.w1 = r.post16; cp.b1 = r.post.bias; cp.rows1 = half;
      cp.mode = 1;
      cp.x1 = x1.p; cp.x1_bs = x1.bs; cp.x1_cs = x1.cs;
      if (ri + 1 < rcls_.size()) {
        const Rcl& nx = rcls_[ri + 1];
        if (nx.in_off != r.out_off) throw std::runtime_error("coupling layers do not alternate halves");
        cp.w2 = nx.pre16; cp.b2 = nx.pre.bias; cp.rows2 = H_;
        cp.out2 = fh.p; cp.o2_bs = fh.bs; cp.o2_cs = fh.cs;
      }
      cp.lens = lens_b_;
      colchain(cp, B, Fmax, 2.0 * fsum * (r.post.macs_per_col + (cp.w2 ? rcls_[ri + 1].pre.macs_per_col : 0)));
    } else {
      conv(r.post, fskip, x1, lens_b_, 1, Fmax, EPI_SUBFROM);
    }
    fl += 2.0 * fsum * (r.pre.macs_per_col + r.post.macs_per_col);
  }
  prof_end(2, fl);

}

void Engine::issue_stage_b() {
  issue_flow();
  double fsum = 0;
  for (int b = 0; b < B_; ++b) fsum += frames_h_[b];
  issue_decoder(zp_, lens_b_, Fg_, fsum, false);     // regulate_kernel zeroed the peak accumulators
}

// streaming: window of z -> window buffer -> generator (lens = window length, in device memory)
void Engine::issue_window() {
  PE_LAUNCH_K("window_copy_kernel", launch::window_copy(dim3((s_wg_ + 63) / 64, C_), stream_, zp_, Fs_, d_win_, zwin_, Fs_, C_));
  issue_decoder(zwin_, d_win_ + 1, s_wg_, (double)s_wg_, true);
}

// HiFiGAN generator + conv_post + int16 on z (already masked by its length semantics). `zsrc` is
// [B][C][Fs_]; `lens` the per-utterance frame counts in device memory; Fmax the grid bound.
void Engine::issue_decoder(const float* zsrc, const int* lens, int Fmax, double fsum, bool zero_absmax) {
  const int B = B_, Fs = Fs_;
  const View none{nullptr, 0, 0};
  const float* cb_dec = nspk_ > 1 ? cond_ + cond_off_dec_ : nullptr;
  double fl = 0;
  bool tail_done = false;      // conv_post + tanh + peak computed inside the last stage's mrf_kernel
  // (zero_absmax marks the streaming window path; the whole-utterance path clears the peaks in regulate_kernel)
  if (zero_absmax) PE_HIP(hipMemsetAsync(absmax_, 0, B * sizeof(unsigned), stream_));
  // ================= HiFiGAN generator (models.py:348-368)
  prof_begin();
  fl = 0;
  {
    View cur{hb_[0], (long)U_ * Fs, Fs};
    conv(dec_pre_, View{const_cast<float*>(zsrc), (long)C_ * Fs, Fs}, cur, lens, 1, Fmax, EPI_STORE, 1.f, ACT_NONE, none, none, 0, 1.f,
         cb_dec, cond_bs_);
    fl += 2.0 * fsum * dec_pre_.macs_per_col;
    int mult = 1;
    int cur_buf = 0;
    const int nk = arch_[A_NRB];
    const float inv_nk = 1.0f / (float)nk;
    for (auto& st : ups_) {
      // pick the five working buffers for this stage: u, ta, tb, tc, xs (all != cur)
      int ids[5], n = 0;
      for (int i = 0; i < 5 && n < 4; ++i)
        if (i != cur_buf) ids[n++] = i;
      const int Lin = mult;            // length multiplier of the input
      mult *= st.rate;
      const long Ls = (long)Fs * mult;
      auto VS = [&](int bi) { return View{hb_[bi], (long)st.ch * Ls, (int)Ls}; };
      const View u = VS(ids[0]), ta = VS(ids[1]), tb = VS(ids[2]), tc = VS(ids[3]);
      const int Lmax = Fmax * mult;
      // One launch per stage (mrf_kernel). Measured (profiles/r03_notes.md): ResBlock2 stages (medium / x-low) win at every
      // batch size (B=1 -3 %, B=16 / 64 +4.5 % end to end over the conv-by-conv schedule); ResBlock1 stages (high) tie at
      // one utterance and lose at batch (its 64-channel stage: 86 vs ~110 TFLOP/s for the conv GEMM kernel on K = 64 * 11
      // convs), so those are fused for one or two utterances and on 32 channels only.
      // (matrix mode bf16x3: the fused kernel is f32; from a few utterances up the conv-by-conv schedule on the bf16 pipe is faster)
      const bool fuse = pol_.mrf_stage(st.mrf_ok, st.mrf_rb1, st.mrf_cp, fsum, matrix_bf3_);
      // the last stage also runs the generator tail (conv_post, tanh, peak) on its MRF mean while it is still on chip
      const bool tail = fuse && pol_.mrf_tail && &st == &ups_.back() && st.mrf_cp == 32 && st.ch == post_cin_ && mult == hop_;
      // leaky_relu(0.1) -> ConvTranspose1d
      // (folding the up-conv into the stage kernel's prologue was built and measured: the window GEMM with its halo
      // recompute on the 209 workgroups of a single round costs what the launch costs -- profiles/r04_notes.md)
      conv(st.up, cur, u, lens, Lin, Fmax * Lin, EPI_CONVT, 0.1f);
      fl += 2.0 * fsum * Lin * st.up.macs_per_col;
      // xs accumulates into the buffer that held the stage input (free once the up-conv is done)
      const View xs{hb_[cur_buf], (long)st.ch * Ls, (int)Ls};
      // One resblock chain, accumulated into xs with the MRF mode. `t` = {c1 output, ping, pong}.
      auto chain = [&](int j, const View (&t)[3], View dst, int accmode) {
        auto& cv = st.rb[j];
        const int last_epi = EPI_ACCUM;
        View xin = u;
        if (arch_[A_RESBLOCK] == 1) {
          // ResBlock1 (modules.py:301-314): x = x + c2(lrelu(c1(lrelu(x)))) per dilation
          const int np = (int)cv.size() / 2;
          for (int d = 0; d < np; ++d) {
            conv(cv[2 * d], xin, t[0], lens, mult, Lmax, EPI_STORE, 0.1f);
            if (d < np - 1) {
              const View nxt = (d & 1) ? t[2] : t[1];
              conv(cv[2 * d + 1], t[0], nxt, lens, mult, Lmax, EPI_RESADD, 0.1f, ACT_NONE, xin);
              xin = nxt;
            } else {
              conv(cv[2 * d + 1], t[0], dst, lens, mult, Lmax, last_epi, 0.1f, ACT_NONE, xin, none, accmode, inv_nk);
            }
            fl += 2.0 * fsum * mult * (cv[2 * d].macs_per_col + cv[2 * d + 1].macs_per_col);
          }
        } else {
          // ResBlock2 (modules.py:355-364): x = x + c(lrelu(x)) per dilation
          const int nc = (int)cv.size();
          for (int d = 0; d < nc; ++d) {
            if (d < nc - 1) {
              const View nxt = (d & 1) ? t[2] : t[1];
              conv(cv[d], xin, nxt, lens, mult, Lmax, EPI_RESADD, 0.1f, ACT_NONE, xin);
              xin = nxt;
            } else {
              conv(cv[d], xin, dst, lens, mult, Lmax, last_epi, 0.1f, ACT_NONE, xin, none, accmode, inv_nk);
            }
            fl += 2.0 * fsum * mult * cv[d].macs_per_col;
          }
        }
      };
      const size_t need = (size_t)B * st.ch * Ls;
      const long blocks64 = (long)((Lmax + 63) / 64) * ((st.ch + 63) / 64) * B;
      // grouped sibling launches are a single-utterance latency measure: measured -24 us (medium) / -4 % (high) at
      // B=1, but +1..2 % at B=2 and B=4, where every conv already fills the chip on its own
      bool grp = pol_.group_stage(B, nk, blocks64, need <= side_floats_);
      for (auto& cv : st.rb) {
        if (cv.size() != st.rb[0].size()) grp = false;
        for (auto& c : cv) grp = grp && can_group(c, Lmax);
      }
      if (fuse) {
        mrf(st, u, xs, lens, mult, Lmax, tail);
        for (auto& cv : st.rb)
          for (auto& c : cv) fl += 2.0 * fsum * mult * c.macs_per_col;
        if (tail) {
          tail_done = true;
          fl += 2.0 * fsum * hop_ * post_cin_ * POST_K;
        }
      } else if (grp) {
        // step d of every resblock in one grouped launch; each resblock keeps its own buffers, one pass sums them
        auto SV = [&](int k) { return View{side_[k], (long)st.ch * Ls, (int)Ls}; };
        View xin[3] = {u, u, u};
        bool summed = false;
        const int nsteps = (int)st.rb[0].size();
        const bool rb1 = arch_[A_RESBLOCK] == 1;
        for (int d = 0; d < nsteps; ++d) {
          group_begin();
          for (int j = 0; j < nk; ++j) {
            auto& cv = st.rb[j];
            const View t0 = j == 0 ? tb : SV(4 * (j - 1)), t1 = j == 0 ? ta : SV(4 * (j - 1) + 1),
                       t2 = j == 0 ? tc : SV(4 * (j - 1) + 2), dst = j == 0 ? SV(8) : SV(4 * (j - 1) + 3);
            if (rb1 && !(d & 1)) {
              conv(cv[d], xin[j], t0, lens, mult, Lmax, EPI_STORE, 0.1f);
            } else {
              const int dd = rb1 ? d / 2 : d, nd = rb1 ? nsteps / 2 : nsteps;
              const View o = dd < nd - 1 ? ((dd & 1) ? t2 : t1) : dst;
              conv(cv[d], rb1 ? t0 : xin[j], o, lens, mult, Lmax, EPI_RESADD, 0.1f, ACT_NONE, xin[j]);
              xin[j] = o;
            }
            fl += 2.0 * fsum * mult * cv[d].macs_per_col;
          }
          // the last step's outputs are only ever summed: one GEMM over the concatenated K writes the mean directly
          if (d == nsteps - 1 && pol_.group_sum() && can_group_sum()) {
            group_end_sum(xs, st.last_bias_sum, inv_nk);
            summed = true;
          } else {
            group_end();
          }
        }
        if (!summed)
          PE_LAUNCH_K("mrf_sum_kernel", launch::mrf_sum(dim3((Lmax + 255) / 256, st.ch, B), stream_, side_[8], side_[3], nk == 3 ? side_[7] : (const float*)nullptr, xs.p, xs.bs, xs.cs, lens, mult, inv_nk));
      } else {
        for (int j = 0; j < nk; ++j) {
          const int accmode = nk == 1 ? 3 : (j == 0 ? 0 : (j == nk - 1 ? 2 : 1));
          const View t[3] = {tb, ta, tc};
          chain(j, t, xs, accmode);
        }
      }
      cur = xs;      // same buffer index cur_buf, new shape
    }
    prof_end(3, fl);

    // ================= conv_post + tanh + peak, int16 (models.py:364-366; piper.cpp:410-431)
    prof_begin();
    const int K = 7, Lmax = Fmax * hop_;
    if (!tail_done)
      PE_LAUNCH_KB("conv_post_kernel", 4.0 * fsum * hop_ * (post_cin_ + 1.0), launch::conv_post(dim3((Lmax + POST_SPB - 1) / POST_SPB, B), stream_, cur.p, cur.bs, cur.cs, post_w_, post_cin_, 0.01f, lens, hop_, audio_, Ss_, absmax_));
    // (the streaming window path delivers per chunk from the device buffer)
    int16_t* zc = (pol_.pcm_zc && !zero_absmax && h_pcm_zc_cap_ >= (size_t)B * (size_t)Ss_) ? h_pcm_zc_ : nullptr;
    PE_LAUNCH_KB("pcm16_kernel", fsum * hop_ * (4.0 + 2.0 + (zc ? 2.0 : 0.0)), launch::pcm16(dim3((Lmax + 255) / 256, B), stream_, audio_, Ss_, absmax_, lens, hop_, pcm_, Ss_, zc));
    prof_end(4, tail_done ? 0.0 : 2.0 * fsum * hop_ * post_cin_ * K);
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
  ensure_stage_b(frame_bucket((int)fmax));
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
  if (spec) ensure_stage_b(fguess);              // before stage A is enqueued: growing the workspace drops every graph
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
    run_stage('C', key);
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
