// Voice -> packed weight arena: every tensor of the voice in the fragment orders the kernels read, carved from ONE
// arena in a deterministic order (Engine::init), plus the phase table / weight stream of the fused MRF stages.
#include "engine_internal.h"
#include <cmath>

namespace pe {

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
  // split matrix modes: the flow / generator conv weights once more as 16-bit term fragments (two terms: the size of the
  // f32 packing; three terms, bf16x6: 1.5x that)
  const int sm = LaunchPolicy::matrix_mode_env();
  return (n * (sm == 2 ? 12 : (sm >= 0 ? 10 : 7)) / 2 + (4u << 20)) * sizeof(float);
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
  if (gate && Cin == 192 && ntaps <= 5 && dil == 1 && split % 32 == 0) {
    // gate4_kernel (kernels/gate4.h): [group of 32 channels][tap][k quad 48][lane][4]; lane l < 32: the tanh row of channel
    // 32 g + l, l >= 32: the sigmoid row of channel 32 g + l - 32; float4 element j of quad q = input channel 4 q + j
    const int ng = split / 32;
    const size_t n4 = (size_t)ng * ntaps * (Cin / 4) * 256;
    std::vector<float> G(skeleton_ ? 0 : n4, 0.f);
    for (int g = 0; g < (skeleton_ ? 0 : ng); ++g)
      for (int tap = 0; tap < ntaps; ++tap)
        for (int q = 0; q < Cin / 4; ++q)
          for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 4; ++j) {
              const int ch = g * 32 + (lane & 31), row = lane < 32 ? ch : split + ch;
              G[((((size_t)g * ntaps + tap) * (Cin / 4) + q) * 64 + lane) * 4 + j] = W[((size_t)row * Cin + 4 * q + j) * ntaps + tap];
            }
    pc.wpg4 = dev_alloc(n4, skeleton_ ? nullptr : G.data());
  }
  if (pack_bf3_now_) {
    // conv_split_kernel (kernels/conv_bf3.h): every weight as the NT 16-bit terms of the engine's split mode, in the
    // A-operand order of v_mfma_f32_32x32x16_{bf16,f16}: [m tile][chunk][tap][term][k-step][lane][8], lane -> row = lane & 31,
    // input channel chunk*32 + 8*(2*kstep + (lane >> 5)) + e. One (tile, chunk, tap) step = NT * 512 floats.
    const int sm = matrix_sm_, nt = sm == 2 ? 3 : 2;
    float wscale = 1.f;
    if (sm == 1 && !skeleton_) {
      // f16's exponent range: the largest weight lands in [2^12, 2^13) (products with activations up to 65504 accumulate in
      // f32), so the low terms of ordinary weights stay far above f16's subnormal step; undone exactly on the accumulators
      float mx = 0.f;
      for (float v : W) mx = std::max(mx, std::fabs(v));
      if (mx > 0.f && std::isfinite(mx)) {
        int e = 0;
        std::frexp(mx, &e);                    // mx = f * 2^e, f in [0.5, 1)
        wscale = std::ldexp(1.f, 13 - e);
      }
    }
    // (the factor that undoes the scale travels WITH the packed weights -- one float behind them in the arena -- so that a
    // skeleton engine, which never sees W, reads the same value after the broadcast)
    const size_t npb = (size_t)pc.mtiles * pc.nchunks * ntaps * nt * 512;           // floats
    std::vector<uint16_t> R(skeleton_ ? 0 : (npb + 64) * 2, 0);
    if (!skeleton_) {
      const float us = 1.f / wscale;
      memcpy(&R[npb * 2], &us, sizeof(float));
    }
    for (int mt = 0; mt < (skeleton_ ? 0 : pc.mtiles); ++mt)
      for (int c = 0; c < pc.nchunks; ++c)
        for (int tap = 0; tap < ntaps; ++tap) {
          const size_t step = (((size_t)mt * pc.nchunks + c) * ntaps + tap) * nt * 1024;      // in 16-bit elements
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
                float rem = W[((size_t)row * Cin + ci) * ntaps + tap] * wscale;
                for (int t = 0; t < nt; ++t) {
                  uint16_t bits;
                  float back;
                  if (sm == 1) {
                    const _Float16 h = (_Float16)rem;
                    memcpy(&bits, &h, 2);
                    back = (float)h;
                  } else {
                    bits = bf16_rne(rem);
                    back = bf16_to_f32(bits);
                  }
                  R[step + ((size_t)(t * 2 + ks) * 64 + lane) * 8 + e] = bits;
                  rem -= back;
                }
              }
            }
        }
    pc.wpb = dev_alloc(npb + 64, skeleton_ ? nullptr : reinterpret_cast<const float*>(R.data()));
    pc.wunscale = pc.wpb + npb;
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

void Engine::init(const WeightSet& ws) {
  pol_.read_env();
  use_graphs_ = !pol_.no_graph;
  matrix_bf3_ = env_bf3();
  matrix_sm_ = std::max(0, LaunchPolicy::matrix_mode_env());
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
  // small calls: enc_p.proj (2 C rows) and dp.pre (H rows) read the same LN(y) -- stacked into ONE matrix for lngemm4_kernel
  // (row parts 0 .. of 192 rows: the last ones are dp.pre's), pack4 order; biases stacked alike
  projpre4_ = nullptr;
  projpre_bias_ = nullptr;
  projpre_split_ = 0;
  if (H_ == 192) {
    const HostTensor& wp = ws.get("enc_p.proj.weight");
    const HostTensor& wd = ws.get("dp.pre.weight");
    const HostTensor& bp = ws.get("enc_p.proj.bias");
    const HostTensor& bd = ws.get("dp.pre.bias");
    if (wp.dims.size() == 3 && wd.dims.size() == 3 && wp.dims[1] == 192 && wd.dims[1] == 192 && wp.dims[2] == 1 && wd.dims[2] == 1 &&
        wp.dims[0] % 192 == 0 && wd.dims[0] == 192) {
      const int rp = (int)wp.dims[0], rd = (int)wd.dims[0];
      std::vector<float> Wst(skeleton_ ? 0 : (size_t)(rp + rd) * 192), Bst(skeleton_ ? 0 : (size_t)(rp + rd));
      if (!skeleton_) {
        std::copy(wp.data.begin(), wp.data.begin() + (size_t)rp * 192, Wst.begin());
        std::copy(wd.data.begin(), wd.data.begin() + (size_t)rd * 192, Wst.begin() + (size_t)rp * 192);
        std::copy(bp.data.begin(), bp.data.begin() + rp, Bst.begin());
        std::copy(bd.data.begin(), bd.data.begin() + rd, Bst.begin() + rp);
      }
      projpre4_ = pack4(Wst, rp + rd, 192);
      projpre_bias_ = dev_alloc((size_t)(rp + rd), skeleton_ ? nullptr : Bst.data());
      projpre_split_ = rp;
    }
  }
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
  // matrix modes bf16x3 / f16x3: the same stream as two 16-bit terms per weight for mrf_split_kernel (kernels/mrf_split.h):
  // [phase][step][16-row tile][term][lane][8], lane -> row = lane & 15, input channel chunk * 32 + 8 * (lane >> 4) + e
  const bool split = matrix_bf3_ && matrix_sm_ < 2;
  std::vector<uint16_t> wsplit;
  std::vector<float> unscale;
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
      if (split) {
        float wscale = 1.f;
        if (matrix_sm_ == 1 && !skeleton_) {          // f16: the conv's largest weight lands in [2^12, 2^13) (engine_pack.cpp pack_matrix)
          float mx = 0.f;
          for (float v : h.w) mx = std::max(mx, std::fabs(v));
          if (mx > 0.f && std::isfinite(mx)) {
            int ex = 0;
            std::frexp(mx, &ex);
            wscale = std::ldexp(1.f, 13 - ex);
          }
        }
        unscale.push_back(1.f / wscale);
        const size_t s0 = wsplit.size();
        wsplit.resize(s0 + (size_t)nsteps * MS * 2 * 512, 0);      // 2 terms x 64 lanes x 8 elements per (step, tile)
        for (int step = 0; step < (skeleton_ ? 0 : nsteps); ++step) {
          const int c = step / h.k, tap = step % h.k;
          for (int ms = 0; ms < MS; ++ms)
            for (int lane = 0; lane < 64; ++lane)
              for (int el = 0; el < 8; ++el) {
                const int row = ms * 16 + (lane & 15), ci = c * KC + 8 * (lane >> 4) + el;
                if (row >= ch || ci >= ch) continue;
                float rem = h.w[((size_t)row * ch + ci) * h.k + tap] * wscale;
                for (int t = 0; t < 2; ++t) {
                  uint16_t bits;
                  float back;
                  if (matrix_sm_ == 1) {
                    const _Float16 hv16 = (_Float16)rem;
                    memcpy(&bits, &hv16, 2);
                    back = (float)hv16;
                  } else {
                    bits = bf16_rne(rem);
                    back = bf16_to_f32(bits);
                  }
                  wsplit[s0 + (((size_t)(step * MS + ms) * 2 + t) * 64 + lane) * 8 + el] = bits;
                  rem -= back;
                }
              }
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
  if (split) {
    st.mrf_wsplit_floats = (int)(wsplit.size() / 2);
    st.mrf_wsplit = dev_alloc(wsplit.size() / 2, reinterpret_cast<const float*>(wsplit.data()));
    unscale.resize(MRF_MAXPH, 1.f);
    st.mrf_unscale = dev_alloc(unscale.size(), unscale.data());
  }
  st.mrf_cp = CP;
  st.mrf_ph = phases;
  st.mrf_hx = hx;
  st.mrf_rb1 = rb1;
  st.mrf_ok = true;
}

}  // namespace pe
