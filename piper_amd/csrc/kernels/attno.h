// attno_kernel: an encoder layer's self-attention AND what follows it up to norm_layers_1 in one launch
// (attentions.py:60-72: y = attn(x); x = norm_layers_1(x + conv_o(y)); :225-272, 292-348 for the attention itself).
// (gfx950 / CDNA4 device code; reference arithmetic cited per kernel, paths relative to
// /root/reference/src/python/piper_train/vits/.)
#pragma once
#include "colchain.h"

namespace pe {

// Small calls ran this as two launches: attn_kernel (32 queries x one head per workgroup: 8 workgroups for a 128-id
// utterance, 13 us) and colchain4_kernel (conv_o + residual + LayerNorm, 6 us). conv_o needs both heads of a column and
// the LayerNorm all 192 channels, so the fused workgroup owns 16 queries of BOTH heads: 8 waves, waves 0-3 = head 0,
// waves 4-7 = head 1, everything on the 16x16x4 MFMA (16 queries = one tile: half the matrix time per workgroup of the
// 32-query form on the same number of workgroups).
//   1. S = (q / sqrt(dk)) k^T: a head's four waves take the 16-key tiles round-robin, K fragments straight from global
//      (the next tile's in flight under the current tile's MFMAs); relative-key partial products next to them.
//   2. band add, softmax (16 lanes per query row).
//   3. O^T = V P^T: a wave owns channel tiles w and w + 4 of its head and takes their V fragments straight from global
//      memory into the MFMA's A registers -- lane (channel, lq) holds keys k0 + 16 c + 4 lq + e (e < 4) of a 64-key half as
//      one 16-byte load per c, the first two halves requested at kernel entry, the matching P[q][key] read from the score
//      rows in the same key order -- so the phase has no LDS transposition and no barrier per chunk (it was 6.5 of the
//      workgroup's 15.8 us with V staged through LDS: profiles/r04_notes.md, calls 34 / 60); the relative-value band as
//      three more k-steps. conv_o's weight row blocks, the residual and the LayerNorm gains are requested before this
//      phase and arrive under it.
//   4. O -> LDS as the [192][16] B operand of conv_o: colchain_kernel's mode 0 from here on (col_gemm16, pe_col_sum16).
// Masked keys (>= len) get weight exactly 0, like the reference's -1e4 fill in fp32. k runs in ascending order inside and
// across the MFMAs of the score and conv_o products (the same fmaf chains as attn_kernel + colchain_kernel); V P^T sums
// the keys of a 16-key group in the order e-major (4 lq + e), a different rounding of the same sum.
constexpr int AO_QB = 16, AO_KCH = 64;
template <int DK>
__global__ __launch_bounds__(512) void attno_kernel(AttnOP p) {
  PE_KTRACE(1);
  constexpr int NH = 2, H = NH * DK, NVT = H / 32, NKS = DK / 4, VS = DK + 1, NDT = DK / 16, NC = 16;
  static_assert(H == 192 && DK % 32 == 0, "compiled for the 192-channel voices (two heads of 96)");
  PE_DYN_SMEM(float, sm);
  const int b = blockIdx.y, i0 = blockIdx.x * AO_QB;
  PE_STAMP(0, 0);
  const int T = p.lens[b];
  if (i0 >= T) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = PE_UNIFORM(tid >> 6), hh = PE_UNIFORM(wv >> 2), w4 = PE_UNIFORM(wv & 3);
  const int t4 = tid & 255, l15 = lane & 15, lq = lane >> 4;
  const int SP = p.SP, nrel = 2 * p.window + 1;
  float* S = sm + hh * AO_QB * SP;                           // this head's scores / probabilities [16][SP]
  float* base = sm + NH * AO_QB * SP;
  // (base .. base + NH * AO_KCH * VS: IN and Z of the conv_o / LayerNorm tail; nothing else lives there)
  float* Qs = base + NH * AO_KCH * VS + hh * DK * AO_QB;     // [DK][16], scaled by 1/sqrt(dk)
  float* RK = base + NH * AO_KCH * VS + NH * DK * AO_QB;     // [nrel][DK] relative-key embeddings (shared by the heads)
  float* RV = RK + nrel * DK;                                // [nrel][DK] relative-value embeddings
  float* part = RV + nrel * DK;                              // [8 waves][16 queries][16 offsets]
  float* red = part + 8 * 256;                               // [2][8][16] (pe_col_sum16)
  float* IN = base;                                          // after phase 3, over both heads' V chunks: [192][16]
  float* Z = IN + H * NC;                                    // [192][16]
  static_assert(2 * H * NC <= NH * AO_KCH * VS, "IN + Z fit the V staging area");
  const float* qb = p.qkv + (long)b * p.q_bs + (long)(hh * DK) * p.q_cs;
  const float* kb = qb + (long)H * p.q_cs;
  const float* vb = kb + (long)H * p.q_cs;
  const int nkt = (T + 15) / 16;
  const pe_rowsrc qd = pe_make_row_u(qb, DK * p.q_cs), kd = pe_make_row_u(kb, DK * p.q_cs), vd = pe_make_row_u(vb, DK * p.q_cs);

  // ---- everything that depends on nothing computed here is requested now: this wave's first key tile, the first V
  // chunk, Q, the relative-position tables
  auto load_k = [&](int kt, float (&kf)[NKS]) {               // B[k = channel][col = key]: channel 4 s + lq, key 16 kt + l15
    const int j = kt * 16 + l15;
    const int o = (kt < nkt && j < T) ? lq * p.q_cs + j : 0x3fffffff;
#pragma unroll
    for (int s = 0; s < NKS; ++s) kf[s] = pe_row_load_so(kd, o, 4 * s * p.q_cs);
  };
  // V fragments of this wave's channel tiles: slots [0, 4) = 64-key half A, [4, 8) = half B (a key group past the length
  // is not read at all; a group that straddles it is masked element by element where it is used)
  // channel tiles wt and wt + 4 of this head: head 1 deals them in the opposite wave order, so the two waves that share a SIMD
  // (w and w + 4) own three of the heads' twelve tiles between them instead of four or two
  const int wt = PE_UNIFORM(hh ? 3 - w4 : w4);
  const bool two = PE_UNIFORM(wt + 4 < NDT);
  f32x4 va0[8], va1[8];
  const int vrow0 = (wt * 16 + l15) * p.q_cs + 4 * lq, vrow1 = ((wt + 4) * 16 + l15) * p.q_cs + 4 * lq;
#define AO_LOAD_HALF(S0, K0)                                                          \
  _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                       \
    const int kk = (K0) + 16 * c;                                                       \
    va0[(S0) + c] = pe_row_load4(vd, (kk + 4 * lq < T) ? vrow0 + kk : -4);              \
    va1[(S0) + c] = pe_row_load4(vd, (two && kk + 4 * lq < T) ? vrow1 + kk : -4);       \
  }
  float kA[NKS], kB[NKS];
  {
    // Q and the relative-position tables first: the first phase waits for them only (the memory counter retires in
    // order), this wave's first two key tiles stay in flight behind them. (The V fragments are requested between the score
    // tiles: 32 more 16-byte loads per lane in front of the LDS stores below cost the first phase 1.7 us of issue time.)
    constexpr int NQ = DK * AO_QB / 256;
    float qv[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
      const int e = t4 + 256 * u, d = e >> 4, i = e & 15;
      qv[u] = pe_row_load(qd, (i0 + i < T) ? d * p.q_cs + i0 + i : -1);
    }
    constexpr int NR = (9 * DK + 511) / 512;                   // window <= 4 (checked by the launcher)
    const pe_rowsrc rkd = pe_make_row(p.relk, nrel * DK), rvd = pe_make_row(p.relv, nrel * DK);
    float rk[NR], rv[NR];
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      rk[u] = pe_row_load(rkd, tid + 512 * u);
      rv[u] = pe_row_load(rvd, tid + 512 * u);
    }
    PE_SCHED_FENCE();
    load_k(w4, kA);
    load_k(w4 + 4, kB);
    PE_SCHED_FENCE();
#pragma unroll
    for (int u = 0; u < NQ; ++u) Qs[t4 + 256 * u] = qv[u] * p.qscale;
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      const int e = tid + 512 * u;
      if (e < nrel * DK) { RK[e] = rk[u]; RV[e] = rv[u]; }
    }
  }
  __syncthreads();
  PE_STAMP(0, 1);

  // ---- 1. relative-key partial products R[q][r] = Q . rel_k^T (a quarter of the channel steps per wave of the head),
  // then the score tiles
  {
    f32x4 racc;
#pragma unroll
    for (int r = 0; r < 4; ++r) racc[r] = 0.f;
    for (int s = w4; s < NKS; s += 4) {
      const int d = 4 * s + lq;
      racc = pe_mfma_16x16x4(Qs[d * AO_QB + l15], l15 < nrel ? RK[l15 * DK + d] : 0.f, racc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) part[(wv * 16 + 4 * lq + r) * 16 + l15] = racc[r];
  }
  {
    auto tile = [&](int kt, const float (&kf)[NKS]) {
      float qf[NKS];
#pragma unroll
      for (int s = 0; s < NKS; ++s) qf[s] = Qs[(4 * s + lq) * AO_QB + l15];
      f32x4 acc;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = 0.f;
      PE_SCHED_FENCE();
#pragma unroll
      for (int s = 0; s < NKS; ++s) acc = pe_mfma_16x16x4(qf[s], kf[s], acc);
      PE_SCHED_FENCE();
#pragma unroll
      for (int r = 0; r < 4; ++r) S[(4 * lq + r) * SP + kt * 16 + l15] = acc[r];
    };
    AO_LOAD_HALF(0, 0)                                         // V: first 64 keys in front of the score tiles, the next 64 behind
    for (int kt = w4; kt < nkt; kt += 8) {                     // (the first two tiles' fragments were requested at entry)
      tile(kt, kA);
      if (kt + 4 < nkt) {
        load_k(kt + 8, kA);                                    // beyond the last tile: zero-length reads
        tile(kt + 4, kB);
        load_k(kt + 12, kB);
      }
    }
    AO_LOAD_HALF(4, AO_KCH)
  }
  __syncthreads();
  PE_STAMP(0, 2);
  // ---- 2a. relative-key band: S[i][i + r - w] += q_i . rel_k[r] (the head's four partial tiles, in wave order)
  for (int e = t4; e < AO_QB * nrel; e += 256) {
    const int i = e & 15, r = e >> 4;
    const int j = i0 + i + r - p.window;
    if (i0 + i < T && j >= 0 && j < T) {
      const float* pp = part + (hh * 4 * 16 + i) * 16 + r;
      S[i * SP + j] += (pp[0] + pp[256]) + (pp[512] + pp[768]);
    }
  }
  __syncthreads();
  PE_STAMP(0, 3);
  // ---- 2b. softmax over the valid keys: row = t4 / 16, 16 adjacent lanes per row
  {
    const int i = t4 >> 4, sj = t4 & 15;
    float* Sr = S + i * SP;
    const int Tpad = (T + AO_KCH - 1) / AO_KCH * AO_KCH;
    if (T <= 128) {
      float ev[8];
      float mx = -3.0e38f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int j = sj + 16 * k;
        ev[k] = j < T ? Sr[j] : -3.0e38f;
        mx = fmaxf(mx, ev[k]);
      }
      for (int m = 8; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m));
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int j = sj + 16 * k;
        ev[k] = j < T ? expf(ev[k] - mx) : 0.f;
        sum += ev[k];
      }
      for (int m = 8; m >= 1; m >>= 1) sum += __shfl_xor(sum, m);
      const float inv = 1.f / sum;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int j = sj + 16 * k;
        if (j < Tpad) Sr[j] = ev[k] * inv;
      }
    } else {
      float mx = -3.0e38f;
      for (int j = sj; j < T; j += 16) mx = fmaxf(mx, Sr[j]);
      for (int m = 8; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m));
      float sum = 0.f;
      for (int j = sj; j < T; j += 16) {
        const float e = expf(Sr[j] - mx);
        Sr[j] = e;
        sum += e;
      }
      for (int m = 8; m >= 1; m >>= 1) sum += __shfl_xor(sum, m);
      const float inv = 1.f / sum;
      for (int j = sj; j < Tpad; j += 16) Sr[j] = (j < T) ? Sr[j] * inv : 0.f;
    }
  }
  PE_STAMP(0, 4);
  // conv_o's weight row blocks and the operands of the LayerNorm tail: in flight under phase 3
  const int col = tid & 15, rl = tid >> 4, t = i0 + col;
  const bool ok = t < T;
  ColW<2 * NVT> gw;
  col_gemm16_fetch<2 * NVT>(gw, p.wo16, p.bo, H, H, H, wv, lane);
  float ov[NVT], gg[NVT], bb[NVT];
  {
    const pe_rowsrc od = pe_make_row(p.x + (long)b * p.x_bs, H * p.x_cs);
    const pe_rowsrc gd = pe_make_row(p.gamma, H), bd = pe_make_row(p.beta, H);
#pragma unroll
    for (int k = 0; k < NVT; ++k) {
      const int c = rl + 32 * k;
      ov[k] = pe_row_load(od, ok ? c * p.x_cs + t : -1);
      gg[k] = pe_row_load(gd, c);
      bb[k] = pe_row_load(bd, c);
    }
  }
  // ---- 3. O^T[d][q] = sum_key V[d][key] P[q][key]: channel tiles wt and wt + 4 of this head, A operand from the registers
  // loaded at entry, B operand P[q = l15][key] from the score rows (zeros from the length to the next multiple of 64)
  f32x4 o0, o1;
#pragma unroll
  for (int r = 0; r < 4; ++r) o0[r] = o1[r] = 0.f;
  __syncthreads();                                             // softmax finished
  // (a half that ends inside the utterance takes its fragments as they are; the one that straddles the length zeroes the
  // stale columns behind it -- a select per element costs matrix-pipe issue slots, profiles/r04_mfma_mix.txt)
#define AO_USE_HALF_(S0, K0, SEL, TWO)                                                \
  _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                       \
    const int key = (K0) + 16 * c + 4 * lq;                                             \
    float pf[4];                                                                        \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) pf[e] = S[l15 * SP + key + e];        \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                     \
      o0 = pe_mfma_16x16x4(SEL(key + e, va0[(S0) + c][e]), pf[e], o0);                  \
      if (TWO) o1 = pe_mfma_16x16x4(SEL(key + e, va1[(S0) + c][e]), pf[e], o1);         \
    }                                                                                   \
  }
#define AO_SEL_ALL(k, v) (v)
#define AO_SEL_LEN(k, v) ((k) < T ? (v) : 0.f)
#define AO_USE_HALF(S0, K0)                                                           \
  if ((K0) + AO_KCH <= T) {                                                           \
    if (two) { AO_USE_HALF_(S0, K0, AO_SEL_ALL, true) } else { AO_USE_HALF_(S0, K0, AO_SEL_ALL, false) }     \
  } else {                                                                            \
    if (two) { AO_USE_HALF_(S0, K0, AO_SEL_LEN, true) } else { AO_USE_HALF_(S0, K0, AO_SEL_LEN, false) }     \
  }
  for (int j0 = 0; j0 < T; j0 += 2 * AO_KCH) {
    AO_USE_HALF(0, j0)
    if (j0 + 2 * AO_KCH < T) { AO_LOAD_HALF(0, j0 + 2 * AO_KCH) }
    if (j0 + AO_KCH < T) {
      AO_USE_HALF(4, j0 + AO_KCH)
      if (j0 + 3 * AO_KCH < T) { AO_LOAD_HALF(4, j0 + 3 * AO_KCH) }
    }
  }
#undef AO_USE_HALF
#undef AO_USE_HALF_
#undef AO_SEL_ALL
#undef AO_SEL_LEN
#undef AO_LOAD_HALF
  {
    // relative-value band as three more k-steps: k index -> relative offset rr, A = rel_v[rr][d], B = p[q][q + rr - w]
    const int q = i0 + l15;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int rr = 4 * s + lq;
      const int j = q + rr - p.window;
      const float bvv = (rr < nrel && q < T && j >= 0 && j < T) ? S[l15 * SP + j] : 0.f;
      const float av0 = rr < nrel ? RV[rr * DK + wt * 16 + l15] : 0.f;
      o0 = pe_mfma_16x16x4(av0, bvv, o0);
      if (two) {
        const float av1 = rr < nrel ? RV[rr * DK + (wt + 4) * 16 + l15] : 0.f;
        o1 = pe_mfma_16x16x4(av1, bvv, o1);
      }
    }
  }
  PE_STAMP(0, 5);
  // (IN overlaps nothing phase 3 reads: no barrier in front of these stores)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    IN[(hh * DK + wt * 16 + 4 * lq + r) * NC + l15] = o0[r];
    if (two) IN[(hh * DK + (wt + 4) * 16 + 4 * lq + r) * NC + l15] = o1[r];
  }
  __syncthreads();
  PE_STAMP(0, 6);
  // ---- 4. conv_o + residual + norm_layers_1 (colchain_kernel mode 0)
  col_gemm16<2 * NVT, true, true>(p.wo16, p.bo, H, H, H, IN, wv, lane, [&](int row, int cc, float v) { Z[row * NC + cc] = v; }, &gw);
  __syncthreads();
  PE_STAMP(0, 7);
  int red_flip = 0;
  auto col_sum = [&](float x) -> float { return pe_col_sum16(x, red, red_flip, wv, lane, col); };
  float v[NVT];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NVT; ++k) {
    const int c = rl + 32 * k;
    v[k] = ok ? Z[c * NC + col] + ov[k] : 0.f;
    s += v[k];
  }
  const float mean = col_sum(s) / (float)H;
  float qq = 0.f;
#pragma unroll
  for (int k = 0; k < NVT; ++k) { const float d = v[k] - mean; qq = fmaf(d, d, qq); }
  const float rstd = 1.f / sqrtf(col_sum(qq) / (float)H + 1e-5f);
  if (!ok) return;
  float* ob = p.x + (long)b * p.x_bs;
#pragma unroll
  for (int k = 0; k < NVT; ++k) {
    const int c = rl + 32 * k;
    ob[(long)c * p.x_cs + t] = (v[k] - mean) * rstd * gg[k] + bb[k];
  }
  PE_STAMP(0, 8);
}

}  // namespace pe
