// dds_layer4_kernel: the DDSConv layer launch of dds.h on FOUR-column workgroups, for calls of a few hundred columns.
// (gfx950 / CDNA4 device code; reference arithmetic cited per kernel, paths relative to
// /root/reference/src/python/piper_train/vits/.)
#pragma once
#include "dds.h"
#include "col4.h"

namespace pe {

// Why: a 128-id utterance gives dds_layer16_kernel 8 workgroups -- 8 of 256 CUs -- and inside each the time goes to the
// element-wise phases (depthwise conv, two LayerNorms, two erf-GELUs over 192 x 16 values on 8 waves: ~5 of ~9 us) and
// to a GEMM that keeps three 16-row tiles per SIMD busy for 3.5 us (profiles/r03_notes.md). Both are bound inside the
// ONE CU that runs the workgroup, so the lever is fewer columns per workgroup on more CUs:
//   * a workgroup = 4 time columns x all 192 channels, 256 threads (one wave per SIMD): thread = (column, channel lane
//     0..63) keeps channels rl, rl + 64, rl + 128 -- a quarter of the element-wise work per SIMD;
//   * the 1x1 conv runs on v_mfma_f32_4x4x1_16B_f32 (16 blocks of a 4x4 outer product: 64 output rows x 4 columns per
//     instruction, 8 cycles -- the same 32 MAC/cycle as the 16x16x4 form, on 4 columns instead of 16). The K = 192 input
//     channels are dealt to the four waves (48 each): a wave runs all three 64-row tiles over its K range (144 MFMAs,
//     three independent accumulator chains) from weights requested at kernel entry (36 float4 per lane), and the four
//     partial tiles meet in LDS in a fixed order (deterministic; within a wave k ascends like the fmaf chain of dds.h);
//   * everything else -- ConvFlow.pre folded into the first layer's input, the second 1x1 conv (dp.proj / ConvFlow.proj)
//     and the rational-quadratic spline behind the last layer -- as in dds_layer16_kernel.
// Weights: engine_pack.cpp pack4 -- [64-row tile][k quad = K/4][lane][4]: lane l <-> row 64 * tile + l, element j <-> input
// channel 4 * quad + j. Each workgroup streams the layer's 147 KB of weights from L2, so the form is for small calls
// only (engine_launch.cpp: Engine::dds; 4x the workgroups of the 16-column form read 4x the weight bytes).
__global__ __launch_bounds__(256) void dds_layer4_kernel(DdsP p) {
  PE_KTRACE(2);
  PE_DYN_SMEM(float, sm);                       // YT[4][KS] | P[4 waves][192][4] | red[2][4][4] | ZL[64][4]
  constexpr int NC = 4, NVT = 3, H = C4_H, C4_KS = Col4W<C4_H>::KS;
  const int b = blockIdx.y;
  const int L = p.lens[b];                      // first used after every operand load is in flight (dds.h)
  const int t0 = c4_tile(blockIdx.x, gridDim.x, p.xcd) * NC;
  const int Lb = p.x_cs;
  float* YT = sm;
  float* P = YT + NC * C4_KS;
  float* red = P + 4 * H * NC;
  float* ZL = red + 32;
  const int tid = threadIdx.x, col = tid & 3, rl = tid >> 2, wv = PE_UNIFORM(tid >> 6), lane = tid & 63;
  const int t = t0 + col;
  const bool okb = t < Lb;
  const float* xb = p.x + (long)b * p.x_bs;
  float* ob = p.out + (long)b * p.o_bs;
  const int pad = (p.dw_k - 1) / 2 * p.dw_dil;
  const bool fold = p.pre_z != nullptr;
  // this wave's 1x1-conv weight fragments: in flight under phase 1, requested BEHIND phase 1's own operands (the memory
  // counter retires in order: in front of them their whole fetch sat on phase 1's critical path -- col4.h, r04_notes.md)
  Col4W<C4_H> gw;

  int red_flip = 0;
  auto col_sum = [&](float x) -> float { return pe_col_sum4(x, red, red_flip, wv, lane, col); };

  // ---- phase 1: depthwise conv, LN1, GELU -> YT (all operands requested up front through descriptors)
  constexpr int MAXK = 3;
  const pe_rowsrc xd = pe_make_row(xb, H * p.x_cs);
  const pe_rowsrc wd = pe_make_row(p.dw_w, H * p.dw_k), bd = pe_make_row(p.dw_b, H);
  const pe_rowsrc g1d = pe_make_row(p.g1, H), b1d = pe_make_row(p.b1, H);
  float v[NVT], xc[NVT], gg[NVT], bb[NVT];
  bool ok;
  {
    float xv[NVT][MAXK], ww[NVT][MAXK], wb[NVT];
    const pe_rowsrc zd = pe_make_row(fold ? p.pre_z + (long)b * p.pre_z_bs : p.dw_b, fold ? Lb : 0);
    const pe_rowsrc pwd = pe_make_row(fold ? p.pre_w : p.dw_b, fold ? H : 0), pbd = pe_make_row(fold ? p.pre_b : p.dw_b, fold ? H : 0);
    float zt[MAXK], pw[NVT], pb[NVT];
#pragma unroll
    for (int kk = 0; kk < MAXK; ++kk) {
      const int tt = t + kk * p.dw_dil - pad;
      const bool tv = okb && kk < p.dw_k && tt >= 0 && tt < Lb;
      zt[kk] = pe_row_load(zd, tv ? tt : -1) * p.z_scale;
    }
#pragma unroll
    for (int k = 0; k < NVT; ++k) {
      const int c = rl + 64 * k;
#pragma unroll
      for (int kk = 0; kk < MAXK; ++kk) {
        const int tt = t + kk * p.dw_dil - pad;
        const bool tv = okb && kk < p.dw_k && tt >= 0 && tt < Lb;
        xv[k][kk] = pe_row_load(xd, tv ? c * p.x_cs + tt : -1);
        ww[k][kk] = pe_row_load(wd, tv ? c * p.dw_k + kk : -1);
      }
      wb[k] = pe_row_load(bd, okb ? c : -1);
      gg[k] = pe_row_load(g1d, c);
      bb[k] = pe_row_load(b1d, c);
      pw[k] = pe_row_load(pwd, okb ? c : -1);
      pb[k] = pe_row_load(pbd, okb ? c : -1);
    }
    PE_SCHED_FENCE();
    col_gemm4_fetch<C4_H>(gw, p.wp4, C4_NT, wv, lane);
    // first use of the length
    if (t0 >= L) return;
#pragma unroll
    for (int kk = 0; kk < MAXK; ++kk) {
      const int tt = t + kk * p.dw_dil - pad;
      const bool in = t < L && tt < L;
      zt[kk] = in ? zt[kk] : 0.f;
#pragma unroll
      for (int k = 0; k < NVT; ++k) xv[k][kk] = in ? xv[k][kk] : 0.f;
    }
    ok = t < L;
    if (fold) {
#pragma unroll
      for (int k = 0; k < NVT; ++k)
#pragma unroll
        for (int kk = 0; kk < MAXK; ++kk) {
          const int tt = t + kk * p.dw_dil - pad;
          const bool tv = ok && kk < p.dw_k && tt >= 0 && tt < L;
          xv[k][kk] = tv ? fmaf(pw[k], zt[kk], pb[k]) + xv[k][kk] : 0.f;
        }
    }
#pragma unroll
    for (int k = 0; k < NVT; ++k) {
      float a = wb[k];
#pragma unroll
      for (int kk = 0; kk < MAXK; ++kk) a = fmaf(ww[k][kk], xv[k][kk], a);
      v[k] = a;
      xc[k] = xv[k][(MAXK - 1) / 2];     // centre tap = x[c][t] (odd kernel, "same" padding)
    }
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NVT; ++k) s += v[k];
  float mean = col_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < NVT; ++k) { const float d = v[k] - mean; q = fmaf(d, d, q); }
  float rstd = 1.f / sqrtf(col_sum(q) / (float)H + 1e-5f);
#pragma unroll
  for (int k = 0; k < NVT; ++k)
    YT[col * C4_KS + rl + 64 * k] = ok ? gelu_erf((v[k] - mean) * rstd * gg[k] + bb[k]) : 0.f;
  // LN2 gains and the 1x1 conv's bias: needed in phase 3, in flight during the GEMM
  const pe_rowsrc g2d = pe_make_row(p.g2, H), b2d = pe_make_row(p.b2, H), cbd = pe_make_row(p.bias, H);
  float cb[NVT];
#pragma unroll
  for (int k = 0; k < NVT; ++k) {
    const int c = rl + 64 * k;
    gg[k] = pe_row_load(g2d, c);
    bb[k] = pe_row_load(b2d, c);
    cb[k] = pe_row_load(cbd, c);
  }
  __syncthreads();

  // ---- phase 2: P <- per-wave partial products of W1x1 . Y
  col_gemm4_run<C4_H>(gw, YT, P, wv, lane);
  // the following 1x1 conv's fragments (last layer of a DDSConv): requested now, in flight under phase 3
  const int post_nt = p.post_w4 ? (p.post_rows + 63) / 64 : 0;
  if (p.post_w4) col_gemm4_fetch<C4_H>(gw, p.post_w4, post_nt, wv, lane);
  __syncthreads();

  // ---- phase 3: LN2, GELU, residual -> out
  s = 0.f;
#pragma unroll
  for (int k = 0; k < NVT; ++k) {
    v[k] = col_gemm4_get(P, rl + 64 * k, col) + cb[k];
    s += v[k];
  }
  mean = col_sum(s) / (float)H;
  q = 0.f;
#pragma unroll
  for (int k = 0; k < NVT; ++k) { const float d = v[k] - mean; q = fmaf(d, d, q); }
  rstd = 1.f / sqrtf(col_sum(q) / (float)H + 1e-5f);
  if (p.post_w4 == nullptr) {
    if (!ok) return;
#pragma unroll
    for (int k = 0; k < NVT; ++k) {
      const int c = rl + 64 * k;
      ob[(long)c * p.o_cs + t] = xc[k] + gelu_erf((v[k] - mean) * rstd * gg[k] + bb[k]);
    }
    return;
  }
  // ---- phase 4 (last layer of a DDSConv): the following 1x1 conv on this workgroup's columns, YT <- layer output
  // (the col_sum barriers above ordered every wave's phase-2 reads of YT before these writes)
#pragma unroll
  for (int k = 0; k < NVT; ++k)
    YT[col * C4_KS + rl + 64 * k] = ok ? xc[k] + gelu_erf((v[k] - mean) * rstd * gg[k] + bb[k]) : 0.f;
  const pe_rowsrc pbd2 = pe_make_row(p.post_bias, p.post_rows);
  float pbv[NVT];
#pragma unroll
  for (int k = 0; k < NVT; ++k) pbv[k] = pe_row_load(pbd2, rl + 64 * k);      // rows >= post_rows: 0
  __syncthreads();                                // YT complete; every phase-3 read of P done
  col_gemm4_run<C4_H>(gw, YT, P, wv, lane);
  __syncthreads();
  if (p.post_out && ok) {
    float* po = p.post_out + (long)b * p.po_bs;
#pragma unroll
    for (int k = 0; k < NVT; ++k) {
      const int c = rl + 64 * k;
      if (c < p.post_rows) po[(long)c * p.po_cs + t] = col_gemm4_get(P, c, col) + pbv[k];
    }
  }
  if (p.zout) {
    // ConvFlow's spline on z1 (z0 passes through, scaled), as in dds_layer16_kernel: ZL[row][4] <- the 29 parameter rows,
    // then 16 lanes per position for the transcendentals, one lane for the order-sensitive sums
    constexpr int NB = SPL_NB;
    ZL[rl * NC + col] = col_gemm4_get(P, rl, col) + pbv[0];      // rows 0..63 (k = 0 slot of every thread)
    __syncthreads();
    float* S = YT;                                     // YT is free: [4 cols][3][16]
    const int scol = tid >> 4, j = tid & 15;           // first 64 threads: 16 consecutive lanes per column
    const int st = t0 + scol;
    if (tid < 64) {
      const float uwj = j < NB ? ZL[j * NC + scol] * p.inv_sqrt_h : -3.0e38f;
      const float uhj = j < NB ? ZL[(NB + j) * NC + scol] * p.inv_sqrt_h : -3.0e38f;
      float mw = uwj, mh = uhj;
#pragma unroll
      for (int m = 8; m >= 1; m >>= 1) { mw = fmaxf(mw, __shfl_xor(mw, m)); mh = fmaxf(mh, __shfl_xor(mh, m)); }
      S[(scol * 3 + 0) * 16 + j] = j < NB ? expf(uwj - mw) : 0.f;
      S[(scol * 3 + 1) * 16 + j] = j < NB ? expf(uhj - mh) : 0.f;
      S[(scol * 3 + 2) * 16 + j] = j <= NB ? spline_deriv((j == 0 || j >= NB) ? 0.f : ZL[(2 * NB + j - 1) * NC + scol], j == 0 || j >= NB) : 0.f;
    }
    __syncthreads();
    if (tid < 64 && j == 0 && st < L) {
      float uw[NB], uh[NB], dv[NB + 1];
#pragma unroll
      for (int i = 0; i < NB; ++i) { uw[i] = S[(scol * 3 + 0) * 16 + i]; uh[i] = S[(scol * 3 + 1) * 16 + i]; }
#pragma unroll
      for (int i = 0; i <= NB; ++i) dv[i] = S[(scol * 3 + 2) * 16 + i];
      const float* zi = p.zin + (long)b * p.zin_bs;
      float* zo = p.zout + (long)b * p.zout_bs;
      const float x1 = zi[(long)p.c1 * p.z_cs + st] * p.z_scale;
      const float x0 = zi[(long)p.c0 * p.z_cs + st] * p.z_scale;
      zo[(long)p.c1 * p.z_cs + st] = (x1 >= -5.0f && x1 <= 5.0f) ? spline_finish(uw, uh, dv, x1) : x1;
      zo[(long)p.c0 * p.z_cs + st] = x0;
    }
  }
}

}  // namespace pe
