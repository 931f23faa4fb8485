// colchain_kernel / lngemm_kernel: short chains of 192-channel 1x1 convs and LayerNorms in one launch.
// (gfx950 / CDNA4 device code; reference arithmetic cited per kernel, paths relative to
// /root/reference/src/python/piper_train/vits/.)
#pragma once
#include "dds.h"

namespace pe {

// Short chains of 1x1 convs whose GEMMs are small enough for one workgroup to own ALL output rows of a 16-column tile,
// so that what follows the GEMM (a LayerNorm over channels, or a second GEMM over the result) needs no second launch
// and no trip through HBM. Small batches only (the tiled conv kernels win when there are columns to fill the chip):
//   mode 0   out = LN(res + W1.in + b1)              attention conv_o + residual + norm_layers_1 (attentions.py:70-72)
//   mode 1   x1 -= W1.in + b1 ; out2 = W2.x1 + b2    ResidualCouplingLayer.post + mean-only reverse update, then the
//                                                    NEXT coupling layer's pre over the updated half -- the Flip between
//                                                    them is folded into the packed weights (modules.py:455-466, 433)
// Same 16x16x4 MFMA GEMM as dds_layer16_kernel: weights in pack16 order, B operand = the input columns in LDS.

template <int NVT>                              // NVT = channel slots per thread: every channel count on the chain <= 32 * NVT
__global__ __launch_bounds__(512) void colchain_kernel(ColP p) {
  PE_KTRACE(3);
  constexpr int NC = 16;
  PE_DYN_SMEM(float, sm);                       // IN[32 NVT][16] | Z[32 NVT][16] | red[2][8][16]
  PE_STAMP(p.mode == 0 ? 3 : (p.w2 ? 6 : 7), 0);
  const int b = blockIdx.y, L = p.lens[b];
  const int t0 = blockIdx.x * NC;
  if (t0 >= L) return;
  PE_STAMP(p.mode == 0 ? 3 : (p.w2 ? 6 : 7), 1);
  float* IN = sm;
  float* Z = IN + 32 * NVT * NC;
  float* red = Z + 32 * NVT * NC;
  const int tid = threadIdx.x, col = tid & 15, rl = tid >> 4, wv = PE_UNIFORM(tid >> 6), lane = tid & 63;
  const int t = t0 + col;
  const bool ok = t < L;
  constexpr int K1p = 32 * NVT;                  // the launcher checks: K1 == 32 NVT, and rows1 == 16 NVT in mode 1
  ColW<2 * NVT> gw;                              // first GEMM's weight row blocks, in flight under the input staging
  col_gemm16_fetch<2 * NVT>(gw, p.w1, p.b1, p.rows1, p.rows1, K1p, wv, lane);

  // operands of the step after the first GEMM are requested before it: residual / previous x1, LN gains
  float ov[NVT], gg[NVT], bb[NVT];
  {
    const pe_rowsrc ind = pe_make_row(p.in1 + (long)b * p.in1_bs, p.K1 * p.in1_cs);
    const int nrow = p.mode == 0 ? p.rows1 : p.rows1;
    const float* ob = p.mode == 0 ? p.res + (long)b * p.res_bs : p.x1 + (long)b * p.x1_bs;
    const int ocs = p.mode == 0 ? p.res_cs : p.x1_cs;
    const pe_rowsrc od = pe_make_row(ob, nrow * ocs);
    const pe_rowsrc gd = pe_make_row(p.mode == 0 ? p.gamma : p.w1, p.mode == 0 ? p.rows1 : 0);
    const pe_rowsrc bd = pe_make_row(p.mode == 0 ? p.beta : p.w1, p.mode == 0 ? p.rows1 : 0);
    float xin[NVT];
#pragma unroll
    for (int k = 0; k < NVT; ++k) {
      const int c = rl + 32 * k;
      xin[k] = pe_row_load(ind, (ok && c < p.K1) ? c * p.in1_cs + t : -1);
      ov[k] = pe_row_load(od, (ok && c < nrow) ? c * ocs + t : -1);
      gg[k] = pe_row_load(gd, c < p.rows1 ? c : -1);
      bb[k] = pe_row_load(bd, c < p.rows1 ? c : -1);
    }
#pragma unroll
    for (int k = 0; k < NVT; ++k) {
      const int c = rl + 32 * k;
      if (c < K1p) IN[c * NC + col] = xin[k];
    }
  }
  PE_STAMP(p.mode == 0 ? 3 : (p.w2 ? 6 : 7), 2);
  __syncthreads();
  PE_STAMP(p.mode == 0 ? 3 : (p.w2 ? 6 : 7), 3);
  col_gemm16<2 * NVT, true, true>(p.w1, p.b1, p.rows1, p.rows1, K1p, IN, wv, lane, [&](int row, int cc, float v) { Z[row * NC + cc] = v; }, &gw);
  // second GEMM's weights (the next layer's pre): in flight under the x1 update
  ColW<NVT> gw2;
  constexpr int K2p = 16 * NVT;
  const bool second = p.mode == 1 && p.w2;
  if (second) col_gemm16_fetch<NVT>(gw2, p.w2, p.b2, p.rows2, p.rows2, K2p, wv, lane);
  PE_STAMP(p.mode == 0 ? 3 : (p.w2 ? 6 : 7), 4);
  __syncthreads();
  PE_STAMP(p.mode == 0 ? 3 : (p.w2 ? 6 : 7), 5);

  if (p.mode == 0) {
    int red_flip = 0;
    auto col_sum = [&](float x) -> float { return pe_col_sum16(x, red, red_flip, wv, lane, col); };
    const int H = p.rows1;
    float v[NVT];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NVT; ++k) {
      const int c = rl + 32 * k;
      v[k] = (ok && c < H) ? Z[c * NC + col] + ov[k] : 0.f;
      s += v[k];
    }
    const float mean = col_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NVT; ++k)
      if (rl + 32 * k < H) { const float d = v[k] - mean; q = fmaf(d, d, q); }
    const float rstd = 1.f / sqrtf(col_sum(q) / (float)H + 1e-5f);
    if (!ok) return;
    float* ob = p.out + (long)b * p.out_bs;
#pragma unroll
    for (int k = 0; k < NVT; ++k) {
      const int c = rl + 32 * k;
      if (c < H) ob[(long)c * p.out_cs + t] = (v[k] - mean) * rstd * gg[k] + bb[k];
    }
    PE_STAMP(p.mode == 0 ? 3 : (p.w2 ? 6 : 7), 6);
    return;
  }

  // mode 1: x1 <- x1 - (post + bias); the updated half is the next layer's x0
  {
    float* xb = p.x1 + (long)b * p.x1_bs;
#pragma unroll
    for (int k = 0; k < NVT; ++k) {
      const int c = rl + 32 * k;
      const float xn = (ok && c < p.rows1) ? ov[k] - Z[c * NC + col] : 0.f;
      if (ok && c < p.rows1) xb[(long)c * p.x1_cs + t] = xn;
      if (c < K2p) IN[c * NC + col] = xn;
    }
    if (!second) return;
    __syncthreads();
    PE_STAMP(p.mode == 0 ? 3 : (p.w2 ? 6 : 7), 7);
    float* o2 = p.out2 + (long)b * p.o2_bs;
    auto st2 = [&](int row, int cc, float v) {
      if (row < p.rows2 && t0 + cc < L) o2[(long)row * p.o2_cs + t0 + cc] = v;
    };
    col_gemm16<NVT, true, true>(p.w2, p.b2, p.rows2, p.rows2, K2p, IN, wv, lane, st2, &gw2);   // K = half the channels
    PE_STAMP(p.mode == 0 ? 3 : (p.w2 ? 6 : 7), 8);
  }
}

// ------------------------------------------------------------------------------------------------
// norm_layers_2 of an encoder layer fused with the 1x1 conv that consumes it -- the next layer's q/k/v conv, or proj
// after the last layer (attentions.py:73-74, 60-69; models.py:207): one workgroup = 16 columns x one 192-row part of
// the GEMM (grid.z = parts: 3 for q/k/v, 2 for proj). Every part normalises its 16 columns itself (cheap next to a
// launch); part 0 also writes LN(y) back for the residual readers. Small batches only, like colchain_kernel.
template <int NVT>                              // channels == 32 * NVT exactly (the launcher checks)
__global__ __launch_bounds__(512) void lngemm_kernel(LnGemmP p) {
  PE_KTRACE(7);
  constexpr int NC = 16, H = 32 * NVT;
  PE_DYN_SMEM(float, sm);                       // IN[H][16] | red[2][8][16]
  const int b = blockIdx.y, L = p.lens[b];
  const int t0 = blockIdx.x * NC;
  if (t0 >= L) return;
  float* IN = sm;
  float* red = IN + H * NC;
  const int tid = threadIdx.x, col = tid & 15, rl = tid >> 4, wv = PE_UNIFORM(tid >> 6), lane = tid & 63;
  const int t = t0 + col;
  const bool ok = t < L;
  const int part = blockIdx.z, row0 = part * H;
  const int rows_here = p.rows - row0 < H ? p.rows - row0 : H;
  const float* wpart = p.w16 + (long)part * (2 * NVT) * (2 * NVT) * 256;      // 2 NVT row tiles of 2 NVT * 256 floats each
  const float* bpart = p.bias ? p.bias + row0 : nullptr;
  ColW<2 * NVT> gw;
  col_gemm16_fetch<2 * NVT>(gw, wpart, bpart, rows_here, rows_here, H, wv, lane);
  float v[NVT], gg[NVT], bb[NVT];
  {
    const pe_rowsrc ind = pe_make_row(p.in + (long)b * p.in_bs, H * p.in_cs);
    const pe_rowsrc gd = pe_make_row(p.gamma, H), bd = pe_make_row(p.beta, H);
#pragma unroll
    for (int k = 0; k < NVT; ++k) {
      const int c = rl + 32 * k;
      v[k] = pe_row_load(ind, ok ? c * p.in_cs + t : -1);
      gg[k] = pe_row_load(gd, c);
      bb[k] = pe_row_load(bd, c);
    }
  }
  int red_flip = 0;
  auto col_sum = [&](float x) -> float { return pe_col_sum16(x, red, red_flip, wv, lane, col); };
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NVT; ++k) s += v[k];
  const float mean = col_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < NVT; ++k) { const float d = v[k] - mean; q = fmaf(d, d, q); }
  const float rstd = 1.f / sqrtf(col_sum(q) / (float)H + 1e-5f);
  float* xo = p.xout + (long)b * p.x_bs;
#pragma unroll
  for (int k = 0; k < NVT; ++k) {
    const int c = rl + 32 * k;
    const float y = ok ? (v[k] - mean) * rstd * gg[k] + bb[k] : 0.f;
    IN[c * NC + col] = y;
    if (part == 0 && ok) xo[(long)c * p.x_cs + t] = y;
  }
  __syncthreads();
  float* ob = p.out + (long)b * p.o_bs + (long)row0 * p.o_cs;
  col_gemm16<2 * NVT, true, true>(wpart, bpart, rows_here, rows_here, H, IN, wv, lane, [&](int row, int cc, float val) {
    if (row < rows_here && t0 + cc < L) ob[(long)row * p.o_cs + t0 + cc] = val;
  }, &gw);
}

}  // namespace pe
