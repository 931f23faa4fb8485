// dds_layer16_kernel: one DDSConv layer per launch (+ optional ConvFlow.pre / proj / spline fusions), col_gemm16.
// (gfx950 / CDNA4 device code; reference arithmetic cited per kernel, paths relative to
// /root/reference/src/python/piper_train/vits/.)
#pragma once
#include "layernorm.h"
#include "spline.h"

namespace pe {

// One whole DDSConv layer (modules.py:119-128) per launch (out of place: neighbouring workgroups read each
// other's halo columns of x, so the result goes to a second buffer):
//     out = x + gelu(LN2(W1x1 . gelu(LN1(dwconv_dil(x) + b_dw)) + b_1x1))
// A workgroup (8 waves) owns 32 time columns and all H <= 256 channels: the depthwise conv + LN1 + GELU run in
// registers (thread = (column, channel lane), 16 channel lanes), the 1x1 conv is an [H x H] x [H x 32] GEMM on
// the f32 MFMAs with the activations in LDS and the pre-packed weights read from L2, LN2 + GELU + residual
// read the GEMM tile back from LDS. Replaces three launches (ln_kernel<2>, conv, ln_kernel<1>) and two
// round trips of the [H x T] activations through memory.
// Sum over the 32 channel lanes x 8 waves that share a column (512-thread, 16-column workgroups): lane pairs by
// shuffle, waves through `red` ([2][8][16] floats). The two halves of `red` alternate between calls, so a call costs
// ONE block barrier: half h is rewritten two calls after it was read, and the barrier of the call in between orders that.
__device__ __forceinline__ float pe_col_sum16(float v, float* red, int& flip, int wv, int lane, int col) {
  constexpr int NC = 16;
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  float* r = red + flip * 8 * NC;
  flip ^= 1;
  if (lane < NC) r[wv * NC + col] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < 8; ++w) s += r[w * NC + col];
  return s;
}

// rows x Kp GEMM over the 16 columns in IN[Kp][16] on the 16x16x4 MFMA; sink(row, col, value + bias[row]).
// Wave w owns the 16-row tiles w and w + 8 (then w + 16, w + 24, ...) and runs such a PAIR together: both weight row
// blocks are requested up front (one memory latency per pair, 2 * NQMAX float4 per lane), the B fragments are read from
// LDS once for both, and the two accumulator chains alternate on the MFMA pipe instead of each waiting on itself.
// k ascends inside and across the instructions of a tile exactly as in a one-tile-at-a-time loop: same fmaf chain.
// EXACT: Kp == 16 * NQMAX is known at compile time (no per-step guards in the unrolled loops: on the common shapes the
// guards were a scalar branch per LDS read, ~200 per launch).
// The weights do not depend on anything the kernel computes: col_gemm16_fetch requests the first pair's row blocks (the
// only pair for <= 256 rows) wherever the caller likes -- at kernel entry, under the phase that produces IN -- and
// col_gemm16<..., PRE = true> starts from them, so the GEMM phase does not open with a memory round trip.
template <int NQMAX>
struct ColW {
  f32x4 w0[NQMAX], w1[NQMAX];
  float bz0[4], bz1[4];
};
template <int NQMAX>
__device__ __forceinline__ void col_gemm16_fetch(ColW<NQMAX>& w, const float* wp16, const float* bias, int nbias,
                                                 int rows, int Kp, int mt, int lane) {
  const int lq = lane >> 4;
  const int nq = Kp / 16, ntile = (rows + 15) / 16, tile_floats = nq * 256;
  const pe_rowsrc biasd = pe_make_row(bias ? bias : wp16, bias ? nbias : 0);
  const bool one = PE_UNIFORM(mt < ntile), two = PE_UNIFORM(mt + 8 < ntile);
  const pe_rowsrc ws0 = pe_make_row_u(wp16 + (long)(one ? mt : 0) * tile_floats, one ? tile_floats : 0);
  const pe_rowsrc ws1 = pe_make_row_u(wp16 + (long)(two ? mt + 8 : 0) * tile_floats, two ? tile_floats : 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    w.bz0[r] = pe_row_load(biasd, one ? mt * 16 + 4 * lq + r : -1);
    w.bz1[r] = pe_row_load(biasd, two ? (mt + 8) * 16 + 4 * lq + r : -1);
  }
#pragma unroll
  for (int qq = 0; qq < NQMAX; ++qq) w.w0[qq] = pe_row_load4(ws0, qq * 256 + lane * 4);     // past nq: zeros
#pragma unroll
  for (int qq = 0; qq < NQMAX; ++qq) w.w1[qq] = pe_row_load4(ws1, qq * 256 + lane * 4);
  PE_SCHED_FENCE();
}
template <int NQMAX, bool EXACT, bool PRE = false, class Sink>
__device__ __forceinline__ void col_gemm16(const float* wp16, const float* bias, int nbias, int rows, int Kp_rt,
                                           const float* IN, int wv, int lane, Sink&& sink, ColW<NQMAX>* pre = nullptr) {
  constexpr int NC = 16;
  const int l15 = lane & 15, lq = lane >> 4;
  const int Kp = EXACT ? 16 * NQMAX : Kp_rt;
  const int nq = Kp / 16, ntile = (rows + 15) / 16;
  auto run_pair = [&](const int mt, const ColW<NQMAX>& W) {
    const bool two = PE_UNIFORM(mt + 8 < ntile);
    f32x4 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc0[r] = acc1[r] = 0.f;
#pragma unroll
    for (int q0 = 0; q0 < NQMAX; q0 += 4) {
      if (q0 < nq) {
        float yv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) yv[u] = (4 * q0 + u < Kp / 4) ? IN[(4 * (4 * q0 + u) + lq) * NC + l15] : 0.f;
        PE_SCHED_FENCE();
        if (two) {
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            acc0 = pe_mfma_16x16x4(W.w0[q0 + (u >> 2)][u & 3], yv[u], acc0);
            acc1 = pe_mfma_16x16x4(W.w1[q0 + (u >> 2)][u & 3], yv[u], acc1);
          }
        } else {
#pragma unroll
          for (int u = 0; u < 16; ++u) acc0 = pe_mfma_16x16x4(W.w0[q0 + (u >> 2)][u & 3], yv[u], acc0);
        }
        PE_SCHED_FENCE();
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) sink(mt * 16 + 4 * lq + r, l15, acc0[r] + W.bz0[r]);
    if (two) {
#pragma unroll
      for (int r = 0; r < 4; ++r) sink((mt + 8) * 16 + 4 * lq + r, l15, acc1[r] + W.bz1[r]);
    }
  };
  int mt = wv;
  if (PRE) {
    if (mt < ntile) run_pair(mt, *pre);
    mt += 16;
  }
  for (; mt < ntile; mt += 16) {
    ColW<NQMAX> wl;
    col_gemm16_fetch<NQMAX>(wl, wp16, bias, nbias, rows, Kp, mt, lane);
    run_pair(mt, wl);
  }
}

// One workgroup = 16 time columns x all channels, 512 threads; the 1x1 conv runs on the 16x16x4 f32 MFMA: the GEMM's
// N matches the column count, its 16-row tiles (Hp/16 = 12 for H = 192) spread evenly over the four SIMDs of the 8
// waves (three each), and a 128-id utterance still gives 8 workgroups. (A first version used 32 columns and the
// 32x32x2 MFMA: six row tiles on eight waves put two tiles on two of the SIMDs; 17.3 vs 11.0 us per launch.) k runs
// over the input channels in ascending order inside and across the instructions: the same fmaf chain. Weights: packed by engine_pack.cpp pack16 as
// [16-row tile][q][lane][4] with lane -> (row = lane & 15, k = lane >> 4) and step s = 4q + j covering ci = 4s + k.
template <int NVT>                              // NVT = channel slots per thread: ceil(Hp / 32)
__device__ __forceinline__ void dds_layer16_body(const DdsP& p, int ctile, int b, float* sm) {
  constexpr int NC = 16;                        // sm: Y[Hp][16] | Z[Hp][16] | red[2][8][16] | S[16][3][16] (spline tail)
  PE_STAMP(2, 0);
  // The utterance length lives in device memory. Nothing below uses it until every operand load has been issued
  // against the row stride instead (Lb): its latency overlaps theirs, and the taps beyond the length are zeroed
  // afterwards -- the conv's zero padding at the end of the utterance.
  const int L = p.lens[b];
  const int t0 = ctile * NC;
  const int Lb = p.x_cs;
  // NVT = 3 / 6: instantiated for exactly Hp = 32 * NVT (the launcher checks); NVT = 8 is the generic form (any Hp <= 256)
  const int H = p.H, Hp = NVT != 8 ? 32 * NVT : p.nchunks * 32;
  float* Y = sm;
  float* Z = Y + Hp * NC;
  float* red = Z + Hp * NC;
  const int tid = threadIdx.x, col = tid & 15, rl = tid >> 4, wv = PE_UNIFORM(tid >> 6), lane = tid & 63;
  const int t = t0 + col;
  const bool okb = t < Lb;
  const float* xb = p.x + (long)b * p.x_bs;
  float* ob = p.out + (long)b * p.o_bs;
  const int pad = (p.dw_k - 1) / 2 * p.dw_dil;
  const bool fold = p.pre_z != nullptr;
  // this wave's 1x1-conv weight row blocks: in flight under phase 1
  ColW<2 * NVT> gw;
  col_gemm16_fetch<2 * NVT>(gw, p.wp16, p.bias, H, Hp, Hp, wv, lane);

  int red_flip = 0;
  auto col_sum = [&](float x) -> float { return pe_col_sum16(x, red, red_flip, wv, lane, col); };

  // ---- phase 1: depthwise conv, LN1, GELU -> Y (all operands requested up front through descriptors)
  constexpr int MAXK = 3;
  const pe_rowsrc xd = pe_make_row(xb, H * p.x_cs);
  const pe_rowsrc wd = pe_make_row(p.dw_w, H * p.dw_k), bd = pe_make_row(p.dw_b, H);
  const pe_rowsrc g1d = pe_make_row(p.g1, H), b1d = pe_make_row(p.b1, H);
  float v[NVT], xc[NVT], gg[NVT], bb[NVT];
  bool ok;                                        // t < L, set once the operand loads are in flight
  {
    float xv[NVT][MAXK], ww[NVT][MAXK], wb[NVT];
    // folded ConvFlow.pre: the three taps' z0 values and this channel's (w, b); zero-length descriptors when unused
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
      const int c = rl + 32 * k;
      const bool cv = okb && c < H;
#pragma unroll
      for (int kk = 0; kk < MAXK; ++kk) {
        const int tt = t + kk * p.dw_dil - pad;
        const bool tv = cv && kk < p.dw_k && tt >= 0 && tt < Lb;
        xv[k][kk] = pe_row_load(xd, tv ? c * p.x_cs + tt : -1);
        ww[k][kk] = pe_row_load(wd, tv ? c * p.dw_k + kk : -1);
      }
      wb[k] = pe_row_load(bd, cv ? c : -1);
      gg[k] = pe_row_load(g1d, c < H ? c : -1);
      bb[k] = pe_row_load(b1d, c < H ? c : -1);
      pw[k] = pe_row_load(pwd, cv ? c : -1);
      pb[k] = pe_row_load(pbd, cv ? c : -1);
    }
    // first use of the length
    if (t0 >= L) return;
    PE_STAMP(2, 1);
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
          const bool tv = ok && rl + 32 * k < H && kk < p.dw_k && tt >= 0 && tt < L;
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
  PE_STAMP(2, 2);
  float mean = col_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < NVT; ++k)
    if (rl + 32 * k < H) { const float d = v[k] - mean; q = fmaf(d, d, q); }
  float rstd = 1.f / sqrtf(col_sum(q) / (float)H + 1e-5f);
  PE_STAMP(2, 3);
#pragma unroll
  for (int k = 0; k < NVT; ++k) {
    const int c = rl + 32 * k;
    if (c < Hp) Y[c * NC + col] = (c < H && ok) ? gelu_erf((v[k] - mean) * rstd * gg[k] + bb[k]) : 0.f;
  }
  // LN2 gains: needed in phase 3, in flight during the GEMM
  const pe_rowsrc g2d = pe_make_row(p.g2, H), b2d = pe_make_row(p.b2, H);
#pragma unroll
  for (int k = 0; k < NVT; ++k) {
    const int c = rl + 32 * k;
    gg[k] = pe_row_load(g2d, c < H ? c : -1);
    bb[k] = pe_row_load(b2d, c < H ? c : -1);
  }
  __syncthreads();

  // ---- phase 2: Z = W1x1 . Y + bias on 16x16x4 MFMAs (col_gemm16: tiles w and w+8 of a wave run as a pair)
  PE_STAMP(2, 4);
  col_gemm16<2 * NVT, NVT != 8, true>(p.wp16, p.bias, H, Hp, Hp, Y, wv, lane, [&](int row, int cc, float val) { Z[row * NC + cc] = val; }, &gw);
  PE_STAMP(2, 5);
  __syncthreads();
  PE_STAMP(2, 6);

  // ---- phase 3: LN2, GELU, residual -> out
  s = 0.f;
#pragma unroll
  for (int k = 0; k < NVT; ++k) {
    const int c = rl + 32 * k;
    v[k] = (c < H) ? Z[c * NC + col] : 0.f;
    s += v[k];
  }
  mean = col_sum(s) / (float)H;
  q = 0.f;
#pragma unroll
  for (int k = 0; k < NVT; ++k)
    if (rl + 32 * k < H) { const float d = v[k] - mean; q = fmaf(d, d, q); }
  rstd = 1.f / sqrtf(col_sum(q) / (float)H + 1e-5f);
  if (p.post_w16 == nullptr) {
    if (!ok) return;
#pragma unroll
    for (int k = 0; k < NVT; ++k) {
      const int c = rl + 32 * k;
      if (c < H) {
        const float y = xc[k] + gelu_erf((v[k] - mean) * rstd * gg[k] + bb[k]);
        ob[(long)c * p.o_cs + t] = y;
      }
    }
    PE_STAMP(2, 7);
    return;
  }
  // ---- phase 4 (last layer of a DDSConv): the following 1x1 conv on this workgroup's columns, Y <- layer output
#pragma unroll
  for (int k = 0; k < NVT; ++k) {
    const int c = rl + 32 * k;
    if (c < Hp) Y[c * NC + col] = (c < H && ok) ? xc[k] + gelu_erf((v[k] - mean) * rstd * gg[k] + bb[k]) : 0.f;
  }
  __syncthreads();
  col_gemm16<2 * NVT, NVT != 8>(p.post_w16, p.post_bias, p.post_rows, p.post_rows, Hp, Y, wv, lane,
                      [&](int row, int cc, float val) { Z[row * NC + cc] = val; });
  __syncthreads();
  if (p.post_out && ok) {
    float* po = p.post_out + (long)b * p.po_bs;
#pragma unroll
    for (int k = 0; k < NVT; ++k) {
      const int c = rl + 32 * k;
      if (c < p.post_rows) {
        po[(long)c * p.po_cs + t] = Z[c * NC + col];
      }
    }
  }
  if (p.zout) {
    // ConvFlow's spline on z1 (z0 passes through, scaled). The ~40 transcendentals of one position are spread over 16
    // lanes (lane j: softmax terms of bin j, derivative j), the order-sensitive sums run in one lane afterwards.
    constexpr int NB = SPL_NB;
    // (a region of its own: it used to alias Y, which is smaller than the 768 floats below 48 padded channels -- the tail of
    // S then ran into the rows of Z that other waves were still reading. No reference quality is that narrow, the tiny test
    // voices are; the emulator's wave-by-wave schedule found it, tests/emu/hip_emu.cpp)
    float* S = red + 2 * 8 * NC;                       // [16 cols][3][16]
    const int scol = tid >> 4, j = tid & 15;           // first 256 threads: 16 consecutive lanes per column
    const int st = t0 + scol;
    if (tid < 256) {
      const float uwj = j < NB ? Z[j * NC + scol] * p.inv_sqrt_h : -3.0e38f;
      const float uhj = j < NB ? Z[(NB + j) * NC + scol] * p.inv_sqrt_h : -3.0e38f;
      float mw = uwj, mh = uhj;
#pragma unroll
      for (int m = 8; m >= 1; m >>= 1) { mw = fmaxf(mw, __shfl_xor(mw, m)); mh = fmaxf(mh, __shfl_xor(mh, m)); }
      S[(scol * 3 + 0) * 16 + j] = j < NB ? expf(uwj - mw) : 0.f;
      S[(scol * 3 + 1) * 16 + j] = j < NB ? expf(uhj - mh) : 0.f;
      S[(scol * 3 + 2) * 16 + j] = j <= NB ? spline_deriv((j == 0 || j >= NB) ? 0.f : Z[(2 * NB + j - 1) * NC + scol], j == 0 || j >= NB) : 0.f;
    }
    __syncthreads();
    if (tid < 256 && j == 0 && st < L) {
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
  (void)ok;
}

template <int NVT>
__global__ __launch_bounds__(512) void dds_layer16_kernel(DdsP p) {
  PE_KTRACE(2);
  PE_DYN_SMEM(float, sm);
  dds_layer16_body<NVT>(p, blockIdx.x, blockIdx.y, sm);
}

}  // namespace pe
