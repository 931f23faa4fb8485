// The 4-column GEMM on v_mfma_f32_4x4x1 and the small-call kernels built on it besides dds_layer4_kernel (dds4.h):
// colchain4_kernel and lngemm4_kernel, the 4-column forms of colchain.h's kernels.
// (gfx950 / CDNA4 device code; reference arithmetic cited per kernel, paths relative to
// /root/reference/src/python/piper_train/vits/.)
#pragma once
#include "dds.h"

namespace pe {

// Why 4 columns: see dds4.h -- a 16-column workgroup of the 192-channel chains is bound inside its ONE CU (element-wise
// phases on 8 waves + three 16-row GEMM tiles per SIMD) while a 128-id utterance occupies 8 of 256 CUs; 4-column
// workgroups put a quarter of that work on each of 4x the CUs. Weights: engine_pack.cpp pack4 -- [64-row tile][k quad = K/4]
// [lane][4]: lane l <-> row 64 * tile + l, element j <-> input channel 4 * quad + j.
// Which 4-column tile a workgroup takes. Workgroups go to the 8 XCDs round-robin by linear id, so with tile = blockIdx.x
// the eight 16-byte pieces of a 128-byte line of any [channel][time] tensor the launch writes would come from eight
// different L2s (measured: the attention kernel behind such a launch ran 4.4 us slower). XCD j takes a contiguous run of
// tiles instead (when the probe at engine creation saw that round-robin): 64 bytes per row from one L2 for a 128-id utterance, what a 16-column workgroup writes.
// `P` = the number of XCDs when the dispatch was seen to be round-robin over them at engine creation (xcc_probe_kernel),
// 0 otherwise: then tile = blockIdx.x. (The round-robin runs over the LINEAR workgroup id, so in row (y, z) of the grid
// workgroup x sits on XCD (base(y, z) + x) mod P: which XCD owns residue class x mod P changes from row to row, but a
// class always sits on ONE XCD -- all the map needs.)
__device__ __forceinline__ int c4_tile(int bx, int nx, int P) { return pe_xcd_tile(bx, nx, P); }

// ---- the 4-column GEMM: rows <= 192 x K (192 or 96) over the 4 columns in YT[4][K + 4] on the 4x4x1 MFMA
constexpr int FFN_MAXS = 16;                          // slices of the fused FFN's hidden dimension (ffn.h): FC <= 768
constexpr int C4_H = 192, C4_NT = C4_H / 64;          // output rows of one call, 64-row tiles
template <int K>
struct Col4W {
  static constexpr int NQ = K / 16;                   // k quads per wave: K / 4 quads dealt to four waves
  static constexpr int KS = K + 4;                    // LDS row stride of the B operand ([column][KS]: 16-byte rows, banks shifted by 4)
  f32x4 w[C4_NT][K / 16];
};
// this wave's weight fragments: tiles [0, nt) of a [rows <= 192][K] matrix in pack4 order (missing tiles: zeros)
template <int K>
__device__ __forceinline__ void col_gemm4_fetch(Col4W<K>& W, const float* wp4, int nt, int wv, int lane) {
  constexpr int NQ = Col4W<K>::NQ, tile_floats = (K / 4) * 256;
#pragma unroll
  for (int m = 0; m < C4_NT; ++m) {
    const bool live = PE_UNIFORM(m < nt);
    const pe_rowsrc ws = pe_make_row_u(wp4 + (long)(live ? m : 0) * tile_floats, live ? tile_floats : 0);
#pragma unroll
    for (int q = 0; q < NQ; ++q) W.w[m][q] = pe_row_load4(ws, ((NQ * wv + q) * 64 + lane) * 4);
  }
  PE_SCHED_FENCE();
}
// partial product of this wave's K range: P[wave][row][4 columns] <- W[:, K range] . Y[K range][4]; YT = [4][K + 4]
template <int K>
__device__ __forceinline__ void col_gemm4_run(const Col4W<K>& W, const float* YT, float* P, int wv, int lane) {
  constexpr int NQ = Col4W<K>::NQ, KS = Col4W<K>::KS;
  f32x4 acc[C4_NT];
#pragma unroll
  for (int m = 0; m < C4_NT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[m][r] = 0.f;
  const float* yp = YT + (lane & 3) * KS + 4 * NQ * wv;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    f32x4 yv;
#pragma unroll
    for (int j = 0; j < 4; ++j) yv[j] = yp[4 * q + j];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int m = 0; m < C4_NT; ++m) acc[m] = pe_mfma_4x4x1(W.w[m][q][j], yv[j], acc[m]);
  }
#pragma unroll
  for (int m = 0; m < C4_NT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) P[((wv * C4_NT + m) * 64 + 4 * (lane >> 2) + r) * 4 + (lane & 3)] = acc[m][r];
}
// row `c` of the product at column `col`: the four waves' partials in wave order
__device__ __forceinline__ float col_gemm4_get(const float* P, int c, int col) {
  const int o = c * 4 + col;
  return ((P[o] + P[C4_H * 4 + o]) + P[2 * C4_H * 4 + o]) + P[3 * C4_H * 4 + o];
}
// Sum over the 64 channel lanes x 3 slots that share a column (256-thread, 4-column workgroups): lanes by shuffle, the
// four waves through `red` ([2][4][4] floats, halves alternating between calls like pe_col_sum16: one barrier per call)
__device__ __forceinline__ float pe_col_sum4(float v, float* red, int& flip, int wv, int lane, int col) {
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 8);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  float* r = red + flip * 16;
  flip ^= 1;
  if (lane < 4) r[wv * 4 + col] = v;
  __syncthreads();
  return ((r[col] + r[4 + col]) + r[8 + col]) + r[12 + col];
}


// ------------------------------------------------------------------------------------------------
// colchain_kernel on 4-column workgroups (same modes, same arguments; weights looked up in pack4 order):
//   mode 0   out = LN(res + W1.in + b1)              attention conv_o + residual + norm_layers_1 (attentions.py:70-72)
//   mode 1   x1 -= W1.in + b1 ; out2 = W2.x1 + b2    coupling post + mean-only reverse update, then the next layer's pre
//   mode 2   (4-column form only) WN res/skip conv (modules.py:200-208), grid.z = 192-row parts of W1:
//            x1 += W1[res rows].in + b  (the WN hidden state, in place) ; out (+)= W1[skip rows].in + b  (the skip sum;
//            `first`: not read). rows1 = 384: part 0 = res, part 1 = skip; rows1 = 192 (last WN layer): all skip.
//   mode 3   (4-column form only) a plain 1x1 conv, grid.z = 192-row parts: out = W1.in + b1 (+ res[utterance][row], a
//            per-utterance bias vector: the speaker conditioning of dp.pre) -- the first encoder layer's q/k/v, dp.pre.
// 192 input channels, rows1 = 192 (mode 0) / 96 (mode 1), second GEMM 192 rows over the 96 updated channels.
// FRONT (mode 1 only): the LAST WN layer's res/skip conv -- 192 skip rows, no hidden-state update -- in front of the chain:
// the post conv's input is (W0.in0 + b0) + in1 instead of in1 (in1 = the skip sum of the layers before; `first`: not
// read). The same additions in the same order as the mode-2 launch it replaces, whose output is not written at all: the
// skip sum of a coupling layer has no other reader.
template <bool FRONT>
__global__ __launch_bounds__(256) void colchain4_kernel(ColP p) {
  PE_KTRACE(3);
  constexpr int NC = 4, NVT = 3, K1 = C4_H, K2 = C4_H / 2, KS1 = Col4W<K1>::KS, KS2 = Col4W<K2>::KS;
  PE_DYN_SMEM(float, sm);                       // YT[4][KS1] | P[4 waves][192][4] | red[2][4][4]
  const int b = blockIdx.y, L = p.lens[b];
  const int t0 = c4_tile(blockIdx.x, gridDim.x, p.xcd) * NC;
  if (t0 >= L) return;
  float* YT = sm;
  float* P = YT + NC * KS1;
  float* red = P + 4 * C4_H * NC;
  const int tid = threadIdx.x, col = tid & 3, rl = tid >> 2, wv = PE_UNIFORM(tid >> 6), lane = tid & 63;
  const int t = t0 + col;
  const bool ok = t < L;
  const int part = p.mode >= 2 ? (int)blockIdx.z : 0, row0 = part * C4_H;
  const int rows_here = p.rows1 - row0 < C4_H ? p.rows1 - row0 : C4_H;
  // Weight fragments are requested BEHIND the (few) input loads of the phase they belong to: a wave reaches its LDS stores
  // only after the vector-memory pipe has taken every load in front of them, and the memory counter retires in order, so
  // 36 16-byte loads per lane in front of the inputs put their whole fetch on the staging's critical path (profiles/r04_notes.md, calls 60-65)
  Col4W<K1> gw0;                                 // FRONT: the res/skip conv's fragments
  Col4W<K1> gw;                                  // first GEMM's fragments
  auto fetch_gw = [&]() { col_gemm4_fetch<K1>(gw, p.w1 + (long)part * C4_NT * (K1 / 4) * 256, (rows_here + 63) / 64, wv, lane); };
  const bool skip_part = p.mode == 2 && (p.rows1 <= C4_H || part == 1);
  float sk[NVT];                                 // FRONT: the completed skip sum of this thread's (channel, column) slots
  if constexpr (FRONT) {
    const pe_rowsrc a0 = pe_make_row(p.in0 + (long)b * p.in0_bs, K1 * p.in0_cs);
    const pe_rowsrc s0 = pe_make_row(p.in1 + (long)b * p.in1_bs, p.first ? 0 : K1 * p.in1_cs);
    const pe_rowsrc b0d = pe_make_row(p.b0 ? p.b0 : p.w0, p.b0 ? K1 : 0);
    float av[NVT], b0v[NVT];
#pragma unroll
    for (int k = 0; k < NVT; ++k) {
      const int c = rl + 64 * k;
      av[k] = pe_row_load(a0, ok ? c * p.in0_cs + t : -1);
      sk[k] = pe_row_load(s0, ok ? c * p.in1_cs + t : -1);
      b0v[k] = pe_row_load(b0d, c);
    }
    PE_SCHED_FENCE();
    col_gemm4_fetch<K1>(gw0, p.w0, C4_NT, wv, lane);
#pragma unroll
    for (int k = 0; k < NVT; ++k) YT[col * KS1 + rl + 64 * k] = av[k];
    __syncthreads();
    fetch_gw();                                  // in flight under the res/skip GEMM
    col_gemm4_run<K1>(gw0, YT, P, wv, lane);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NVT; ++k) sk[k] = ok ? (col_gemm4_get(P, rl + 64 * k, col) + b0v[k]) + sk[k] : 0.f;
  }

  // operands of the step after the first GEMM are requested before it: residual / previous x1, LN gains, bias
  float ov[NVT], gg[NVT], bb[NVT], b1v[NVT];
  {
    const pe_rowsrc ind = pe_make_row(p.in1 + (long)b * p.in1_bs, p.K1 * p.in1_cs);      // K1 < 192 (mode 3): the missing channels read as zeros
    const float* obp = p.mode == 0 ? p.res + (long)b * p.res_bs : (skip_part ? p.out + (long)b * p.out_bs : p.x1 + (long)b * p.x1_bs);
    const int ocs = p.mode == 0 ? p.res_cs : (skip_part ? p.out_cs : p.x1_cs);
    const pe_rowsrc od = p.mode == 3 ? pe_make_row(p.res ? p.res + (long)b * p.res_bs + row0 : p.w1, p.res ? rows_here : 0)
                                     : pe_make_row(obp, (skip_part && p.first) ? 0 : rows_here * ocs);
    const pe_rowsrc gd = pe_make_row(p.mode == 0 ? p.gamma : p.w1, p.mode == 0 ? p.rows1 : 0);
    const pe_rowsrc bd = pe_make_row(p.mode == 0 ? p.beta : p.w1, p.mode == 0 ? p.rows1 : 0);
    const pe_rowsrc b1d = pe_make_row(p.b1 ? p.b1 + row0 : p.w1, p.b1 ? rows_here : 0);
    float xin[NVT];
#pragma unroll
    for (int k = 0; k < NVT; ++k) {
      const int c = rl + 64 * k;
      xin[k] = FRONT ? sk[k] : pe_row_load(ind, ok ? c * p.in1_cs + t : -1);
      ov[k] = pe_row_load(od, (ok && c < rows_here) ? (p.mode == 3 ? c : c * ocs + t) : -1);
      gg[k] = pe_row_load(gd, c < rows_here ? c : -1);
      bb[k] = pe_row_load(bd, c < rows_here ? c : -1);
      b1v[k] = pe_row_load(b1d, c < rows_here ? c : -1);
    }
    PE_SCHED_FENCE();
    if constexpr (!FRONT) fetch_gw();
#pragma unroll
    for (int k = 0; k < NVT; ++k) YT[col * KS1 + rl + 64 * k] = xin[k];
  }
  __syncthreads();
  col_gemm4_run<K1>(gw, YT, P, wv, lane);
  // second GEMM's fragments (the next layer's pre): in flight under the x1 update
  Col4W<K2> gw2;
  const bool second = p.mode == 1 && p.w2;
  if (second) col_gemm4_fetch<K2>(gw2, p.w2, (p.rows2 + 63) / 64, wv, lane);
  __syncthreads();

  if (p.mode >= 2) {
    if (!ok) return;
    float* tp = (skip_part || p.mode == 3) ? p.out + (long)b * p.out_bs + (p.mode == 3 ? (long)row0 * p.out_cs : 0) : p.x1 + (long)b * p.x1_bs;
    const int tcs = (skip_part || p.mode == 3) ? p.out_cs : p.x1_cs;
#pragma unroll
    for (int k = 0; k < NVT; ++k) {
      const int c = rl + 64 * k;
      if (c < rows_here) {
        const float val = (col_gemm4_get(P, c, col) + b1v[k]) + ov[k];
        tp[(long)c * tcs + t] = val;
        if (p.mode == 3 && p.kT && part == 1) p.kT[(long)b * p.kt_bs + ((long)(c >> 2) * p.out_cs + t) * 4 + (c & 3)] = val;      // [channel quad][column][4]
        if (p.mode == 3 && p.vQ && part == 2) p.vQ[(long)b * p.kt_bs + ((long)(t >> 2) * C4_H + c) * 4 + (t & 3)] = val;
      }
    }
    return;
  }
  if (p.mode == 0) {
    int red_flip = 0;
    auto col_sum = [&](float x) -> float { return pe_col_sum4(x, red, red_flip, wv, lane, col); };
    const int H = p.rows1;
    float v[NVT];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NVT; ++k) {
      const int c = rl + 64 * k;
      v[k] = (ok && c < H) ? (col_gemm4_get(P, c, col) + b1v[k]) + ov[k] : 0.f;
      s += v[k];
    }
    const float mean = col_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NVT; ++k)
      if (rl + 64 * k < H) { const float d = v[k] - mean; q = fmaf(d, d, q); }
    const float rstd = 1.f / sqrtf(col_sum(q) / (float)H + 1e-5f);
    if (!ok) return;
    float* ob = p.out + (long)b * p.out_bs;
#pragma unroll
    for (int k = 0; k < NVT; ++k) {
      const int c = rl + 64 * k;
      if (c < H) ob[(long)c * p.out_cs + t] = (v[k] - mean) * rstd * gg[k] + bb[k];
    }
    return;
  }

  // mode 1: x1 <- x1 - (post + bias); the updated half is the next layer's x0
  {
    float* xb = p.x1 + (long)b * p.x1_bs;
    float* YT2 = YT;                             // [4][KS2]: the first GEMM's reads of YT ended before the barrier above
    const pe_rowsrc b2d = pe_make_row(second ? p.b2 : p.w1, (second && p.b2) ? p.rows2 : 0);
    float b2v[NVT];
#pragma unroll
    for (int k = 0; k < NVT; ++k) b2v[k] = pe_row_load(b2d, rl + 64 * k);
#pragma unroll
    for (int k = 0; k < NVT; ++k) {
      const int c = rl + 64 * k;
      if (c < K2) {
        const float xn = (ok && c < p.rows1) ? ov[k] - (col_gemm4_get(P, c, col) + b1v[k]) : 0.f;
        if (ok && c < p.rows1) xb[(long)c * p.x1_cs + t] = xn;
        YT2[col * KS2 + c] = xn;
      }
    }
    if (!second) return;
    __syncthreads();                             // YT2 complete; every read of the first product done
    col_gemm4_run<K2>(gw2, YT2, P, wv, lane);
    __syncthreads();
    if (!ok) return;
    float* o2 = p.out2 + (long)b * p.o2_bs;
#pragma unroll
    for (int k = 0; k < NVT; ++k) {
      const int c = rl + 64 * k;
      if (c < p.rows2) o2[(long)c * p.o2_cs + t] = col_gemm4_get(P, c, col) + b2v[k];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// lngemm_kernel on 4-column workgroups: norm_layers_2 of an encoder layer fused with the 1x1 conv that consumes it
// (attentions.py:73-74, 60-69; models.py:207); grid.z = 192-row parts of the GEMM, every part normalises its 4 columns
// itself, part 0 writes LN(y) back for the residual readers.
__global__ __launch_bounds__(256) void lngemm4_kernel(LnGemmP p) {
  PE_KTRACE(7);
  constexpr int NC = 4, NVT = 3, H = C4_H, KS = Col4W<H>::KS;
  PE_DYN_SMEM(float, sm);                       // YT[4][KS] | P[4 waves][192][4] | red[2][4][4]
  const int b = blockIdx.y, L = p.lens[b];
  const int t0 = c4_tile(blockIdx.x, gridDim.x, p.xcd) * NC;
  if (t0 >= L) return;
  float* YT = sm;
  float* P = YT + NC * KS;
  float* red = P + 4 * H * NC;
  const int tid = threadIdx.x, col = tid & 3, rl = tid >> 2, wv = PE_UNIFORM(tid >> 6), lane = tid & 63;
  const int t = t0 + col;
  const bool ok = t < L;
  const int part = blockIdx.z, row0 = part * H;
  const int rows_here = p.rows - row0 < H ? p.rows - row0 : H;
  Col4W<H> gw;                                   // requested behind the input loads (see colchain4_kernel)
  auto fetch_gw = [&]() { col_gemm4_fetch<H>(gw, p.w16 + (long)part * C4_NT * (H / 4) * 256, (rows_here + 63) / 64, wv, lane); };
  float v[NVT], gg[NVT], bb[NVT], cb[NVT];
  {
    const pe_rowsrc ind = pe_make_row(p.in + (long)b * p.in_bs, H * p.in_cs);
    const pe_rowsrc gd = pe_make_row(p.gamma, H), bd = pe_make_row(p.beta, H);
    const pe_rowsrc cbd = pe_make_row(p.bias ? p.bias + row0 : p.gamma, p.bias ? rows_here : 0);
#pragma unroll
    for (int k = 0; k < NVT; ++k) {
      const int c = rl + 64 * k;
      v[k] = pe_row_load(ind, ok ? c * p.in_cs + t : -1);
      gg[k] = pe_row_load(gd, c);
      bb[k] = pe_row_load(bd, c);
      cb[k] = pe_row_load(cbd, c);
    }
    if (p.parts) {
      // y = x + (conv_2 bias + the FFN's partial outputs, summed in slice order): every load in flight at once
      const pe_rowsrc pbd = pe_make_row(p.pbias, H);
      float pv[NVT][FFN_MAXS];
      float pbv[NVT];
#pragma unroll
      for (int k = 0; k < NVT; ++k) {
        const int c = rl + 64 * k;
        pbv[k] = pe_row_load(pbd, c);
        // this workgroup's tile: one contiguous [slice][192][4] block
        const pe_rowsrc pd = pe_make_row(p.parts + (long)b * p.p_bs + (long)(t0 >> 2) * p.nparts * (H * 4), p.nparts * (H * 4));
#pragma unroll
        for (int sl = 0; sl < FFN_MAXS; ++sl) pv[k][sl] = pe_row_load(pd, (ok && sl < p.nparts) ? sl * (H * 4) + c * 4 + col : -1);
      }
      PE_SCHED_FENCE();
      fetch_gw();
#pragma unroll
      for (int k = 0; k < NVT; ++k) {
        float a = pv[k][0];
#pragma unroll
        for (int sl = 1; sl < FFN_MAXS; ++sl) a += pv[k][sl];
        v[k] = ok ? v[k] + (a + pbv[k]) : 0.f;
      }
    } else {
      PE_SCHED_FENCE();
      fetch_gw();
    }
  }
  int red_flip = 0;
  auto col_sum = [&](float x) -> float { return pe_col_sum4(x, red, red_flip, wv, lane, col); };
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
    const int c = rl + 64 * k;
    const float y = ok ? (v[k] - mean) * rstd * gg[k] + bb[k] : 0.f;
    YT[col * KS + c] = y;
    if (part == 0 && ok) xo[(long)c * p.x_cs + t] = y;
  }
  __syncthreads();
  col_gemm4_run<H>(gw, YT, P, wv, lane);
  __syncthreads();
  if (!ok) return;
  const bool second = p.out2 != nullptr && row0 >= p.split;        // a row part of the stacked second conv (192-row parts: never mixed)
  float* ob = second ? p.out2 + (long)b * p.o2_bs + (long)(row0 - p.split) * p.o2_cs : p.out + (long)b * p.o_bs + (long)row0 * p.o_cs;
  const int ocs = second ? p.o2_cs : p.o_cs;
  const pe_rowsrc c2d = pe_make_row((second && p.bias2) ? p.bias2 + (long)b * p.bias2_bs + (row0 - p.split) : p.gamma, (second && p.bias2) ? rows_here : 0);
#pragma unroll
  for (int k = 0; k < NVT; ++k) {
    const int c = rl + 64 * k;
    if (c < rows_here) {
      const float val = (col_gemm4_get(P, c, col) + cb[k]) + pe_row_load(c2d, c);
      ob[(long)c * ocs + t] = val;
      if (p.kT && part == 1) p.kT[(long)b * p.kt_bs + ((long)(c >> 2) * p.x_cs + t) * 4 + (c & 3)] = val;      // [channel quad][column][4]
      if (p.vQ && part == 2) p.vQ[(long)b * p.kt_bs + ((long)(t >> 2) * H + c) * 4 + (t & 3)] = val;      // [column quad][192][4]
    }
  }
}

}  // namespace pe
