// ffn_kernel: an encoder layer's FFN (conv k3 -> ReLU -> conv k3) in ONE launch for small calls.
// (gfx950 / CDNA4 device code; reference arithmetic cited per kernel, paths relative to
// /root/reference/src/python/piper_train/vits/.)
#pragma once
#include "../pe_rt.h"
#include "params.h"

namespace pe {

// attentions.py:386-407 (FFN.forward, kernel_size 3, "same" padding, no activation after conv_2; dropout is identity at
// inference):   y = conv_2(relu(conv_1(x * mask)) * mask) * mask,   x: [192][T] -> hidden [FC][T] -> [192][T].
// As two launches (conv_splitk_kernel, conv_splitk16_kernel) a 128-id utterance costs 8.5 + 12.2 us per layer: conv_2 is a
// K = 3 * FC = 2304 GEMM on 48 workgroups, matrix-pipe bound inside each (3.8 us of MFMA issue per CU) while 200 CUs
// idle, and the FC x T hidden tensor makes a round trip through memory between the two. Here the HIDDEN dimension is dealt
// to the workgroups instead: workgroup (column tile j, slice s) computes its 48 hidden rows of conv_1 on 16 columns and
// immediately multiplies them into conv_2 -- a partial product over its 48 x 3 slice of conv_2's K -- for 12 of the 14
// output columns those 16 hidden columns cover (conv_2 needs h at t-1, t, t+1; 12 = three of the consumer's 4-column tiles). The FC/48 partial outputs of a column are summed by
// the consumer, which is lngemm4_kernel (norm_layers_2 + the next q/k/v conv; col4.h): it reads residual + bias + the
// partials in slice order (deterministic) where it used to read one tensor. 11 x 16 = 176 workgroups of 4 waves for T = 128.
// Partial outputs are laid out for that consumer: [utterance][4-column tile][slice][192][4], so a consumer workgroup reads
// ONE contiguous 48 KB block and every 3 KB (tile, slice) block is written whole by one workgroup here (with [slice][192][T]
// rows a consumer workgroup touched 3072 cache lines for 16 useful bytes each: 38 us instead of 5).
//   * conv_1 slice: [48 rows] x [K = 192 ch x 3 taps] x [16 cols] on v_mfma_f32_16x16x4_f32; the 192 channels are dealt to
//     the four waves (48 each, all three taps, all three 16-row tiles: 108 MFMAs), partial tiles meet in LDS in wave order;
//     B operand = the x window [192][18 columns] in LDS, a tap is a shifted read.
//   * h = relu(sum + b1), zero outside [0, L) (conv_2's zero padding and the mask), [48][18] in LDS (columns 16, 17 zero).
//   * conv_2 partial: [192 rows] x [K = 48 hidden x 3 taps] x [16 cols]: wave w owns row tiles 3w .. 3w + 2 over the
//     whole K (108 MFMAs), no reduction; columns 0..11 are stored.
// Both weight slices (27 float4 per lane each) are requested at kernel entry / under the first GEMM. Weights: engine_pack.cpp
// pack_ffn -- conv_1 [slice][tile 3][wave 4][tap 3][step quad 3][lane][4], lane -> (row = lane & 15, k = lane >> 4),
// element j of quad Q = channel 48 wave + 4 (4Q + j) + k; conv_2 [slice][row tile 12][tap 3][step quad 3][lane][4] with
// hidden channel 48 slice + 4 (4Q + j) + k.
constexpr int FFN_SL = 48, FFN_H = 192, FFN_NC = 16, FFN_NO = 12, FFN_XS = 48;      // slice rows, channels, MFMA columns, output columns per tile, LDS row stride (== 16 mod 32)

__global__ __launch_bounds__(256) void ffn_kernel(FfnP p) {
  PE_KTRACE(9);
  PE_DYN_SMEM(float, sm);                         // XS[192][48] | PA[4][48][16] | HS[48][48]
  float* XS = sm;
  float* PA = XS + FFN_H * FFN_XS;
  float* HS = PA + 4 * FFN_SL * FFN_NC;
  // (column tile, slice), slice-major over the XCDs: an XCD's workgroups share one or two slices' weights (pe_rt.h)
  int bx = blockIdx.x, by = blockIdx.y;
  pe_xcd_xy(p.xcd, bx, by);
  const int b = blockIdx.z, s = PE_UNIFORM(by);
  const int o0 = PE_UNIFORM(bx) * FFN_NO;             // first output column of the tile; hidden columns o0 - 1 .. o0 + 14, x columns o0 - 2 .. o0 + 15
                                                  // (outputs o0 .. o0 + 11 are kept)
  const int tid = threadIdx.x, lane = tid & 63, wv = PE_UNIFORM(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  // ---- conv_1 fragments of this wave: [tile][tap][quad], requested BEHIND the x window (the memory counter retires in
  // order: in front of it the window's LDS stores waited for the whole fragment fetch -- col4.h, profiles/r04_notes.md)
  f32x4 a1[3][3][3];
  // the utterance length lives in device memory: the window is requested against the row stride and the columns beyond
  // the length are zeroed when it is stored, so the length's latency overlaps the window's
  const int L = p.lens[b];
  // ---- x window -> XS[ch][c], c = 0..17 <-> column o0 - 2 + c; zero outside [0, L)
  {
    const pe_rowsrc xd = pe_make_row(p.x + (long)b * p.x_bs, FFN_H * p.x_cs);
    float xv[14];
#pragma unroll
    for (int i = 0; i < 14; ++i) {
      const int idx = tid + 256 * i, ch = idx / 18, c = idx - ch * 18, t = o0 - 2 + c;
      xv[i] = pe_row_load(xd, (idx < FFN_H * 18 && t >= 0 && t < p.x_cs) ? ch * p.x_cs + t : -1);
    }
    PE_SCHED_FENCE();
    {
      const pe_rowsrc wd = pe_make_row_u(p.w1p + (long)s * (FFN_SL * FFN_H * 3), FFN_SL * FFN_H * 3);
#pragma unroll
      for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int tp = 0; tp < 3; ++tp)
#pragma unroll
          for (int q = 0; q < 3; ++q) a1[m][tp][q] = pe_row_load4(wd, ((((m * 4 + wv) * 3 + tp) * 3 + q) * 64 + lane) * 4);
    }
    PE_SCHED_FENCE();
    if (o0 >= L) return;
#pragma unroll
    for (int i = 0; i < 14; ++i) {
      const int idx = tid + 256 * i, ch = idx / 18, c = idx - ch * 18;
      if (idx < FFN_H * 18) XS[ch * FFN_XS + c] = (o0 - 2 + c < L) ? xv[i] : 0.f;
    }
  }
  __syncthreads();
  // ---- conv_1: this wave's 48 channels x 3 taps into the three 16-row tiles
  {
    f32x4 acc[3];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[m][r] = 0.f;
    const float* xp = XS + (48 * wv + lq) * FFN_XS + l15;
#pragma unroll
    for (int tp = 0; tp < 3; ++tp)
#pragma unroll
      for (int st = 0; st < 12; ++st) {
        const float bv = xp[4 * st * FFN_XS + tp];
#pragma unroll
        for (int m = 0; m < 3; ++m) acc[m] = pe_mfma_16x16x4(a1[m][tp][st >> 2][st & 3], bv, acc[m]);
      }
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) PA[(wv * FFN_SL + 16 * m + 4 * lq + r) * FFN_NC + l15] = acc[m][r];
  }
  // ---- conv_2 fragments of this wave (row tiles 3 wv .. 3 wv + 2): in flight under the reduction
  f32x4 a2[3][3][3];
  {
    const pe_rowsrc wd = pe_make_row_u(p.w2p + (long)s * (FFN_H * FFN_SL * 3), FFN_H * FFN_SL * 3);
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int tp = 0; tp < 3; ++tp)
#pragma unroll
        for (int q = 0; q < 3; ++q) a2[m][tp][q] = pe_row_load4(wd, ((((3 * wv + m) * 3 + tp) * 3 + q) * 64 + lane) * 4);
  }
  const float b1v0 = p.b1[s * FFN_SL + (tid >> 4)], b1v1 = p.b1[s * FFN_SL + 16 + (tid >> 4)], b1v2 = p.b1[s * FFN_SL + 32 + (tid >> 4)];
  __syncthreads();
  // ---- h = relu(conv_1 + b1) on hidden columns inside the utterance, else 0; HS[row][c], c = 0..17 <-> column o0 - 1 + c
  {
    const int n = tid & 15, r0 = tid >> 4;                       // thread -> column n, rows r0, r0 + 16, r0 + 32
    const int th = o0 - 1 + n;
    const bool in = th >= 0 && th < L;
    const float bb[3] = {b1v0, b1v1, b1v2};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int row = r0 + 16 * i, o = row * FFN_NC + n;
      float v = ((PA[o] + PA[FFN_SL * FFN_NC + o]) + PA[2 * FFN_SL * FFN_NC + o]) + PA[3 * FFN_SL * FFN_NC + o];
      v += bb[i];
      HS[row * FFN_XS + n] = (in && v > 0.f) ? v : 0.f;
    }
    if (tid < 2 * FFN_SL) HS[(tid >> 1) * FFN_XS + 16 + (tid & 1)] = 0.f;
  }
  __syncthreads();
  // ---- conv_2 partial product over this slice: rows 48 wv .. 48 wv + 47, output column n <-> hidden columns n, n+1, n+2
  {
    f32x4 acc[3];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[m][r] = 0.f;
    const float* hp = HS + lq * FFN_XS + l15;
#pragma unroll
    for (int tp = 0; tp < 3; ++tp)
#pragma unroll
      for (int st = 0; st < 12; ++st) {
        const float bv = hp[4 * st * FFN_XS + tp];
#pragma unroll
        for (int m = 0; m < 3; ++m) acc[m] = pe_mfma_16x16x4(a2[m][tp][st >> 2][st & 3], bv, acc[m]);
      }
    const int t = o0 + l15;
    if (l15 < FFN_NO && t < L) {
      // [utterance][tile t / 4][slice][row][t % 4]
      float* pp = p.parts + (long)b * p.p_bs + ((long)(t >> 2) * p.nslices + s) * (FFN_H * 4) + (t & 3);
#pragma unroll
      for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) pp[(48 * wv + 16 * m + 4 * lq + r) * 4] = acc[m][r];
    }
  }
}

}  // namespace pe
