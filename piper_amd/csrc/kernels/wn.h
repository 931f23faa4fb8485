// wn_kernel: one WN layer of a coupling flow (gate conv k5 -> tanh * sigmoid -> res/skip 1x1 conv) in ONE launch for
// small calls. OPT-IN (PIPER_HIP_WN=1): verified against the oracle on the emulator, not yet measured on the hardware.
// (gfx950 / CDNA4 device code; reference arithmetic cited per kernel, paths relative to
// /root/reference/src/python/piper_train/vits/.)
#pragma once
#include "../pe_rt.h"
#include "params.h"

namespace pe {

// modules.py:184-209 (WN.forward), commons.py:99-106 (fused_add_tanh_sigmoid_multiply), one layer i:
//     x_in = in_layer_i(x) (+ g_i)                      conv k5, 192 -> 384, "same" padding
//     acts = tanh(x_in[:192]) * sigmoid(x_in[192:])
//     rs   = res_skip_i(acts)                           1x1, 192 -> 384 (last layer: 192 -> 192, all skip)
//     x    = (x + rs[:192]) * mask ; output += rs[192:]
// As two launches (conv_splitk16_kernel<true,12,2>, colchain4_kernel mode 2) a 417-frame utterance costs 10.2 + 5.3 us per
// layer, 16 layers per step: the gate conv is 162 workgroups of 3.2 us of MFMA issue each. Here -- like ffn_kernel does for
// the FFN -- the GATED CHANNELS are dealt to the workgroups: workgroup (16-column tile, slice) computes the 24 tanh + 24
// sigmoid pre-activation rows of its 24 channels over the whole K = 192 x 5, gates them, and multiplies the 24 x 16
// activations straight into the res/skip conv: a partial product over its 24-channel slice of that conv's K for all 384
// rows. 27 x 8 = 216 workgroups of 4 waves for 417 frames, 0.88 MMAC each.
// The partial products are summed by their consumers, in slice order (deterministic):
//   * res rows: the NEXT layer's launch of this kernel, while it stages its x window -- x_{i+1} = x_i + b_res + sum over
//     the 8 slices; its slice-0 workgroups also write x_{i+1} back, for the layer after. Layout per 16-column tile,
//     [tile][slice][192][16]: a window touches 64-byte rows of three tiles.
//   * skip rows: colchain4_kernel mode 1 (the coupling layer's post conv), which reads b_skip_sum + the 4 x 8 partials of
//     all four layers where it used to read the skip sum; layout per ITS 4-column tile, [tile][layer * 8 + slice][192][4].
// Gate GEMM: 48 rows (three 16-row tiles; tanh rows 0..23, sigmoid rows 24..47) on v_mfma_f32_16x16x4_f32, the 192
// channels dealt to the four waves (48 each, five taps: 60 k-steps x 3 tiles), all 45 weight float4 of a lane requested at
// kernel entry, B operand = the x window [192][20 columns] in LDS; partial tiles meet in LDS in
// wave order. Res/skip GEMM: K = 24 (padded to 32), wave w owns row tiles 6w .. 6w + 5 (waves 0, 1: res rows, 2, 3: skip
// rows), no reduction. Weights: engine.cpp pack_wn_gate / pack_wn_rs.
constexpr int WN_H = 192, WN_S = 24, WN_NS = WN_H / WN_S, WN_NC = 16, WN_XS = 48, WN_TAPS = 5;

__global__ __launch_bounds__(256) void wn_kernel(WnP p) {
  PE_KTRACE(22);
  PE_DYN_SMEM(float, sm);                         // XS[192][48] | PA[4][48][16] | AS[32][16]
  float* XS = sm;
  float* PA = XS + WN_H * WN_XS;
  float* AS = PA + 4 * 48 * WN_NC;
  const int b = blockIdx.z, s = blockIdx.y, tile = blockIdx.x;
  const int t0 = tile * WN_NC;
  const int tid = threadIdx.x, lane = tid & 63, wv = PE_UNIFORM(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  // ---- gate fragments: [slice][tap][tile 3][wave 4][quad 3][lane][4]
  const pe_rowsrc gd = pe_make_row_u(p.wg + (long)s * (48 * WN_H * WN_TAPS), 48 * WN_H * WN_TAPS);
  // all five taps' fragments (45 float4 per lane) are requested here, in front of the window staging: one memory latency
  // for the whole gate GEMM, like ffn_kernel's (one wave per SIMD: the registers are there). The fence keeps the compiler
  // from sinking each load next to its use (it did: 70 VGPRs, a wait in front of every few MFMAs).
  f32x4 g[WN_TAPS][3][3];
#pragma unroll
  for (int tp = 0; tp < WN_TAPS; ++tp)
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int q = 0; q < 3; ++q) g[tp][m][q] = pe_row_load4(gd, ((((tp * 3 + m) * 4 + wv) * 3 + q) * 64 + lane) * 4);
  PE_SCHED_FENCE();
  const int L = p.lens[b];
  // ---- x window -> XS[ch][c], c = 0..19 <-> frame t0 - 2 + c; x = previous state (+ previous layer's res bias and its
  // partial products); zero outside [0, L). 192 x 20 values as 960 float4 along the frames: 4 rounds of 256 threads.
  {
    const pe_rowsrc xd = pe_make_row(p.xprev + (long)b * p.x_bs, WN_H * p.x_cs);
    const bool pp = p.prev_parts != nullptr;
    // previous partials: [tile][slice][192][16]; this window = columns 14, 15 of tile - 1, all of tile, 0..1 of tile + 1
    const long tb = (long)WN_NS * WN_H * 16;
    const pe_rowsrc pd = pe_make_row(pp ? p.prev_parts + (long)b * p.pr_bs : p.xprev, pp ? (int)((long)p.ntiles * tb) : 0);
    const pe_rowsrc bd = pe_make_row(pp ? p.prev_bias : p.xprev, pp ? WN_H : 0);
#pragma unroll
    for (int rd = 0; rd < 4; ++rd) {
      const int idx = tid + 256 * rd;               // (channel, group of 4 window columns): 192 x 5
      const int ch = idx / 5, g4 = idx - ch * 5;
      if (idx < WN_H * 5) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int t = t0 - 2 + 4 * g4 + e;
          const bool in = t >= 0 && t < L;
          float a = pe_row_load(xd, in ? ch * p.x_cs + t : -1);
          if (pp) {
            const int tl = t >> 4, cc = t & 15;
            float ps = pe_row_load(pd, in ? (int)(((long)tl * WN_NS) * (WN_H * 16)) + ch * 16 + cc : -1);
#pragma unroll
            for (int sl = 1; sl < WN_NS; ++sl)
              ps += pe_row_load(pd, in ? (int)(((long)tl * WN_NS + sl) * (WN_H * 16)) + ch * 16 + cc : -1);
            a = in ? a + (ps + pe_row_load(bd, ch)) : 0.f;
          }
          v[e] = a;
          XS[ch * WN_XS + 4 * g4 + e] = a;
        }
        // slice 0 writes the state back for the layer after the next (central 16 columns only)
        if (p.xout && s == 0) {
          float* xo = p.xout + (long)b * p.x_bs + (long)ch * p.x_cs;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int c = 4 * g4 + e, t = t0 - 2 + c;
            if (c >= 2 && c < 18 && t < L) xo[t] = v[e];
          }
        }
      }
    }
  }
  if (t0 >= L) return;
  __syncthreads();
  // ---- gate GEMM: this wave's 48 channels x 5 taps into the three 16-row tiles
  {
    f32x4 acc[3];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[m][r] = 0.f;
    const float* xp = XS + (48 * wv + lq) * WN_XS + l15;
#pragma unroll
    for (int tp = 0; tp < WN_TAPS; ++tp)
#pragma unroll
      for (int st = 0; st < 12; ++st) {
        const float bv = xp[4 * st * WN_XS + tp];
#pragma unroll
        for (int m = 0; m < 3; ++m) acc[m] = pe_mfma_16x16x4(g[tp][m][st >> 2][st & 3], bv, acc[m]);
      }
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) PA[(wv * 48 + 16 * m + 4 * lq + r) * WN_NC + l15] = acc[m][r];
  }
  // ---- res/skip fragments of this wave (row tiles 6 wv .. 6 wv + 5, K = 24 padded to 32): in flight under the gating
  const int ntile_rs = p.rs_rows / 16;            // 24 (res + skip) or 12 (last layer: skip only)
  f32x4 ra[6][2];
  {
    const pe_rowsrc rd = pe_make_row_u(p.wr + (long)s * (p.rs_rows * 32), p.rs_rows * 32);
#pragma unroll
    for (int m = 0; m < 6; ++m)
#pragma unroll
      for (int q = 0; q < 2; ++q) ra[m][q] = pe_row_load4(rd, (6 * wv + m < ntile_rs) ? (((6 * wv + m) * 2 + q) * 64 + lane) * 4 : -1);
  }
  // gate biases (+ the utterance's conditioning) of this thread's two channels
  const int n = tid & 15, r0 = tid >> 4;          // thread -> column n, channels r0 and r0 + 16 of the slice (r0 + 16 < 24 for r0 < 8)
  float bt[2], bs[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ch = s * WN_S + r0 + 16 * i;
    const bool ok = r0 + 16 * i < WN_S;
    bt[i] = ok ? p.bg[ch] : 0.f;
    bs[i] = ok ? p.bg[WN_H + ch] : 0.f;
    if (ok && p.bias2) {
      bt[i] += p.bias2[(long)b * p.bias2_bs + ch];
      bs[i] += p.bias2[(long)b * p.bias2_bs + WN_H + ch];
    }
  }
  __syncthreads();
  // ---- acts = tanh(a) * sigmoid(b), zero beyond the utterance; AS[k][n], rows 24..31 zero
  {
    const bool in = t0 + n < L;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = r0 + 16 * i;
      float v = 0.f;
      if (r < WN_S) {
        const int oa = r * WN_NC + n, ob = (WN_S + r) * WN_NC + n;
        const float ta = ((PA[oa] + PA[48 * WN_NC + oa]) + PA[2 * 48 * WN_NC + oa]) + PA[3 * 48 * WN_NC + oa] + bt[i];
        const float sa = ((PA[ob] + PA[48 * WN_NC + ob]) + PA[2 * 48 * WN_NC + ob]) + PA[3 * 48 * WN_NC + ob] + bs[i];
        v = in ? tanhf(ta) * (1.f / (1.f + expf(-sa))) : 0.f;
      }
      AS[r * WN_NC + n] = v;                        // r = r0, r0 + 16: covers rows 0..31
    }
  }
  __syncthreads();
  // ---- res/skip partial products: rows 96 wv .. 96 wv + 95 over this slice's 24 channels
  {
    f32x4 acc[6];
#pragma unroll
    for (int m = 0; m < 6; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[m][r] = 0.f;
    const float* ap = AS + lq * WN_NC + l15;
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      const float bv = ap[4 * st * WN_NC];
#pragma unroll
      for (int m = 0; m < 6; ++m) acc[m] = pe_mfma_16x16x4(ra[m][st >> 2][st & 3], bv, acc[m]);
    }
    const int t = t0 + l15;
    const bool two = p.rs_rows > WN_H;            // res + skip rows; else skip only
#pragma unroll
    for (int m = 0; m < 6; ++m) {
      if (6 * wv + m >= ntile_rs) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 96 * wv + 16 * m + 4 * lq + r;
        if (two && row < WN_H) {
          // res partial: [tile][slice][192][16] (whole tiles: columns beyond the utterance carry zeros)
          p.pr_out[(long)b * p.pr_bs + (((long)tile * WN_NS + s) * WN_H + row) * 16 + l15] = acc[m][r];
        } else if (t < L) {
          const int srow = two ? row - WN_H : row;
          // skip partial: [4-column tile][layer * 8 + slice][192][4]
          p.ps_out[(long)b * p.ps_bs + ((((long)(t >> 2)) * p.ps_n + p.layer * WN_NS + s) * WN_H + srow) * 4 + (t & 3)] = acc[m][r];
        }
      }
    }
  }
}

}  // namespace pe
