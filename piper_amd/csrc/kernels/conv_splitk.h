// Split-K forms of the conv GEMM for launches with few column tiles (one or a few utterances).
// (gfx950 / CDNA4 device code; reference arithmetic cited per kernel, paths relative to
// /root/reference/src/python/piper_train/vits/.)
#pragma once
#include "conv_common.h"

namespace pe {

// Same GEMM for launches that would otherwise fill only a few CUs (one utterance through the text
// encoder / duration predictor / flow / first generator stage: 128..3500 columns). These launches are pure
// latency chains, so the kernel is organised around having every memory request in flight as early as
// possible:
//   * one 32*MT x 32 output tile per workgroup; its NW (4 or 8) waves split the K-chunks between them
//     (wave w takes chunks w, w+NW, ...), each with a private x slab in LDS;
//   * the epilogue operands (bias, residual, previous value) of the slots a wave will finish do not depend
//     on the GEMM and are requested first;
//   * weight fragments come through a buffer descriptor as float4 loads into a ring of D steps, issued
//     unconditionally (past the end the descriptor returns zeros) so that the wait counts stay exact and a
//     wave with <= D steps has its whole K range in flight at once;
//   * partial tiles are summed through LDS in a fixed order (deterministic).
// XW: columns of a wave's x slab: 64 (halo (taps-1)*dil <= 32), or 128 for the long-dilation resblock convs that are
// launched in a group with their siblings (halo <= 96).
// MS (multi-segment): the K dimension is the concatenation of up to three convs that share the launch shape and are
// SUMMED -- the last convs of an MRF stage's sibling resblocks, out = (sum_j (t_j + c_j(lrelu(t_j)))) / n -- each with its
// own input tensor, kernel size, dilation and weights (ConvP::seg*): chunk c of the virtual 3 * Cin channels belongs to
// segment c / nchunks. With 4 chunk lanes and 4 chunks per segment every wave gets one chunk of each conv.
template <int MT, bool GATE, int NW, int D, int XW, bool MS = false>
__device__ __forceinline__ void conv_splitk_body(const ConvP& p, const int b, float* sm) {

  constexpr int BN = 32, KH = KC / 2, XB = XW / 64;
  constexpr int NS = GATE ? (16 + NW - 1) / NW : (MT * 16 + NW - 1) / NW;    // epilogue slots per wave
  // sm: NW x [KC][XW] slabs, then NW x [MT*16][64] partial tiles
  PE_STAMP(1, 0);
  // The utterance length lives in device memory (one graph per shape bucket). Nothing below touches it until the
  // x slab and the first weight fragments are requested, so its latency overlaps theirs instead of preceding them.
  const int L = p.lens[b] * p.len_mul;
  const int n0 = blockIdx.x * BN;
  const int tid = threadIdx.x, lane = tid & 63, wv = PE_UNIFORM(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int mtile0 = blockIdx.y * MT;
  const int col = n0 + l31;
  const int ntaps = p.ntaps, nchunks = MS ? p.nchunks * p.nseg : p.nchunks;     // MS: chunks of the concatenated K
  const int wstride_mt = p.nchunks * ntaps * KH * 64;
  const float* xb = p.x + (long)b * p.x_bs;
  const float slope = p.in_slope;
  const bool act_in = slope != 1.f;               // inputs without a pre-activation skip its two VALU ops per element
  const pe_rowsrc wsrc = pe_make_row(p.wp + (long)mtile0 * wstride_mt, MT * wstride_mt);
  // per-segment views (MS): taps, dilation, left padding, weights, input of the segment that owns global chunk c
  // (no integer division on the step path: nseg <= 3, two compares)
  auto seg_of = [&](int c) { return MS ? (c >= p.nchunks) + (c >= 2 * p.nchunks) + (c >= 3 * p.nchunks) : 0; };
  auto taps_of = [&](int sg) { return MS ? p.seg_ntaps[sg < 3 ? sg : 0] : ntaps; };
  float* xw = sm + wv * KC * XW;                  // this wave's slab: 32 channels x XW columns
  // K is dealt to the waves as (chunk lane, tap group): with p.tgroups == 1 wave w takes chunks w, w+NW, ...
  // and every tap; with 2 groups the waves form two halves that share the chunks and split the taps (a 5-tap
  // conv with 6 chunks then keeps 12 waves busy with 3 / 2 steps each instead of 6 waves with 5)
  const int CL = NW / p.tgroups;                  // chunk lanes
  const int wi = wv % CL, wg = wv / CL;
  const int tpg = (ntaps + p.tgroups - 1) / p.tgroups;
  const int tap_lo = wg * tpg, tap_hi = (tap_lo + tpg < ntaps) ? tap_lo + tpg : ntaps;
  const int mytaps = tap_hi > tap_lo ? tap_hi - tap_lo : 0;
  const int myc = (wi < nchunks && mytaps > 0) ? (nchunks - wi + CL - 1) / CL : 0;   // chunks wi, wi+CL, ...
  int nsteps = myc * mytaps;
  if (MS) {                                       // every tap of every chunk (tgroups == 1), taps differ per segment
    nsteps = 0;
    for (int k = 0; k < myc; ++k) nsteps += taps_of(seg_of(wi + CL * k));
  }

  // x slab: one descriptor over the utterance's [Cin][stride] tensor, a per-lane column offset (poisoned outside
  // the row) and a wave-uniform row offset -- independent of L; columns >= L are zeroed when the slab is stored
  // (the launcher sends only convs of whole 32-channel chunks here -- policy.h: the row offset of a load rides in the SGPR
  // offset, which the hardware does not range-check; a chunk that does not exist reads through a zero-length descriptor)
  float xr[XB][KC];
  pe_rowsrc xd = pe_make_row(xb, p.Cin * p.x_cs);
  const pe_rowsrc xz = pe_make_row(xb, 0);
  int xcol = n0 - p.padl + lane;
  int xoff[XB];
#pragma unroll
  for (int h = 0; h < XB; ++h) xoff[h] = (xcol + 64 * h >= 0 && xcol + 64 * h < p.x_cs) ? xcol + 64 * h : 0x3fffffff;
  int ld_dil = p.dil, cur_dil = p.dil;            // dilation of the chunk in xr / of the chunk in the LDS slab
  auto load_x = [&](int c) {
    int cc = c;
    if (MS) {
      const int sgr = seg_of(c), sg = sgr < p.nseg ? sgr : 0;
      cc = sgr < p.nseg ? c - sg * p.nchunks : 0;              // (a wave without work: zero-length descriptor, zeros)
      xd = pe_make_row(p.seg_x[sg] + (long)b * p.x_bs, sgr < p.nseg ? p.Cin * p.x_cs : 0);
      xcol = n0 - p.seg_padl[sg] + lane;
      ld_dil = p.seg_dil[sg];
#pragma unroll
      for (int h = 0; h < XB; ++h) xoff[h] = (xcol + 64 * h >= 0 && xcol + 64 * h < p.x_cs) ? xcol + 64 * h : 0x3fffffff;
    }
    const bool exists = MS || c < nchunks;
    const pe_rowsrc& src = exists ? xd : xz;
    cc = exists ? cc : 0;
#pragma unroll
    for (int h = 0; h < XB; ++h)
#pragma unroll
      for (int r = 0; r < KC; ++r) xr[h][r] = pe_row_load_so(src, xoff[h], (cc * KC + r) * p.x_cs);
  };
  auto store_x = [&]() {
    cur_dil = ld_dil;
#pragma unroll
    for (int h = 0; h < XB; ++h) {
      const bool live = xcol + 64 * h < L;
      if (act_in) {
#pragma unroll
        for (int r = 0; r < KC; ++r) xw[r * XW + 64 * h + lane] = pe_lrelu(live ? xr[h][r] : 0.f, slope);
      } else {
#pragma unroll
        for (int r = 0; r < KC; ++r) xw[r * XW + 64 * h + lane] = live ? xr[h][r] : 0.f;
      }
    }
  };
  // weight ring: slot d holds the fragments of step (s with s % D == d); the load cursor runs D steps ahead
  float a[D][MT][KH];
  int lk = 0, ltap = tap_lo;
  auto load_ring = [&](float (&dst)[MT][KH]) {
    if (MS) {
      const int c = wi + CL * lk, sgr = seg_of(c), sg = sgr < p.nseg ? sgr : 0;
      const int nt = p.seg_ntaps[sg], ws = p.nchunks * nt * KH * 64;
      // past the last chunk the descriptor has length 0: zeros, like the single-conv form's reads beyond its matrix
      const pe_rowsrc wsg = pe_make_row(p.seg_wp[sg] + (long)mtile0 * ws, sgr < p.nseg ? MT * ws : 0);
      const int off = PE_UNIFORM(((c - sg * p.nchunks) * nt + ltap) * (KH * 64));
#pragma unroll
      for (int i = 0; i < MT; ++i) load_frags<KH>(wsg, off + i * ws, lane, dst[i]);
      if (++ltap >= nt) { ltap = 0; ++lk; }
      return;
    }
    const int off = PE_UNIFORM(((wi + CL * lk) * ntaps + ltap) * (KH * 64));
#pragma unroll
    for (int i = 0; i < MT; ++i) load_frags<KH>(wsrc, off + i * wstride_mt, lane, dst[i]);
    if (++ltap >= tap_hi) { ltap = tap_lo; ++lk; }
  };
  f32x16 acc[MT];
  auto mma = [&](int tap, const float (&af)[MT][KH]) {
    const float* xp = xw + lhi * XW + tap * (MS ? cur_dil : p.dil) + l31;
    float bv[KH];
#pragma unroll
    for (int kk = 0; kk < KH; ++kk) bv[kk] = xp[2 * kk * XW];
#pragma unroll
    for (int kk = 0; kk < KH; ++kk)
#pragma unroll
      for (int i = 0; i < MT; ++i) acc[i] = pe_mfma_32x32x2(af[i][kk], bv[kk], acc[i]);
  };

  load_x(myc > 0 ? wi : nchunks);   // unconditional (zeros for a wave without work): keeps the wait counts exact
#pragma unroll
  for (int d = 0; d < D; ++d) load_ring(a[d]);
  PE_SCHED_FENCE();
  const int ncols = (p.epi == EPI_CONVT) ? L + 1 : L;
  if (n0 >= ncols) return;
  PE_STAMP(1, 1);
  const EpiFlags ef = epi_flags(p);
  // ---- epilogue operands of this wave's slots: four independent loads per slot, combined only in the
  // epilogue (adding them here would wait for each load in turn)
  float e_b1[NS], e_b2[NS], e_o1[NS], e_o2[NS];
  float* e_dst[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int s = wv + NW * i;
    e_b1[i] = 0.f; e_b2[i] = 0.f; e_o1[i] = 0.f; e_o2[i] = 0.f; e_dst[i] = nullptr;
    if constexpr (GATE) {
      // commons.py:99-106: (b1, b2) = tanh-half bias + speaker bias, (o1, o2) = the sigmoid half's
      const int ch = (mtile0 >> 1) * 32 + (s & 3) + 8 * (s >> 2) + 4 * lhi;
      if (s < 16 && ch < p.split && col < ncols) {
        e_b1[i] = p.bias[ch];
        e_o1[i] = p.bias[p.split + ch];
        if (p.bias2) {
          const float* b2 = p.bias2 + (long)b * p.bias2_bs;
          e_b2[i] = b2[ch];
          e_o2[i] = b2[p.split + ch];
        }
        e_dst[i] = p.out + (long)b * p.o_bs + (long)ch * p.o_cs + col;
      }
    } else {
      const int r = s & 15;
      const int row = (mtile0 + (s >> 4)) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      if (s < MT * 16 && row < p.rows && col < ncols) {
        if (p.epi == EPI_CONVT) {
          const int co = row / p.up, ph = row - co * p.up;
          const int t = col * p.up + ph - p.padT;
          if (t >= 0 && t < L * p.up) {
            if (p.bias) e_b1[i] = p.bias[co];
            e_dst[i] = p.out + (long)b * p.o_bs + (long)co * p.o_cs + t;
          }
        } else {
          if (p.bias) e_b1[i] = p.bias[row];
          if (p.bias2) e_b2[i] = p.bias2[(long)b * p.bias2_bs + row];
          const bool to_skip = p.epi == EPI_WNRS && row >= p.split;
          const bool rd_old = to_skip ? (p.mode != 1) : (ef.use_old || p.epi == EPI_WNRS);
          float* d = to_skip ? p.out2 + (long)b * p.o2_bs + (long)(row - p.split) * p.o2_cs + col
                             : p.out + (long)b * p.o_bs + (long)row * p.o_cs + col;
          if (rd_old) e_o1[i] = *d;
          if (ef.use_res) e_o2[i] = p.res[(long)b * p.r_bs + (long)row * p.r_cs + col];
          if (MS && p.res2) e_b2[i] = p.res2[(long)b * p.r_bs + (long)row * p.r_cs + col];     // the other segments' residuals
          if (MS && p.res3) e_o1[i] = p.res3[(long)b * p.r_bs + (long)row * p.r_cs + col];
          e_dst[i] = d;
        }
      }
    }
  }

#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  {
    int k = 0, tap = tap_lo;
    for (int s0 = 0; s0 < nsteps; s0 += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        if (s0 + d < nsteps) {
          if (tap == tap_lo) {          // new chunk: its slab is in xr
            PE_WAVE_SYNC();             // all lanes done reading the previous slab
            store_x();
            PE_WAVE_SYNC();
            if (k + 1 < myc) load_x(wi + CL * (k + 1));
          }
          mma(tap, a[d]);
          if (++tap >= (MS ? taps_of(seg_of(wi + CL * k)) : tap_hi)) { tap = tap_lo; ++k; }
        }
        PE_SCHED_FENCE();
        load_ring(a[d]);
        PE_SCHED_FENCE();
      }
    }
  }
  // ---- cross-wave reduction through LDS (fixed order w = 0..NW-1)
  PE_STAMP(1, 2);
  __syncthreads();
  PE_STAMP(1, 3);
  float* red = sm;                                // [NW waves][MT*16 slots][64 lanes]
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wv * MT * 16 + i * 16 + r) * 64 + lane] = acc[i][r];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int s = wv + NW * i;
    if constexpr (GATE) {
      float ta = 0.f, sa = 0.f;
      if (s < 16) {
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          ta += red[(w * MT * 16 + s) * 64 + lane];
          sa += red[(w * MT * 16 + (MT - 1) * 16 + s) * 64 + lane];
        }
      }
      if (e_dst[i]) {
        ta += e_b1[i] + e_b2[i];
        sa += e_o1[i] + e_o2[i];
        *e_dst[i] = tanhf(ta) * (1.f / (1.f + expf(-sa)));
      }
    } else {
      float v = 0.f;
      if (s < MT * 16) {
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[(w * MT * 16 + s) * 64 + lane];
      }
      if (e_dst[i]) {
        v = ((v + (e_b1[i] + e_b2[i])) * ef.sign + (e_o1[i] + e_o2[i])) * ef.alpha;
        if (ef.relu) v = v > 0.f ? v : 0.f;
        *e_dst[i] = v;
      }
    }
  }
  PE_STAMP(1, 4);
}

template <int MT, bool GATE, int NW, int D>
__global__ __launch_bounds__(64 * NW) void conv_splitk_kernel(ConvP p) {
  PE_KTRACE(1);
  PE_DYN_SMEM(float, sm);
  conv_splitk_body<MT, GATE, NW, D, 64>(p, blockIdx.z, sm);
}

// Up to three INDEPENDENT convs of the same launch shape in one launch (grid.z = group x utterance): the sibling
// resblocks of an MRF stage read the same input and are each a latency chain of ~10 us on a fraction of the CUs when
// launched one after the other; together they fill the chip once (models.py:356-363 runs them in a Python loop).
// Launch bounds ask for 4 workgroups per CU with the 64-column slab (<= 128 registers, 32 KB of LDS each): a group of
// 3 x ~420 workgroups then runs as ~1.2 rounds over the chip instead of 1.6-2.5.
template <int NW, int D, int XW>
__global__ __launch_bounds__(64 * NW, XW == 64 ? 4 : 2) void conv_splitk_group_kernel(ConvG g) {
  PE_KTRACE(6);
  PE_DYN_SMEM(float, sm);
  const int gi = PE_UNIFORM((int)blockIdx.z / g.B);
  const ConvP& p = g.c[gi];
  if ((int)blockIdx.y * 32 >= p.rows) return;              // a sibling with fewer row tiles than the grid
  conv_splitk_body<1, false, NW, D, XW>(p, (int)blockIdx.z - gi * g.B, sm);
}
// The siblings' LAST convs, whose outputs the MRF sums: one GEMM over the concatenated K (MS form of the body), one
// output tensor -- no per-sibling outputs, no summing pass.
// (D = 16 -- a wave's whole K range in flight at kernel entry, 256 registers, one workgroup per CU -- measured 59 us against
// the 2-deep ring's 30 on the medium voice's 128-channel stage and was removed: profiles/r04_notes.md.)
template <int NW, int D>
__global__ __launch_bounds__(64 * NW, 2) void conv_splitk_sum_kernel(ConvP p) {
  PE_KTRACE(8);
  PE_DYN_SMEM(float, sm);
  conv_splitk_body<1, false, NW, D, 128, true>(p, blockIdx.z, sm);
}

// The split-K kernel on 16 output columns with the 16x16x4 f32 MFMA, for launches that are MFMA-pipe bound inside a
// workgroup although most CUs idle (one utterance through the WN gate conv: 84 workgroups of 960 MFMAs): half the
// columns per workgroup = half the matrix time per CU and twice the workgroups. One workgroup = MT16 sixteen-row
// sub-tiles (4 for the gate: tanh a, tanh b, sigmoid a, sigmoid b of one 32-channel group; 2 otherwise) x 16 columns;
// K is dealt to the waves exactly as in conv_splitk_kernel. Weights: engine_pack.cpp pack16 --
// [16-row sub-tile][chunk][tap][q = 0..1][lane][4] with lane -> (row = lane & 15, k = lane >> 4), float4 element j of
// group q = k-step s = 4q + j, input channel chunk*32 + 4s + k; ascending k inside and across instructions, i.e. the
// same fmaf chain as the 32x32x2 form.
// GT (gate only): 16-row sub-tiles per workgroup -- 4 = a whole 32-channel group (tanh a, tanh b, sigmoid a, sigmoid b), 2 =
// half of one (tanh h, sigmoid h; blockIdx.y = 2 * group + h): twice the workgroups with half the matrix time each, for
// launches whose workgroups fit the chip either way (one utterance through the WN gate conv: 162 -> 324).
template <bool GATE, int NW, int D, int GT = 4>
__global__ __launch_bounds__(64 * NW) void conv_splitk16_kernel(ConvP p) {
  PE_KTRACE(4);
  constexpr int BN = 16, XW = 64, KS8 = KC / 4, MT16 = GATE ? GT : 2;
  constexpr int TSTR = (GATE && GT == 2) ? 2 : 1;             // sub-tile stride of a workgroup's tiles in the packed order
  constexpr int NSLOT = GATE ? 2 * GT : MT16 * 4;             // result slots per lane position (gate: tanh/sigmoid pairs)
  constexpr int NS = (NSLOT + NW - 1) / NW;                   // epilogue slots per wave
  PE_DYN_SMEM(float, sm);                         // NW x [KC][XW] slabs, then NW x [MT16*4][64] partial tiles
  const int b = blockIdx.z;
  const int L = p.lens[b] * p.len_mul;            // first used after the loads below are in flight
  const int n0 = blockIdx.x * BN;
  const int tid = threadIdx.x, lane = tid & 63, wv = PE_UNIFORM(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  // first 16-row sub-tile of this workgroup (half-group gate form: group (y >> 1), tiles (tanh h, sigmoid h) two apart)
  const int st0 = (GATE && GT == 2) ? ((int)blockIdx.y >> 1) * 4 + ((int)blockIdx.y & 1) : (int)blockIdx.y * MT16;
  const int col = n0 + l15;
  const int ntaps = p.ntaps, nchunks = p.nchunks;
  const int sub_stride = nchunks * ntaps * KS8 * 64;          // floats per 16-row sub-tile
  const float* xb = p.x + (long)b * p.x_bs;
  const float slope = p.in_slope;
  const bool act_in = slope != 1.f;
  const pe_rowsrc wsrc = pe_make_row(p.wp16 + (long)st0 * sub_stride, ((MT16 - 1) * TSTR + 1) * sub_stride);
  float* xw = sm + wv * KC * XW;
  const int CL = NW / p.tgroups;
  const int wi = wv % CL, wg = wv / CL;
  const int tpg = (ntaps + p.tgroups - 1) / p.tgroups;
  const int tap_lo = wg * tpg, tap_hi = (tap_lo + tpg < ntaps) ? tap_lo + tpg : ntaps;
  const int mytaps = tap_hi > tap_lo ? tap_hi - tap_lo : 0;
  const int myc = (wi < nchunks && mytaps > 0) ? (nchunks - wi + CL - 1) / CL : 0;
  const int nsteps = myc * mytaps;

  float xr[KC];
  const pe_rowsrc xd = pe_make_row(xb, p.Cin * p.x_cs), xz = pe_make_row(xb, 0);
  const int xcol = n0 - p.padl + lane;
  const int xoff = (xcol >= 0 && xcol < p.x_cs) ? xcol : 0x3fffffff;
  auto load_x = [&](int c) {       // (whole 32-channel chunks only, like conv_splitk_body; a chunk that does not exist: zeros)
    const pe_rowsrc& src = c < nchunks ? xd : xz;
    const int cc = c < nchunks ? c : 0;
#pragma unroll
    for (int r = 0; r < KC; ++r) xr[r] = pe_row_load_so(src, xoff, (cc * KC + r) * p.x_cs);
  };
  auto store_x = [&]() {
    const bool live = xcol < L;
    if (act_in) {
#pragma unroll
      for (int r = 0; r < KC; ++r) xw[r * XW + lane] = pe_lrelu(live ? xr[r] : 0.f, slope);
    } else {
#pragma unroll
      for (int r = 0; r < KC; ++r) xw[r * XW + lane] = live ? xr[r] : 0.f;
    }
  };
  float a[D][MT16][KS8];
  int lk = 0, ltap = tap_lo;
  auto load_ring = [&](float (&dst)[MT16][KS8]) {
    const int off = PE_UNIFORM(((wi + CL * lk) * ntaps + ltap) * (KS8 * 64));
#pragma unroll
    for (int i = 0; i < MT16; ++i)
#pragma unroll
      for (int q = 0; q < KS8 / 4; ++q) {
        const f32x4 t = pe_row_load4(wsrc, off + i * TSTR * sub_stride + q * 256 + lane * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[i][4 * q + j] = t[j];
      }
    if (++ltap >= tap_hi) { ltap = tap_lo; ++lk; }
  };
  f32x4 acc[MT16];
  auto mma = [&](int tap, const float (&af)[MT16][KS8]) {
    const float* xp = xw + lq * XW + tap * p.dil + l15;
    float bv[KS8];
#pragma unroll
    for (int s8 = 0; s8 < KS8; ++s8) bv[s8] = xp[4 * s8 * XW];
#pragma unroll
    for (int s8 = 0; s8 < KS8; ++s8)
#pragma unroll
      for (int i = 0; i < MT16; ++i) acc[i] = pe_mfma_16x16x4(af[i][s8], bv[s8], acc[i]);
  };

  load_x(myc > 0 ? wi : nchunks);
#pragma unroll
  for (int d = 0; d < D; ++d) load_ring(a[d]);
  PE_SCHED_FENCE();
  const int ncols = L;
  if (n0 >= ncols) return;
  const EpiFlags ef = epi_flags(p);

  // ---- epilogue operands of this wave's slots (slot s -> sub-tile s >> 2, register s & 3; the lane adds row 4*lq)
  float e_b1[NS], e_b2[NS], e_o1[NS], e_o2[NS];
  float* e_dst[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int s = wv + NW * i;
    e_b1[i] = 0.f; e_b2[i] = 0.f; e_o1[i] = 0.f; e_o2[i] = 0.f; e_dst[i] = nullptr;
    if constexpr (GATE) {
      const int ch = GT == 2 ? ((int)blockIdx.y >> 1) * 32 + ((int)blockIdx.y & 1) * 16 + 4 * lq + (s & 3)
                             : (int)blockIdx.y * 32 + (s >> 2) * 16 + 4 * lq + (s & 3);
      if (s < NSLOT && ch < p.split && col < ncols) {
        e_b1[i] = p.bias[ch];
        e_o1[i] = p.bias[p.split + ch];
        if (p.bias2) {
          const float* b2 = p.bias2 + (long)b * p.bias2_bs;
          e_b2[i] = b2[ch];
          e_o2[i] = b2[p.split + ch];
        }
        e_dst[i] = p.out + (long)b * p.o_bs + (long)ch * p.o_cs + col;
      }
    } else {
      const int row = (st0 + (s >> 2)) * 16 + 4 * lq + (s & 3);
      if (s < NSLOT && row < p.rows && col < ncols) {
        if (p.bias) e_b1[i] = p.bias[row];
        if (p.bias2) e_b2[i] = p.bias2[(long)b * p.bias2_bs + row];
        const bool to_skip = p.epi == EPI_WNRS && row >= p.split;
        const bool rd_old = to_skip ? (p.mode != 1) : (ef.use_old || p.epi == EPI_WNRS);
        float* d = to_skip ? p.out2 + (long)b * p.o2_bs + (long)(row - p.split) * p.o2_cs + col
                           : p.out + (long)b * p.o_bs + (long)row * p.o_cs + col;
        if (rd_old) e_o1[i] = *d;
        if (ef.use_res) e_o2[i] = p.res[(long)b * p.r_bs + (long)row * p.r_cs + col];
        e_dst[i] = d;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MT16; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  {
    int k = 0, tap = tap_lo;
    for (int s0 = 0; s0 < nsteps; s0 += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        if (s0 + d < nsteps) {
          if (tap == tap_lo) {
            PE_WAVE_SYNC();
            store_x();
            PE_WAVE_SYNC();
            if (k + 1 < myc) load_x(wi + CL * (k + 1));
          }
          mma(tap, a[d]);
          if (++tap >= tap_hi) { tap = tap_lo; ++k; }
        }
        PE_SCHED_FENCE();
        load_ring(a[d]);
        PE_SCHED_FENCE();
      }
    }
  }
  // ---- cross-wave reduction through LDS (fixed order w = 0..NW-1)
  __syncthreads();
  float* red = sm;                                // [NW waves][MT16*4 slots][64 lanes]
#pragma unroll
  for (int i = 0; i < MT16; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(wv * MT16 * 4 + i * 4 + r) * 64 + lane] = acc[i][r];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int s = wv + NW * i;
    if constexpr (GATE) {
      float ta = 0.f, sa = 0.f;
      if (s < NSLOT) {
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          ta += red[(w * MT16 * 4 + s) * 64 + lane];
          sa += red[(w * MT16 * 4 + NSLOT + s) * 64 + lane];
        }
      }
      if (e_dst[i]) {
        ta += e_b1[i] + e_b2[i];
        sa += e_o1[i] + e_o2[i];
        *e_dst[i] = tanhf(ta) * (1.f / (1.f + expf(-sa)));
      }
    } else {
      float v = 0.f;
      if (s < NSLOT) {
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[(w * MT16 * 4 + s) * 64 + lane];
      }
      if (e_dst[i]) {
        v = ((v + (e_b1[i] + e_b2[i])) * ef.sign + (e_o1[i] + e_o2[i])) * ef.alpha;
        if (ef.relu) v = v > 0.f ? v : 0.f;
        *e_dst[i] = v;
      }
    }
  }
}

}  // namespace pe
