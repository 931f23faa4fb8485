// Shared pieces of the conv GEMM kernels: launch parameters, A-operand fragment loads, the fused epilogues.
// (gfx950 / CDNA4 device code; reference arithmetic cited per kernel, paths relative to
// /root/reference/src/python/piper_train/vits/.)
#pragma once
#include "../pe_rt.h"
#include "params.h"

namespace pe {


// A-operand fragments (engine_pack.cpp: pack_matrix). One (m tile, chunk, tap) step is 1024 floats:
// [q = 0..3][lane][j = 0..3] holds fragment kk = 4q + j of `lane`, so NK fragments are NK/4 float4 loads.
template <int NK>
__device__ __forceinline__ void load_frags(const float* step_base, int lane, int kk0, float (&a)[NK]) {
#pragma unroll
  for (int q = 0; q < NK / 4; ++q) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(step_base + (kk0 / 4 + q) * 256 + lane * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) a[4 * q + j] = t[j];
  }
}
template <int NK>
__device__ __forceinline__ void load_frags(const pe_rowsrc& w, int step_off, int lane, float (&a)[NK], int kk0 = 0) {
#pragma unroll
  for (int q = 0; q < NK / 4; ++q) {
    const f32x4 t = pe_row_load4(w, step_off + (kk0 / 4 + q) * 256 + lane * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) a[4 * q + j] = t[j];
  }
}



// ---- shared epilogue of the conv GEMM kernels: one accumulator element (row, col) of utterance b.
// Every non-transposed mode is the same straight-line form
//     dst = alpha * ( old*use_old + (res*use_res + (acc + bias)*sign) )
// with per-launch uniform flags, which keeps the unrolled epilogue small:
//   STORE  : dst=out                         RESADD : +res            SUBFROM: old - v  (modules.py:464)
//   ACCUM  : MRF sum/scale (models.py:356-363)       WNRS: rows<split h += v, else skip (+)= v (modules.py:201-208)
template <bool V> struct pe_bool { static constexpr bool value = V; };
struct EpiFlags {
  float sign, alpha;
  bool use_res, use_old, relu;
};
__device__ __forceinline__ EpiFlags epi_flags(const ConvP& p) {
  EpiFlags f{1.f, 1.f, false, false, false};
  switch (p.epi) {
    case EPI_STORE: f.relu = p.act == ACT_RELU; break;
    case EPI_RESADD: f.use_res = true; break;
    case EPI_SUBFROM: f.sign = -1.f; f.use_old = true; break;
    case EPI_ACCUM:
      f.use_res = true;
      f.use_old = (p.mode == 1 || p.mode == 2);
      if (p.mode >= 2) f.alpha = p.alpha;
      break;
    default: break;
  }
  return f;
}
__device__ __forceinline__ void conv_store(const ConvP& p, const EpiFlags& f, int b, int row, int col, float v, int L) {
  if (p.epi == EPI_CONVT) {
    const int co = row / p.up, ph = row - co * p.up;
    const int t = col * p.up + ph - p.padT;
    if (t >= 0 && t < L * p.up) p.out[(long)b * p.o_bs + (long)co * p.o_cs + t] = v + (p.bias ? p.bias[co] : 0.f);
    return;
  }
  if (p.bias) v += p.bias[row];
  if (p.bias2) v += p.bias2[(long)b * p.bias2_bs + row];
  float* d = p.out + (long)b * p.o_bs + (long)row * p.o_cs + col;
  bool use_old = f.use_old;
  if (p.epi == EPI_WNRS) {
    if (row < p.split) use_old = true;
    else {
      d = p.out2 + (long)b * p.o2_bs + (long)(row - p.split) * p.o2_cs + col;
      use_old = p.mode != 1;
    }
  }
  v *= f.sign;
  if (f.use_res) v += p.res[(long)b * p.r_bs + (long)row * p.r_cs + col];
  if (use_old) v += *d;
  v *= f.alpha;
  if (f.relu) v = v > 0.f ? v : 0.f;
  *d = v;
}
// One 32x32 accumulator tile (16 values per lane) through the epilogue, branch-free: every operand stream
// (bias, speaker bias, previous value, residual) is a buffer descriptor whose length is 0 when the stream
// is not used and rows*stride otherwise, so unused operands read as 0; invalid columns (and, in a matrix's partial last
// row tile, invalid rows) poison the lane offset, so their loads give 0 and their stores are dropped. Addresses are one
// per-lane offset (row base, column) shared by the 16 elements plus a wave-uniform k*stride that rides in
// an SGPR: no per-element VALU address arithmetic. All loads are issued before the first store (out and
// res may alias). WNRS relies on split % 32 == 0 (checked at load): a tile lies on one side of the split.
__device__ __forceinline__ void conv_store_tile(const ConvP& p, const EpiFlags& f, int b, int row0, int col, int lhi,
                                                int L, int ncols, const f32x16& acc) {
  constexpr int OOB = 0x3fffffff;                // element index beyond any descriptor
  int rb = row0 + 4 * lhi;
  PE_OPAQUE(rb);       // keeps the (tile-invariant) row addressing from being hoisted out of the tile loop
  if (p.epi == EPI_CONVT) {
    // row = co*up + phase; output sample t = col*up + phase - padT (models.py:321-332, polyphase form)
    const pe_rowsrc od = pe_make_row_u(p.out + (long)b * p.o_bs, (p.rows / p.up) * p.o_cs);
    const pe_rowsrc bd = pe_make_row_u(p.bias, p.bias ? p.rows / p.up : 0);
    const int tmax = L * p.up;
    if (p.up_vec == 4) {
      // up % 4 == 0: rows rb + 8g .. + 3 (rb = row0 + 4 lhi) are four consecutive phases of ONE output channel, i.e. the
      // lane's accumulators 4g .. 4g + 3 are four consecutive output samples: one 16-byte store instead of four 4-byte
      // stores `up` samples apart from the next lane's (any dword alignment: stride 4 with padding 2 lands on 8-byte
      // boundaries). Medium voice at 64 utterances: the three up-convs 349 / 643 / 790 -> 316 / 544 / 674 us, against
      // both the element-wise stores and the tile transposed through LDS (profiles/r04_notes.md, call 11). Groups that
      // straddle the ends of the utterance go element by element.
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int row = rb + 8 * g;
        const int co = (int)(((unsigned long long)(unsigned)row * p.up_magic) >> 32);     // row / up
        const int t0 = col * p.up + (row - co * p.up) - p.padT;
        const bool ok = row < p.rows && col < ncols;
        const float bz = pe_row_load(bd, ok ? co : OOB);
        const float v0 = acc[4 * g] + bz, v1 = acc[4 * g + 1] + bz, v2 = acc[4 * g + 2] + bz, v3 = acc[4 * g + 3] + bz;
        const int o0 = co * p.o_cs + t0;
        if (ok && t0 >= 0 && t0 + 3 < tmax) {
          pe_row_store4(od, o0, v0, v1, v2, v3);
        } else if (ok) {
          if (t0 >= 0 && t0 < tmax) pe_row_store_so(od, o0, 0, v0);
          if (t0 + 1 >= 0 && t0 + 1 < tmax) pe_row_store_so(od, o0 + 1, 0, v1);
          if (t0 + 2 >= 0 && t0 + 2 < tmax) pe_row_store_so(od, o0 + 2, 0, v2);
          if (t0 + 3 >= 0 && t0 + 3 < tmax) pe_row_store_so(od, o0 + 3, 0, v3);
        }
      }
      return;
    }
    if (p.up_vec == 2) {
      // up == 2: rows rb + 8g .. + 3 are both phases of TWO channels: two 8-byte stores
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int row = rb + 8 * g + 2 * h, co = row >> 1;
          const int t0 = 2 * col - p.padT;
          const bool ok = row < p.rows && col < ncols;
          const float bz = pe_row_load(bd, ok ? co : OOB);
          const float v0 = acc[4 * g + 2 * h] + bz, v1 = acc[4 * g + 2 * h + 1] + bz;
          const int o0 = co * p.o_cs + t0;
          if (ok && t0 >= 0 && t0 + 1 < tmax) {
            pe_row_store2(od, o0, v0, v1);
          } else if (ok) {
            if (t0 >= 0 && t0 < tmax) pe_row_store_so(od, o0, 0, v0);
            if (t0 + 1 >= 0 && t0 + 1 < tmax) pe_row_store_so(od, o0 + 1, 0, v1);
          }
        }
      return;
    }
    float bv[16];
    int off[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = rb + (r & 3) + 8 * (r >> 2);
      const int co = (int)(((unsigned long long)(unsigned)row * p.up_magic) >> 32);     // row / up
      const int t = col * p.up + (row - co * p.up) - p.padT;
      off[r] = (row < p.rows && col < ncols && t >= 0 && t < tmax) ? co * p.o_cs + t : OOB;
      bv[r] = pe_row_load(bd, row < p.rows ? co : OOB);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) pe_row_store_so(od, off[r], 0, acc[r] + bv[r]);
    return;
  }
  const bool to_skip = p.epi == EPI_WNRS && row0 >= p.split;        // uniform per tile
  const bool rd_old = to_skip ? (p.mode != 1) : (f.use_old || p.epi == EPI_WNRS);
  const int orow0 = to_skip ? p.split : 0;                           // first GEMM row of the destination tensor
  const int orows = to_skip ? p.rows - p.split : (p.epi == EPI_WNRS ? p.split : p.rows);
  const int ocs = to_skip ? p.o2_cs : p.o_cs;
  float* ob = to_skip ? p.out2 + (long)b * p.o2_bs : p.out + (long)b * p.o_bs;
  const pe_rowsrc od = pe_make_row_u(ob, orows * ocs);
  const pe_rowsrc old = pe_make_row_u(ob, rd_old ? orows * ocs : 0);
  const pe_rowsrc rd = pe_make_row_u(p.res + (long)b * p.r_bs, f.use_res ? p.rows * p.r_cs : 0);
  const pe_rowsrc bd = pe_make_row_u(p.bias, p.bias ? p.rows : 0);
  const pe_rowsrc b2d = pe_make_row_u(p.bias2 + (long)b * p.bias2_bs, p.bias2 ? p.rows : 0);
  const bool cok = col < ncols;
  const int ooff = cok ? (rb - orow0) * ocs + col : OOB;
  const int roff = cok ? rb * p.r_cs + col : OOB;
  // A lane's 16 elements share ONE per-lane offset; the element's row rides in the SGPR offset, which the hardware's range
  // check does not include (gfx9 / CDNA raw buffers: only the VGPR + immediate part is compared with num_records). Rows
  // beyond the GEMM exist only in the LAST row tile of a matrix whose row count is not a multiple of 32: that tile
  // (wave-uniform test) poisons the per-element lane offset instead; every other tile runs without per-element VALU.
  const int row_end = to_skip ? p.rows : (p.epi == EPI_WNRS ? p.split : p.rows);      // first GEMM row NOT of this tensor
  auto body = [&](auto partialc) {
    constexpr bool PARTIAL = decltype(partialc)::value;
    // two groups of eight elements (register budget of the 4-waves-per-SIMD instantiations); an element only
    // ever reads its own location, so a group's stores cannot disturb the next group's loads
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      float b1[8], b2[8], o1[8], o2[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int r = 8 * g + e, kr = (r & 3) + 8 * (r >> 2);
        const bool rok = !PARTIAL || rb + kr < row_end;
        b1[e] = pe_row_load(bd, rb + kr);
        b2[e] = pe_row_load(b2d, rb + kr);
        o1[e] = pe_row_load_so(old, rok ? ooff : OOB, kr * ocs);
        o2[e] = pe_row_load_so(rd, rok ? roff : OOB, kr * p.r_cs);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int r = 8 * g + e, kr = (r & 3) + 8 * (r >> 2);
        const bool rok = !PARTIAL || rb + kr < row_end;
        float v = ((acc[r] + (b1[e] + b2[e])) * f.sign + (o1[e] + o2[e])) * f.alpha;
        if (f.relu) v = v > 0.f ? v : 0.f;
        pe_row_store_so(od, rok ? ooff : OOB, kr * ocs, v);
      }
      PE_SCHED_FENCE();
    }
  };
  if (row0 + 32 <= row_end) body(pe_bool<false>{});
  else body(pe_bool<true>{});
}
// commons.py:99-106 fused_add_tanh_sigmoid_multiply on a (tanh-tile, sigmoid-tile) accumulator pair
__device__ __forceinline__ void conv_store_gate(const ConvP& p, int b, int ch, int col, float ta, float sa) {
  ta += p.bias[ch];
  sa += p.bias[p.split + ch];
  if (p.bias2) {
    const float* b2 = p.bias2 + (long)b * p.bias2_bs;
    ta += b2[ch];
    sa += b2[p.split + ch];
  }
  p.out[(long)b * p.o_bs + (long)ch * p.o_cs + col] = tanhf(ta) * (1.f / (1.f + expf(-sa)));
}

}  // namespace pe
