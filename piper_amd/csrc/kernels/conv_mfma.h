// conv_mfma_kernel: Conv1d / ConvTranspose1d as a tiled implicit GEMM on the f32 MFMA (batched launches).
// (gfx950 / CDNA4 device code; reference arithmetic cited per kernel, paths relative to
// /root/reference/src/python/piper_train/vits/.)
#pragma once
#include "conv_common.h"

namespace pe {

// Conv1d / ConvTranspose1d as an implicit GEMM on the f32 matrix cores.
//   D[row][col] = sum_{ci,k} W[row][ci][k] * act(x[ci][col + k*dil - padl])
// rows -> MFMA M, cols (time) -> MFMA N, K = (ci, tap). v_mfma_f32_32x32x2_f32 keeps the reference's
// fp32 arithmetic exactly (k-ordered fmaf chain).
//   * B operand (activations): one [KC x (BN+halo)] slab per K-chunk in LDS, double-buffered; the next
//     chunk is fetched into registers while the current one feeds the MFMAs, so a dilated tap is just
//     a shifted LDS read and the pre-activation (leaky-relu) is applied once per element.
//   * A operand (weights): pre-packed at load time in fragment order and read through a buffer
//     descriptor as float4 per lane; the fragments of the next unit are prefetched into a second register
//     set (ping-pong) while the current unit's MFMAs issue.
//   * Every global load sits at an unconditional position of the loop nest (a slab or a unit that does not
//     exist is read through a zero-length descriptor / wraps to unit 0) and scheduling fences keep the
//     prefetch block ahead of the MFMAs: with a branch between a load and its use the compiler's wait-count
//     bookkeeping collapses to "drain everything" in every unit.
// Covers every groups=1 Conv1d of attentions.py / modules.py / models.py and (EPI_CONVT) the polyphase
// form of Generator.ups ConvTranspose1d (models.py:321-332) where k = 2*stride.
// second launch-bound argument = waves per SIMD the register allocation must leave room for: latency here is
// hidden across workgroups (profiles/r01_ablation.txt), so small wave tiles are held to 128 / 168 registers
template <int WM, int WN, int MT, int NT, int KS, bool GATE, int HALO>
__device__ __forceinline__ void conv_mfma_body(const ConvP& p, const int bx, const int by, const int b) {
  constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
  constexpr int NCOL = (BN + HALO + 63) / 64;    // staging columns per lane; (taps-1)*dilation <= HALO
  constexpr int KH = KC / 2;
  constexpr int NSUB = KH / KS;                   // A prefetch sets per (chunk, tap)
  static_assert(WM * WN == 4, "4 waves per block");
  static_assert(NSUB * KS == KH && KS % 4 == 0, "KS must divide KC/2 in float4 groups");
  static_assert(!GATE || MT == 2, "gate epilogue pairs two M tiles");
  constexpr int XS = NCOL * 64;                   // LDS row stride (compile time: taps become immediates)
  PE_DYN_SMEM(float, xs);                         // 2 x [KC][XS]
  const int L = p.lens[b] * p.len_mul;
  const int ncols = (p.epi == EPI_CONVT) ? L + 1 : L;
  // a workgroup walks p.tpb consecutive column tiles (1 by default, profiles/r01_tpb_sweep.txt)
  const int tile0 = bx * p.tpb;
  const int ntile_all = (ncols + BN - 1) / BN;
  if (tile0 >= ntile_all) return;
  const int ntl = (ntile_all - tile0) < p.tpb ? (ntile_all - tile0) : p.tpb;
  const int m0 = by * BM;
  const int tid = threadIdx.x, lane = tid & 63, wv = PE_UNIFORM(tid >> 6);
  const int wm = wv / WN, wn = wv % WN;
  const int l31 = lane & 31, lhi = lane >> 5;

  f32x16 acc[MT][NT];
  const float* xb = p.x + (long)b * p.x_bs;
  const int mtile0 = m0 / 32 + wm * MT;
  const float slope = p.in_slope;
  const int ntaps = p.ntaps, nchunks = p.nchunks;
  const int upc = ntaps * NSUB;                   // units per chunk
  const int nunits = nchunks * upc;
  const int nslabs = ntl * nchunks;
  const int wstride_mt = nchunks * ntaps * KH * 64;
  const pe_rowsrc wsrc = pe_make_row(p.wp + (long)mtile0 * wstride_mt, MT * wstride_mt);

  float xr[KC / 4][NCOL];
  // Branch-free staging: rows are read through buffer descriptors (hardware range check returns 0 for
  // the halo, the tail and padded channels) and the LDS rows are NCOL*64 wide so every lane stores
  // unconditionally. A slab that does not exist (`live` false) reads through zero-length descriptors.
  auto load_x = [&](int s, bool live) {
    const int tl = s / nchunks, c = s - tl * nchunks;
    const int tbase = (tile0 + tl) * BN - p.padl + lane;
#pragma unroll
    for (int rr = 0; rr < KC / 4; ++rr) {
      const int ci = c * KC + wv + 4 * rr;
      const pe_rowsrc row = pe_make_row(xb + (long)ci * p.x_cs, (live && ci < p.Cin) ? L : 0);
#pragma unroll
      for (int cc = 0; cc < NCOL; ++cc) xr[rr][cc] = pe_row_load(row, tbase + 64 * cc);
    }
  };
  // (the pre-activation costs matrix-pipe time -- a VALU op takes 5-7 cycles out of the MFMA stream, profiles/r04_mfma_mix.txt --
  // so it is two ops per element, max(v, v * slope), and none at all for inputs that take no activation: slope == 1)
  auto store_x = [&](int buf) {
    float* dst = xs + buf * KC * XS + wv * XS + lane;
    if (slope != 1.f) {
#pragma unroll
      for (int rr = 0; rr < KC / 4; ++rr)
#pragma unroll
        for (int cc = 0; cc < NCOL; ++cc) dst[4 * rr * XS + 64 * cc] = pe_lrelu(xr[rr][cc], slope);
    } else {
#pragma unroll
      for (int rr = 0; rr < KC / 4; ++rr)
#pragma unroll
        for (int cc = 0; cc < NCOL; ++cc) dst[4 * rr * XS + 64 * cc] = xr[rr][cc];
    }
  };
  // unit u of a tile = (chunk, tap, sub): KS fragments per M tile
  auto load_a = [&](int u, float (&a)[MT][KS]) {
    const int ut = u / NSUB, sub = u - ut * NSUB;
    const int off = PE_UNIFORM(ut * (KH * 64));
#pragma unroll
    for (int i = 0; i < MT; ++i) load_frags<KS>(wsrc, off + i * wstride_mt, lane, a[i], sub * KS);
  };
  auto read_b = [&](int tap, int sub, const float* xbuf, float (&bv)[KS][NT]) {
    const float* xp = xbuf + (lhi + 2 * KS * sub) * XS + tap * p.dil + wn * NT * 32 + l31;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
#pragma unroll
      for (int j = 0; j < NT; ++j) bv[kk][j] = xp[2 * kk * XS + j * 32];
  };
  auto mma = [&](const float (&a)[MT][KS], const float (&bv)[KS][NT]) {
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = pe_mfma_32x32x2(a[i][kk], bv[kk][j], acc[i][j]);
  };

  float aA[MT][KS], aB[MT][KS];
  load_x(0, true);
  load_a(0, aA);
  store_x(0);
  __syncthreads();
  const EpiFlags ef = epi_flags(p);
  for (int tl = 0; tl < ntl; ++tl) {
    const int n0 = (tile0 + tl) * BN;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int u = 0;                                    // unit index inside the tile
    for (int c = 0; c < nchunks; ++c) {
      const int s = tl * nchunks + c;
      const float* xbuf = xs + (s & 1) * KC * XS;
      load_x(s + 1, s + 1 < nslabs);              // next slab: in flight for the whole chunk
      int tap = 0, sub = 0;
      auto next_unit = [&]() { ++u; if (++sub == NSUB) { sub = 0; ++tap; } };
      for (int j = 0; j < upc; j += 2) {
        float bv[KS][NT];
        load_a(u + 1 == nunits ? 0 : u + 1, aB);  // wraps to the next tile's first unit
        read_b(tap, sub, xbuf, bv);
        PE_SCHED_FENCE();
        mma(aA, bv);
        PE_SCHED_FENCE();
        next_unit();
        if (j + 1 < upc) {
          load_a(u + 1 == nunits ? 0 : u + 1, aA);
          read_b(tap, sub, xbuf, bv);
          PE_SCHED_FENCE();
          mma(aB, bv);
          PE_SCHED_FENCE();
          next_unit();
        }
      }
      if (upc & 1) {        // odd unit count: the next unit's fragments were prefetched into aB
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int kk = 0; kk < KS; ++kk) aA[i][kk] = aB[i][kk];
      }
      if (s + 1 < nslabs) {
        store_x((s + 1) & 1);
        __syncthreads();
      }
    }
    // ---- epilogue of this tile
    if constexpr (GATE) {
      const int q = mtile0 >> 1;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int col = n0 + (wn * NT + j) * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int ch = q * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          PE_OPAQUE(ch);
          if (ch < p.split && col < ncols) conv_store_gate(p, b, ch, col, acc[0][j][r], acc[MT - 1][j][r]);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          conv_store_tile(p, ef, b, (mtile0 + i) * 32, n0 + (wn * NT + j) * 32 + l31, lhi, L, ncols, acc[i][j]);
    }
  }
}

#define PE_CONV_MFMA_BOUNDS(MT, NT, WN, GATE, HALO) \
  __launch_bounds__(256, (MT * NT == 1 ? ((HALO == 128 && WN == 4) ? 3 : 4) : ((GATE && MT * NT == 2) ? 3 : 2)))
template <int WM, int WN, int MT, int NT, int KS, bool GATE, int HALO>
__global__ PE_CONV_MFMA_BOUNDS(MT, NT, WN, GATE, HALO) void conv_mfma_kernel(ConvP p) {
  PE_KTRACE(11);
  conv_mfma_body<WM, WN, MT, NT, KS, GATE, HALO>(p, blockIdx.x, blockIdx.y, blockIdx.z);
}
// Up to three INDEPENDENT convs of the same tile configuration in one launch (grid.z = conv x utterance): the sibling
// resblocks of an MRF stage (models.py:356-363) read the same input, and a single one of them is a few workgroups per CU --
// 834 tiles on 256 CUs are 3.26 per CU, so the CUs that draw a fourth tile set the launch time (81 % of the chip's
// rate) -- while three of them together are ~10 per CU, dispatched longest kernel first. (The split-K form of the same
// idea: conv_splitk_group_kernel.)
template <int WM, int WN, int MT, int NT, int KS, int HALO>
__global__ PE_CONV_MFMA_BOUNDS(MT, NT, WN, false, HALO) void conv_mfma_group_kernel(ConvG g) {
  PE_KTRACE(11);
  const int gi = PE_UNIFORM((int)blockIdx.z / g.B);
  const ConvP& p = g.c[gi];
  if ((int)blockIdx.y * (WM * MT * 32) >= p.rows) return;      // a sibling with fewer row blocks than the grid
  conv_mfma_body<WM, WN, MT, NT, KS, false, HALO>(p, blockIdx.x, blockIdx.y, (int)blockIdx.z - gi * g.B);
}

}  // namespace pe
