// conv_small_kernel: the tiled conv GEMM for launches whose whole K fits in registers + LDS at once (<= 8 chunk-tap units:
// the generator's polyphase up-convs). OPT-IN (PIPER_HIP_UPPRE=1): verified on the emulator, not yet measured.
// (gfx950 / CDNA4 device code; reference arithmetic cited per kernel, paths relative to
// /root/reference/src/python/piper_train/vits/.)
#pragma once
#include "conv_common.h"

namespace pe {

// Same implicit GEMM, weight packing, tile shape (64 x 64 per workgroup, one 32x32 MFMA tile per wave) and epilogue as
// conv_mfma_kernel<2,2,1,1,16,false,64>, which runs the two late up-convs of a one-utterance call (models.py:321-332,
// polyphase form: K = Cin x 2 taps) at 21 us each for 7 us of MFMA time on the busiest CU: its slab pipeline was built for
// long K -- one 32-channel x slab in flight while the previous one feeds the MFMAs, weight fragments one unit ahead -- and
// with two taps of work per slab every slab's and every fragment's memory latency is exposed (device trace: 16 us inside
// a workgroup). Here EVERYTHING the workgroup will read is requested before anything is used: all <= 8 units' weight
// fragments (128 registers) and all <= 4 chunks' x slabs (64 KB of LDS), one barrier, then the MFMAs back to back.
constexpr int CS_MAXC = 4, CS_MAXU = 8;

__global__ __launch_bounds__(256) void conv_small_kernel(ConvP p) {
  PE_KTRACE(23);
  constexpr int BN = 64, NCOL = 2, XS = NCOL * 64, KH = KC / 2;
  PE_DYN_SMEM(float, xs);                         // [nchunks <= 4][KC][XS]
  const int b = blockIdx.z;
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * 64;
  const int tid = threadIdx.x, lane = tid & 63, wv = PE_UNIFORM(tid >> 6);
  const int wm = wv >> 1, wn = wv & 1;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int ntaps = p.ntaps, nchunks = p.nchunks, nunits = nchunks * ntaps;
  const int mtile = m0 / 32 + wm;
  const int wstride_mt = nunits * KH * 64;
  const pe_rowsrc wsrc = pe_make_row(p.wp + (long)mtile * wstride_mt, wstride_mt);
  // ---- every unit's A fragments (a unit past the end reads zeros)
  float a[CS_MAXU][KH];
#pragma unroll
  for (int u = 0; u < CS_MAXU; ++u) load_frags<KH>(wsrc, u < nunits ? u * (KH * 64) : wstride_mt, lane, a[u]);
  // ---- every chunk's x slab: rows wv, wv + 4, ... of the chunk, two 64-column halves per lane; requested against the row
  // stride (the length arrives meanwhile), zeroed beyond the length when stored
  const float* xb = p.x + (long)b * p.x_bs;
  const int tbase = n0 - p.padl + lane;
  float xr[CS_MAXC][KC / 4][NCOL];
#pragma unroll
  for (int c = 0; c < CS_MAXC; ++c)
#pragma unroll
    for (int rr = 0; rr < KC / 4; ++rr) {
      const int ci = c * KC + wv + 4 * rr;
      const pe_rowsrc row = pe_make_row(xb + (long)ci * p.x_cs, (c < nchunks && ci < p.Cin) ? p.x_cs : 0);
#pragma unroll
      for (int cc = 0; cc < NCOL; ++cc) xr[c][rr][cc] = pe_row_load(row, tbase + 64 * cc);
    }
  PE_SCHED_FENCE();
  const int L = p.lens[b] * p.len_mul;
  const int ncols = (p.epi == EPI_CONVT) ? L + 1 : L;
  if (n0 >= ncols) return;
  const float slope = p.in_slope;
#pragma unroll
  for (int c = 0; c < CS_MAXC; ++c)
    if (c < nchunks) {
#pragma unroll
      for (int rr = 0; rr < KC / 4; ++rr)
#pragma unroll
        for (int cc = 0; cc < NCOL; ++cc) {
          float v = (tbase + 64 * cc < L) ? xr[c][rr][cc] : 0.f;
          v = v > 0.f ? v : v * slope;
          xs[(c * KC + wv + 4 * rr) * XS + lane + 64 * cc] = v;
        }
    }
  __syncthreads();
  // ---- the whole K, unit = (chunk, tap) in the packed order (chunk-major)
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int u = 0; u < CS_MAXU; ++u)
    if (u < nunits) {
      const int c = u / ntaps, tap = u - c * ntaps;
      const float* xp = xs + (c * KC + lhi) * XS + tap * p.dil + wn * 32 + l31;
#pragma unroll
      for (int kk = 0; kk < KH; ++kk) acc = pe_mfma_32x32x2(a[u][kk], xp[2 * kk * XS], acc);
    }
  const EpiFlags ef = epi_flags(p);
  conv_store_tile(p, ef, b, mtile * 32, n0 + wn * 32 + l31, lhi, L, ncols, acc);
}

}  // namespace pe
