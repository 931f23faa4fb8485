// conv_bf3_kernel: the tiled conv GEMM on the bf16 matrix pipe with split operands (opt-in, PIPER_HIP_MATRIX=bf16x3).
// (gfx950 / CDNA4 device code; reference arithmetic cited per kernel, paths relative to
// /root/reference/src/python/piper_train/vits/.)
#pragma once
#include "conv_common.h"

namespace pe {

// Same implicit GEMM as conv_mfma_kernel (conv_mfma.h):
//   D[row][col] = sum_{ci,k} W[row][ci][k] * act(x[ci][col + k*dil - padl])
// but every f32 operand is split into two bf16 terms, v = hi + lo with hi = bf16(v), lo = bf16(v - hi), and the
// product runs as three v_mfma_f32_32x32x16_bf16 (hi*hi + hi*lo + lo*hi, f32 accumulate; the lo*lo term, 2^-16
// relative, is dropped): 16 mantissa bits per operand at 16x the f32 matrix rate, i.e. 6 instructions of 32 cycles per
// (32 channels, tap) step and 32x32 tile where the f32 kernel issues 16 of 64 cycles. Not bit-identical to the
// reference's fp32 arithmetic (relative error ~1e-5 per product, averaging out over K): used for the coupling flow and
// the generator only, never for the text encoder / duration predictor (durations stay exact), and never by default.
//   * B operand (activations): staged per 32-channel chunk through registers, pre-activation and the split applied
//     ONCE per element there, written to LDS as [part hi|lo][k group of 8 channels][column][8 bf16]: a lane's B
//     fragment of one k-step (8 consecutive channels of its column) is one ds_read_b128, a dilated tap a shifted column.
//   * A operand (weights): split and packed at load time (engine_pack.cpp pack_matrix) as
//     [m tile][chunk][tap][part][k-step 0|1][lane][8 bf16]: 1024 floats per step like the f32 packing, four 16-byte
//     loads per lane and (m tile, chunk, tap), prefetched one unit ahead (ping-pong).
//   * accumulator layout == v_mfma_f32_32x32x2_f32's: the epilogues of conv_common.h are shared.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#ifdef PE_EMU
#define pe_mfma_bf16_32x32x16(a, b, c) emu_mfma_bf16_32x32x16((a), (b), (c))
#else
#define pe_mfma_bf16_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#endif

// f32 <-> bf16, round to nearest even (v_cvt_pk_bf16_f32 on the GPU; the emulator build spells it out in integer ops)
#ifdef PE_EMU
inline __bf16 pe_f2bf(float v) {
  unsigned u = __float_as_uint(v);
  u += 0x7fffu + ((u >> 16) & 1u);
  return __builtin_bit_cast(__bf16, (unsigned short)(u >> 16));
}
inline float pe_bf2f(__bf16 h) { return __uint_as_float((unsigned)__builtin_bit_cast(unsigned short, h) << 16); }
#else
__device__ __forceinline__ __bf16 pe_f2bf(float v) { return (__bf16)v; }
__device__ __forceinline__ float pe_bf2f(__bf16 h) { return (float)h; }
#endif
// v (8 floats) -> hi / lo bf16 vectors
__device__ __forceinline__ void bf3_split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 h = pe_f2bf(v[i]);
    hi[i] = h;
    lo[i] = pe_f2bf(v[i] - pe_bf2f(h));
  }
}

template <int WM, int WN, int MT, int NT, bool GATE, int HALO>
__global__ __launch_bounds__(256, (MT * NT >= 4 ? 2 : 3))
void conv_bf3_kernel(ConvP p) {
  PE_KTRACE(12);
  constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
  constexpr int NCOL = (BN + HALO + 63) / 64;    // staging columns per lane; (taps-1)*dilation <= HALO
  static_assert(WM * WN == 4, "4 waves per block");
  static_assert(!GATE || MT == 2, "gate epilogue pairs two M tiles");
  static_assert(KC == 32, "a chunk is four k groups of 8 channels (two k-steps of the 32x32x16 MFMA)");
  constexpr int XS = NCOL * 64;                   // LDS columns per k group
  constexpr int PART = 4 * XS;                    // bf16x8 elements per part (hi | lo) of one slab
  PE_DYN_SMEM(bf16x8, xs);                        // 2 x [2 parts][4 k groups][XS] x 16 bytes
  const int b = blockIdx.z;
  const int L = p.lens[b] * p.len_mul;
  const int ncols = (p.epi == EPI_CONVT) ? L + 1 : L;
  const int tile0 = blockIdx.x;
  if (tile0 * BN >= ncols) return;
  (void)BM;
  const int m0 = blockIdx.y * BM;
  const int tid = threadIdx.x, lane = tid & 63, wv = PE_UNIFORM(tid >> 6);
  const int wm = wv / WN, wn = wv % WN;
  const int l31 = lane & 31, lhi = lane >> 5;

  f32x16 acc[MT][NT];
  const float* xb = p.x + (long)b * p.x_bs;
  const int mtile0 = m0 / 32 + wm * MT;
  const float slope = p.in_slope;
  const int ntaps = p.ntaps, nchunks = p.nchunks;
  const int nunits = nchunks * ntaps;             // unit = (chunk, tap)
  const int wstride_mt = nunits * 1024;
  const pe_rowsrc wsrc = pe_make_row(p.wpb + (long)mtile0 * wstride_mt, MT * wstride_mt);
  const int n0 = tile0 * BN;

  float xr[8][NCOL];
  // wave wv stages k group wv of the chunk (channels 8 wv .. 8 wv + 7), lane -> column: a thread holds the 8 channels
  // of its column, i.e. exactly one B fragment per staging column. Rows through buffer descriptors (halo, tail and
  // padded channels read 0), zero-length descriptors when the slab does not exist.
  auto load_x = [&](int c, bool live) {
    const int tbase = n0 - p.padl + lane;
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int ci = c * KC + 8 * wv + rr;
      const pe_rowsrc row = pe_make_row(xb + (long)ci * p.x_cs, (live && ci < p.Cin) ? L : 0);
#pragma unroll
      for (int cc = 0; cc < NCOL; ++cc) xr[rr][cc] = pe_row_load(row, tbase + 64 * cc);
    }
  };
  auto store_x = [&](int buf) {
    bf16x8* dst = xs + buf * 2 * PART + wv * XS + lane;
#pragma unroll
    for (int cc = 0; cc < NCOL; ++cc) {
      float v[8];
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const float t = xr[rr][cc];
        v[rr] = t > 0.f ? t : t * slope;
      }
      bf16x8 hi, lo;
      bf3_split8(v, hi, lo);
      dst[64 * cc] = hi;
      dst[PART + 64 * cc] = lo;
    }
  };
  // A fragments of unit u: [part][k-step] per M tile
  auto load_a = [&](int u, bf16x8 (&a)[MT][4]) {
    const int off = PE_UNIFORM(u * 1024);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int f = 0; f < 4; ++f)
        a[i][f] = __builtin_bit_cast(bf16x8, pe_row_load4(wsrc, off + i * wstride_mt + f * 256 + lane * 4));
  };
  auto read_b = [&](int tap, const bf16x8* xbuf, bf16x8 (&bv)[NT][4]) {
    const bf16x8* xp = xbuf + lhi * XS + tap * p.dil + wn * NT * 32 + l31;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int f = 0; f < 4; ++f) bv[j][f] = xp[(f >> 1) * PART + (f & 1) * 2 * XS + j * 32];
  };
  // small terms first: lo*hi and hi*lo, then hi*hi, k-step by k-step
  auto mma = [&](const bf16x8 (&a)[MT][4], const bf16x8 (&bv)[NT][4]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = pe_mfma_bf16_32x32x16(a[i][2 + ks], bv[j][ks], acc[i][j]);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = pe_mfma_bf16_32x32x16(a[i][ks], bv[j][2 + ks], acc[i][j]);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = pe_mfma_bf16_32x32x16(a[i][ks], bv[j][ks], acc[i][j]);
    }
  };

  bf16x8 aA[MT][4], aB[MT][4];
  load_x(0, true);
  load_a(0, aA);
  store_x(0);
  __syncthreads();
  const EpiFlags ef = epi_flags(p);
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  int u = 0;
  for (int c = 0; c < nchunks; ++c) {
    const bf16x8* xbuf = xs + (c & 1) * 2 * PART;
    load_x(c + 1, c + 1 < nchunks);               // next slab: in flight for the whole chunk
    for (int tap = 0; tap < ntaps; tap += 2) {
      bf16x8 bv[NT][4];
      load_a(u + 1 == nunits ? 0 : u + 1, aB);
      read_b(tap, xbuf, bv);
      PE_SCHED_FENCE();
      mma(aA, bv);
      PE_SCHED_FENCE();
      ++u;
      if (tap + 1 < ntaps) {
        load_a(u + 1 == nunits ? 0 : u + 1, aA);
        read_b(tap + 1, xbuf, bv);
        PE_SCHED_FENCE();
        mma(aB, bv);
        PE_SCHED_FENCE();
        ++u;
      }
    }
    if (ntaps & 1) {        // odd tap count: the next unit's fragments were prefetched into aB
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int f = 0; f < 4; ++f) aA[i][f] = aB[i][f];
    }
    if (c + 1 < nchunks) {
      store_x((c + 1) & 1);
      __syncthreads();
    }
  }
  // ---- epilogue (shared with conv_mfma_kernel)
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

}  // namespace pe
