// conv_split_kernel: the tiled conv GEMM on the 16-bit matrix pipe with split f32 operands (opt-in, PIPER_HIP_MATRIX=bf16x3 | f16x3 | bf16x6).
// (gfx950 / CDNA4 device code; reference arithmetic cited per kernel, paths relative to
// /root/reference/src/python/piper_train/vits/.)
#pragma once
#include "conv_common.h"

namespace pe {

// Same implicit GEMM as conv_mfma_kernel (conv_mfma.h):
//   D[row][col] = sum_{ci,k} W[row][ci][k] * act(x[ci][col + k*dil - padl])
// but every f32 operand is split into NT low-precision terms and the product runs as a few 16-bit MFMAs with f32
// accumulate (16x the f32 matrix rate per instruction). Three split modes SM, selected by PIPER_HIP_MATRIX:
//   SM 0 "bf16x3": v = h + l, h = bf16(v), l = bf16(v - h); products hh + hl + lh (ll, 2^-16 relative, dropped):
//                  16 significand bits per operand, 3 x v_mfma_f32_32x32x16_bf16.
//   SM 1 "f16x3":  v = h + l, h = f16(v), l = f16(v - h); the same three products on v_mfma_f32_32x32x16_f16:
//                  22 significand bits per operand (the f32 significand has 24). f16's narrow exponent is handled by
//                  scaling: weights are packed times a power of two per conv (largest magnitude near 2^13, undone
//                  exactly on the accumulators), activations clamp at +-65504 (never reached by a voice; a term
//                  below f16's subnormal step 2^-24 is dropped, an ABSOLUTE error of 6e-8 per element).
//   SM 2 "bf16x6": v = h + m + l (three bf16 terms = the whole 24-bit significand, operands exact); products
//                  hh + hm + mh + hl + lh + mm (the dropped ml, lm, ll are 2^-24 relative: f32 rounding level),
//                  6 x v_mfma_f32_32x32x16_bf16.
// Accuracy against the f32 oracle at full size (scripts/split_study.py, CPU model of this arithmetic; the f32 oracle
// itself is 6e-7 from an f64 run): max|d audio| 0.9-2.1e-5 (bf16x3), 0.7-2.5e-6 (f16x3), 0.3-1.4e-6 (bf16x6); the f32
// HIP path's gate is 2e-4. Used for the coupling flow and the generator only, never for the text encoder / duration
// predictor (the integer durations are the f32 path's), and never by default.
//   * B operand (activations): staged per 32-channel chunk through registers, pre-activation and the split applied
//     ONCE per element there, written to LDS as [term][k group of 8 channels][column][8 x 16 bit]: a lane's B
//     fragment of one k-step (8 consecutive channels of its column) is one ds_read_b128, a dilated tap a shifted column.
//   * A operand (weights): split and packed at load time (engine_pack.cpp pack_matrix) as
//     [m tile][chunk][tap][term][k-step 0|1][lane][8 x 16 bit]: NT * 512 floats per step, 2 * NT 16-byte loads per
//     lane and (m tile, chunk, tap), prefetched one unit ahead (ping-pong).
//   * accumulator layout == v_mfma_f32_32x32x2_f32's: the epilogues of conv_common.h are shared.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int frag16 __attribute__((ext_vector_type(4)));     // eight 16-bit terms as the MFMA takes them
#ifdef PE_EMU
#define pe_mfma_bf16_32x32x16(a, b, c) emu_mfma_bf16_32x32x16((a), (b), (c))
#define pe_mfma_f16_32x32x16(a, b, c) emu_mfma_f16_32x32x16((a), (b), (c))
#else
#define pe_mfma_bf16_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define pe_mfma_f16_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
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

constexpr int split_terms(int sm) { return sm == 2 ? 3 : 2; }
static constexpr float F16_MAX = 65504.f;

// v (8 floats) -> the NT term vectors of split mode SM, largest term first
template <int SM>
__device__ __forceinline__ void split8(const float (&v)[8], frag16 (&t)[split_terms(SM)]) {
  if constexpr (SM == 1) {
    f16x8 hi, lo;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float c = fminf(fmaxf(v[i], -F16_MAX), F16_MAX);
      const _Float16 h = (_Float16)c;
      hi[i] = h;
      lo[i] = (_Float16)(c - (float)h);
    }
    t[0] = __builtin_bit_cast(frag16, hi);
    t[1] = __builtin_bit_cast(frag16, lo);
  } else {
    bf16x8 q[split_terms(SM)];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float r = v[i];
#pragma unroll
      for (int k = 0; k < split_terms(SM); ++k) {
        const __bf16 h = pe_f2bf(r);
        q[k][i] = h;
        r -= pe_bf2f(h);                        // exact: h holds the leading bits of r
      }
    }
#pragma unroll
    for (int k = 0; k < split_terms(SM); ++k) t[k] = __builtin_bit_cast(frag16, q[k]);
  }
}
template <int SM>
__device__ __forceinline__ f32x16 split_mfma(frag16 a, frag16 b, f32x16 c) {
  if constexpr (SM == 1) return pe_mfma_f16_32x32x16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c);
  else return pe_mfma_bf16_32x32x16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c);
}

template <int SM, int WM, int WN, int MT, int NT, bool GATE, int HALO>
__global__ __launch_bounds__(256, ((MT * NT >= 4 || SM == 2) ? 2 : 3))
void conv_split_kernel(ConvP p) {
  PE_KTRACE(12);
  constexpr int NTM = split_terms(SM);            // terms per operand
  constexpr int NF = 2 * NTM;                     // fragments per (tile, unit): [term][k-step]
  constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
  constexpr int NCOL = (BN + HALO + 63) / 64;    // staging columns per lane; (taps-1)*dilation <= HALO
  static_assert(WM * WN == 4, "4 waves per block");
  static_assert(!GATE || MT == 2, "gate epilogue pairs two M tiles");
  static_assert(KC == 32, "a chunk is four k groups of 8 channels (two k-steps of the 32x32x16 MFMA)");
  constexpr int XS = NCOL * 64;                   // LDS columns per k group
  constexpr int PART = 4 * XS;                    // fragments per term of one slab
  PE_DYN_SMEM(frag16, xs);                        // 2 x [NTM terms][4 k groups][XS] x 16 bytes
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
  constexpr int USTEP = NTM * 512;                // floats per (tile, unit)
  const int wstride_mt = nunits * USTEP;
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
    frag16* dst = xs + buf * NTM * PART + wv * XS + lane;
#pragma unroll
    for (int cc = 0; cc < NCOL; ++cc) {
      float v[8];
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const float t = xr[rr][cc];
        v[rr] = t > 0.f ? t : t * slope;
      }
      frag16 t[NTM];
      split8<SM>(v, t);
#pragma unroll
      for (int k = 0; k < NTM; ++k) dst[k * PART + 64 * cc] = t[k];
    }
  };
  // A fragments of unit u: [term][k-step] per M tile
  auto load_a = [&](int u, frag16 (&a)[MT][NF]) {
    const int off = PE_UNIFORM(u * USTEP);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int f = 0; f < NF; ++f)
        a[i][f] = __builtin_bit_cast(frag16, pe_row_load4(wsrc, off + i * wstride_mt + f * 256 + lane * 4));
  };
  auto read_b = [&](int tap, const frag16* xbuf, frag16 (&bv)[NT][NF]) {
    const frag16* xp = xbuf + lhi * XS + tap * p.dil + wn * NT * 32 + l31;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int f = 0; f < NF; ++f) bv[j][f] = xp[(f >> 1) * PART + (f & 1) * 2 * XS + j * 32];
  };
  // one term product over all tiles of the wave: A term ta x B term tb, k-step ks
  auto prod = [&](const frag16 (&a)[MT][NF], const frag16 (&bv)[NT][NF], int ta, int tb, int ks) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = split_mfma<SM>(a[i][2 * ta + ks], bv[j][2 * tb + ks], acc[i][j]);
  };
  // small terms first, k-step by k-step
  auto mma = [&](const frag16 (&a)[MT][NF], const frag16 (&bv)[NT][NF]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if constexpr (NTM == 3) {
        prod(a, bv, 1, 1, ks);
        prod(a, bv, 2, 0, ks);
        prod(a, bv, 0, 2, ks);
      }
      prod(a, bv, 1, 0, ks);
      prod(a, bv, 0, 1, ks);
      prod(a, bv, 0, 0, ks);
    }
  };

  frag16 aA[MT][NF], aB[MT][NF];
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
    const frag16* xbuf = xs + (c & 1) * NTM * PART;
    load_x(c + 1, c + 1 < nchunks);               // next slab: in flight for the whole chunk
    for (int tap = 0; tap < ntaps; tap += 2) {
      frag16 bv[NT][NF];
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
        for (int f = 0; f < NF; ++f) aA[i][f] = aB[i][f];
    }
    if (c + 1 < nchunks) {
      store_x((c + 1) & 1);
      __syncthreads();
    }
  }
  if constexpr (SM == 1) {
    // the weights were packed times a power of two (f16's exponent range): undo it, exactly
    const float us = p.wunscale[0];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] *= us;
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
