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

// leaky-relu for 0 < slope <= 1 as max(v, v * slope) in exactly two VALU instructions (the compiler's lowering of fmaxf /
// fmed3 adds a canonicalising max(v, v) in front of each; inputs here are finite products of finite values)
#ifdef PE_EMU
inline float pe_lrelu2(float v, float slope) { return v > 0.f ? v : v * slope; }
#else
__device__ __forceinline__ float pe_lrelu2(float v, float slope) {
  const float t = v * slope;
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(t));
  return r;
}
#endif

// f32 pair -> f16 pair, round toward zero (v_cvt_pkrtz_f16_f32: ONE instruction per two elements; a magnitude beyond
// f16's range lands on +-65504, the largest finite value, so no clamp is needed in front of it)
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#ifdef PE_EMU
inline _Float16 pe_f2h_rtz(float v) {
  _Float16 h = (_Float16)v;                               // round to nearest even, then step back towards zero where that rounded away
  unsigned short u;
  memcpy(&u, &h, 2);
  if ((u & 0x7fffu) == 0x7c00u && std::isfinite(v)) u = (unsigned short)((u & 0x8000u) | 0x7bffu);      // overflow -> 65504
  else if (std::fabs((float)h) > std::fabs(v)) u = (unsigned short)(u - 1);
  memcpy(&h, &u, 2);
  return h;
}
inline f16x2 pe_cvt_pkrtz(float a, float b) { return f16x2{pe_f2h_rtz(a), pe_f2h_rtz(b)}; }
#else
__device__ __forceinline__ f16x2 pe_cvt_pkrtz(float a, float b) { return __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(a, b)); }
#endif

// v (8 floats) -> the NT term vectors of split mode SM, largest term first
template <int SM>
__device__ __forceinline__ void split8(const float (&v)[8], frag16 (&t)[split_terms(SM)]) {
  if constexpr (SM == 1) {
    // hi = v rounded TOWARD ZERO to f16 (packed conversion, saturating: no clamp), lo = f16(v - hi): hi + lo carries 21-22
    // significand bits either way, and the staging costs 2.5 VALU instructions per element instead of 6
    f16x8 hi, lo;
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
      const f16x2 h = pe_cvt_pkrtz(v[i], v[i + 1]);
      hi[i] = h[0];
      hi[i + 1] = h[1];
      lo[i] = (_Float16)(v[i] - (float)h[0]);
      lo[i + 1] = (_Float16)(v[i + 1] - (float)h[1]);
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

// The epilogue of the split kernels. conv_store_tile (conv_common.h) issues FIVE memory instructions per output element --
// bias, speaker bias, previous value, residual (zero-length descriptors when absent) and the store -- which the f32 kernels
// hide behind 64-cycle MFMAs; behind a K loop that is 5x shorter it cost as much as a 3-tap K loop (SQ counters, call 7 of
// profiles/r06_notes.md: 9 VALU / memory instructions per MFMA). Here the absent streams are skipped behind kernel-uniform
// branches and the bias comes as four 16-byte loads (a lane's rows are four groups of four consecutive rows). Same
// arithmetic, same order: ((acc + (bias + bias2)) * sign + (old + res)) * alpha. Partial row tiles and the polyphase
// scatter keep the general routine.
__device__ __forceinline__ void conv_store_tile_lean(const ConvP& p, const EpiFlags& f, int b, int row0, int col, int lhi,
                                                     int L, int ncols, const f32x16& acc) {
  const bool to_skip = p.epi == EPI_WNRS && row0 >= p.split;        // uniform per tile
  const int row_end = to_skip ? p.rows : (p.epi == EPI_WNRS ? p.split : p.rows);
  if (p.epi == EPI_CONVT || row0 + 32 > row_end) {
    conv_store_tile(p, f, b, row0, col, lhi, L, ncols, acc);
    return;
  }
  constexpr int OOB = 0x3fffffff;
  int rb = row0 + 4 * lhi;
  PE_OPAQUE(rb);
  const bool rd_old = to_skip ? (p.mode != 1) : (f.use_old || p.epi == EPI_WNRS);
  const int orow0 = to_skip ? p.split : 0;
  const int orows = to_skip ? p.rows - p.split : (p.epi == EPI_WNRS ? p.split : p.rows);
  const int ocs = to_skip ? p.o2_cs : p.o_cs;
  float* ob = to_skip ? p.out2 + (long)b * p.o2_bs : p.out + (long)b * p.o_bs;
  const pe_rowsrc od = pe_make_row_u(ob, orows * ocs);
  const bool cok = col < ncols;
  const int ooff = cok ? (rb - orow0) * ocs + col : OOB;
  f32x4 bz[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bz[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (p.bias) {
    const pe_rowsrc bd = pe_make_row_u(p.bias, p.rows);
#pragma unroll
    for (int g = 0; g < 4; ++g) bz[g] = pe_row_load4(bd, rb + 8 * g);
  }
  if (p.bias2) {
    const pe_rowsrc b2d = pe_make_row_u(p.bias2 + (long)b * p.bias2_bs, p.rows);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 t = pe_row_load4(b2d, rb + 8 * g);
#pragma unroll
      for (int j = 0; j < 4; ++j) bz[g][j] += t[j];
    }
  }
  float o[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
  if (rd_old) {
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = pe_row_load_so(od, ooff, ((r & 3) + 8 * (r >> 2)) * ocs);
  }
  if (f.use_res) {
    const pe_rowsrc rd = pe_make_row_u(p.res + (long)b * p.r_bs, p.rows * p.r_cs);
    const int roff = cok ? rb * p.r_cs + col : OOB;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] += pe_row_load_so(rd, roff, ((r & 3) + 8 * (r >> 2)) * p.r_cs);
  }
  if (f.sign == 1.f && f.alpha == 1.f && !f.relu) {          // (kernel-uniform; x * 1 is exact, so this is the same arithmetic)
#pragma unroll
    for (int r = 0; r < 16; ++r) pe_row_store_so(od, ooff, ((r & 3) + 8 * (r >> 2)) * ocs, (acc[r] + bz[r >> 2][r & 3]) + o[r]);
    return;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float v = ((acc[r] + bz[r >> 2][r & 3]) * f.sign + o[r]) * f.alpha;
    if (f.relu) v = v > 0.f ? v : 0.f;
    pe_row_store_so(od, ooff, ((r & 3) + 8 * (r >> 2)) * ocs, v);
  }
}
// tanh(a) * sigmoid(s) (commons.py:99-106) on the hardware's exp2 / reciprocal units: tanh(a) = 1 - 2 / (1 + e^(2a)),
// sigmoid(s) = 1 / (1 + e^(-s)); ~1e-7 absolute against tanhf / expf (the split modes' own error is 1e-6). The library forms
// are ~60 VALU instructions per element -- 40 % of the split gate conv's time; the f32 gate kernels keep them.
__device__ __forceinline__ float split_gate(float ta, float sa) {
#ifdef PE_EMU
  const float e1 = exp2f(2.8853900817779268f * ta), e2 = exp2f(-1.4426950408889634f * sa);
  return (1.f - 2.f / (1.f + e1)) * (1.f / (1.f + e2));
#else
  const float e1 = __builtin_amdgcn_exp2f(2.8853900817779268f * ta), e2 = __builtin_amdgcn_exp2f(-1.4426950408889634f * sa);
  return (1.f - 2.f * __builtin_amdgcn_rcpf(1.f + e1)) * __builtin_amdgcn_rcpf(1.f + e2);
#endif
}
__device__ __forceinline__ void conv_store_gate_split(const ConvP& p, int b, int ch, int col, float ta, float sa) {
  ta += p.bias[ch];
  sa += p.bias[p.split + ch];
  if (p.bias2) {
    const float* b2 = p.bias2 + (long)b * p.bias2_bs;
    ta += b2[ch];
    sa += b2[p.split + ch];
  }
  p.out[(long)b * p.o_bs + (long)ch * p.o_cs + col] = split_gate(ta, sa);
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
      for (int rr = 0; rr < 8; ++rr) v[rr] = pe_lrelu2(xr[rr][cc], slope);         // (slope 1 = no activation: max(t, t))
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
        if (ch < p.split && col < ncols) conv_store_gate_split(p, b, ch, col, acc[0][j][r], acc[MT - 1][j][r]);
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        conv_store_tile_lean(p, ef, b, (mtile0 + i) * 32, n0 + (wn * NT + j) * 32 + l31, lhi, L, ncols, acc[i][j]);
  }
}

}  // namespace pe
