// mrf_split_kernel: the fused MRF stage of mrf.h on the 16-bit matrix pipe with split f32 operands (matrix modes bf16x3 and
// f16x3 of conv_bf3.h; opt-in, PIPER_HIP_MATRIX).
// (gfx950 / CDNA4 device code; reference arithmetic: models.py:356-363, modules.py:301-314 / :355-364, paths relative to
// /root/reference/src/python/piper_train/vits/.)
//
// Same plan as mrf_kernel -- a workgroup owns N output columns of one utterance and runs every conv of every resblock of
// the stage out of LDS on shrinking windows; a wave owns the same 16x16 output units in every phase, so the raw running x
// of a resblock chain and the MRF sum stay in its f32 registers; weights are one flat stream in execution order read
// straight from L2; the only workgroup barriers are the phase boundaries -- with these differences:
//   * the matrix instruction is v_mfma_f32_16x16x32_{bf16,f16}: ONE instruction per term product covers a whole
//     (32-channel chunk, tap) step of a 16x16 unit (the f32 kernel issues eight 16x16x4 MFMAs of twice the cycles). A step
//     is three products (hi*hi, hi*lo, lo*hi; small terms first) = 48 matrix cycles against 256.
//   * LDS holds the ACTIVATED tensors already SPLIT, as [term][k group of 8 channels][column][8 x 16 bit]: a lane's B
//     fragment of a step (8 consecutive channels of its column) is one ds_read_b128 per term, a dilated tap a shifted
//     column. Two 16-bit terms are the 4 bytes per element of the f32 layout: the same windows fit the 160 KB.
//   * the producer splits once per element: stage_x (8 channels of a column per thread, one 16-byte store per term) and the
//     phase epilogue (a lane's 4 consecutive rows of a column = half a k group: one 8-byte store per term).
//   * the weight stream is [phase][step][16-row tile][term][lane][8 x 16 bit] (the bytes of the f32 stream); in mode f16x3
//     every conv's weights are packed times a power of two that the epilogue undoes exactly (p.wunscale[phase]).
// Residuals, biases, the MRF sum, the mean and the fused generator tail (conv_post + tanh + peak on the mean, out of LDS)
// are f32 exactly as in mrf_kernel. Parity: the f32 path's own gate (tests/test_gpu_batched.py).
#pragma once
#include "conv_bf3.h"
#include "mrf.h"

namespace pe {

#ifdef PE_EMU
#define pe_mfma_bf16_16x16x32(a, b, c) emu_mfma_bf16_16x16x32((a), (b), (c))
#define pe_mfma_f16_16x16x32(a, b, c) emu_mfma_f16_16x16x32((a), (b), (c))
#else
#define pe_mfma_bf16_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#define pe_mfma_f16_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
#endif

typedef unsigned int frag8 __attribute__((ext_vector_type(2)));      // four 16-bit terms
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template <int SM>
__device__ __forceinline__ f32x4 split_mfma16(frag16 a, frag16 b, f32x4 c) {
  if constexpr (SM == 1) return pe_mfma_f16_16x16x32(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c);
  else return pe_mfma_bf16_16x16x32(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c);
}
// v (4 floats) -> hi / lo term vectors of the two-term modes
template <int SM>
__device__ __forceinline__ void split4(const float (&v)[4], frag8& hi, frag8& lo) {
  if constexpr (SM == 1) {
    f16x4 h, l;
#pragma unroll
    for (int i = 0; i < 4; i += 2) {          // hi rounded toward zero (packed, saturating), lo = f16(v - hi): conv_bf3.h split8
      const f16x2 t = pe_cvt_pkrtz(v[i], v[i + 1]);
      h[i] = t[0];
      h[i + 1] = t[1];
      l[i] = (_Float16)(v[i] - (float)t[0]);
      l[i + 1] = (_Float16)(v[i + 1] - (float)t[1]);
    }
    hi = __builtin_bit_cast(frag8, h);
    lo = __builtin_bit_cast(frag8, l);
  } else {
    bf16x4 h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const __bf16 t = pe_f2bf(v[i]);
      h[i] = t;
      l[i] = pe_f2bf(v[i] - pe_bf2f(t));
    }
    hi = __builtin_bit_cast(frag8, h);
    lo = __builtin_bit_cast(frag8, l);
  }
}

template <int SM, int CP, int OU, int HU>
__global__ __launch_bounds__(64 * MRF_NW) void mrf_split_kernel(MrfP p) {
  PE_KTRACE(22);
  static_assert(SM == 0 || SM == 1, "two-term split modes");
  constexpr int NW = MRF_NW, WS = mrf_ws(CP, OU), MS = CP / 16, MSW = 2, NRG = MS / MSW, NCG = NW / NRG, NT = 64 * NW;
  constexpr int UPW = OU + HU, NCH = CP / KC, NTM = 2, QN = CP / 8;
  constexpr int STEPF = MS * NTM * 256;            // floats of the weight stream per step
  constexpr int TS = QN * WS;                      // fragments per term of one activation buffer
  constexpr int bufsz = NTM * TS;                  // fragments per buffer: 4 * CP * WS bytes, the f32 kernel's CP x WS floats
  static_assert(MS % MSW == 0 && NW % NRG == 0, "waves split evenly over the row groups");
  PE_DYN_SMEM(float, sm);
  const int b = blockIdx.y;
  const int L = p.lens[b] * p.len_mul;
  if (blockIdx.x * p.stride >= L) return;
  const int n0 = blockIdx.x * p.stride - p.n0off;
  // LDS: [pad][buffer 0][buffer 1][pad][phase table], the f32 kernel's map. Reads of never-used columns may leave a row
  // on the left / right: they land in the pads / a neighbouring row (finite 16-bit patterns feeding unused accumulators).
  frag16* bufs = reinterpret_cast<frag16*>(sm + MRF_PAD);
  int* tph = reinterpret_cast<int*>(sm + MRF_PAD + 2 * CP * WS + MRF_PAD);
  const int tid = threadIdx.x, lane = tid & 63, wv = PE_UNIFORM(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int ms0 = (wv % NRG) * MSW, cg = wv / NRG;
  const int g0 = n0 - p.hxa;                       // global column of window column 0
  const float slope = p.slope;
  const int C = p.C;
  const int wcols = p.wcols;

  // ---- weight stream: this wave's A fragments of a step = MSW tiles x NTM terms, one 16-byte load each
  const pe_rowsrc wd = pe_make_row(p.wstream, p.wfloats);
  const int wlane = ms0 * NTM * 256 + lane * 4;
  const int wlast = PE_UNIFORM(p.wfloats - STEPF);
  auto load_a = [&](int woff, frag16 (&a)[MSW][NTM]) {
    woff = woff < wlast ? woff : wlast;
#pragma unroll
    for (int m = 0; m < MSW; ++m)
#pragma unroll
      for (int t = 0; t < NTM; ++t) a[m][t] = __builtin_bit_cast(frag16, pe_row_load4_so(wd, wlane + (m * NTM + t) * 256, woff));
  };
  frag16 aA[MSW][NTM], aB[MSW][NTM];
  int wnext = 0;
  load_a(wnext, aA);
  wnext += STEPF;
  {
    const int* gp = reinterpret_cast<const int*>(p.phases);
    for (int i = tid; i < p.nphases * 12; i += NT) tph[i] = gp[i];
    if (tid < MRF_PAD) { sm[tid] = 0.f; sm[MRF_PAD + 2 * CP * WS + tid] = 0.f; }
  }

  // ---- stage the activated, split input window: buffer 0 <- split(lrelu(x[g0 + c])), zero outside [0, L) and for rows >= C.
  // Item = (k group q, 64-column block j): a thread holds the 8 channels of its column = one fragment per term.
  const float* xb = p.x + (long)b * p.x_bs;
  auto stage_x = [&]() {
    constexpr int NCC = (WS + 63) / 64, NIT = QN * NCC, IPW = (NIT + NW - 1) / NW;
    float v[IPW][8];
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
      const int it = wv + NW * i, q = it % QN, j = it / QN;
      const int c = lane + 64 * j;
      const bool live = it < NIT && c < wcols;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int row = 8 * q + r;
        const pe_rowsrc rd = pe_make_row(xb + (long)row * p.x_cs, (it < NIT && row < C) ? L : 0);
        v[i][r] = pe_row_load(rd, live ? g0 + c : -1);
      }
    }
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
      const int it = wv + NW * i, q = it % QN, j = it / QN;
      const int c = lane + 64 * j;
      if (it < NIT && c < WS) {
        float a[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) a[r] = pe_lrelu2(v[i][r], slope);
        frag16 t[NTM];
        split8<SM>(a, t);
#pragma unroll
        for (int k = 0; k < NTM; ++k) bufs[k * TS + q * WS + c] = t[k];
      }
    }
  };

  // ---- this wave's units: window column block cu[u] (-1: the wave has no such unit) x row tiles ms0 + m
  f32x4 acc[MSW][UPW], rawc[MSW][UPW], tot[MSW][OU];
  int cu[UPW];
  {
    const int cuo0 = p.hxa / 16, nout = p.N / 16;
#pragma unroll
    for (int u = 0; u < OU; ++u) cu[u] = cuo0 + cg + NCG * u;
#pragma unroll
    for (int v = 0; v < HU; ++v) {
      const int h = cg + NCG * v;
      cu[OU + v] = h >= p.nhalo ? -1 : (h < p.nleft ? p.cu_lo + h : cuo0 + nout + (h - p.nleft));
    }
#pragma unroll
    for (int m = 0; m < MSW; ++m) {
#pragma unroll
      for (int u = 0; u < UPW; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) rawc[m][u][r] = 0.f;
#pragma unroll
      for (int u = 0; u < OU; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) tot[m][u][r] = 0.f;
    }
  }
  const pe_rowsrc xd = pe_make_row(xb, C * p.x_cs);
  stage_x();

  for (int ph = 0; ph < p.nphases; ++ph) {
    __syncthreads();            // table + window (first phase) / the previous phase's activations are in LDS
    MrfPhase P;
    {
      const int* t = tph + ph * 12;
      const unsigned lo = PE_UNIFORM((unsigned)t[0]), hi = PE_UNIFORM((unsigned)t[1]);
      P.bias = reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
      P.ntaps = PE_UNIFORM(t[2]); P.dil = PE_UNIFORM(t[3]); P.e = PE_UNIFORM(t[4]);
      P.src = PE_UNIFORM(t[5]); P.dst = PE_UNIFORM(t[6]); P.flags = PE_UNIFORM(t[7]);
    }
    if (P.flags & MRF_RESTAGE) {       // ResBlock1 rewrites buffer 0 in place: a new chain starts from the stage input
      stage_x();
      __syncthreads();
    }
    if (P.flags & MRF_INIT) {          // running x of the chain <- raw stage input of the owned units (f32, L2-hot)
#pragma unroll
      for (int u = 0; u < UPW; ++u) {
        const int g = g0 + 16 * cu[u] + l15;
        int voff = (cu[u] >= 0 && g >= 0 && g < L) ? 4 * lq * p.x_cs + g : 0x3fffffff;
        PE_OPAQUE(voff);
#pragma unroll
        for (int m = 0; m < MSW; ++m) {
          const int vm = (C == CP || (ms0 + m) * 16 + 4 * lq < C) ? voff : 0x3fffffff;
#pragma unroll
          for (int r = 0; r < 4; ++r) rawc[m][u][r] = pe_row_load_so(xd, vm, ((ms0 + m) * 16 + r) * p.x_cs);
        }
      }
    }
    const pe_rowsrc bd = pe_make_row(P.bias, P.bias ? C : 0);
    float bz[MSW][4];
#pragma unroll
    for (int m = 0; m < MSW; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) bz[m][r] = pe_row_load(bd, (ms0 + m) * 16 + 4 * lq + r);
    // mode f16x3: this conv's weights were packed times 1 / unscale (a power of two); bf16x3: 1
    const float unscale = SM == 1 ? p.wunscale[ph] : 1.f;
    const int hh = P.dil * (P.ntaps - 1) / 2;
    const int wlo = p.hxa - P.e, whi = p.hxa + p.N + P.e;          // columns this phase must produce
    bool act[UPW];
#pragma unroll
    for (int u = 0; u < UPW; ++u) {
      act[u] = cu[u] >= 0 && 16 * cu[u] < whi && 16 * cu[u] + 16 > wlo;
#pragma unroll
      for (int m = 0; m < MSW; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[m][u][r] = 0.f;
    }
    const frag16* src = bufs + P.src * bufsz;
    // The K loop, specialised at compile time on WHICH halo units take part in this phase (as in mrf_kernel).
    auto k_loop = [&](auto maskc) {
      constexpr int MASK = decltype(maskc)::value;
      const int nsteps = NCH * P.ntaps;
      // per-unit LDS base of (k group lq, tap 0): the step adds chunk * 4 * WS + tap * dil
      const frag16* ub[UPW];
#pragma unroll
      for (int u = 0; u < UPW; ++u) ub[u] = src + lq * WS + l15 + 16 * (cu[u] < 0 ? 0 : cu[u]) - hh;
      auto read_b = [&](int soff, frag16 (&bv)[UPW][NTM]) {
#pragma unroll
        for (int u = 0; u < UPW; ++u)
          if (u < OU || ((MASK >> (u - OU)) & 1)) {
            const frag16* bp = ub[u] + soff;
#pragma unroll
            for (int t = 0; t < NTM; ++t) bv[u][t] = bp[t * TS];
          }
      };
      auto mma = [&](const frag16 (&a)[MSW][NTM], const frag16 (&bv)[UPW][NTM]) {
        // small terms first; unit-interleaved: consecutive MFMAs hit different accumulators
#pragma unroll
        for (int pr = 0; pr < 3; ++pr) {
          const int ta = pr == 0 ? 1 : 0, tb = pr == 1 ? 1 : 0;          // lo*hi, hi*lo, hi*hi
#pragma unroll
          for (int u = 0; u < UPW; ++u)
            if (u < OU || ((MASK >> (u - OU)) & 1)) {
#pragma unroll
              for (int m = 0; m < MSW; ++m) acc[m][u] = split_mfma16<SM>(a[m][ta], bv[u][tb], acc[m][u]);
            }
        }
      };
      int ntap = 0, nchunk = 0;
      auto advance = [&]() -> int {
        if (++ntap == P.ntaps) { ntap = 0; if (++nchunk == NCH) nchunk = 0; }
        return PE_UNIFORM(nchunk * 4 * WS + ntap * P.dil);
      };
      constexpr int NU = OU + ((MASK & 1) ? 1 : 0) + ((MASK & 2) ? 1 : 0);       // units taking part
      constexpr int NMF = 3 * MSW * NU, NVM = NTM * MSW, NDS = NTM * NU;         // MFMAs / weight fetches / ds_read_b128 per step
      constexpr int KI = NMF / (NVM + NDS + 1) > 0 ? NMF / (NVM + NDS + 1) : 1;
      frag16 bA[UPW][NTM], bB[UPW][NTM];
      read_b(0, bA);
      int st = 0;
      for (; st + 1 < nsteps; st += 2) {
        PE_SCHED_FENCE();
        load_a(wnext, aB);
        read_b(advance(), bB);
        wnext += STEPF;
        mma(aA, bA);
        mrf_interleave<NVM, NDS, KI>();
        PE_SCHED_FENCE();
        load_a(wnext, aA);      // behind the phase's last step: the first step of the next phase
        read_b(advance(), bA);
        wnext += STEPF;
        mma(aB, bB);
        mrf_interleave<NVM, NDS, KI>();
        PE_SCHED_FENCE();
      }
      if (st < nsteps) {        // odd step count: the last step, and the next phase's first fragments move to aA
        load_a(wnext, aB);
        wnext += STEPF;
        mma(aA, bA);
        mrf_interleave<NVM, 0, KI>();
        PE_SCHED_FENCE();
#pragma unroll
        for (int m = 0; m < MSW; ++m)
#pragma unroll
          for (int t = 0; t < NTM; ++t) aA[m][t] = aB[m][t];
      }
    };
    {
      int mask = 0;
#pragma unroll
      for (int v = 0; v < HU; ++v) mask |= act[OU + v] ? (1 << v) : 0;
      mask = PE_UNIFORM(mask);
      if (HU == 1) {
        if (mask) k_loop(pe_int<1>{}); else k_loop(pe_int<0>{});
      } else {
        switch (mask) {
          case 0: k_loop(pe_int<0>{}); break;
          case 1: k_loop(pe_int<1>{}); break;
          case 2: k_loop(pe_int<2>{}); break;
          default: k_loop(pe_int<3>{}); break;
        }
      }
    }
    // ---- epilogue of the phase: f32 bias / residual / sum; the next conv's operand leaves split, half a k group per lane
    frag16* dstb = bufs + (P.dst < 0 ? 0 : P.dst) * bufsz;
#pragma unroll
    for (int u = 0; u < UPW; ++u)
      if (act[u]) {
        const int col = 16 * cu[u] + l15;
        const int g = g0 + col;
        const bool inside = g >= 0 && g < L;           // intermediates only exist on [0, L): zero padding
#pragma unroll
        for (int m = 0; m < MSW; ++m) {
          f32x4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float t = acc[m][u][r] * unscale + bz[m][r];
            if (P.flags & MRF_RES) t += rawc[m][u][r];
            v[r] = inside ? t : 0.f;
          }
          if (P.flags & MRF_KEEP) rawc[m][u] = v;
          if (u < OU && (P.flags & MRF_FINAL)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) tot[m][u < OU ? u : 0][r] += v[r];
          }
          if (P.dst >= 0) {
            // rows (ms0 + m) * 16 + 4 lq + r: k group 2 (ms0 + m) + (lq >> 1), its lower / upper four channels
            float a4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) a4[r] = pe_lrelu2(v[r], slope);
            frag8 hi, lo;
            split4<SM>(a4, hi, lo);
            frag8* d8 = reinterpret_cast<frag8*>(dstb + (2 * (ms0 + m) + (lq >> 1)) * WS + col) + (lq & 1);
            d8[0] = hi;
            d8[2 * TS] = lo;                           // (the lo term's plane: TS fragments = 2 * TS half-fragments on)
          }
        }
      }
  }
  if (p.post_w == nullptr) {
    // ---- MRF mean of the owned output units
    float* ob = p.out + (long)b * p.o_bs;
#pragma unroll
    for (int u = 0; u < OU; ++u) {
      const int g = g0 + 16 * cu[u] + l15;
      if (g >= L) continue;
#pragma unroll
      for (int m = 0; m < MSW; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = (ms0 + m) * 16 + 4 * lq + r;
          if (row < C) ob[(long)row * p.o_cs + g] = tot[m][u][r] * p.alpha;
        }
    }
    return;
  }
  // ---- last stage with the generator tail fused: f32, as in mrf_kernel (the activation buffers are re-used as plain f32
  // rows: buffer 0 <- lrelu(mean, 0.01) [CP][WS], buffer 1 <- [POST_CG][N] partial sums)
  __syncthreads();
  float* mb = sm + MRF_PAD;
  float* part = mb + CP * WS;
#pragma unroll
  for (int u = 0; u < OU; ++u) {
    const int col = 16 * cu[u] + l15;
#pragma unroll
    for (int m = 0; m < MSW; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) mb[((ms0 + m) * 16 + 4 * lq + r) * WS + col] = pe_lrelu(tot[m][u][r] * p.alpha, p.post_slope);
  }
  __syncthreads();
  static_assert(POST_CG * POST_CU == 32 && POST_OPT == 4 && POST_K == 7, "tail mapping: 4 channel groups x 128 threads x 4 samples");
  const int nsamp = p.N - (POST_K - 1);
  {
    const int pg = PE_UNIFORM(tid >> 7), ct = tid & 127;
    if (4 * ct < nsamp) {
      float a4[POST_OPT];
#pragma unroll
      for (int o = 0; o < POST_OPT; ++o) a4[o] = 0.f;
#pragma unroll
      for (int cc = 0; cc < POST_CU; ++cc) {
        const int c = pg * POST_CU + cc;
        if (c < C) {
          const float* rp = mb + c * WS + p.hxa + 4 * ct;
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(rp), v1 = *reinterpret_cast<const f32x4*>(rp + 4);
          const float v[POST_OPT + POST_K - 1] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3], rp[8], rp[9]};
          const float* wc = p.post_w + c * POST_K;
#pragma unroll
          for (int k = 0; k < POST_K; ++k) {
            const float wk = wc[k];
#pragma unroll
            for (int o = 0; o < POST_OPT; ++o) a4[o] = fmaf(wk, v[o + k], a4[o]);
          }
        }
      }
#pragma unroll
      for (int o = 0; o < POST_OPT; ++o) part[pg * p.N + 4 * ct + o] = a4[o];
    }
  }
  __syncthreads();
  float pk = 0.f;
  if (tid < nsamp) {
    float sum = 0.f;
#pragma unroll
    for (int g = 0; g < POST_CG; ++g) sum += part[g * p.N + tid];
    const float y = tanhf(sum);
    const int n = n0 + (POST_K - 1) / 2 + tid;
    if (n < L) {
      p.audio[(long)b * p.a_bs + n] = y;
      pk = fabsf(y);
    }
  }
  for (int o = 32; o >= 1; o >>= 1) pk = fmaxf(pk, __shfl_xor(pk, o));
  float* wmax = part + POST_CG * p.N;
  if (lane == 0) wmax[wv] = pk;
  __syncthreads();
  if (tid == 0) {
    float m = wmax[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) m = fmaxf(m, wmax[w]);
    atomicMax(p.absmax + b, __float_as_uint(m));
  }
}

}  // namespace pe
