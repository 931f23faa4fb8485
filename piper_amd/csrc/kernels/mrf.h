// One whole MRF stage of the HiFiGAN generator per launch (models.py:356-363: xs = sum_j resblock_j(x) / n;
// modules.py:301-314 ResBlock1, :355-364 ResBlock2) for stages with <= 64 channels.
//
// A workgroup owns N output columns of one utterance and runs every conv of every resblock of the stage out of LDS
// (halo recompute on shrinking windows), so the stage costs ONE launch, one read of x and one write of the MRF mean
// instead of one launch and ~3 tensor round trips per conv:
//   * LDS holds the ACTIVATED tensors (leaky-relu applied once, by the producer): the MFMA B operand is a plain ds_read,
//     no VALU on the read side. Residuals never come from LDS: a wave owns the same 16x16 output units
//     (v_mfma_f32_16x16x4_f32) in every phase (static map unit -> wave), so the raw running x of a resblock chain and the
//     MRF sum stay in its registers; the raw stage input of a chain is re-read from L2 when the chain starts.
//   * 8 waves of <= 256 registers: a wave owns two 16-row tiles x (OU + HU) column units (one B operand feeds two MFMAs),
//     up to 80 MFMAs per (chunk, tap) step from 4 + 40 operand registers.
//   * Weights never touch LDS: they are one flat stream in execution order ([phase][step][16-row tile][2][lane][4], step =
//     one (32-channel chunk, tap)) and every wave reads its own A fragments of the NEXT step straight from L2 (a few
//     hundred KB shared by all workgroups; the waves of a row group hit the same lines in L1) into a second register set
//     while the current step's MFMAs issue. The only workgroup barriers are the phase boundaries.
//   * B operands (LDS) are software-pipelined the same way, and the loads of step s + 1 are INTERLEAVED between the MFMAs
//     of step s (sched_group_barrier): the two waves of a SIMD run in lockstep, so a block of loads in front of the MFMA
//     burst would be a bubble in both at once. All prefetches sit at unconditional positions of the 2x unrolled ping-pong
//     step loop (exact wait counts); the fragments of the next PHASE's first step are fetched during the last step.
//   * The K loop is specialised at compile time on which halo units take part in a phase; (chunk, tap) are counters.
//   * The row stride WS is a per-width constant (immediates for the 8 k-rows of a step); the number of output units per
//     wave (OU = 1..3 -> N = 16 * NCG * OU columns per workgroup) is picked per launch (engine_launch.cpp: Engine::mrf).
// Every step is the same k-ordered f32 fmaf chain as the conv kernels (chunk-major, tap-minor, ascending channel).
// History (profiles/r02_notes.md, r03_notes.md): generation 1 applied the leaky-relu on the read side (3 VALU per MFMA);
// generation 2 staged the weights through a double-buffered LDS ring with one workgroup barrier per segment (16 waves,
// matrix pipe 61 % busy); this is generation 3 (70 %).
#pragma once
#include "../pe_rt.h"
#include "params.h"
#include "conv_common.h"

namespace pe {

template <int V> struct pe_int { static constexpr int value = V; };



// One step's schedule: the NEXT step's loads (NVM weight fetches, then NDS LDS reads) spread between this step's NMF
// MFMAs, K MFMAs per load. The two waves of a SIMD run in lockstep (same work, fair pipe arbitration), so a block of
// loads in front of the MFMA burst is a bubble in BOTH at the same time; a load issued while the wave waits for the
// matrix pipe anyway costs nothing.
template <int NVM, int NDS, int K>
__device__ __forceinline__ void mrf_interleave() {
  if constexpr (NVM + NDS > 0) {
    PE_SCHED_GROUP(0x8, K);
    if constexpr (NVM > 0) {
      PE_SCHED_GROUP(0x20, 1);
      mrf_interleave<NVM - 1, NDS, K>();
    } else {
      PE_SCHED_GROUP(0x100, 1);
      mrf_interleave<0, NDS - 1, K>();
    }
  }
}


// which instantiations buffer the B operand in half steps (see the K loop): the five-unit ones
#ifndef PE_MRF_HALF
#define PE_MRF_HALF 1          // 0: none, 2: all (A/B builds)
#endif
template <int CP, int OU, int HU> struct MRF_HALF_B { static constexpr bool value = PE_MRF_HALF == 2 || (PE_MRF_HALF == 1 && OU + HU >= 5); };

template <int CP, int OU, int HU>
__global__ __launch_bounds__(64 * MRF_NW) void mrf_kernel(MrfP p) {
  PE_KTRACE(20);
  constexpr int NW = MRF_NW, WS = mrf_ws(CP, OU), MS = CP / 16, MSW = 2, NRG = MS / MSW, NCG = NW / NRG, NT = 64 * NW;
  constexpr int UPW = OU + HU, NCH = CP / KC, STEPF = MS * 512;
  static_assert(MS % MSW == 0 && NW % NRG == 0, "waves split evenly over the row groups");
  static_assert(WS % 32 == 16, "row stride == 16 (mod 32): the two k rows of a half-wave hit disjoint banks");
  PE_DYN_SMEM(float, sm);
  const int b = blockIdx.y;
  const int L = p.lens[b] * p.len_mul;
  if (blockIdx.x * p.stride >= L) return;
  const int n0 = blockIdx.x * p.stride - p.n0off;     // first column of the window's N output columns (fused tail: -3 + ...)
  // LDS: [pad][buffer 0: CP x WS][buffer 1][pad][phase table]. Reads of never-used columns may leave a buffer on the
  // left / right: they land in the pads / the neighbouring buffer.
  float* bufs = sm + MRF_PAD;
  constexpr int bufsz = CP * WS;
  int* tph = reinterpret_cast<int*>(bufs + 2 * bufsz + MRF_PAD);       // [MAXPH][12] ints (MrfPhase is 12 ints wide)
  const int tid = threadIdx.x, lane = tid & 63, wv = PE_UNIFORM(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int ms0 = (wv % NRG) * MSW, cg = wv / NRG;
  const int g0 = n0 - p.hxa;                       // global column of window column 0
  const float slope = p.slope;
  const int C = p.C;
  const int wcols = p.wcols;

  // ---- weight stream: this wave's A fragments of a step = MSW tiles x 2 float4 per lane
  const pe_rowsrc wd = pe_make_row(p.wstream, p.wfloats);
  const int wlane = ms0 * 512 + lane * 4;
  // (the fetch behind the stream's last step re-reads that step -- its values are never used; the stream offset rides in the
  // SGPR offset, which the hardware does not range-check, so it must not run past the stream)
  const int wlast = PE_UNIFORM(p.wfloats - STEPF);
  auto load_a = [&](int woff, f32x4 (&a)[MSW][2]) {
    woff = woff < wlast ? woff : wlast;
#pragma unroll
    for (int m = 0; m < MSW; ++m)
#pragma unroll
      for (int q = 0; q < 2; ++q) a[m][q] = pe_row_load4_so(wd, wlane + (m * 2 + q) * 256, woff);
  };
  f32x4 aA[MSW][2], aB[MSW][2];
  int wnext = 0;                                   // float offset of the next step to fetch (the stream is in execution order)
  load_a(wnext, aA);
  wnext += STEPF;
  {
    const int* gp = reinterpret_cast<const int*>(p.phases);
    for (int i = tid; i < p.nphases * 12; i += NT) tph[i] = gp[i];
    if (tid < MRF_PAD) { sm[tid] = 0.f; bufs[2 * bufsz + tid] = 0.f; }
  }

  // ---- stage the activated input window: buffer 0 <- lrelu(x[g0 + c]), zero outside [0, L) and for rows >= C
  const float* xb = p.x + (long)b * p.x_bs;
  auto stage_x = [&]() {        // every load of the window in flight before the first store: one memory latency
    constexpr int NCC = (WS + 63) / 64, RPW = CP / NW;
    float v[RPW][NCC];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int row = wv + NW * i;
      const pe_rowsrc rd = pe_make_row(xb + (long)row * p.x_cs, row < C ? L : 0);
#pragma unroll
      for (int j = 0; j < NCC; ++j) v[i][j] = pe_row_load(rd, (lane + 64 * j < wcols) ? g0 + lane + 64 * j : -1);
    }
#pragma unroll
    for (int i = 0; i < RPW; ++i)
#pragma unroll
      for (int j = 0; j < NCC; ++j) {
        const int c = lane + 64 * j;
        if (c < WS) bufs[(wv + NW * i) * WS + c] = pe_lrelu(v[i][j], slope);
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
  constexpr int SK = CP == 64 ? 6 : 7;     // tuning build: stamps of the 64- / 32-channel stage (entry, window staged, then per phase: K loop starts, K loop done, epilogue done)
  PE_STAMP(SK, 0);
  stage_x();
  PE_STAMP(SK, 1);

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
    if (P.flags & MRF_INIT) {          // running x of the chain <- raw stage input of the owned units (L2-hot; used in the
                                        // epilogue, so the loads fly under the K loop). One lane offset per unit + an SGPR
                                        // row offset: no per-element address registers
#pragma unroll
      for (int u = 0; u < UPW; ++u) {
        const int g = g0 + 16 * cu[u] + l15;
        int voff = (cu[u] >= 0 && g >= 0 && g < L) ? 4 * lq * p.x_cs + g : 0x3fffffff;
        PE_OPAQUE(voff);
#pragma unroll
        for (int m = 0; m < MSW; ++m) {
          // rows (ms0 + m) * 16 + 4 lq + r; C is a multiple of 4 (launcher), so a lane's four rows exist together. The row
          // offset is an SGPR offset, outside the hardware's range check: lanes whose rows are >= C are poisoned here
          // (only stages narrower than the padded width CP have such lanes)
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
    const float* src = bufs + P.src * bufsz;
    if (ph < 7) PE_STAMP(SK, 2 + 3 * ph);
    // The K loop, specialised at compile time on WHICH halo units take part in this phase (bit v of MASK = halo unit
    // v; output units always do): the hot loop is straight-line code, the choice is one wave-uniform switch per phase.
    // Every variant issues the same weight fetches.
    auto k_loop = [&](auto maskc) {
      constexpr int MASK = decltype(maskc)::value;
      const int nsteps = NCH * P.ntaps;
      // per-unit LDS base of (row lq, tap 0): the step adds chunk * KC * WS + tap * dil
      const float* ub[UPW];
#pragma unroll
      for (int u = 0; u < UPW; ++u) ub[u] = src + lq * WS + l15 + 16 * (cu[u] < 0 ? 0 : cu[u]) - hh;
      auto read_b = [&](int soff, float (&bv)[UPW][8]) {
#pragma unroll
        for (int u = 0; u < UPW; ++u)
          if (u < OU || ((MASK >> (u - OU)) & 1)) {
            const float* bp = ub[u] + soff;
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) bv[u][s8] = bp[4 * s8 * WS];
          }
      };
      auto mma = [&](const f32x4 (&a)[MSW][2], const float (&bv)[UPW][8]) {
        // unit-interleaved: consecutive MFMAs hit different accumulators (no dependent-issue stall)
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8)
#pragma unroll
          for (int u = 0; u < UPW; ++u)
            if (u < OU || ((MASK >> (u - OU)) & 1)) {
#pragma unroll
              for (int m = 0; m < MSW; ++m) acc[m][u] = pe_mfma_16x16x4(a[m][s8 >> 2][s8 & 3], bv[u][s8], acc[m][u]);
            }
      };
      // (chunk, tap) of the NEXT step as an LDS offset; wraps to step 0 behind the last one (a harmless re-read)
      int ntap = 0, nchunk = 0;
      auto advance = [&]() -> int {
        if (++ntap == P.ntaps) { ntap = 0; if (++nchunk == NCH) nchunk = 0; }
        return PE_UNIFORM(nchunk * KC * WS + ntap * P.dil);
      };
      constexpr int NU = OU + ((MASK & 1) ? 1 : 0) + ((MASK & 2) ? 1 : 0);       // units taking part
      constexpr int NMF = 8 * MSW * NU, NVM = 2 * MSW, NDS = 4 * NU;             // MFMAs / weight fetches / ds_read2 per step
      if constexpr (MRF_HALF_B<CP, OU, HU>::value) {
        // B operands in HALF steps (four of a step's eight k-groups = the A fragment's q): two buffers of UPW x 4 registers
        // instead of UPW x 8 -- 40 registers fewer with five units per wave, which is what takes the five-unit instantiations
        // (<64,3,2>, <32,4,1>) from 23 / 12 spilled VGPRs to none: their spills were reloaded in every phase epilogue and in
        // the prologue (profiles/r04_notes.md, call 48). Same MFMA order per accumulator.
        auto read_bh = [&](int soff, int h, float (&bv)[UPW][4]) {
#pragma unroll
          for (int u = 0; u < UPW; ++u)
            if (u < OU || ((MASK >> (u - OU)) & 1)) {
              const float* bp = ub[u] + soff;
#pragma unroll
              for (int s4 = 0; s4 < 4; ++s4) bv[u][s4] = bp[4 * (4 * h + s4) * WS];
            }
        };
        auto mma_h = [&](const f32x4 (&a)[MSW][2], int h, const float (&bv)[UPW][4]) {
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int u = 0; u < UPW; ++u)
              if (u < OU || ((MASK >> (u - OU)) & 1)) {
#pragma unroll
                for (int m = 0; m < MSW; ++m) acc[m][u] = pe_mfma_16x16x4(a[m][h][s4], bv[u][s4], acc[m][u]);
              }
        };
        constexpr int KH0 = (NMF / 2) / (NVM + NDS / 2 + 1), KH1 = (NMF / 2) / (NDS / 2 + 1);
        float b0[UPW][4], b1[UPW][4];
        int cur = 0;                                   // LDS offset of the step whose halves are being consumed
        read_bh(0, 0, b0);
        int st = 0;
        for (; st + 1 < nsteps; st += 2) {
          PE_SCHED_FENCE();
          load_a(wnext, aB);
          read_bh(cur, 1, b1);
          wnext += STEPF;
          mma_h(aA, 0, b0);
          mrf_interleave<NVM, NDS / 2, KH0>();
          PE_SCHED_FENCE();
          cur = advance();
          read_bh(cur, 0, b0);
          mma_h(aA, 1, b1);
          mrf_interleave<0, NDS / 2, KH1>();
          PE_SCHED_FENCE();
          load_a(wnext, aA);      // behind the phase's last step: the first step of the next phase
          read_bh(cur, 1, b1);
          wnext += STEPF;
          mma_h(aB, 0, b0);
          mrf_interleave<NVM, NDS / 2, KH0>();
          PE_SCHED_FENCE();
          cur = advance();
          read_bh(cur, 0, b0);
          mma_h(aB, 1, b1);
          mrf_interleave<0, NDS / 2, KH1>();
          PE_SCHED_FENCE();
        }
        if (st < nsteps) {        // odd step count: the last step, and the next phase's first fragments move to aA
          load_a(wnext, aB);
          read_bh(cur, 1, b1);
          wnext += STEPF;
          mma_h(aA, 0, b0);
          mrf_interleave<NVM, NDS / 2, KH0>();
          PE_SCHED_FENCE();
          mma_h(aA, 1, b1);
          PE_SCHED_FENCE();
#pragma unroll
          for (int m = 0; m < MSW; ++m)
#pragma unroll
            for (int q = 0; q < 2; ++q) aA[m][q] = aB[m][q];
        }
      } else {
      constexpr int KI = NMF / (NVM + NDS + 1);
      float bA[UPW][8], bB[UPW][8];
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
          for (int q = 0; q < 2; ++q) aA[m][q] = aB[m][q];
      }
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
    if (ph < 7) PE_STAMP(SK, 3 + 3 * ph);
    // ---- epilogue of the phase
    float* dstb = bufs + (P.dst < 0 ? 0 : P.dst) * bufsz;
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
            float t = acc[m][u][r] + bz[m][r];
            if (P.flags & MRF_RES) t += rawc[m][u][r];
            v[r] = inside ? t : 0.f;
          }
          if (P.flags & MRF_KEEP) rawc[m][u] = v;
          if (u < OU && (P.flags & MRF_FINAL)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) tot[m][u < OU ? u : 0][r] += v[r];
          }
          if (P.dst >= 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dstb[((ms0 + m) * 16 + 4 * lq + r) * WS + col] = pe_lrelu(v[r], slope);
          }
        }
        if (ph == 4 && u < 4) PE_STAMP(SK, 20 + u);
      }
    if (ph < 7) PE_STAMP(SK, 4 + 3 * ph);
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
  // ---- last stage with the generator tail fused (models.py:364-366: leaky_relu(0.01) -> conv_post (k = 7, one output
  // channel, no bias) -> tanh; piper.cpp:410-418: the utterance's max |sample|): the MRF mean never leaves the chip. The
  // windows of neighbouring workgroups overlap by POST_K - 1 columns (launch stride N - 6, first window at column -3), so
  // every tap of the workgroup's N - 6 samples is one of its own N mean columns: no halo units, no exchange. Arithmetic
  // and summation order are conv_post_kernel's (post.h): four channel groups of POST_CU channels, channel-major /
  // tap-minor fmaf chains, partials summed in group order -- bit-identical to the separate launch.
  __syncthreads();                       // every wave is done with the activation buffers
  float* mb = bufs;                      // buffer 0 <- lrelu(mean, 0.01) on window columns [hxa, hxa + N)
  float* part = bufs + bufsz;            // buffer 1 <- [POST_CG][N] partial sums
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
  const int nsamp = p.N - (POST_K - 1);  // samples of this workgroup: global n0 + 3 + j, j in [0, nsamp)
  {
    const int pg = PE_UNIFORM(tid >> 7), ct = tid & 127;     // channel group (two waves each), 4 consecutive samples
    if (4 * ct < nsamp) {
      float a4[POST_OPT];
#pragma unroll
      for (int o = 0; o < POST_OPT; ++o) a4[o] = 0.f;
#pragma unroll
      for (int cc = 0; cc < POST_CU; ++cc) {
        const int c = pg * POST_CU + cc;
        if (c < C) {
          const float* rp = mb + c * WS + p.hxa + 4 * ct;    // 16-byte aligned: taps of sample j = rp[j .. j + 6]
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
