// One whole MRF stage of the HiFiGAN generator per launch, second generation (models.py:356-363:
// xs = sum_j resblock_j(x) / n; modules.py:301-314 ResBlock1, :355-364 ResBlock2) for stages with <= 64 channels.
//
// A workgroup owns N output columns of one utterance and runs every conv of every resblock of the stage out of LDS
// (halo recompute on shrinking windows), so the stage costs ONE launch, one read of x and one write of the MRF mean
// instead of one launch and ~3 tensor round trips per conv. What changed against mrf_fused_kernel (kernels.h):
//   * LDS holds the ACTIVATED tensors (leaky-relu applied once, by the producer): the MFMA B operand is a plain
//     ds_read, no VALU on the read side. Residuals never come from LDS: a wave owns the same 16x16 output units in
//     every phase (static map unit -> wave), so the raw value of the running x of a resblock chain stays in its
//     registers, as do the raw stage input (residual of every chain's first conv) and the MRF sum.
//   * 16 waves (4 per SIMD) on v_mfma_f32_16x16x4_f32 units of 16 channels x 16 columns: the static map gives every
//     wave 2-4 units per phase, its unit accumulators interleave in the matrix pipe (no dependent-issue stall), and
//     the halo over-compute is quantised to 16 columns instead of 32.
//   * Weights are a flat stream in execution order ([segment][step][16-row tile][2][lane][4], step = one (32-channel
//     chunk, tap)); each thread fetches one float4 of the NEXT segment into a register while the current one feeds the
//     MFMAs from a double-buffered LDS ring: weights cross L2 -> CU once per workgroup, shared by all 16 waves, and
//     the only barrier is one per segment (it also orders the activation buffers between phases).
// Every step is the same k-ordered f32 fmaf chain as the conv kernels (chunk-major, tap-minor, ascending channel).
#pragma once
#include "../pe_rt.h"
#include "conv_common.h"

namespace pe {

template <int V> struct pe_int { static constexpr int value = V; };

enum { MRF2_RES = 1, MRF2_KEEP = 2, MRF2_FINAL = 4, MRF2_INIT = 8, MRF2_RESTAGE = 16 };

struct Mrf2Phase {       // one conv of one resblock chain; 12 ints wide (the kernel copies the table to LDS as ints)
  const float* bias;
  int ntaps, dil;
  int e;                 // columns of halo its OUTPUT still needs (0 for the last conv of a resblock)
  int src, dst;          // LDS activation buffers (0 = stage input window, 1 = chain buffer); dst < 0: none
  int flags;             // RES: + running x (registers); KEEP: result becomes the running x; FINAL: add to the MRF sum;
                         // INIT: running x = stage input (first conv of a resblock); RESTAGE: reload buffer 0 first
  int seg0, nseg;        // its weight segments in the stream
  int pad0, pad1;
};
static_assert(sizeof(Mrf2Phase) == 48, "Mrf2Phase is read as 12 ints");
struct Mrf2Seg { int step0, nsteps, woff, pad; };
struct Mrf2P {
  const float* x; long x_bs; int x_cs;
  float* out; long o_bs; int o_cs;
  const int* lens; int len_mul;
  const Mrf2Phase* phases; int nphases;
  const Mrf2Seg* segs; int nsegs;
  const float* wstream; int wfloats;
  int C;                 // real channels (<= CP)
  int N;                 // output columns per workgroup (multiple of 16)
  int WS;                // LDS row stride = window width, == 16 (mod 32): the two k rows of a half-wave hit disjoint banks
  int hxa;               // window column of the first output column (halo rounded up to 16)
  int cu_lo, cu_hi;      // 16-column units any phase needs: [cu_lo, cu_hi)
  int nleft, nhalo;      // halo units left of the output columns / in total
  float slope, alpha;
};

// A wave owns MSW 16-row tiles (ms0 .. ms0 + MSW - 1) of OU output + HU halo column blocks: column unit u < OU is output
// block cg + NCG * u of the N / 16 output blocks (N == 16 * NCG * OU); unit OU + v is halo block h = cg + NCG * v of the
// nhalo blocks around them (left ones first). With MSW = 2 (64-channel stages) one B operand feeds two MFMAs.
// WS (LDS row stride, == 16 mod 32) is a compile-time constant: the eight k-rows of a step are immediates of one base.
// NWR = float4 per thread per weight segment (ring half = NWR * 16 KiB at 16 waves).
template <int CP, int NW, int MSW, int OU, int HU, int WS, int NWR>
__global__ __launch_bounds__(64 * NW) void mrf2_kernel(Mrf2P p) {
  PE_KTRACE(19);
  constexpr int MS = CP / 16, NRG = MS / MSW, NCG = NW / NRG, NT = 64 * NW, UPW = OU + HU;
  constexpr int RINGF = 4 * NT * NWR;              // floats per ring half
  constexpr int MAXPH = 24, MAXSEG = 64;           // table capacities (engine.cpp: build_mrf2 checks them)
  static_assert(MS % MSW == 0 && NW % NRG == 0, "waves split evenly over the row groups");
  PE_DYN_SMEM(float, sm);
  const int b = blockIdx.y;
  const int L = p.lens[b] * p.len_mul;
  const int n0 = blockIdx.x * p.N;
  if (n0 >= L) return;
  static_assert(WS % 32 == 16, "row stride == 16 (mod 32)");
  float* ring = sm;                                // 2 x RINGF
  float* bufs = sm + 2 * RINGF;                    // 2 x [CP][WS]; reads that leave a buffer on the left / right land
  constexpr int bufsz = CP * WS;                   // in the ring / the other buffer / the tail pad (unused columns only)
  // phase / segment tables: copied to LDS once, so that no phase or segment starts with a global-memory round trip
  int* tph = reinterpret_cast<int*>(bufs + 2 * bufsz + 128);     // [MAXPH][12] ints (Mrf2Phase is 12 ints wide)
  int* tsg = tph + MAXPH * 12;                                   // [MAXSEG][4]
  const int tid = threadIdx.x, lane = tid & 63, wv = PE_UNIFORM(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int ms0 = (wv % NRG) * MSW, cg = wv / NRG;
  const int g0 = n0 - p.hxa;                       // global column of window column 0
  const float slope = p.slope;
  const int C = p.C;

  // ---- weight stream: segment s -> ring half s & 1, one float4 per thread, fetched one segment ahead
  const pe_rowsrc wd = pe_make_row(p.wstream, p.wfloats);
  f32x4 wreg[NWR];
  auto wfetch_at = [&](int woff) {
#pragma unroll
    for (int k = 0; k < NWR; ++k) wreg[k] = pe_row_load4(wd, woff + (tid + k * NT) * 4);
  };
  auto wfetch = [&](int seg) { wfetch_at(PE_UNIFORM(seg < p.nsegs ? tsg[seg * 4 + 2] : 0x3ffffff0)); };   // past the end: zeros
  wfetch_at(0);                 // segment 0 starts the stream
  {
    const int* gp = reinterpret_cast<const int*>(p.phases);
    const int* gs = reinterpret_cast<const int*>(p.segs);
    for (int i = tid; i < p.nphases * 12; i += NT) tph[i] = gp[i];
    for (int i = tid; i < p.nsegs * 4; i += NT) tsg[i] = gs[i];
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
      for (int j = 0; j < NCC; ++j) v[i][j] = pe_row_load(rd, g0 + lane + 64 * j);
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
  f32x4 acc[MSW][UPW], rawx[MSW][UPW], rawc[MSW][UPW], tot[MSW][OU];
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
    const pe_rowsrc xd = pe_make_row(xb, C * p.x_cs);
#pragma unroll
    for (int m = 0; m < MSW; ++m)
#pragma unroll
      for (int u = 0; u < UPW; ++u) {
        const int g = g0 + 16 * cu[u] + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = (ms0 + m) * 16 + 4 * lq + r;
          rawx[m][u][r] = pe_row_load(xd, (cu[u] >= 0 && g >= 0 && g < L && row < C) ? row * p.x_cs + g : 0x3fffffff);
          rawc[m][u][r] = 0.f;
        }
      }
#pragma unroll
    for (int m = 0; m < MSW; ++m)
#pragma unroll
      for (int u = 0; u < OU; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) tot[m][u][r] = 0.f;
  }
  stage_x();
  __syncthreads();              // tables and window are in LDS

  int seg = 0;
  for (int ph = 0; ph < p.nphases; ++ph) {
    Mrf2Phase P;
    {
      const int* t = tph + ph * 12;
      const unsigned lo = PE_UNIFORM((unsigned)t[0]), hi = PE_UNIFORM((unsigned)t[1]);
      P.bias = reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
      P.ntaps = PE_UNIFORM(t[2]); P.dil = PE_UNIFORM(t[3]); P.e = PE_UNIFORM(t[4]);
      P.src = PE_UNIFORM(t[5]); P.dst = PE_UNIFORM(t[6]); P.flags = PE_UNIFORM(t[7]);
      P.seg0 = PE_UNIFORM(t[8]); P.nseg = PE_UNIFORM(t[9]);
    }
    if (P.flags & MRF2_RESTAGE) {       // ResBlock1 rewrites buffer 0 in place: a new chain starts from the stage input
      __syncthreads();
      stage_x();
    }
    if (P.flags & MRF2_INIT) {
#pragma unroll
      for (int m = 0; m < MSW; ++m)
#pragma unroll
        for (int u = 0; u < UPW; ++u) rawc[m][u] = rawx[m][u];
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
    // The K loop, specialised at compile time on WHICH halo units take part in this phase (bit v of MASK = halo unit
    // v; output units always do): the hot loop is straight-line code, the choice is one wave-uniform switch per phase.
    // Every variant runs the same segments and barriers.
    auto k_loop = [&](auto maskc) {
      constexpr int MASK = decltype(maskc)::value;
      for (int s = 0; s < P.nseg; ++s, ++seg) {
        const int step0 = PE_UNIFORM(tsg[seg * 4]), nsteps = PE_UNIFORM(tsg[seg * 4 + 1]);
        float* half = ring + (seg & 1) * RINGF;
#pragma unroll
        for (int k = 0; k < NWR; ++k) *reinterpret_cast<f32x4*>(half + (tid + k * NT) * 4) = wreg[k];
        __syncthreads();      // this segment's weights (and the previous phase's activations) are visible; every wave
                              // is done with the segment before, whose ring half the NEXT store overwrites
        wfetch(seg + 1);
#pragma unroll 1
        for (int st = 0; st < nsteps; ++st) {
          const int step = step0 + st;
          const int c = step / P.ntaps, tap = step - c * P.ntaps;
          float a[MSW][8];
#pragma unroll
          for (int m = 0; m < MSW; ++m) {
            const float* ap = half + (st * MS + ms0 + m) * 512 + lane * 4;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(ap);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(ap + 256);
#pragma unroll
            for (int j = 0; j < 4; ++j) { a[m][j] = a0[j]; a[m][4 + j] = a1[j]; }
          }
          const float* bp0 = src + (c * KC + lq) * WS + tap * P.dil - hh + l15;
          float bv[UPW][8];
#pragma unroll
          for (int u = 0; u < UPW; ++u)
            if (u < OU || ((MASK >> (u - OU)) & 1)) {
              const float* bp = bp0 + 16 * cu[u];
#pragma unroll
              for (int s8 = 0; s8 < 8; ++s8) bv[u][s8] = bp[4 * s8 * WS];
            }
          // unit-interleaved: consecutive MFMAs hit different accumulators (no dependent-issue stall)
#pragma unroll
          for (int s8 = 0; s8 < 8; ++s8)
#pragma unroll
            for (int u = 0; u < UPW; ++u)
              if (u < OU || ((MASK >> (u - OU)) & 1)) {
#pragma unroll
                for (int m = 0; m < MSW; ++m) acc[m][u] = pe_mfma_16x16x4(a[m][s8], bv[u][s8], acc[m][u]);
              }
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
            if (P.flags & MRF2_RES) t += rawc[m][u][r];
            v[r] = inside ? t : 0.f;
          }
          if (P.flags & MRF2_KEEP) rawc[m][u] = v;
          if (u < OU && (P.flags & MRF2_FINAL)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) tot[m][u < OU ? u : 0][r] += v[r];
          }
          if (P.dst >= 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dstb[((ms0 + m) * 16 + 4 * lq + r) * WS + col] = pe_lrelu(v[r], slope);
          }
        }
      }
  }
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
}

}  // namespace pe
