// Device kernels of the synthesis path (gfx950 / CDNA4). Activations are fp32, channels-first
// [B][C][Lstride] with a per-utterance valid length: every kernel treats columns >= len[b] as
// non-existent (read as zero, never written), which reproduces the reference's B=1 zero padding
// and x_mask semantics for every utterance of a batch.
//
// Reference arithmetic each kernel implements is cited per kernel (paths relative to
// /root/reference/src/python/piper_train/vits/).
#pragma once
#include "pe_rt.h"

namespace pe {

static constexpr int KC = 32;           // input channels staged per K-chunk of the conv GEMM

// A-operand fragments (engine.cpp: pack_matrix). One (m tile, chunk, tap) step is 1024 floats:
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

enum Epi { EPI_STORE = 0, EPI_RESADD = 1, EPI_GATE = 2, EPI_WNRS = 3, EPI_SUBFROM = 4,
           EPI_ACCUM = 5, EPI_CONVT = 6 };
enum Act { ACT_NONE = 0, ACT_RELU = 1 };

struct ConvP {
  const float* x; long x_bs; int x_cs;          // input  x[b][ci][t]
  const float* wp;                              // packed weights (engine.cpp: pack_conv)
  const float* wp16;                            // conv_splitk16_kernel: the same in 16x16x4 fragment order, or null
  const float* bias;                            // per output channel or null
  const float* bias2; int bias2_bs;             // per-utterance extra bias (speaker cond) or null
  float* out; long o_bs; int o_cs;
  const float* res; long r_bs; int r_cs;        // residual input (may alias out)
  float* out2; long o2_bs; int o2_cs;           // second output (WN skip accumulator)
  const int* lens; int len_mul;                 // valid input length = lens[b]*len_mul
  int Cin, rows;                                // real input channels; GEMM rows (Cout, or Cout*up)
  int nchunks;                                  // ceil(Cin/KC)
  int ntaps, dil, padl;                         // tap k reads x[t + k*dil - padl]
  int xhalo;                                    // (ntaps-1)*dil
  float in_slope;                               // leaky-relu slope applied to x while staging (1 = none)
  int epi, act;
  int split;                                    // GATE: H ; WNRS: rows < split go to h, rest to skip
  int up, padT;                                 // CONVT: stride and padding
  unsigned up_magic;                            // CONVT: ceil(2^32 / up): row / up == (row * up_magic) >> 32 for row < 2^16
  int mode;                                     // ACCUM: 0 first,1 middle,2 last,3 only ; WNRS: 1 = first layer
  float alpha;                                  // ACCUM last/only: scale
  int tpb;                                      // conv_mfma_kernel: column tiles walked by one workgroup
  int tgroups;                                  // conv_splitk_kernel: 1, or 2 = two halves of the waves split the taps
  // conv_splitk_kernel only: LayerNorm over the input channels applied while staging x (modules.py:23-26; the encoder's
  // norm_layers_1/2 folded into the conv that consumes them). The workgroups of row tile 0 write LN(x) back for the
  // later residual readers. Requires one chunk per wave (Cin <= 32 * waves).
  const float* ln_g; const float* ln_b;
  float* ln_out; long ln_o_bs; int ln_o_cs;
  // conv_splitk_body<..., MS = true> only: K = the concatenation of nseg convs of one shape whose outputs are summed
  // (segment 0 repeats x / wp / ntaps / dil / padl); res2 / res3 = the residual tensors of segments 1 / 2
  int nseg;
  const float* seg_x[3]; const float* seg_wp[3];
  int seg_ntaps[3], seg_dil[3], seg_padl[3];
  const float* res2; const float* res3;
};

// ---- shared epilogue of the conv GEMM kernels: one accumulator element (row, col) of utterance b.
// Every non-transposed mode is the same straight-line form
//     dst = alpha * ( old*use_old + (res*use_res + (acc + bias)*sign) )
// with per-launch uniform flags, which keeps the unrolled epilogue small:
//   STORE  : dst=out                         RESADD : +res            SUBFROM: old - v  (modules.py:464)
//   ACCUM  : MRF sum/scale (models.py:356-363)       WNRS: rows<split h += v, else skip (+)= v (modules.py:201-208)
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
// is not used and rows*stride otherwise, so unused operands and rows beyond the GEMM read as 0 and such
// stores are dropped by the range check; invalid columns poison the lane offset. Addresses are one
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
  // two groups of eight elements (register budget of the 4-waves-per-SIMD instantiations); an element only
  // ever reads its own location, so a group's stores cannot disturb the next group's loads
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    float b1[8], b2[8], o1[8], o2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int r = 8 * g + e, kr = (r & 3) + 8 * (r >> 2);
      b1[e] = pe_row_load(bd, rb + kr);
      b2[e] = pe_row_load(b2d, rb + kr);
      o1[e] = pe_row_load_so(old, ooff, kr * ocs);
      o2[e] = pe_row_load_so(rd, roff, kr * p.r_cs);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int r = 8 * g + e, kr = (r & 3) + 8 * (r >> 2);
      float v = ((acc[r] + (b1[e] + b2[e])) * f.sign + (o1[e] + o2[e])) * f.alpha;
      if (f.relu) v = v > 0.f ? v : 0.f;
      pe_row_store_so(od, ooff, kr * ocs, v);
    }
    PE_SCHED_FENCE();
  }
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
__global__ __launch_bounds__(256, (MT * NT == 1 ? ((HALO == 128 && WN == 4) ? 3 : 4) : ((GATE && MT * NT == 2) ? 3 : 2)))
void conv_mfma_kernel(ConvP p) {
  PE_KTRACE(11);
  constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
  constexpr int NCOL = (BN + HALO + 63) / 64;    // staging columns per lane; (taps-1)*dilation <= HALO
  constexpr int KH = KC / 2;
  constexpr int NSUB = KH / KS;                   // A prefetch sets per (chunk, tap)
  static_assert(WM * WN == 4, "4 waves per block");
  static_assert(NSUB * KS == KH && KS % 4 == 0, "KS must divide KC/2 in float4 groups");
  static_assert(!GATE || MT == 2, "gate epilogue pairs two M tiles");
  constexpr int XS = NCOL * 64;                   // LDS row stride (compile time: taps become immediates)
  PE_DYN_SMEM(float, xs);                         // 2 x [KC][XS]
  const int b = blockIdx.z;
  const int L = p.lens[b] * p.len_mul;
  const int ncols = (p.epi == EPI_CONVT) ? L + 1 : L;
  // a workgroup walks p.tpb consecutive column tiles (1 by default, profiles/r01_tpb_sweep.txt)
  const int tile0 = blockIdx.x * p.tpb;
  const int ntile_all = (ncols + BN - 1) / BN;
  if (tile0 >= ntile_all) return;
  const int ntl = (ntile_all - tile0) < p.tpb ? (ntile_all - tile0) : p.tpb;
  const int m0 = blockIdx.y * BM;
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
  auto store_x = [&](int buf) {
    float* dst = xs + buf * KC * XS + wv * XS + lane;
#pragma unroll
    for (int rr = 0; rr < KC / 4; ++rr)
#pragma unroll
      for (int cc = 0; cc < NCOL; ++cc) {
        float v = xr[rr][cc];
        v = v > 0.f ? v : v * slope;
        dst[4 * rr * XS + 64 * cc] = v;
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
  float xr[XB][KC];
  pe_rowsrc xd = pe_make_row(xb, p.Cin * p.x_cs);
  int xcol = n0 - p.padl + lane;
  int xoff[XB];
#pragma unroll
  for (int h = 0; h < XB; ++h) xoff[h] = (xcol + 64 * h >= 0 && xcol + 64 * h < p.x_cs) ? xcol + 64 * h : 0x3fffffff;
  int ld_dil = p.dil, cur_dil = p.dil;            // dilation of the chunk in xr / of the chunk in the LDS slab
  auto load_x = [&](int c) {
    int cc = c;
    if (MS) {
      const int sgr = seg_of(c), sg = sgr < p.nseg ? sgr : 0;
      cc = sgr < p.nseg ? c - sg * p.nchunks : p.nchunks;      // (a wave without work loads "chunk nchunks": zeros)
      xd = pe_make_row(p.seg_x[sg] + (long)b * p.x_bs, p.Cin * p.x_cs);
      xcol = n0 - p.seg_padl[sg] + lane;
      ld_dil = p.seg_dil[sg];
#pragma unroll
      for (int h = 0; h < XB; ++h) xoff[h] = (xcol + 64 * h >= 0 && xcol + 64 * h < p.x_cs) ? xcol + 64 * h : 0x3fffffff;
    }
#pragma unroll
    for (int h = 0; h < XB; ++h)
#pragma unroll
      for (int r = 0; r < KC; ++r) xr[h][r] = pe_row_load_so(xd, xoff[h], (cc * KC + r) * p.x_cs);
  };
  auto store_x = [&]() {
    cur_dil = ld_dil;
#pragma unroll
    for (int h = 0; h < XB; ++h) {
      const bool live = xcol + 64 * h < L;
#pragma unroll
      for (int r = 0; r < KC; ++r) {
        float v = live ? xr[h][r] : 0.f;
        v = v > 0.f ? v : v * slope;
        xw[r * XW + 64 * h + lane] = v;
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
  if (!GATE && XW == 64 && !MS && p.ln_g) {
    // LayerNorm of the staged columns over ALL input channels: every wave holds one 32-channel chunk of the same 64
    // columns (lane = column); two fixed-order cross-wave sums (mean, then centred second moment, like ln_kernel).
    float* red1 = sm + NW * (KC * XW > MT * 16 * 64 ? KC * XW : MT * 16 * 64);     // behind the slabs / partial tiles
    float* red2 = red1 + NW * 64;
    const bool mine = myc > 0;
    const int c0 = wi * KC;
    float s1 = 0.f;
#pragma unroll
    for (int r = 0; r < KC; ++r) s1 += (mine && c0 + r < p.Cin) ? xr[0][r] : 0.f;
    red1[wv * 64 + lane] = s1;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) tot += red1[w * 64 + lane];
    const float mean = tot / (float)p.Cin;
    float s2 = 0.f;
#pragma unroll
    for (int r = 0; r < KC; ++r) {
      const float dlt = xr[0][r] - mean;
      s2 += (mine && c0 + r < p.Cin) ? dlt * dlt : 0.f;
    }
    red2[wv * 64 + lane] = s2;
    __syncthreads();
    float tot2 = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) tot2 += red2[w * 64 + lane];
    const float rstd = 1.f / sqrtf(tot2 / (float)p.Cin + 1e-5f);
    const bool wb = p.ln_out != nullptr && blockIdx.y == 0 && mine && xcol >= n0 && xcol < n0 + BN && xcol < L;
    float* ob = p.ln_out + (long)b * p.ln_o_bs + xcol;
#pragma unroll
    for (int r = 0; r < KC; ++r) {
      const int ci = c0 + r;
      const bool cv = mine && ci < p.Cin;
      const float g = cv ? p.ln_g[ci] : 0.f, be = cv ? p.ln_b[ci] : 0.f;      // wave-uniform: scalar loads
      xr[0][r] = xcol >= 0 ? (xr[0][r] - mean) * rstd * g + be : 0.f;       // left halo: zero padding comes AFTER the norm
      if (wb && cv) ob[(long)ci * p.ln_o_cs] = xr[0][r];
    }
  }

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
struct ConvG {
  ConvP c[3];
  int n, B;
};
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
// K is dealt to the waves exactly as in conv_splitk_kernel. Weights: engine.cpp pack16 --
// [16-row sub-tile][chunk][tap][q = 0..1][lane][4] with lane -> (row = lane & 15, k = lane >> 4), float4 element j of
// group q = k-step s = 4q + j, input channel chunk*32 + 4s + k; ascending k inside and across instructions, i.e. the
// same fmaf chain as the 32x32x2 form.
template <bool GATE, int NW, int D>
__global__ __launch_bounds__(64 * NW) void conv_splitk16_kernel(ConvP p) {
  PE_KTRACE(4);
  constexpr int BN = 16, XW = 64, KS8 = KC / 4, MT16 = GATE ? 4 : 2;
  constexpr int NSLOT = GATE ? 8 : MT16 * 4;                  // result slots per lane position (gate: tanh/sigmoid pairs)
  constexpr int NS = (NSLOT + NW - 1) / NW;                   // epilogue slots per wave
  PE_DYN_SMEM(float, sm);                         // NW x [KC][XW] slabs, then NW x [MT16*4][64] partial tiles
  const int b = blockIdx.z;
  const int L = p.lens[b] * p.len_mul;            // first used after the loads below are in flight
  const int n0 = blockIdx.x * BN;
  const int tid = threadIdx.x, lane = tid & 63, wv = PE_UNIFORM(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int st0 = blockIdx.y * MT16;              // first 16-row sub-tile of this workgroup
  const int col = n0 + l15;
  const int ntaps = p.ntaps, nchunks = p.nchunks;
  const int sub_stride = nchunks * ntaps * KS8 * 64;          // floats per 16-row sub-tile
  const float* xb = p.x + (long)b * p.x_bs;
  const float slope = p.in_slope;
  const pe_rowsrc wsrc = pe_make_row(p.wp16 + (long)st0 * sub_stride, MT16 * sub_stride);
  float* xw = sm + wv * KC * XW;
  const int CL = NW / p.tgroups;
  const int wi = wv % CL, wg = wv / CL;
  const int tpg = (ntaps + p.tgroups - 1) / p.tgroups;
  const int tap_lo = wg * tpg, tap_hi = (tap_lo + tpg < ntaps) ? tap_lo + tpg : ntaps;
  const int mytaps = tap_hi > tap_lo ? tap_hi - tap_lo : 0;
  const int myc = (wi < nchunks && mytaps > 0) ? (nchunks - wi + CL - 1) / CL : 0;
  const int nsteps = myc * mytaps;

  float xr[KC];
  const pe_rowsrc xd = pe_make_row(xb, p.Cin * p.x_cs);
  const int xcol = n0 - p.padl + lane;
  const int xoff = (xcol >= 0 && xcol < p.x_cs) ? xcol : 0x3fffffff;
  auto load_x = [&](int c) {
#pragma unroll
    for (int r = 0; r < KC; ++r) xr[r] = pe_row_load_so(xd, xoff, (c * KC + r) * p.x_cs);
  };
  auto store_x = [&]() {
    const bool live = xcol < L;
#pragma unroll
    for (int r = 0; r < KC; ++r) {
      float v = live ? xr[r] : 0.f;
      v = v > 0.f ? v : v * slope;
      xw[r * XW + lane] = v;
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
        const f32x4 t = pe_row_load4(wsrc, off + i * sub_stride + q * 256 + lane * 4);
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
      const int ch = blockIdx.y * 32 + (s >> 2) * 16 + 4 * lq + (s & 3);
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
          sa += red[(w * MT16 * 4 + 8 + s) * 64 + lane];
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

// ------------------------------------------------------------------------------------------------
// Text-encoder embedding: x[b][h][t] = emb[id][h] * sqrt(H)  (models.py:199-200)
// The first kernel of every pipeline run also advances the RNG call counter (state[1]) that both randn sites of
// the run read afterwards, so a replayed graph draws fresh noise on every run without a host copy.
__global__ void embed_kernel(const int* ids, int ids_bs, const int* lens, const float* emb, int H,
                             float scale, float* out, long o_bs, int o_cs, unsigned long long* rng_state) {
  PE_KTRACE(10);
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) rng_state[1] += 1ull;
  const int b = blockIdx.z;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= lens[b]) return;
  const int id = ids[b * ids_bs + t];
  const float* e = emb + (long)id * H;
  float* o = out + (long)b * o_bs + t;
  const int h0 = blockIdx.y * 16;
#pragma unroll
  for (int k = 0; k < 16; ++k)
    if (h0 + k < H) o[(long)(h0 + k) * o_cs] = e[h0 + k] * scale;
}

// ------------------------------------------------------------------------------------------------
// Windowed relative-position multi-head self-attention (attentions.py:225-272, 292-348).
// qkv: [B][3H][Ts] rows [0,H)=q, [H,2H)=k, [2H,3H)=v. The reference's pad/reshape "relative to
// absolute" trick is evaluated directly as a band: logits[i][j] += q_i . rel_k[j-i+w] and
// out_i += sum_r p[i][i+r] rel_v[r+w] for |r| <= w. Masked keys (>= len) get weight exactly 0, which is
// what the reference's -1e4 fill yields in fp32.
struct AttnP {
  const float* qkv; long q_bs; int q_cs;
  const float* relk; const float* relv;     // [2w+1][dk]
  float* out; long o_bs; int o_cs;
  const int* lens;
  int H, dk, window;
  int SP;                                   // score row stride in LDS: odd, >= round_up(max len, 64)
  float qscale;
};
static constexpr int ATT_QB = 32;           // queries per workgroup (one MFMA tile)
static constexpr int ATT_KCH = 64;          // keys staged per V chunk
static constexpr int ATT_MAXDK = 128;

// One workgroup = 32 queries of one (utterance, head); 4 waves.
//   1. S = (q/sqrt(dk)) k^T on the f32 MFMAs: wave w owns key tiles w, w+4, ...; both operands are read
//      from global memory directly in fragment order (q and k rows are contiguous along time).
//   2. banded relative-key logits, softmax over the valid keys (8 lanes per query row).
//   3. O^T = V P^T on the MFMAs (V chunk transposed through LDS so the A fragment is contiguous; P read
//      from the score slab with an odd stride), wave w owns channel tiles w, w+4, ...; banded
//      relative-value term added before the coalesced store.
// DKT: channels per head known at compile time (96 for the 192-channel voices, 48 for x-low): every unrolled loop has its
// exact trip count. DKT = 0: any even dk <= ATT_MAXDK, loops sized for the maximum and guarded per step (on the common
// shapes those guards were ~190 scalar branches per workgroup, a third of the kernel's time).
template <int DKT>
__global__ __launch_bounds__(256) void attn_kernel(AttnP p) {
  constexpr int MAXDK = DKT ? DKT : ATT_MAXDK;
  PE_KTRACE(0);
  PE_DYN_SMEM(float, sm);
  PE_STAMP(0, 0);
  const int b = blockIdx.z, h = blockIdx.y, i0 = blockIdx.x * ATT_QB;
  const int T = p.lens[b];
  if (i0 >= T) return;
  PE_STAMP(0, 1);
  const int dk = DKT ? DKT : p.dk, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int SP = p.SP, VS = dk + 1 + (dk & 1);       // odd strides -> conflict-free column reads
  float* S = sm;                                      // [32][SP]
  float* Vt = S + ATT_QB * SP;                        // [KCH][VS]
  float* Qs = Vt + ATT_KCH * VS;                      // [dk][32], scaled by 1/sqrt(dk)
  const int nrel = 2 * p.window + 1;
  float* RK = Qs + dk * ATT_QB;                       // [nrel][dk] relative-key embeddings
  float* RV = RK + nrel * dk;                         // [nrel][dk] relative-value embeddings
  const float* qb = p.qkv + (long)b * p.q_bs + (long)(h * dk) * p.q_cs;
  const float* kb = qb + (long)p.H * p.q_cs;
  const float* vb = kb + (long)p.H * p.q_cs;
  const int nkt = (T + 31) / 32;
  const int nk2 = dk / 2;                             // MFMA k-steps over channels (dk even)

  // ---- 1. scores
  // all global reads go through buffer descriptors with index -1 for masked elements (hardware returns 0), so
  // each staging step issues its loads back to back: one memory latency per step instead of one per element
  const pe_rowsrc qd = pe_make_row(qb, dk * p.q_cs), kd = pe_make_row(kb, dk * p.q_cs), vd = pe_make_row(vb, dk * p.q_cs);
  constexpr int NKF = MAXDK / 2;
  float kf[NKF];
  auto load_k = [&](int kt) {
    // one per-lane base (channel parity, key) + a wave-uniform 2*u*stride in an SGPR: no VALU per load; rows
    // beyond dk fall outside the descriptor and read 0
    const int j = kt * 32 + l31;
    const int base = (kt < nkt && j < T) ? lhi * p.q_cs + j : 0x3fffffff;
#pragma unroll
    for (int u = 0; u < NKF; ++u) kf[u] = pe_row_load_so(kd, base, 2 * u * p.q_cs);
  };
  // Every global operand of the kernel that does not depend on earlier phases is requested NOW, together: this wave's
  // first key tile, the first V chunk, then Q and the relative-position tables -- one memory latency instead of three
  // serialised ones (the barriers below wait for all of them anyway).
  // V chunk staging: thread -> key jj = tid&63, channel group tid>>6
  float vv[(MAXDK + 31) / 32][8];
  auto load_v = [&](int j0) {
    const int jj = tid & 63;
    const int base = (j0 + jj < T) ? (tid >> 6) * 8 * p.q_cs + j0 + jj : 0x3fffffff;   // rows >= dk read 0
#pragma unroll
    for (int g = 0; g < (MAXDK + 31) / 32; ++g)
#pragma unroll
      for (int u = 0; u < 8; ++u) vv[g][u] = pe_row_load_so(vd, base, (32 * g + u) * p.q_cs);
  };
  auto store_v = [&]() {
    const int jj = tid & 63;
#pragma unroll
    for (int g = 0; g < (MAXDK + 31) / 32; ++g)
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int d = (tid >> 6) * 8 + 32 * g + u;
        if (d < dk) Vt[jj * VS + d] = vv[g][u];
      }
  };
  load_k(wv);
  load_v(0);
  {
    constexpr int NQ = MAXDK * ATT_QB / 256;     // 16 elements per thread at dk = 128
    float qv[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
      const int e = tid + 256 * u, d = e >> 5, i = e & 31;
      qv[u] = pe_row_load(qd, (d < dk && i0 + i < T) ? d * p.q_cs + i0 + i : -1);
    }
    // the two small relative-position tables ride along: the band loops below then never touch global memory
    constexpr int NR = 5;                              // (2*4+1) * 128 / 256 rounded up
    const pe_rowsrc rkd = pe_make_row(p.relk, nrel * dk), rvd = pe_make_row(p.relv, nrel * dk);
    float rk[NR], rv[NR];
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      rk[u] = pe_row_load(rkd, tid + 256 * u);
      rv[u] = pe_row_load(rvd, tid + 256 * u);
    }
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
      const int e = tid + 256 * u;
      if (e < dk * ATT_QB) Qs[e] = qv[u] * p.qscale;
    }
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      const int e = tid + 256 * u;
      if (e < nrel * dk) { RK[e] = rk[u]; RV[e] = rv[u]; }
    }
  }
  PE_STAMP(0, 2);
  __syncthreads();
  PE_STAMP(0, 3);
  // relative-key partial products R[q][r] = Q . rel_k^T (a 32 x (2w+1) GEMM over dk, a quarter of the channel steps
  // per wave): computed here, next to the score tiles -- both only need Q and the tables in LDS -- so that one barrier
  // publishes the scores and the partials together; they are added onto the band in step 2a
  float* part = RV + nrel * dk;                        // [4 waves][32 queries][16 offsets]
  {
    f32x16 racc;
#pragma unroll
    for (int r = 0; r < 16; ++r) racc[r] = 0.f;
    for (int s2 = wv; s2 < nk2; s2 += 4) {
      const int d = 2 * s2 + lhi;
      racc = pe_mfma_32x32x2(Qs[d * ATT_QB + l31], l31 < nrel ? RK[l31 * dk + d] : 0.f, racc);
    }
    if (l31 < 16) {
#pragma unroll
      for (int r = 0; r < 16; ++r) part[(wv * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * 16 + l31] = racc[r];
    }
  }
  {
    for (int kt = wv; kt < nkt; kt += 4) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      // all K fragments of a key tile are requested at once (<= 64 loads; the wave's first tile already at kernel
      // start), then each batch of 32 steps reads its Q operands from LDS in one go and issues its MFMAs back to back
      if (kt != wv) load_k(kt);
#pragma unroll
      for (int s0 = 0; s0 < NKF; s0 += 32) {
        if (s0 < nk2) {
          float qf[32];
#pragma unroll
          for (int u = 0; u < 32; ++u) qf[u] = (s0 + u < nk2) ? Qs[(2 * (s0 + u) + lhi) * ATT_QB + l31] : 0.f;
          PE_SCHED_FENCE();
#pragma unroll
          for (int u = 0; u < 32; ++u)
            if (s0 + u < nk2) acc = pe_mfma_32x32x2(qf[u], kf[s0 + u], acc);
          PE_SCHED_FENCE();
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) S[((r & 3) + 8 * (r >> 2) + 4 * lhi) * SP + kt * 32 + l31] = acc[r];
    }
  }
  PE_STAMP(0, 4);
  __syncthreads();
  PE_STAMP(0, 5);
  // ---- 2a. relative-key band: S[i][i+r-w] += (q_i/sqrt(dk)) . rel_k[r]: the four waves' partial tiles (above) meet
  // here and are scattered onto the band.
  {
    PE_STAMP(0, 6);
    for (int e = tid; e < ATT_QB * nrel; e += 256) {
      const int i = e % ATT_QB, r = e / ATT_QB;
      const int j = i0 + i + r - p.window;
      if (i0 + i < T && j >= 0 && j < T)
        S[i * SP + j] += (part[i * 16 + r] + part[(32 + i) * 16 + r]) + (part[(64 + i) * 16 + r] + part[(96 + i) * 16 + r]);
    }
  }
  __syncthreads();
  PE_STAMP(0, 7);
  // ---- 2b. softmax over valid keys: row = tid/8, 8 adjacent lanes per row (values stay in registers for the
  // common T <= 128)
  {
    const int i = tid >> 3, sj = tid & 7;
    float* Sr = S + i * SP;
    const int Tpad = (T + ATT_KCH - 1) / ATT_KCH * ATT_KCH;
    if (T <= 128) {
      float ev[16];
      float mx = -3.0e38f;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int j = sj + 8 * k;
        ev[k] = j < T ? Sr[j] : -3.0e38f;
        mx = fmaxf(mx, ev[k]);
      }
      for (int m = 4; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m));
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int j = sj + 8 * k;
        ev[k] = j < T ? expf(ev[k] - mx) : 0.f;
        sum += ev[k];
      }
      for (int m = 4; m >= 1; m >>= 1) sum += __shfl_xor(sum, m);
      const float inv = 1.f / sum;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int j = sj + 8 * k;
        if (j < Tpad) Sr[j] = ev[k] * inv;
      }
    } else {
      float mx = -3.0e38f;
      for (int j = sj; j < T; j += 8) mx = fmaxf(mx, Sr[j]);
      for (int m = 4; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m));
      float sum = 0.f;
      for (int j = sj; j < T; j += 8) {
        const float e = expf(Sr[j] - mx);
        Sr[j] = e;
        sum += e;
      }
      for (int m = 4; m >= 1; m >>= 1) sum += __shfl_xor(sum, m);
      const float inv = 1.f / sum;
      for (int j = sj; j < Tpad; j += 8) Sr[j] = (j < T) ? Sr[j] * inv : 0.f;
    }
  }
  __syncthreads();
  PE_STAMP(0, 8);
  // ---- 3. O^T[d][q] = sum_key V[d][key] P[q][key]
  const int ndt = (dk + 31) / 32;
  f32x16 oacc;                                        // this wave's channel tile (wv < ndt), one tile per wave pass
  for (int dt0 = 0; dt0 < ndt; dt0 += 4) {
    const int dt = dt0 + wv;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
    for (int j0 = 0; j0 < T; j0 += ATT_KCH) {
      __syncthreads();                                // previous chunk consumed / softmax finished
      PE_STAMP(0, 9 + 3 * (j0 / ATT_KCH));
      store_v();
      // next chunk (of this pass, or the first one of the next channel pass) in flight under the MFMAs
      if (j0 + ATT_KCH < T) load_v(j0 + ATT_KCH);
      else if (dt0 + 4 < ndt) load_v(0);
      __syncthreads();
      PE_STAMP(0, 10 + 3 * (j0 / ATT_KCH));
      if (dt < ndt) {
        const int d = dt * 32 + l31;
        float af[ATT_KCH / 2], pf[ATT_KCH / 2];
#pragma unroll
        for (int s2 = 0; s2 < ATT_KCH / 2; ++s2) {
          const int key = 2 * s2 + lhi;
          af[s2] = d < dk ? Vt[key * VS + d] : 0.f;
          pf[s2] = S[l31 * SP + j0 + key];
        }
        PE_SCHED_FENCE();
#pragma unroll
        for (int s2 = 0; s2 < ATT_KCH / 2; ++s2) oacc = pe_mfma_32x32x2(af[s2], pf[s2], oacc);
        PE_SCHED_FENCE();
      }
      PE_STAMP(0, 11 + 3 * (j0 / ATT_KCH));
    }
    if (dt < ndt) {
      // relative-value band as five more k-steps of the same accumulation: key index -> relative offset rr,
      // A = rel_v[rr][d], B = p[q][q + rr - w] (zero outside the band / the utterance)
      const int q = i0 + l31;
      const int d0 = dt * 32 + l31;
      constexpr int MAXREL = 9;
#pragma unroll
      for (int s2 = 0; s2 < (MAXREL + 1) / 2; ++s2) {
        const int rr = 2 * s2 + lhi;
        const int j = q + rr - p.window;
        const float av = (rr < nrel && d0 < dk) ? RV[rr * dk + d0] : 0.f;
        const float bvv = (rr < nrel && q < T && j >= 0 && j < T) ? S[l31 * SP + j] : 0.f;
        oacc = pe_mfma_32x32x2(av, bvv, oacc);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (d < dk && q < T) p.out[(long)b * p.o_bs + (long)(h * dk + d) * p.o_cs + q] = oacc[r];
      }
    }
  }
  PE_STAMP(0, 20);
}

// ------------------------------------------------------------------------------------------------
// LayerNorm over channels (modules.py:23-26) with the fusions the graph needs:
//   MODE 0: out = LN(in)                                   (encoder norm_layers_1/2; in = x + y)
//   MODE 1: out = res + gelu(LN(in))                       (DDSConv second half, modules.py:123-128)
//   MODE 2: out = gelu(LN(dwconv_k(in; dil) + b))          (DDSConv first half, modules.py:120-123)
// One workgroup = 8 time columns x all channels; thread (col = tid&7, rl = tid>>3) keeps channels
// rl, rl+32, ... in registers (C <= 256), so the input is read (and the depthwise conv evaluated) once.
struct LnP {
  const float* in; long i_bs; int i_cs;
  const float* res; long r_bs; int r_cs;
  float* out; long o_bs; int o_cs;
  const float* gamma; const float* beta;
  const float* dw_w; const float* dw_b; int dw_k, dw_dil;
  const int* lens;
  int C;
};
static constexpr int LN_COLS = 8, LN_NV = 8;

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

template <int MODE>
__global__ __launch_bounds__(256) void ln_kernel(LnP p) {
  PE_KTRACE(5);
  __shared__ float red[4][LN_COLS];
  PE_STAMP(5, 0);
  const int b = blockIdx.y, L = p.lens[b];
  const int t0 = blockIdx.x * LN_COLS;
  if (t0 >= L) return;
  const int col = threadIdx.x & 7, rl = threadIdx.x >> 3, wv = threadIdx.x >> 6;
  const int t = t0 + col;
  const bool ok = t < L;
  const float* ib = p.in + (long)b * p.i_bs;
  float v[LN_NV];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < LN_NV; ++k) {
    const int c = rl + 32 * k;
    float x = 0.f;
    if (ok && c < p.C) {
      if (MODE == 2) {
        x = p.dw_b[c];
        const float* xr = ib + (long)c * p.i_cs;
        const int pad = (p.dw_k - 1) / 2 * p.dw_dil;
        for (int kk = 0; kk < p.dw_k; ++kk) {
          const int tt = t + kk * p.dw_dil - pad;
          if (tt >= 0 && tt < L) x = fmaf(p.dw_w[c * p.dw_k + kk], xr[tt], x);
        }
      } else {
        x = ib[(long)c * p.i_cs + t];
      }
    }
    v[k] = x;
    s += x;
  }
  // reduce over rl: lanes differing in bits 3..5 within the wave, then across the 4 waves
  auto block_sum = [&](float x) -> float {
    x += __shfl_xor(x, 8);
    x += __shfl_xor(x, 16);
    x += __shfl_xor(x, 32);
    __syncthreads();
    if ((threadIdx.x & 63) < LN_COLS) red[wv][col] = x;
    __syncthreads();
    return red[0][col] + red[1][col] + red[2][col] + red[3][col];
  };
  const float mean = block_sum(s) / (float)p.C;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < LN_NV; ++k) {
    const int c = rl + 32 * k;
    if (c < p.C) { const float d = v[k] - mean; q = fmaf(d, d, q); }
  }
  const float var = block_sum(q) / (float)p.C;
  const float rstd = 1.f / sqrtf(var + 1e-5f);
  if (!ok) return;
#pragma unroll
  for (int k = 0; k < LN_NV; ++k) {
    const int c = rl + 32 * k;
    if (c < p.C) {
      float y = (v[k] - mean) * rstd * p.gamma[c] + p.beta[c];
      if (MODE >= 1) y = gelu_erf(y);
      if (MODE == 1) y += p.res[(long)b * p.r_bs + (long)c * p.r_cs + t];
      p.out[(long)b * p.o_bs + (long)c * p.o_cs + t] = y;
    }
  }
  PE_STAMP(5, 1);
}

// ------------------------------------------------------------------------------------------------
// ConvFlow.pre (1 -> H channels, 1x1) fused with DDSConv's "x = x + g" (modules.py:504-505,118-119):
//   h[c][t] = w[c] * z0[t] + b[c] + g[c][t]
__global__ void cf_pre_kernel(const float* z0, long z_bs, const float* w, const float* bia,
                              const float* g, long g_bs, int g_cs, float* out, long o_bs, int o_cs,
                              const int* lens, int H) {
  PE_KTRACE(12);
  const int b = blockIdx.z, c = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= lens[b] || c >= H) return;
  out[(long)b * o_bs + (long)c * o_cs + t] =
      fmaf(w[c], z0[(long)b * z_bs + t], bia[c]) + g[(long)b * g_bs + (long)c * g_cs + t];
}

// ------------------------------------------------------------------------------------------------
// Inverse piecewise rational-quadratic spline with linear tails, 10 bins, bound 5
// (transforms.py:50-98 unconstrained_rational_quadratic_spline(inverse=True) over :101-191;
// the per-position parameters are ConvFlow.proj's 29 outputs, modules.py:508-517): one element.
static constexpr int SPL_NB = 10;
// The cheap, order-sensitive tail: from the un-normalised softmax terms ew / eh (= exp(u - max u)) and the derivatives dv
// to the transformed value. Kept separate so that the ~40 transcendentals in front of it can be spread over lanes
// (dds_layer16_kernel) while the sums keep the reference's sequential order.
__device__ __forceinline__ float spline_finish(const float (&uw)[SPL_NB], const float (&uh)[SPL_NB],
                                               const float (&dv)[SPL_NB + 1], float x) {
  constexpr int NB = SPL_NB;
  constexpr float TB = 5.0f, MINB = 1e-3f;
  float sw = 0.f, sh = 0.f;
  for (int i = 0; i < NB; ++i) { sw += uw[i]; sh += uh[i]; }
  // cumulative widths / heights scaled to [-TB, TB], end knots pinned
  float cw[NB + 1], ch[NB + 1];
  cw[0] = -TB; ch[0] = -TB;
  float aw = 0.f, ah = 0.f;
  for (int i = 0; i < NB; ++i) {
    aw += MINB + (1.f - MINB * NB) * (uw[i] / sw);
    ah += MINB + (1.f - MINB * NB) * (uh[i] / sh);
    cw[i + 1] = 2.f * TB * aw - TB;
    ch[i + 1] = 2.f * TB * ah - TB;
  }
  cw[NB] = TB; ch[NB] = TB;
  // searchsorted on heights (transforms.py:44-47): last edge + 1e-6
  int bin = -1;
  for (int i = 0; i <= NB; ++i) {
    const float e = (i == NB) ? ch[i] + 1e-6f : ch[i];
    bin += (x >= e) ? 1 : 0;
  }
  bin = bin < 0 ? 0 : (bin > NB - 1 ? NB - 1 : bin);
  float in_cw = 0.f, in_w = 0.f, in_ch = 0.f, in_h = 0.f, d0 = 0.f, d1 = 0.f;
  for (int i = 0; i < NB; ++i)
    if (i == bin) {
      in_cw = cw[i]; in_w = cw[i + 1] - cw[i];
      in_ch = ch[i]; in_h = ch[i + 1] - ch[i];
      d0 = dv[i]; d1 = dv[i + 1];
    }
  const float delta = in_h / in_w;
  const float y = x - in_ch;
  const float s = d0 + d1 - 2.f * delta;
  const float a = y * s + in_h * (delta - d0);
  const float bq = in_h * d0 - y * s;
  const float c = -delta * y;
  const float disc = bq * bq - 4.f * a * c;
  const float root = (2.f * c) / (-bq - sqrtf(disc));
  return root * in_w + in_cw;
}
// derivative i of the spline (0 and NB are the linear tails' constant): min + softplus(u)
__device__ __forceinline__ float spline_deriv(float u, bool boundary) {
  constexpr float MIND = 1e-3f;
  // boundary u = log(exp(1-min)-1) -> derivative exactly ~1
  if (boundary) u = logf(expf(1.f - MIND) - 1.f);
  return MIND + (u > 20.f ? u : log1pf(expf(u)));
}
__device__ __forceinline__ float spline_inverse(const float (&raw)[3 * SPL_NB - 1], float x, float inv_sqrt_h) {
  constexpr int NB = SPL_NB;
  constexpr float TB = 5.0f;
  if (!(x >= -TB && x <= TB)) return x;          // identity outside the interval
  float uw[NB], uh[NB], dv[NB + 1];
  float mw = -3.0e38f, mh = -3.0e38f;
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    uw[i] = raw[i] * inv_sqrt_h;
    uh[i] = raw[NB + i] * inv_sqrt_h;
    mw = fmaxf(mw, uw[i]);
    mh = fmaxf(mh, uh[i]);
  }
  for (int i = 0; i < NB; ++i) {
    uw[i] = expf(uw[i] - mw);
    uh[i] = expf(uh[i] - mh);
  }
  for (int i = 0; i <= NB; ++i) dv[i] = spline_deriv((i == 0 || i == NB) ? 0.f : raw[2 * NB + i - 1], i == 0 || i == NB);
  return spline_finish(uw, uh, dv, x);
}
// One thread per (utterance, position). z1 is transformed in place; z0 is the untouched half.
__global__ void spline_inverse_kernel(const float* hproj, long h_bs, int h_cs, float* z1, long z_bs,
                                      const int* lens, float inv_sqrt_h) {
  PE_KTRACE(21);
  const int b = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= lens[b]) return;
  // all 3*NB-1 spline parameters of this element are requested together with x (one memory round trip)
  const float* hp = hproj + (long)b * h_bs + t;
  float raw[3 * SPL_NB - 1];
#pragma unroll
  for (int i = 0; i < 3 * SPL_NB - 1; ++i) raw[i] = hp[(long)i * h_cs];
  const float x = z1[(long)b * z_bs + t];
  z1[(long)b * z_bs + t] = spline_inverse(raw, x, inv_sqrt_h);
}

// ------------------------------------------------------------------------------------------------
// One whole DDSConv layer (modules.py:119-128) per launch (out of place: neighbouring workgroups read each
// other's halo columns of x, so the result goes to a second buffer):
//     out = x + gelu(LN2(W1x1 . gelu(LN1(dwconv_dil(x) + b_dw)) + b_1x1))
// A workgroup (8 waves) owns 32 time columns and all H <= 256 channels: the depthwise conv + LN1 + GELU run in
// registers (thread = (column, channel lane), 16 channel lanes), the 1x1 conv is an [H x H] x [H x 32] GEMM on
// the f32 MFMAs with the activations in LDS and the pre-packed weights read from L2, LN2 + GELU + residual
// read the GEMM tile back from LDS. Replaces three launches (ln_kernel<2>, conv, ln_kernel<1>) and two
// round trips of the [H x T] activations through memory.
struct DdsP {
  const float* x; long x_bs; int x_cs;
  float* out; long o_bs; int o_cs;
  const float* dw_w; const float* dw_b; int dw_k, dw_dil;
  const float* g1; const float* b1; const float* g2; const float* b2;
  const float* bias;                        // 1x1 conv bias
  const float* wp16;                        // 1x1 conv weights in the 16x16x4 fragment order (engine.cpp)
  int nchunks;                              // ceil(H / 32)
  const int* lens;
  int H;
  // Optional fold of ConvFlow.pre + DDSConv's "x = x + g" into the layer input (modules.py:504-505, 118-119), first
  // layer of a ConvFlow: the input is  pre_w[c] * (z0[t] * z_scale) + pre_b[c] + x[c][t]  with x = the conditioning g.
  const float* pre_z; long pre_z_bs;        // z0 row of utterance b (null: no fold)
  const float* pre_w; const float* pre_b;
  float z_scale;                            // noise_scale_w on the first flow (z is still the raw N(0,1) draw), else 1
  // Optional second 1x1 conv on the layer's output columns (last layer of a DDSConv: dp.proj / ConvFlow.proj,
  // models.py:65, modules.py:507), weights in the 16x16x4 fragment order; the layer output itself is then not stored.
  const float* post_w16; const float* post_bias; int post_rows;
  float* post_out; long po_bs; int po_cs;   // plain store of the post conv (dp.proj), or null
  // Optional spline epilogue (ConvFlow, modules.py:508-526): the post conv's 29 rows are the per-position parameters;
  // z1 <- rq_spline_inverse(z1 * z_scale), z0 <- z0 * z_scale (pass-through), both [2][Ts] tensors may alias.
  const float* zin; long zin_bs; int z_cs; int c0, c1;
  float* zout; long zout_bs;
  float inv_sqrt_h;
  // Halo exchange between the column tiles of ONE launch (dp_persist_kernel): a tile reads its own columns from the
  // tensors above and its neighbours' boundary columns from 8-byte {tag, value} granules the neighbours publish --
  // [utterance][tile][slot][side][channel][9]; side 0 = the owner's columns 0..8, side 1 = its columns 7..15.
  // Slots 0 / 1: layer outputs (alternating), 2: the conditioning g (dp.proj output). Tags = epoch base + layer + 1.
  // The flow variable z ([2][T]): [utterance][tile][parity][side][row] granules of its columns 0 and 15.
  // (The arenas themselves are in DdsG, shared by the layers; per layer only these few bytes.)
  signed char gin_slot;                     // input halo (-1: the input was written by an EARLIER kernel: plain loads)
  signed char gout_slot, gout_d;            // layer output: slot (-1: not published), columns per side the reader needs
  signed char pg_slot;                      // post_out (g) boundary columns (-1: none)
  signed char zin_par, zout_par;            // z read by the folded ConvFlow.pre (-1: written by an earlier kernel) / published
  signed char zin_row;                      // physical row of z the folded ConvFlow.pre reads (pre_z points at it)
  unsigned char gin_tag, gout_tag, pg_tag, zin_tag, zout_tag;
};
struct DdsG {
  unsigned long long* gx; long gx_bs; int gx_ts;
  unsigned long long* gz; long gz_bs; int gz_ts;
};
static constexpr int DDS_HALO = 9;          // widest depthwise halo of the duration predictor (kernel 3, dilation 9)
__device__ __forceinline__ unsigned long long pe_gran(unsigned tag, float v) {
  return ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
}
// Sum over the 32 channel lanes x 8 waves that share a column (512-thread, 16-column workgroups): lane pairs by
// shuffle, waves through `red` ([2][8][16] floats). The two halves of `red` alternate between calls, so a call costs
// ONE block barrier: half h is rewritten two calls after it was read, and the barrier of the call in between orders that.
__device__ __forceinline__ float pe_col_sum16(float v, float* red, int& flip, int wv, int lane, int col) {
  constexpr int NC = 16;
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  float* r = red + flip * 8 * NC;
  flip ^= 1;
  if (lane < NC) r[wv * NC + col] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < 8; ++w) s += r[w * NC + col];
  return s;
}

// rows x Kp GEMM over the 16 columns in IN[Kp][16] on the 16x16x4 MFMA; sink(row, col, value + bias[row]).
// Wave w owns the 16-row tiles w and w + 8 (then w + 16, w + 24, ...) and runs such a PAIR together: both weight row
// blocks are requested up front (one memory latency per pair, 2 * NQMAX float4 per lane), the B fragments are read from
// LDS once for both, and the two accumulator chains alternate on the MFMA pipe instead of each waiting on itself.
// k ascends inside and across the instructions of a tile exactly as in a one-tile-at-a-time loop: same fmaf chain.
// EXACT: Kp == 16 * NQMAX is known at compile time (no per-step guards in the unrolled loops: on the common shapes the
// guards were a scalar branch per LDS read, ~200 per launch).
// The weights do not depend on anything the kernel computes: col_gemm16_fetch requests the first pair's row blocks (the
// only pair for <= 256 rows) wherever the caller likes -- at kernel entry, under the phase that produces IN -- and
// col_gemm16<..., PRE = true> starts from them, so the GEMM phase does not open with a memory round trip.
template <int NQMAX>
struct ColW {
  f32x4 w0[NQMAX], w1[NQMAX];
  float bz0[4], bz1[4];
};
template <int NQMAX>
__device__ __forceinline__ void col_gemm16_fetch(ColW<NQMAX>& w, const float* wp16, const float* bias, int nbias,
                                                 int rows, int Kp, int mt, int lane) {
  const int lq = lane >> 4;
  const int nq = Kp / 16, ntile = (rows + 15) / 16, tile_floats = nq * 256;
  const pe_rowsrc biasd = pe_make_row(bias ? bias : wp16, bias ? nbias : 0);
  const bool one = PE_UNIFORM(mt < ntile), two = PE_UNIFORM(mt + 8 < ntile);
  const pe_rowsrc ws0 = pe_make_row_u(wp16 + (long)(one ? mt : 0) * tile_floats, one ? tile_floats : 0);
  const pe_rowsrc ws1 = pe_make_row_u(wp16 + (long)(two ? mt + 8 : 0) * tile_floats, two ? tile_floats : 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    w.bz0[r] = pe_row_load(biasd, one ? mt * 16 + 4 * lq + r : -1);
    w.bz1[r] = pe_row_load(biasd, two ? (mt + 8) * 16 + 4 * lq + r : -1);
  }
#pragma unroll
  for (int qq = 0; qq < NQMAX; ++qq) w.w0[qq] = pe_row_load4(ws0, qq * 256 + lane * 4);     // past nq: zeros
#pragma unroll
  for (int qq = 0; qq < NQMAX; ++qq) w.w1[qq] = pe_row_load4(ws1, qq * 256 + lane * 4);
  PE_SCHED_FENCE();
}
template <int NQMAX, bool EXACT, bool PRE = false, class Sink>
__device__ __forceinline__ void col_gemm16(const float* wp16, const float* bias, int nbias, int rows, int Kp_rt,
                                           const float* IN, int wv, int lane, Sink&& sink, ColW<NQMAX>* pre = nullptr) {
  constexpr int NC = 16;
  const int l15 = lane & 15, lq = lane >> 4;
  const int Kp = EXACT ? 16 * NQMAX : Kp_rt;
  const int nq = Kp / 16, ntile = (rows + 15) / 16;
  auto run_pair = [&](const int mt, const ColW<NQMAX>& W) {
    const bool two = PE_UNIFORM(mt + 8 < ntile);
    f32x4 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc0[r] = acc1[r] = 0.f;
#pragma unroll
    for (int q0 = 0; q0 < NQMAX; q0 += 4) {
      if (q0 < nq) {
        float yv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) yv[u] = (4 * q0 + u < Kp / 4) ? IN[(4 * (4 * q0 + u) + lq) * NC + l15] : 0.f;
        PE_SCHED_FENCE();
        if (two) {
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            acc0 = pe_mfma_16x16x4(W.w0[q0 + (u >> 2)][u & 3], yv[u], acc0);
            acc1 = pe_mfma_16x16x4(W.w1[q0 + (u >> 2)][u & 3], yv[u], acc1);
          }
        } else {
#pragma unroll
          for (int u = 0; u < 16; ++u) acc0 = pe_mfma_16x16x4(W.w0[q0 + (u >> 2)][u & 3], yv[u], acc0);
        }
        PE_SCHED_FENCE();
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) sink(mt * 16 + 4 * lq + r, l15, acc0[r] + W.bz0[r]);
    if (two) {
#pragma unroll
      for (int r = 0; r < 4; ++r) sink((mt + 8) * 16 + 4 * lq + r, l15, acc1[r] + W.bz1[r]);
    }
  };
  int mt = wv;
  if (PRE) {
    if (mt < ntile) run_pair(mt, *pre);
    mt += 16;
  }
  for (; mt < ntile; mt += 16) {
    ColW<NQMAX> wl;
    col_gemm16_fetch<NQMAX>(wl, wp16, bias, nbias, rows, Kp, mt, lane);
    run_pair(mt, wl);
  }
}

// One workgroup = 16 time columns x all channels, 512 threads; the 1x1 conv runs on the 16x16x4 f32 MFMA: the GEMM's
// N matches the column count, its 16-row tiles (Hp/16 = 12 for H = 192) spread evenly over the four SIMDs of the 8
// waves (three each), and a 128-id utterance still gives 8 workgroups. (A first version used 32 columns and the
// 32x32x2 MFMA: six row tiles on eight waves put two tiles on two of the SIMDs; 17.3 vs 11.0 us per launch.) k runs
// over the input channels in ascending order inside and across the instructions: the same fmaf chain. Weights: packed by engine.cpp pack_dds16 as
// [16-row tile][q][lane][4] with lane -> (row = lane & 15, k = lane >> 4) and step s = 4q + j covering ci = 4s + k.
// SC1: the layer's activations travel between workgroups of ONE launch (dp_persist_kernel): agent-scope loads / stores.
template <int NVT, bool SC1>                    // NVT = channel slots per thread: ceil(Hp / 32)
__device__ __forceinline__ void dds_layer16_body(const DdsP& p, int ctile, int b, float* sm, const DdsG* G = nullptr,
                                                 unsigned gbase = 0, int* gerr = nullptr) {
  constexpr int NC = 16;                        // sm: Y[Hp][16] | Z[Hp][16] | red[2][8][16]
  PE_STAMP(2, 0);
  // The utterance length lives in device memory. As a separate launch (!SC1) nothing below uses it until every operand
  // load has been issued against the row stride instead (Lb): its latency overlaps theirs, and the taps beyond the
  // length are zeroed afterwards -- the conv's zero padding at the end of the utterance.
  const int L = p.lens[b];
  const int t0 = ctile * NC;
  if (SC1 && t0 >= L) return;
  const int Lb = SC1 ? L : p.x_cs;
  // SC1 = the layer runs inside dp_persist_kernel: its own columns travel through memory between the waves of this
  // workgroup only (plain stores and loads: a CU's vector L1 is coherent for its own waves, workgroup scope needs no
  // cache policy), its neighbours' boundary columns arrive as granules, and only z -- which the duration step of ANOTHER
  // workgroup reads at the end -- uses agent-scope accesses.
  auto ldx = [&](const pe_rowsrc& r, int idx) { return pe_row_load(r, idx); };
  auto stg = [&](float* q, float v) { *q = v; };
  auto stz = [&](float* q, float v) { if (SC1) pe_st_sc1(q, v); else *q = v; };
  // NVT = 3 / 6: instantiated for exactly Hp = 32 * NVT (the launcher checks); NVT = 8 is the generic form (any Hp <= 256)
  const int H = p.H, Hp = NVT != 8 ? 32 * NVT : p.nchunks * 32;
  // granule (slot, side, channel c, j) of column tile `tile`; z granule (parity, side, row)
  auto gxa = [&](int tile, int slot, int side, int c, int j) {
    return G->gx + (long)b * G->gx_bs + (long)tile * G->gx_ts + ((long)(slot * 2 + side) * Hp + c) * DDS_HALO + j;
  };
  auto gza = [&](int tile, int par, int side, int row) {
    return G->gz + (long)b * G->gz_bs + (long)tile * G->gz_ts + (par * 2 + side) * 2 + row;
  };
  float* Y = sm;
  float* Z = Y + Hp * NC;
  float* red = Z + Hp * NC;
  const int tid = threadIdx.x, col = tid & 15, rl = tid >> 4, wv = PE_UNIFORM(tid >> 6), lane = tid & 63;
  const int t = t0 + col;
  const bool okb = t < Lb;
  const float* xb = p.x + (long)b * p.x_bs;
  float* ob = p.out + (long)b * p.o_bs;
  const int pad = (p.dw_k - 1) / 2 * p.dw_dil;
  const bool fold = p.pre_z != nullptr;
  // this wave's 1x1-conv weight row blocks: in flight under phase 1 (in the persistent kernel they are requested after
  // the halo granules have arrived: the granule bookkeeping and 96 weight registers do not fit together)
  ColW<2 * NVT> gw;
  if (!SC1) col_gemm16_fetch<2 * NVT>(gw, p.wp16, p.bias, H, Hp, Hp, wv, lane);

  int red_flip = 0;
  auto col_sum = [&](float x) -> float { return pe_col_sum16(x, red, red_flip, wv, lane, col); };

  // ---- phase 1: depthwise conv, LN1, GELU -> Y (all operands requested up front through descriptors)
  constexpr int MAXK = 3;
  const pe_rowsrc xd = pe_make_row(xb, H * p.x_cs);
  const pe_rowsrc wd = pe_make_row(p.dw_w, H * p.dw_k), bd = pe_make_row(p.dw_b, H);
  const pe_rowsrc g1d = pe_make_row(p.g1, H), b1d = pe_make_row(p.b1, H);
  float v[NVT], xc[NVT], gg[NVT], bb[NVT];
  bool ok;                                        // t < L, set once the operand loads are in flight
  {
    float xv[NVT][MAXK], ww[NVT][MAXK], wb[NVT];
    // folded ConvFlow.pre: the three taps' z0 values and this channel's (w, b); zero-length descriptors when unused
    const pe_rowsrc zd = pe_make_row(fold ? p.pre_z + (long)b * p.pre_z_bs : p.dw_b, fold ? Lb : 0);
    const pe_rowsrc pwd = pe_make_row(fold ? p.pre_w : p.dw_b, fold ? H : 0), pbd = pe_make_row(fold ? p.pre_b : p.dw_b, fold ? H : 0);
    float zt[MAXK], pw[NVT], pb[NVT];
    // taps outside this workgroup's 16 columns (kk = 0 and kk = 2 only: the halo is at most 9 columns) come from
    // the neighbours' granules when the producer ran in THIS launch
    const bool xhalo = SC1 && p.gin_slot >= 0, zhalo = SC1 && fold && p.zin_par >= 0;
    auto own = [&](int tt) { return tt >= t0 && tt < t0 + NC; };
#pragma unroll
    for (int kk = 0; kk < MAXK; ++kk) {
      const int tt = t + kk * p.dw_dil - pad;
      const bool tv = okb && kk < p.dw_k && tt >= 0 && tt < Lb && (!zhalo || own(tt));
      zt[kk] = (SC1 && fold && p.zin_par >= 0 ? (tv ? pe_ld_sc1(p.pre_z + (long)b * p.pre_z_bs + tt) : 0.f)
                                              : ldx(zd, tv ? tt : -1)) * p.z_scale;
    }
#pragma unroll
    for (int k = 0; k < NVT; ++k) {
      const int c = rl + 32 * k;
      const bool cv = okb && c < H;
#pragma unroll
      for (int kk = 0; kk < MAXK; ++kk) {
        const int tt = t + kk * p.dw_dil - pad;
        const bool tv = cv && kk < p.dw_k && tt >= 0 && tt < Lb;
        xv[k][kk] = ldx(xd, (tv && (!xhalo || own(tt))) ? c * p.x_cs + tt : -1);
        ww[k][kk] = pe_row_load(wd, tv ? c * p.dw_k + kk : -1);
      }
      wb[k] = pe_row_load(bd, cv ? c : -1);
      gg[k] = pe_row_load(g1d, c < H ? c : -1);
      bb[k] = pe_row_load(b1d, c < H ? c : -1);
      pw[k] = pe_row_load(pwd, cv ? c : -1);
      pb[k] = pe_row_load(pbd, cv ? c : -1);
    }
    if (!SC1) {
      // first use of the length
      if (t0 >= L) return;
      PE_STAMP(2, 1);
#pragma unroll
      for (int kk = 0; kk < MAXK; ++kk) {
        const int tt = t + kk * p.dw_dil - pad;
        const bool in = t < L && tt < L;
        zt[kk] = in ? zt[kk] : 0.f;
#pragma unroll
        for (int k = 0; k < NVT; ++k) xv[k][kk] = in ? xv[k][kk] : 0.f;
      }
    }
    ok = t < L;
    if (SC1 && (xhalo || zhalo)) {
      // all granule loads go out together; the ones whose tag is still old are re-read until it arrives. Slot k = NVT
      // is z; bit 2k + h of `need` = tap h (0: left, 1: right) of slot k comes from a neighbour.
      auto gaddr = [&](int k, int h) -> const unsigned long long* {
        const int tt = t + 2 * h * p.dw_dil - pad;
        const int side = h == 0 ? 1 : 0, tile = ctile + (h == 0 ? -1 : 1);
        const int j = h == 0 ? tt - (t0 - NC) - (NC - DDS_HALO) : tt - (t0 + NC);
        return k == NVT ? gza(tile, p.zin_par, side, p.zin_row) : gxa(tile, p.gin_slot, side, rl + 32 * k, j);
      };
      unsigned long long gv[NVT + 1][2];
      unsigned need = 0;
      const unsigned wantx = gbase + p.gin_tag, wantz = gbase + p.zin_tag;
#pragma unroll
      for (int k = 0; k <= NVT; ++k)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int tt = t + 2 * h * p.dw_dil - pad;
          const bool nd = ok && p.dw_k == 3 && tt >= 0 && tt < L && !own(tt) && (k == NVT ? zhalo : (xhalo && rl + 32 * k < H));
          if (nd) need |= 1u << (2 * k + h);
          gv[k][h] = nd ? pe_ld_gran(gaddr(k, h)) : 0ull;
        }
      for (long spins = 0;; ++spins) {
        bool all = true;
#pragma unroll
        for (int k = 0; k <= NVT; ++k)
#pragma unroll
          for (int h = 0; h < 2; ++h)
            if (((need >> (2 * k + h)) & 1u) && (unsigned)(gv[k][h] >> 32) < (k == NVT ? wantz : wantx)) {
              all = false;
              gv[k][h] = pe_ld_gran(gaddr(k, h));
            }
        if (all) break;
        pe_spin_pause();
        if (spins > (1L << 23)) { if (gerr) *gerr = 1; break; }
      }
#pragma unroll
      for (int k = 0; k < NVT; ++k)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          if ((need >> (2 * k + h)) & 1u) xv[k][2 * h] = __uint_as_float((unsigned)gv[k][h]);
#pragma unroll
      for (int h = 0; h < 2; ++h)
        if ((need >> (2 * NVT + h)) & 1u) zt[2 * h] = __uint_as_float((unsigned)gv[NVT][h]) * p.z_scale;
    }
    if (SC1) col_gemm16_fetch<2 * NVT>(gw, p.wp16, p.bias, H, Hp, Hp, wv, lane);
    if (fold) {
#pragma unroll
      for (int k = 0; k < NVT; ++k)
#pragma unroll
        for (int kk = 0; kk < MAXK; ++kk) {
          const int tt = t + kk * p.dw_dil - pad;
          const bool tv = ok && rl + 32 * k < H && kk < p.dw_k && tt >= 0 && tt < L;
          xv[k][kk] = tv ? fmaf(pw[k], zt[kk], pb[k]) + xv[k][kk] : 0.f;
        }
    }
#pragma unroll
    for (int k = 0; k < NVT; ++k) {
      float a = wb[k];
#pragma unroll
      for (int kk = 0; kk < MAXK; ++kk) a = fmaf(ww[k][kk], xv[k][kk], a);
      v[k] = a;
      xc[k] = xv[k][(MAXK - 1) / 2];     // centre tap = x[c][t] (odd kernel, "same" padding)
    }
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NVT; ++k) s += v[k];
  PE_STAMP(2, 2);
  float mean = col_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < NVT; ++k)
    if (rl + 32 * k < H) { const float d = v[k] - mean; q = fmaf(d, d, q); }
  float rstd = 1.f / sqrtf(col_sum(q) / (float)H + 1e-5f);
  PE_STAMP(2, 3);
#pragma unroll
  for (int k = 0; k < NVT; ++k) {
    const int c = rl + 32 * k;
    if (c < Hp) Y[c * NC + col] = (c < H && ok) ? gelu_erf((v[k] - mean) * rstd * gg[k] + bb[k]) : 0.f;
  }
  // LN2 gains: needed in phase 3, in flight during the GEMM
  const pe_rowsrc g2d = pe_make_row(p.g2, H), b2d = pe_make_row(p.b2, H);
#pragma unroll
  for (int k = 0; k < NVT; ++k) {
    const int c = rl + 32 * k;
    gg[k] = pe_row_load(g2d, c < H ? c : -1);
    bb[k] = pe_row_load(b2d, c < H ? c : -1);
  }
  __syncthreads();

  // ---- phase 2: Z = W1x1 . Y + bias on 16x16x4 MFMAs (col_gemm16: tiles w and w+8 of a wave run as a pair)
  PE_STAMP(2, 4);
  col_gemm16<2 * NVT, NVT != 8, true>(p.wp16, p.bias, H, Hp, Hp, Y, wv, lane, [&](int row, int cc, float val) { Z[row * NC + cc] = val; }, &gw);
  PE_STAMP(2, 5);
  __syncthreads();
  PE_STAMP(2, 6);

  // ---- phase 3: LN2, GELU, residual -> out
  s = 0.f;
#pragma unroll
  for (int k = 0; k < NVT; ++k) {
    const int c = rl + 32 * k;
    v[k] = (c < H) ? Z[c * NC + col] : 0.f;
    s += v[k];
  }
  mean = col_sum(s) / (float)H;
  q = 0.f;
#pragma unroll
  for (int k = 0; k < NVT; ++k)
    if (rl + 32 * k < H) { const float d = v[k] - mean; q = fmaf(d, d, q); }
  rstd = 1.f / sqrtf(col_sum(q) / (float)H + 1e-5f);
  if (p.post_w16 == nullptr) {
    if (!ok) return;
#pragma unroll
    for (int k = 0; k < NVT; ++k) {
      const int c = rl + 32 * k;
      if (c < H) {
        const float y = xc[k] + gelu_erf((v[k] - mean) * rstd * gg[k] + bb[k]);
        stg(ob + (long)c * p.o_cs + t, y);
        if (SC1 && p.gout_slot >= 0) {       // boundary columns for the neighbours' next layer
          if (col < p.gout_d) pe_st_gran(gxa(ctile, p.gout_slot, 0, c, col), pe_gran(gbase + p.gout_tag, y));
          if (col >= NC - p.gout_d)
            pe_st_gran(gxa(ctile, p.gout_slot, 1, c, col - (NC - DDS_HALO)), pe_gran(gbase + p.gout_tag, y));
        }
      }
    }
    PE_STAMP(2, 7);
    return;
  }
  // ---- phase 4 (last layer of a DDSConv): the following 1x1 conv on this workgroup's columns, Y <- layer output
#pragma unroll
  for (int k = 0; k < NVT; ++k) {
    const int c = rl + 32 * k;
    if (c < Hp) Y[c * NC + col] = (c < H && ok) ? xc[k] + gelu_erf((v[k] - mean) * rstd * gg[k] + bb[k]) : 0.f;
  }
  __syncthreads();
  col_gemm16<2 * NVT, NVT != 8>(p.post_w16, p.post_bias, p.post_rows, p.post_rows, Hp, Y, wv, lane,
                      [&](int row, int cc, float val) { Z[row * NC + cc] = val; });
  __syncthreads();
  if (p.post_out && ok) {
    float* po = p.post_out + (long)b * p.po_bs;
#pragma unroll
    for (int k = 0; k < NVT; ++k) {
      const int c = rl + 32 * k;
      if (c < p.post_rows) {
        const float y = Z[c * NC + col];
        stg(po + (long)c * p.po_cs + t, y);
        if (SC1 && p.pg_slot >= 0) {         // g is read with a one-column halo by every flow's first layer
          if (col == 0) pe_st_gran(gxa(ctile, p.pg_slot, 0, c, 0), pe_gran(gbase + p.pg_tag, y));
          if (col == NC - 1) pe_st_gran(gxa(ctile, p.pg_slot, 1, c, DDS_HALO - 1), pe_gran(gbase + p.pg_tag, y));
        }
      }
    }
  }
  if (p.zout) {
    // ConvFlow's spline on z1 (z0 passes through, scaled). The ~40 transcendentals of one position are spread over 16
    // lanes (lane j: softmax terms of bin j, derivative j), the order-sensitive sums run in one lane afterwards.
    constexpr int NB = SPL_NB;
    float* S = Y;                                      // Y is free: [16 cols][3][16]
    const int scol = tid >> 4, j = tid & 15;           // first 256 threads: 16 consecutive lanes per column
    const int st = t0 + scol;
    if (tid < 256) {
      const float uwj = j < NB ? Z[j * NC + scol] * p.inv_sqrt_h : -3.0e38f;
      const float uhj = j < NB ? Z[(NB + j) * NC + scol] * p.inv_sqrt_h : -3.0e38f;
      float mw = uwj, mh = uhj;
#pragma unroll
      for (int m = 8; m >= 1; m >>= 1) { mw = fmaxf(mw, __shfl_xor(mw, m)); mh = fmaxf(mh, __shfl_xor(mh, m)); }
      S[(scol * 3 + 0) * 16 + j] = j < NB ? expf(uwj - mw) : 0.f;
      S[(scol * 3 + 1) * 16 + j] = j < NB ? expf(uhj - mh) : 0.f;
      S[(scol * 3 + 2) * 16 + j] = j <= NB ? spline_deriv((j == 0 || j >= NB) ? 0.f : Z[(2 * NB + j - 1) * NC + scol], j == 0 || j >= NB) : 0.f;
    }
    __syncthreads();
    if (tid < 256 && j == 0 && st < L) {
      float uw[NB], uh[NB], dv[NB + 1];
#pragma unroll
      for (int i = 0; i < NB; ++i) { uw[i] = S[(scol * 3 + 0) * 16 + i]; uh[i] = S[(scol * 3 + 1) * 16 + i]; }
#pragma unroll
      for (int i = 0; i <= NB; ++i) dv[i] = S[(scol * 3 + 2) * 16 + i];
      const float* zi = p.zin + (long)b * p.zin_bs;
      float* zo = p.zout + (long)b * p.zout_bs;
      const float x1 = (SC1 ? pe_ld_sc1(zi + (long)p.c1 * p.z_cs + st) : zi[(long)p.c1 * p.z_cs + st]) * p.z_scale;
      const float x0 = (SC1 ? pe_ld_sc1(zi + (long)p.c0 * p.z_cs + st) : zi[(long)p.c0 * p.z_cs + st]) * p.z_scale;
      const float y1 = (x1 >= -5.0f && x1 <= 5.0f) ? spline_finish(uw, uh, dv, x1) : x1;
      stz(zo + (long)p.c1 * p.z_cs + st, y1);
      stz(zo + (long)p.c0 * p.z_cs + st, x0);
      if (SC1 && p.zout_par >= 0 && (scol == 0 || scol == NC - 1)) {
        const int side = scol == 0 ? 0 : 1;
        pe_st_gran(gza(ctile, p.zout_par, side, p.c1), pe_gran(gbase + p.zout_tag, y1));
        pe_st_gran(gza(ctile, p.zout_par, side, p.c0), pe_gran(gbase + p.zout_tag, x0));
      }
    }
  }
  (void)ok;
}

template <int NVT>
__global__ __launch_bounds__(512) void dds_layer16_kernel(DdsP p) {
  PE_KTRACE(2);
  PE_DYN_SMEM(float, sm);
  dds_layer16_body<NVT, false>(p, blockIdx.x, blockIdx.y, sm);
}

// ------------------------------------------------------------------------------------------------
// Short chains of 1x1 convs whose GEMMs are small enough for one workgroup to own ALL output rows of a 16-column tile,
// so that what follows the GEMM (a LayerNorm over channels, or a second GEMM over the result) needs no second launch
// and no trip through HBM. Small batches only (the tiled conv kernels win when there are columns to fill the chip):
//   mode 0   out = LN(res + W1.in + b1)              attention conv_o + residual + norm_layers_1 (attentions.py:70-72)
//   mode 1   x1 -= W1.in + b1 ; out2 = W2.x1 + b2    ResidualCouplingLayer.post + mean-only reverse update, then the
//                                                    NEXT coupling layer's pre over the updated half -- the Flip between
//                                                    them is folded into the packed weights (modules.py:455-466, 433)
// Same 16x16x4 MFMA GEMM as dds_layer16_kernel: weights in pack16 order, B operand = the input columns in LDS.
struct ColP {
  const float* in1; long in1_bs; int in1_cs; int K1;
  const float* w1; const float* b1; int rows1;
  int mode;
  const float* res; long res_bs; int res_cs;            // mode 0
  const float* gamma; const float* beta;
  float* out; long out_bs; int out_cs;
  float* x1; long x1_bs; int x1_cs;                     // mode 1 (updated in place)
  const float* w2; const float* b2; int rows2;          // w2 == null: no second GEMM (last coupling layer)
  float* out2; long o2_bs; int o2_cs;
  const int* lens;
};

template <int NVT>                              // NVT = channel slots per thread: every channel count on the chain <= 32 * NVT
__global__ __launch_bounds__(512) void colchain_kernel(ColP p) {
  PE_KTRACE(3);
  constexpr int NC = 16;
  PE_DYN_SMEM(float, sm);                       // IN[32 NVT][16] | Z[32 NVT][16] | red[2][8][16]
  PE_STAMP(p.mode == 0 ? 3 : (p.w2 ? 6 : 7), 0);
  const int b = blockIdx.y, L = p.lens[b];
  const int t0 = blockIdx.x * NC;
  if (t0 >= L) return;
  PE_STAMP(p.mode == 0 ? 3 : (p.w2 ? 6 : 7), 1);
  float* IN = sm;
  float* Z = IN + 32 * NVT * NC;
  float* red = Z + 32 * NVT * NC;
  const int tid = threadIdx.x, col = tid & 15, rl = tid >> 4, wv = PE_UNIFORM(tid >> 6), lane = tid & 63;
  const int t = t0 + col;
  const bool ok = t < L;
  constexpr int K1p = 32 * NVT;                  // the launcher checks: K1 == 32 NVT, and rows1 == 16 NVT in mode 1
  ColW<2 * NVT> gw;                              // first GEMM's weight row blocks, in flight under the input staging
  col_gemm16_fetch<2 * NVT>(gw, p.w1, p.b1, p.rows1, p.rows1, K1p, wv, lane);

  // operands of the step after the first GEMM are requested before it: residual / previous x1, LN gains
  float ov[NVT], gg[NVT], bb[NVT];
  {
    const pe_rowsrc ind = pe_make_row(p.in1 + (long)b * p.in1_bs, p.K1 * p.in1_cs);
    const int nrow = p.mode == 0 ? p.rows1 : p.rows1;
    const float* ob = p.mode == 0 ? p.res + (long)b * p.res_bs : p.x1 + (long)b * p.x1_bs;
    const int ocs = p.mode == 0 ? p.res_cs : p.x1_cs;
    const pe_rowsrc od = pe_make_row(ob, nrow * ocs);
    const pe_rowsrc gd = pe_make_row(p.mode == 0 ? p.gamma : p.w1, p.mode == 0 ? p.rows1 : 0);
    const pe_rowsrc bd = pe_make_row(p.mode == 0 ? p.beta : p.w1, p.mode == 0 ? p.rows1 : 0);
    float xin[NVT];
#pragma unroll
    for (int k = 0; k < NVT; ++k) {
      const int c = rl + 32 * k;
      xin[k] = pe_row_load(ind, (ok && c < p.K1) ? c * p.in1_cs + t : -1);
      ov[k] = pe_row_load(od, (ok && c < nrow) ? c * ocs + t : -1);
      gg[k] = pe_row_load(gd, c < p.rows1 ? c : -1);
      bb[k] = pe_row_load(bd, c < p.rows1 ? c : -1);
    }
#pragma unroll
    for (int k = 0; k < NVT; ++k) {
      const int c = rl + 32 * k;
      if (c < K1p) IN[c * NC + col] = xin[k];
    }
  }
  PE_STAMP(p.mode == 0 ? 3 : (p.w2 ? 6 : 7), 2);
  __syncthreads();
  PE_STAMP(p.mode == 0 ? 3 : (p.w2 ? 6 : 7), 3);
  col_gemm16<2 * NVT, true, true>(p.w1, p.b1, p.rows1, p.rows1, K1p, IN, wv, lane, [&](int row, int cc, float v) { Z[row * NC + cc] = v; }, &gw);
  // second GEMM's weights (the next layer's pre): in flight under the x1 update
  ColW<NVT> gw2;
  constexpr int K2p = 16 * NVT;
  const bool second = p.mode == 1 && p.w2;
  if (second) col_gemm16_fetch<NVT>(gw2, p.w2, p.b2, p.rows2, p.rows2, K2p, wv, lane);
  PE_STAMP(p.mode == 0 ? 3 : (p.w2 ? 6 : 7), 4);
  __syncthreads();
  PE_STAMP(p.mode == 0 ? 3 : (p.w2 ? 6 : 7), 5);

  if (p.mode == 0) {
    int red_flip = 0;
    auto col_sum = [&](float x) -> float { return pe_col_sum16(x, red, red_flip, wv, lane, col); };
    const int H = p.rows1;
    float v[NVT];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NVT; ++k) {
      const int c = rl + 32 * k;
      v[k] = (ok && c < H) ? Z[c * NC + col] + ov[k] : 0.f;
      s += v[k];
    }
    const float mean = col_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NVT; ++k)
      if (rl + 32 * k < H) { const float d = v[k] - mean; q = fmaf(d, d, q); }
    const float rstd = 1.f / sqrtf(col_sum(q) / (float)H + 1e-5f);
    if (!ok) return;
    float* ob = p.out + (long)b * p.out_bs;
#pragma unroll
    for (int k = 0; k < NVT; ++k) {
      const int c = rl + 32 * k;
      if (c < H) ob[(long)c * p.out_cs + t] = (v[k] - mean) * rstd * gg[k] + bb[k];
    }
    PE_STAMP(p.mode == 0 ? 3 : (p.w2 ? 6 : 7), 6);
    return;
  }

  // mode 1: x1 <- x1 - (post + bias); the updated half is the next layer's x0
  {
    float* xb = p.x1 + (long)b * p.x1_bs;
#pragma unroll
    for (int k = 0; k < NVT; ++k) {
      const int c = rl + 32 * k;
      const float xn = (ok && c < p.rows1) ? ov[k] - Z[c * NC + col] : 0.f;
      if (ok && c < p.rows1) xb[(long)c * p.x1_cs + t] = xn;
      if (c < K2p) IN[c * NC + col] = xn;
    }
    if (!second) return;
    __syncthreads();
    PE_STAMP(p.mode == 0 ? 3 : (p.w2 ? 6 : 7), 7);
    float* o2 = p.out2 + (long)b * p.o2_bs;
    auto st2 = [&](int row, int cc, float v) {
      if (row < p.rows2 && t0 + cc < L) o2[(long)row * p.o2_cs + t0 + cc] = v;
    };
    col_gemm16<NVT, true, true>(p.w2, p.b2, p.rows2, p.rows2, K2p, IN, wv, lane, st2, &gw2);   // K = half the channels
    PE_STAMP(p.mode == 0 ? 3 : (p.w2 ? 6 : 7), 8);
  }
}

// ------------------------------------------------------------------------------------------------
// norm_layers_2 of an encoder layer fused with the 1x1 conv that consumes it -- the next layer's q/k/v conv, or proj
// after the last layer (attentions.py:73-74, 60-69; models.py:207): one workgroup = 16 columns x one 192-row part of
// the GEMM (grid.z = parts: 3 for q/k/v, 2 for proj). Every part normalises its 16 columns itself (cheap next to a
// launch); part 0 also writes LN(y) back for the residual readers. Small batches only, like colchain_kernel.
struct LnGemmP {
  const float* in; long in_bs; int in_cs;        // y = x + ffn(x)
  const float* gamma; const float* beta;
  float* xout; long x_bs; int x_cs;              // LN(y)
  const float* w16; const float* bias; int rows; // pack16 order, all parts; part z owns rows [32 NVT z, 32 NVT (z + 1))
  float* out; long o_bs; int o_cs;
  const int* lens;
};
template <int NVT>                              // channels == 32 * NVT exactly (the launcher checks)
__global__ __launch_bounds__(512) void lngemm_kernel(LnGemmP p) {
  PE_KTRACE(7);
  constexpr int NC = 16, H = 32 * NVT;
  PE_DYN_SMEM(float, sm);                       // IN[H][16] | red[2][8][16]
  const int b = blockIdx.y, L = p.lens[b];
  const int t0 = blockIdx.x * NC;
  if (t0 >= L) return;
  float* IN = sm;
  float* red = IN + H * NC;
  const int tid = threadIdx.x, col = tid & 15, rl = tid >> 4, wv = PE_UNIFORM(tid >> 6), lane = tid & 63;
  const int t = t0 + col;
  const bool ok = t < L;
  const int part = blockIdx.z, row0 = part * H;
  const int rows_here = p.rows - row0 < H ? p.rows - row0 : H;
  const float* wpart = p.w16 + (long)part * (2 * NVT) * (2 * NVT) * 256;      // 2 NVT row tiles of 2 NVT * 256 floats each
  const float* bpart = p.bias ? p.bias + row0 : nullptr;
  ColW<2 * NVT> gw;
  col_gemm16_fetch<2 * NVT>(gw, wpart, bpart, rows_here, rows_here, H, wv, lane);
  float v[NVT], gg[NVT], bb[NVT];
  {
    const pe_rowsrc ind = pe_make_row(p.in + (long)b * p.in_bs, H * p.in_cs);
    const pe_rowsrc gd = pe_make_row(p.gamma, H), bd = pe_make_row(p.beta, H);
#pragma unroll
    for (int k = 0; k < NVT; ++k) {
      const int c = rl + 32 * k;
      v[k] = pe_row_load(ind, ok ? c * p.in_cs + t : -1);
      gg[k] = pe_row_load(gd, c);
      bb[k] = pe_row_load(bd, c);
    }
  }
  int red_flip = 0;
  auto col_sum = [&](float x) -> float { return pe_col_sum16(x, red, red_flip, wv, lane, col); };
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NVT; ++k) s += v[k];
  const float mean = col_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < NVT; ++k) { const float d = v[k] - mean; q = fmaf(d, d, q); }
  const float rstd = 1.f / sqrtf(col_sum(q) / (float)H + 1e-5f);
  float* xo = p.xout + (long)b * p.x_bs;
#pragma unroll
  for (int k = 0; k < NVT; ++k) {
    const int c = rl + 32 * k;
    const float y = ok ? (v[k] - mean) * rstd * gg[k] + bb[k] : 0.f;
    IN[c * NC + col] = y;
    if (part == 0 && ok) xo[(long)c * p.x_cs + t] = y;
  }
  __syncthreads();
  float* ob = p.out + (long)b * p.o_bs + (long)row0 * p.o_cs;
  col_gemm16<2 * NVT, true, true>(wpart, bpart, rows_here, rows_here, H, IN, wv, lane, [&](int row, int cc, float val) {
    if (row < rows_here && t0 + cc < L) ob[(long)row * p.o_cs + t0 + cc] = val;
  }, &gw);
}

// ------------------------------------------------------------------------------------------------
// ElementwiseAffine reverse + durations (modules.py:407-409; models.py:702-704):
//   logw = (z0 - m0) * exp(-logs0); w = exp(logw) * length_scale; d = ceil(w);
//   cum = inclusive prefix sum; frames = max(sum d, 1).   One block per utterance.
// Sums run in 64 bits and are clamped to MAX_FRAMES + 1 (a single duration to 1e6): an absurd length_scale cannot
// overflow `cum`, and the host rejects frames > MAX_FRAMES before sizing stage B from it.
static constexpr int MAX_FRAMES = 60000;      // per-utterance activations stay below the 2 GiB descriptor range
struct DurP {
  const float* z0; long z_bs; float m0, es0, length_scale;
  const int* lens; int* dur; int* cum; int d_bs; int* frames; float* logw_out;
  int* frames_host; int* frames_clamped; int frame_cap;
};
// `part`: 256 long longs of LDS. The first 256 threads of the workgroup work, all of them must call (barriers).
template <bool SC1>
__device__ __forceinline__ void duration_body(const DurP& p, int b, long long* part) {
  const int T = p.lens[b], tid = threadIdx.x;
  const int per = (T + 255) / 256;
  const int lo = tid * per, hi = (lo + per < T) ? lo + per : T;
  long long s = 0;
  if (tid < 256) {
    for (int t = lo; t < hi; ++t) {
      const float zv = SC1 ? pe_ld_sc1(p.z0 + (long)b * p.z_bs + t) : p.z0[(long)b * p.z_bs + t];
      const float logw = (zv - p.m0) * p.es0;
      const float w = expf(logw) * p.length_scale;
      float c = ceilf(w);
      c = c < 0.f ? 0.f : (c > 1.0e6f ? 1.0e6f : c);
      const int d = (int)c;
      p.dur[b * p.d_bs + t] = d;
      if (p.logw_out) p.logw_out[(long)b * p.d_bs + t] = logw;
      s += d;
    }
    part[tid] = s;
  }
  __syncthreads();
  if (tid == 0) {
    long long run = 0;
    for (int i = 0; i < 256; ++i) { const long long v = part[i]; part[i] = run; run += v; }
    const int f = run < 1 ? 1 : (run > MAX_FRAMES ? MAX_FRAMES + 1 : (int)run);
    p.frames[b] = f;
    // the host sizes stage B from this count: written straight into pinned host memory (visible once the stream is
    // synchronised), which saves the device-to-host copy node behind this kernel
    if (p.frames_host) p.frames_host[b] = f;
    // speculative stage B (launched before the host has seen f): lengths clamped to the allocated frame capacity
    p.frames_clamped[b] = f < p.frame_cap ? f : p.frame_cap;
  }
  __syncthreads();
  if (tid < 256) {
    long long run = part[tid];
    for (int t = lo; t < hi; ++t) {
      run += p.dur[b * p.d_bs + t];
      p.cum[b * p.d_bs + t] = run > MAX_FRAMES ? MAX_FRAMES + 1 : (int)run;
    }
  }
}
__global__ __launch_bounds__(256) void duration_kernel(DurP p) {
  PE_KTRACE(13);
  __shared__ long long part[256];
  duration_body<false>(p, blockIdx.x, part);
}

// ------------------------------------------------------------------------------------------------
// The stochastic duration predictor's DDSConv chain as ONE launch (models.py:63-71,108-117; modules.py:117-129,
// 496-527): every DDSConv layer of dp.convs and of the ConvFlows (with ConvFlow.pre folded in and dp.proj / proj +
// spline fused behind, see dds_layer16_body) and the duration step at the end. A workgroup owns 16 time columns of one
// utterance for the whole chain and keeps reading / writing them itself (plain accesses, own L2). The only thing a layer
// needs from other workgroups is the depthwise conv's halo -- <= 9 boundary columns of the two neighbouring tiles -- and
// those travel as 8-byte {tag, value} granules: the producer stores them (relaxed, agent scope) right behind its own
// columns, the consumer issues all its granule loads at once and re-reads the ones whose tag is still old. The data is
// its own flag: ONE fabric round trip per layer, no drain, no fence, no progress word (a first version -- payload,
// drain, flag, poll, reload -- cost three round trips per layer and was no faster than 12 launches,
// profiles/r02_notes.md). Tags = epoch * 64 + layer + 1 only ever grow inside a slot, so nothing is reset between runs;
// the epoch lives in device memory and is advanced by the workgroup that finishes last (the host clears the arenas
// every 2^24 runs, long before the 32-bit tag wraps). Slot reuse is safe because a tile can publish layer l + 2 only
// after it has read its neighbours' layer l + 1, which they published after reading this tile's layer l. Residency: the
// host launches this kernel only for grids of at most one workgroup per CU. A neighbour that never shows up (cannot
// happen on a resident grid) ends the spin after ~1e7 polls with an error code instead of a hang.
static constexpr int DP_MAX_LAYERS = 12;
struct DpPersistP {
  DdsP layer[DP_MAX_LAYERS];
  int nlayers;
  DurP dur;
  DdsG g;                               // halo granule arenas
  unsigned* state;                      // [0] epoch, [1] finished workgroups of this run, [2] error code, [4 + b] finished tiles of utterance b
  int* err_host;                        // pinned host word: set when a neighbour wait gave up
};
static_assert(sizeof(DpPersistP) <= 4096, "kernel arguments are limited to 4 KB");
template <int NVT>
__global__ __launch_bounds__(512) void dp_persist_kernel(DpPersistP p) {
  PE_KTRACE(14);
  PE_DYN_SMEM(float, sm);
  const int ct = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int L = p.layer[0].lens[b];
  const int nct = (L + 15) / 16;
  unsigned* flag = reinterpret_cast<unsigned*>(sm);          // LDS word 0: broadcast slot (the layer body uses sm from word 16 on)
  float* lsm = sm + 16;
  if (tid == 0) flag[0] = pe_ld_flag(p.state);
  __syncthreads();
  const unsigned epoch = flag[0];
  const unsigned base = epoch * 64u;
  __syncthreads();
  if (ct < nct) {
    int gerr = 0;
    for (int l = 0; l < p.nlayers; ++l) {
      dds_layer16_body<NVT, true>(p.layer[l], ct, b, lsm, &p.g, base, &gerr);
      // the layer's own columns are re-read by other waves of this workgroup in the next layer (plain stores -> plain loads)
      pe_drain_stores();
      __syncthreads();
    }
    if (gerr) { pe_st_flag(p.state + 2, 1u); *p.err_host = 1; }
    // ---- durations: the workgroup that completes the utterance's last tile (every spline epilogue is in memory then)
    if (tid == 0) flag[0] = pe_atomic_inc(p.state + 4 + b);
    __syncthreads();
    const bool last_of_utt = flag[0] == (unsigned)(nct - 1);        // per-run counters: the run's last workgroup zeroes them
    __syncthreads();
    if (last_of_utt) duration_body<true>(p.dur, b, reinterpret_cast<long long*>(lsm));
  }
  // ---- the last workgroup of the run advances the epoch
  __syncthreads();
  if (tid == 0) {
    const unsigned done = pe_atomic_inc(p.state + 1);
    if (done == (unsigned)(gridDim.x * gridDim.y) - 1u) {
      for (unsigned i = 0; i < gridDim.y; ++i) pe_st_flag(p.state + 4 + i, 0u);
      pe_st_flag(p.state + 1, 0u);
      pe_st_flag(p.state, epoch + 1u);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Counter-based N(0,1) generator for the two sampling sites (models.py:111 and :718) when the caller
// does not inject noise: Philox-4x32-10 keyed by the engine seed, Box-Muller on the four outputs.
__device__ __forceinline__ void philox4x32(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                           unsigned k1, unsigned* o) {
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
// state = {seed, call counter} in device memory (so a captured graph draws fresh noise on replay);
// site 0 = duration noise, 1 = prior noise.
// the four draws of counter block q (elements 4q .. 4q+3 of a site's flat stream)
__device__ __forceinline__ void randn4(long q, const unsigned long long* state, int site, float (&g)[4]) {
  const unsigned long long seed = state[0], stream = state[1] * 2ull + (unsigned long long)site;
  unsigned r[4];
  philox4x32((unsigned)q, (unsigned)((unsigned long long)q >> 32), (unsigned)stream, (unsigned)(stream >> 32),
             (unsigned)seed, (unsigned)(seed >> 32), r);
  for (int h = 0; h < 2; ++h) {
    const float u1 = ((float)r[2 * h] + 1.0f) * 2.3283064365386963e-10f;   // (0,1]
    const float u2 = (float)r[2 * h + 1] * 2.3283064365386963e-10f;
    const float rad = sqrtf(-2.f * logf(u1));
    g[2 * h] = rad * cosf(6.283185307179586f * u2);
    g[2 * h + 1] = rad * sinf(6.283185307179586f * u2);
  }
}
__global__ void randn_kernel(float* out, long n, const unsigned long long* state, int site) {
  PE_KTRACE(15);
  const long i4 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 >= n) return;
  float g[4];
  randn4(i4 >> 2, state, site, g);
  for (int k = 0; k < 4 && i4 + k < n; ++k) out[i4 + k] = g[k];
}

// ------------------------------------------------------------------------------------------------
// Length regulator + prior sample (models.py:705-718, commons.py:116-129). The reference multiplies
// by a one-hot path matrix; the same result is a gather: frame f takes id i with cum[i-1] <= f < cum[i].
//   z_p[c][f] = m_p[c][i] + noise[c][f] * exp(logs_p[c][i]) * noise_scale
struct RegP {
  const float* stats; long s_bs; int s_cs;     // [B][2C][Ts]: m_p rows [0,C), logs_p rows [C,2C)
  const int* cum; int d_bs;
  const int* tlens; const int* frames;
  const float* noise; long n_bs; int n_cs;     // [B][C][>=F] or null
  float noise_scale;
  float* out; long o_bs; int o_cs;
  int C;
  unsigned* absmax;                            // per-utterance peak accumulator of conv_post_kernel: zeroed here
};
// At batch 1 this launch is a latency chain, so: `cum` is copied to LDS once (the 7-step binary search then never
// leaves the CU) and the 3 x 16 operands of a thread's channels are requested together through row descriptors.
static constexpr int REG_MAXT = 4096;          // ids whose cumulative durations fit the LDS copy; longer: search in global memory
__global__ __launch_bounds__(64) void regulate_kernel(RegP p) {
  PE_KTRACE(16);
  __shared__ int scum[REG_MAXT];
  const int b = blockIdx.z;
  if (p.absmax && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) p.absmax[b] = 0u;
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  const int F = p.frames[b], T = p.tlens[b];
  if ((int)(blockIdx.x * blockDim.x) >= F) return;
  const int* cum = p.cum + b * p.d_bs;
  const bool in_lds = T <= REG_MAXT;
  if (in_lds) {
    for (int i = threadIdx.x; i < T; i += 64) scum[i] = cum[i];
    __syncthreads();
  }
  if (f >= F) return;
  int lo = 0, hi = T;                      // first i with cum[i] > f
  if (in_lds) {
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (scum[mid] > f) hi = mid; else lo = mid + 1;
    }
  } else {
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cum[mid] > f) hi = mid; else lo = mid + 1;
    }
  }
  const bool hit = lo < T;                 // false only when every duration is 0 (frames clamped to 1)
  const int c0 = blockIdx.y * 16;
  const pe_rowsrc sd = pe_make_row(p.stats + (long)b * p.s_bs, 2 * p.C * p.s_cs);
  const pe_rowsrc nd = pe_make_row(p.noise ? p.noise + (long)b * p.n_bs : p.stats, p.noise ? p.C * p.n_cs : 0);
  float m[16], lg[16], nz[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int c = c0 + k;
    const bool cv = c < p.C;
    m[k] = pe_row_load(sd, (hit && cv) ? c * p.s_cs + lo : -1);
    lg[k] = pe_row_load(sd, (hit && cv) ? (p.C + c) * p.s_cs + lo : -1);
    nz[k] = pe_row_load(nd, cv ? c * p.n_cs + f : -1);
  }
  float* ob = p.out + (long)b * p.o_bs + f;
#pragma unroll
  for (int k = 0; k < 16; ++k)
    if (c0 + k < p.C) ob[(long)(c0 + k) * p.o_cs] = m[k] + nz[k] * expf(lg[k]) * p.noise_scale;
}

// ------------------------------------------------------------------------------------------------
// Generator tail (models.py:364-366): leaky_relu(0.01) -> conv_post (k=7, no bias, 1 output channel)
// -> tanh, fused with the per-utterance max|x| that the int16 conversion needs (piper.cpp:410-418).
// HBM-bound (4*Cin bytes in, 4 out per sample), and at batch 1 a latency chain: a workgroup = 256 samples x 4 channel
// groups (one wave each); a thread owns POST_OPT consecutive samples of POST_CU channels per pass (one pass for
// Cin <= 32) and requests all its POST_CU * (POST_OPT + 6) inputs at once through row descriptors (zero padding = range
// check; neighbouring threads' overlap is served by L1). The weights are wave-uniform scalars. The channel-group
// partials meet in LDS and are summed in a fixed order; one peak atomic per workgroup. (A first version walked all channels in one thread: 8 dependent memory round
// trips and 104 workgroups for a 4.8 s utterance, 22.9 us; profiles/r02_notes.md.)
static constexpr int POST_K = 7, POST_OPT = 4, POST_CU = 8, POST_CG = 4, POST_SPB = 64 * POST_OPT;
__global__ __launch_bounds__(256) void conv_post_kernel(const float* x, long x_bs, int x_cs, const float* __restrict__ w,
                                                        int Cin, float slope, const int* lens,
                                                        int len_mul, float* audio, long a_bs,
                                                        unsigned* absmax) {
  PE_KTRACE(17);
  constexpr int NIN = POST_OPT + POST_K - 1;
  __shared__ float part[POST_CG][POST_SPB];
  PE_STAMP(4, 0);
  const int b = blockIdx.y, L = lens[b] * len_mul;
  if (blockIdx.x * POST_SPB >= L) return;
  const int sg = threadIdx.x & 63, cg = PE_UNIFORM(threadIdx.x >> 6);
  const int t0 = blockIdx.x * POST_SPB + sg * POST_OPT;
  const float* xb = x + (long)b * x_bs;
  float acc[POST_OPT];
#pragma unroll
  for (int o = 0; o < POST_OPT; ++o) acc[o] = 0.f;
  for (int c0 = cg * POST_CU; c0 < Cin; c0 += POST_CG * POST_CU) {
    float v[POST_CU][NIN];
#pragma unroll
    for (int cc = 0; cc < POST_CU; ++cc) {
      const pe_rowsrc row = pe_make_row(xb + (long)(c0 + cc) * x_cs, c0 + cc < Cin ? L : 0);
#pragma unroll
      for (int j = 0; j < NIN; ++j) v[cc][j] = pe_row_load(row, t0 - (POST_K - 1) / 2 + j);
    }
#pragma unroll
    for (int cc = 0; cc < POST_CU; ++cc) {
      const float* wc = w + (c0 + cc < Cin ? c0 + cc : 0) * POST_K;
#pragma unroll
      for (int j = 0; j < NIN; ++j) v[cc][j] = pe_lrelu(v[cc][j], slope);
#pragma unroll
      for (int k = 0; k < POST_K; ++k) {
        const float wk = wc[k];
#pragma unroll
        for (int o = 0; o < POST_OPT; ++o) acc[o] = fmaf(wk, v[cc][o + k], acc[o]);   // channel-major, tap-minor
      }
    }
  }
#pragma unroll
  for (int o = 0; o < POST_OPT; ++o) part[cg][sg * POST_OPT + o] = acc[o];
  __shared__ float wmax[4];
  PE_STAMP(4, 1);
  __syncthreads();
  PE_STAMP(4, 2);
  const int t = blockIdx.x * POST_SPB + threadIdx.x;
  float sum = 0.f;
#pragma unroll
  for (int g = 0; g < POST_CG; ++g) sum += part[g][threadIdx.x];
  const float y = tanhf(sum);
  float m = 0.f;
  if (t < L) {
    audio[(long)b * a_bs + t] = y;
    m = fabsf(y);
  }
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  PE_STAMP(4, 3);
  __syncthreads();
  if (threadIdx.x == 0)
    atomicMax(absmax + b, __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
  PE_STAMP(4, 4);
}

// float -> int16 exactly as piper.cpp:420-431 (scale 32767/max(0.01,peak), clamp, truncate)
// `host`: pinned host memory that also receives the samples, utterances packed back to back (zero-copy delivery), or null
__global__ void pcm16_kernel(const float* audio, long a_bs, const unsigned* absmax, const int* lens,
                             int len_mul, short* pcm, long p_bs, short* host) {
  PE_KTRACE(18);
  const int b = blockIdx.y, L = lens[b] * len_mul;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= L) return;
  const float peak = fmaxf(0.01f, __uint_as_float(absmax[b]));
  const float scale = 32767.0f / peak;
  float v = audio[(long)b * a_bs + t] * scale;
  v = fminf(fmaxf(v, -32768.0f), 32767.0f);
  pcm[(long)b * p_bs + t] = (short)v;
  if (host) {
    // the utterances are packed back to back in the host buffer, as pe_result.sample_offsets describes them
    long off = 0;
    for (int u = 0; u < b; ++u) off += (long)lens[u] * len_mul;
    host[off + t] = (short)v;
  }
}

// Streaming decode: copy frames [win[0], win[0]+win[1]) of z [C][zs] into the window buffer [C][ws]
// (window bounds live in device memory so one captured graph serves every chunk).
__global__ void window_copy_kernel(const float* z, int zs, const int* win, float* out, int ws, int C) {
  const int c = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C && t < win[1]) out[(long)c * ws + t] = z[(long)c * zs + win[0] + t];
}

// MRF combine for the parallel-branch schedule: out = ((r0 + r1) + r2) * scale  (models.py:356-363)
__global__ void mrf_sum_kernel(const float* r0, const float* r1, const float* r2, float* out, long bs, int cs,
                               const int* lens, int len_mul, float scale) {
  const int b = blockIdx.z, c = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= lens[b] * len_mul) return;
  const long i = (long)b * bs + (long)c * cs + t;
  float v = r0[i] + r1[i];
  if (r2) v += r2[i];
  out[i] = v * scale;
}

// ------------------------------------------------------------------------------------------------
// One whole MRF stage of the HiFiGAN generator in one launch (models.py:356-363: xs = sum_j resblock_j(x) / n;
// modules.py:301-314 ResBlock1, :355-364 ResBlock2) for the stages with <= 64 channels, which are at the HBM
// ridge when run conv by conv (x, residual and accumulator round trips per conv). A workgroup owns N = 32*NT
// output columns of one utterance: the input window x[n0-hx, n0+N+hx) is staged once in LDS (hx = the widest
// resblock's receptive half-width) and every resblock chain runs out of LDS on shrinking column windows
// (halo recompute), intermediates ping-pong between LDS buffers, the residual is read from LDS in the
// epilogue, and the MRF sum lives in the accumulator registers of the wave that owns the output tile.
// HBM traffic per stage: one read of x (+halo) and one write, instead of ~3 accesses per conv.
// The host flattens the stage into a list of conv steps (engine.cpp: build_mrf); the kernel is agnostic of
// the resblock type. Every step is the same GEMM as conv_mfma_kernel (weights in the same packed fragment
// order, f32 MFMA, k-ordered), so results match the unfused path up to the order of the final MRF adds.
struct MrfStep {
  const float* wp;      // packed conv weights (pack_matrix order)
  const float* bias;
  int ntaps, dil;
  int e;                // columns of halo still needed after this conv (0 for the last conv of a resblock)
  int src, dst, res;    // LDS buffer ids (0 = x, 1, 2); dst < 0: add into the output accumulators; res < 0: none
};
struct MrfP {
  const float* x; long x_bs; int x_cs;
  float* out; long o_bs; int o_cs;
  const int* lens; int len_mul;
  const MrfStep* steps; int nsteps;     // device table
  int C;                                // real channels (<= CP)
  int hx;                               // window halo (the kernel's WS template argument >= N + 2*hx + 32)
  float slope, alpha;
};

template <int CP, int NT, int NW, int WS>
__global__ __launch_bounds__(64 * NW, 2) void mrf_fused_kernel(MrfP p) {
  PE_KTRACE(20);
  constexpr int N = NT * 32, MTL = CP / 32, NCH = CP / KC;
  constexpr int FU = (NT * MTL + NW - 1) / NW;     // output tiles owned by one wave
  constexpr int RG = 4, NCC = WS / 64;             // staging: rows per register batch, 64-column groups per row
  constexpr int KH = KC / 2;                       // MFMAs per (chunk, tap) step
  static_assert(WS % 64 == 0, "row stride");
  PE_DYN_SMEM(float, sm);                          // nbuf x [CP][WS]
  const int b = blockIdx.z;
  const int L = p.lens[b] * p.len_mul;
  const int n0 = blockIdx.x * N;
  if (n0 >= L) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = PE_UNIFORM(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int hx = p.hx;
  constexpr int bufsz = CP * WS;
  const float slope = p.slope;
  const int nph = p.nsteps;
  // table rows are wave-uniform; say so, or every descriptor built from them is loaded in a waterfall loop
  auto get_step = [&](int ph) {
    MrfStep s = p.steps[ph];
    s.wp = pe_uniform_ptr(s.wp); s.bias = pe_uniform_ptr(s.bias);
    s.ntaps = PE_UNIFORM(s.ntaps); s.dil = PE_UNIFORM(s.dil); s.e = PE_UNIFORM(s.e);
    s.src = PE_UNIFORM(s.src); s.dst = PE_UNIFORM(s.dst); s.res = PE_UNIFORM(s.res);
    return s;
  };

  // weights and bias are read through buffer descriptors: a uniform base + lane offset + immediate, vmcnt
  // only (a flat load would also hold up every LDS wait), zero for rows beyond C
  float aA[KH], aB[KH], bA[KH], bB[KH];
  auto load_a = [&](const pe_rowsrc& w, int step, float (&a)[KH]) {
    load_frags<KH>(w, PE_UNIFORM(step * KH * 64), lane, a);
  };
  // unit u of a phase -> (m tile, column tile); waves take units round-robin, rotated per phase so the idle
  // slots of the uneven phases move around the SIMDs
  auto first_unit = [&](int ph, bool fin) { return fin ? wv : (wv + ph) % NW; };
  auto unit_w = [&](const MrfStep& st, int u) {
    const int per_mt = NCH * st.ntaps * KH * 64;
    return pe_make_row(st.wp + (long)(u % MTL) * per_mt, per_mt);
  };
  auto phase_units = [&](const MrfStep& st) { return (st.dst < 0 ? NT : (N + 2 * st.e + 31) / 32) * MTL; };

  // the first weight fragments travel while the window is staged
  bool have = false;
  {
    const MrfStep s0 = get_step(0);
    const int u0 = first_unit(0, s0.dst < 0);
    if (u0 < phase_units(s0)) { load_a(unit_w(s0, u0), 0, aA); have = true; }
  }
  // ---- stage x[n0-hx, n0-hx+WS) of every channel: raw values (the residual needs them), zero outside [0, L)
  {
    const float* xb = p.x + (long)b * p.x_bs;
    const int g0 = n0 - hx + lane;
    for (int r0 = wv * RG; r0 < CP; r0 += NW * RG) {
      float v[RG][NCC];
#pragma unroll
      for (int i = 0; i < RG; ++i) {
        const int ci = r0 + i;
        const pe_rowsrc row = pe_make_row(xb + (long)ci * p.x_cs, ci < p.C ? L : 0);
#pragma unroll
        for (int cc = 0; cc < NCC; ++cc) v[i][cc] = pe_row_load(row, g0 + 64 * cc);
      }
#pragma unroll
      for (int i = 0; i < RG; ++i)
#pragma unroll
        for (int cc = 0; cc < NCC; ++cc) sm[(r0 + i) * WS + lane + 64 * cc] = v[i][cc];
    }
  }

  f32x16 tot[FU];
#pragma unroll
  for (int i = 0; i < FU; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) tot[i][r] = 0.f;

  for (int ph = 0; ph < nph; ++ph) {
    const MrfStep st = get_step(ph);
    const bool fin = st.dst < 0;
    const int nunits = phase_units(st);
    const int ntaps = st.ntaps, dil = st.dil;
    const int hh = dil * (ntaps - 1) / 2;
    const int nst = NCH * ntaps;
    const float* src = sm + st.src * bufsz;
    const float* rs = sm + (st.res < 0 ? 0 : st.res) * bufsz;
    float* dp = sm + (fin ? 0 : st.dst) * bufsz;
    const pe_rowsrc brow = pe_make_row(st.bias, st.bias ? p.C : 0);
    // where this wave's weights come from after the unit at hand: its next unit of this phase, else its
    // first unit of the next phase
    const bool more = ph + 1 < nph;
    MrfStep sn = st;
    if (more) sn = get_step(ph + 1);
    const int nfirst = first_unit(ph + 1, sn.dst < 0);
    const bool nhas = more && nfirst < phase_units(sn);

    __syncthreads();      // the previous phase's LDS writes (or the staging) are visible, its reads are done

    // one 32x32 output tile: K loop over (chunk, tap) steps of 16 MFMAs; the A fragments (L2) and the B
    // values (LDS) of the next step are in flight while this step's MFMAs issue
    auto gemm = [&](int u, f32x16& acc, float (&bs)[16]) {
      const int mt = u % MTL, j0 = hx - st.e + 32 * (u / MTL);
      const pe_rowsrc wb = unit_w(st, u);
      if (!have) load_a(wb, 0, aA);
#pragma unroll
      for (int r = 0; r < 16; ++r) bs[r] = pe_row_load(brow, mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* xp0 = src + lhi * WS + j0 + l31 - hh;
      auto load_b = [&](int c, int tap, float (&bv)[KH]) {
        const float* xp = xp0 + c * KC * WS + tap * dil;
#pragma unroll
        for (int kk = 0; kk < KH; ++kk) bv[kk] = xp[2 * kk * WS];
      };
      auto mma = [&](const float (&a)[KH], const float (&bv)[KH]) {
#pragma unroll
        for (int kk = 0; kk < KH; ++kk) acc = pe_mfma_32x32x2(a[kk], pe_lrelu(bv[kk], slope), acc);
      };
      load_b(0, 0, bA);
      int c = 0, tap = 0;            // (chunk, tap) of the step being prefetched
      auto advance = [&]() { if (++tap == ntaps) { tap = 0; if (++c == NCH) c = 0; } };
      // The prefetches are unconditional -- past the last step the descriptor returns zeros and the LDS
      // position wraps to step 0 -- so that no branch sits between a load and its use: with one, the
      // compiler's wait-count bookkeeping degrades to "drain everything" at every step.
      for (int s = 0; s < nst; s += 2) {
        advance(); load_a(wb, s + 1, aB); load_b(c, tap, bB);
        PE_SCHED_FENCE();
        mma(aA, bA);
        PE_SCHED_FENCE();
        if (s + 1 < nst) {
          advance(); load_a(wb, s + 2, aA); load_b(c, tap, bA);
          PE_SCHED_FENCE();
          mma(aB, bB);
          PE_SCHED_FENCE();
        }
      }
      // next unit's first fragments: in flight during the epilogue (and the barrier)
      have = false;
      if (u + NW < nunits) { load_a(unit_w(st, u + NW), 0, aA); have = true; }
      else if (nhas) { load_a(unit_w(sn, nfirst), 0, aA); have = true; }
    };

    if (fin) {
#pragma unroll
      for (int i = 0; i < FU; ++i) {
        const int u = wv + NW * i;
        if (u < nunits) {
          f32x16 acc;
          float bs[16];
          gemm(u, acc, bs);
          const float* rp = rs + ((u % MTL) * 32 + 4 * lhi) * WS + hx + 32 * (u / MTL) + l31;
#pragma unroll
          for (int r = 0; r < 16; ++r)     // conv_store_tile's ACCUM association
            tot[i][r] = (acc[r] + bs[r]) + (tot[i][r] + rp[((r & 3) + 8 * (r >> 2)) * WS]);
        }
      }
    } else {
      for (int u = first_unit(ph, false); u < nunits; u += NW) {
        f32x16 acc;
        float bs[16];
        gemm(u, acc, bs);
        const int j = hx - st.e + 32 * (u / MTL) + l31;
        const int g = n0 - hx + j;
        const bool inside = g >= 0 && g < L;      // intermediates only exist on [0, L): zero padding
        const int off0 = ((u % MTL) * 32 + 4 * lhi) * WS + j;
        const bool has_res = st.res >= 0;
        float rv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[r] = has_res ? rs[off0 + ((r & 3) + 8 * (r >> 2)) * WS] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = (acc[r] + bs[r]) + rv[r];
          dp[off0 + ((r & 3) + 8 * (r >> 2)) * WS] = inside ? v : 0.f;
        }
      }
    }
  }
  // ---- MRF mean of the owned output tiles
  float* ob = p.out + (long)b * p.o_bs;
#pragma unroll
  for (int i = 0; i < FU; ++i) {
    const int u = wv + NW * i;
    if (u >= NT * MTL) continue;
    const int g = n0 + 32 * (u / MTL) + l31;
    if (g >= L) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (u % MTL) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      if (row < p.C) ob[(long)row * p.o_cs + g] = tot[i][r] * p.alpha;
    }
  }
}

__global__ void scale_kernel(const float* in, float* out, long n, float s) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] * s;
}

// Speaker conditioning (models.py:692-696 emb_g; :66-68 dp.cond; modules.py:188-199 WN.cond_layer;
// models.py:349-351 dec.cond): g is a length-1 sequence, so every 1x1 cond conv reduces to a
// per-utterance bias vector  out[b][r] = W[r][:] . emb_g[sid_b] + bias[r].
__global__ void cond_kernel(const float* emb_g, int gin, const int* sids, const float* w, const float* bias,
                            int rows, float* out, int o_bs) {
  const int b = blockIdx.y;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const float* g = emb_g + (long)sids[b] * gin;
  float s = bias ? bias[r] : 0.f;
  for (int i = 0; i < gin; ++i) s = fmaf(w[(long)r * gin + i], g[i], s);
  out[(long)b * o_bs + r] = s;
}

}  // namespace pe
