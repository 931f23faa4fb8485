// conv1x1_kernel: the batched 1x1 convs (one tap, no halo) as a GEMM whose B operand never touches LDS.
// (gfx950 / CDNA4 device code; reference arithmetic: the kernel_size = 1 Conv1d modules of attentions.py (q / k / v, o),
// modules.py (WN res_skip_layers :201-208, ResidualCouplingLayer pre / post :447-466) and models.py (TextEncoder.proj);
// paths relative to /root/reference/src/python/piper_train/vits/.)
#pragma once
#include "conv_common.h"

namespace pe {

// Why a form of its own: in conv_mfma_kernel a 32-channel chunk of a ONE-tap conv feeds 16 MFMAs per wave, and behind
// them stand 16 staging loads, 16 LDS stores, the leaky-relu these inputs do not need (32-48 VALU) and a workgroup barrier --
// about half the chunk's matrix time in instructions that are taken out of the MFMA stream (profiles/r04_mfma_mix.txt: a
// VALU op costs the pipe 5-7 cycles, a global load 20, an LDS read nothing): 60 TFLOP/s on the WN res/skip convs of a
// 64-utterance call against 100-115 on the k = 3 ... 7 convs of the same kernel.
// Without taps there is nothing to share between the MFMA's k rows: lane (n = l & 31, k = l >> 5) of v_mfma_f32_32x32x2
// needs x[2 kk + k][col0 + n] -- for the 32 lanes of a k row that is ONE 128-byte piece of the channel's time row, so the
// B operand is loaded straight into its register: one dword load per k-step through a descriptor over the utterance's
// [Cin][stride] tensor (per-lane column offset, poisoned beyond the utterance's length; row offset in an SGPR, which the
// hardware's range check does not see: the launcher sends only convs of whole 32-channel chunks here, policy.h), the same count as the staging loads and nothing else. No LDS, no barrier;
// chunk c + 1's operands (A fragments as float4 descriptor loads, conv_mfma_kernel's packing; B as above) are in flight
// while chunk c's MFMAs issue, at unconditional positions of a 2x unrolled ping-pong (exact wait counts; a chunk that
// does not exist reads B through a zero-length descriptor, so its products are exact zeros).
//   workgroup = 2 x 2 waves = (64 MT) rows x 64 columns; k ascends inside and across the chunks: the fmaf chain of the
//   tiled kernel, bit for bit. Epilogue: conv_store_tile (every non-transposed mode).
// Measured (profiles/r04_notes.md, call 23; medium voice, 64 x 128 ids): WN res/skip 63.9 -> 48.2 us per launch (58.9 -> 78.1
// TFLOP/s), q/k/v 34.3 -> 24.2 us, step 18.55 -> 18.18 ms; MT = 2 (128 x 64 tiles, three waves per SIMD) 49.8 / 26.4 us, 18.31 ms:
// only MT = 1 is compiled.
template <int MT>
__global__ __launch_bounds__(256, MT == 1 ? 4 : 3) void conv1x1_kernel(ConvP p) {
  PE_KTRACE(12);
  constexpr int KH = KC / 2, OOB = 0x3fffffff;
  const int b = blockIdx.z;
  const int L = p.lens[b] * p.len_mul;
  const int n0 = blockIdx.x * 64;
  if (n0 >= L) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = PE_UNIFORM(tid >> 6);
  const int wm = wv >> 1, wn = wv & 1, l31 = lane & 31, lhi = lane >> 5;
  const int mtiles = (p.rows + 31) / 32;
  const int mtile0 = (int)blockIdx.y * (2 * MT) + wm * MT;
  if (mtile0 >= mtiles) return;                       // (no barriers in this kernel: a wave without rows just leaves)
  const int nchunks = p.nchunks;
  const int wstride_mt = nchunks * KH * 64;
  const int live_mt = mtiles - mtile0 < MT ? mtiles - mtile0 : MT;
  const pe_rowsrc wsrc = pe_make_row_u(p.wp + (long)mtile0 * wstride_mt, live_mt * wstride_mt);
  const float* xb = p.x + (long)b * p.x_bs;
  const pe_rowsrc xd = pe_make_row(xb, p.Cin * p.x_cs), xz = pe_make_row(xb, 0);
  const int col = n0 + wn * 32 + l31;
  int voff = col < L ? lhi * p.x_cs + col : OOB;
  PE_OPAQUE(voff);
  const float slope = p.in_slope;

  f32x16 acc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  auto load_ab = [&](int c, bool live, float (&a)[MT][KH], float (&bv)[KH]) {
    const int cc = live ? c : 0;
    const int off = PE_UNIFORM(cc * (KH * 64));
#pragma unroll
    for (int i = 0; i < MT; ++i) load_frags<KH>(wsrc, off + i * wstride_mt, lane, a[i]);
    const pe_rowsrc& src = live ? xd : xz;
#pragma unroll
    for (int kk = 0; kk < KH; ++kk) bv[kk] = pe_row_load_so(src, voff, (cc * KC + 2 * kk) * p.x_cs);
  };
  auto mma = [&](const float (&a)[MT][KH], float (&bv)[KH]) {
    if (slope != 1.f) {
#pragma unroll
      for (int kk = 0; kk < KH; ++kk) bv[kk] = bv[kk] > 0.f ? bv[kk] : bv[kk] * slope;
    }
#pragma unroll
    for (int kk = 0; kk < KH; ++kk)
#pragma unroll
      for (int i = 0; i < MT; ++i) acc[i] = pe_mfma_32x32x2(a[i][kk], bv[kk], acc[i]);
  };
  float aA[MT][KH], aB[MT][KH], bA[KH], bB[KH];
  load_ab(0, true, aA, bA);
  for (int c = 0; c < nchunks; c += 2) {
    PE_SCHED_FENCE();
    load_ab(c + 1, c + 1 < nchunks, aB, bB);
    PE_SCHED_FENCE();
    mma(aA, bA);
    PE_SCHED_FENCE();
    load_ab(c + 2, c + 2 < nchunks, aA, bA);
    PE_SCHED_FENCE();
    mma(aB, bB);
  }
  const EpiFlags ef = epi_flags(p);
#pragma unroll
  for (int i = 0; i < MT; ++i)
    if (i < live_mt) conv_store_tile(p, ef, b, (mtile0 + i) * 32, col, lhi, L, L, acc[i]);
}

}  // namespace pe
