// Generator tail: conv_post + tanh + peak, int16 conversion; small glue kernels.
// (gfx950 / CDNA4 device code; reference arithmetic cited per kernel, paths relative to
// /root/reference/src/python/piper_train/vits/.)
#pragma once
#include "../pe_rt.h"
#include "params.h"

namespace pe {

// ------------------------------------------------------------------------------------------------
// Generator tail (models.py:364-366): leaky_relu(0.01) -> conv_post (k=7, no bias, 1 output channel)
// -> tanh, fused with the per-utterance max|x| that the int16 conversion needs (piper.cpp:410-418).
// HBM-bound (4*Cin bytes in, 4 out per sample), and at batch 1 a latency chain: a workgroup = 256 samples x 4 channel
// groups (one wave each); a thread owns POST_OPT consecutive samples of POST_CU channels per pass (one pass for
// Cin <= 32) and requests all its POST_CU * (POST_OPT + 6) inputs at once through row descriptors (zero padding = range
// check; neighbouring threads' overlap is served by L1). The weights are wave-uniform scalars. The channel-group
// partials meet in LDS and are summed in a fixed order; one peak atomic per workgroup. (A first version walked all channels in one thread: 8 dependent memory round
// trips and 104 workgroups for a 4.8 s utterance, 22.9 us; profiles/r02_notes.md.)
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

}  // namespace pe
