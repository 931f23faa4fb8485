// LayerNorm over channels (encoder widths the fused 192-channel kernels do not cover).
// (gfx950 / CDNA4 device code; reference arithmetic cited per kernel, paths relative to
// /root/reference/src/python/piper_train/vits/.)
#pragma once
#include "../pe_rt.h"
#include "params.h"

namespace pe {

// LayerNorm over channels (modules.py:23-26): out = LN(in) (encoder norm_layers_1/2; in = x + y).
// One workgroup = 8 time columns x all channels; thread (col = tid&7, rl = tid>>3) keeps channels
// rl, rl+32, ... in registers (C <= 256), so the input is read once. (Batched calls and encoder widths other than
// 192: small calls of the 192-channel voices fold the norms into colchain_kernel / lngemm_kernel.)

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

__global__ __launch_bounds__(256) void ln_kernel(LnP p) {
  PE_KTRACE(5);
  __shared__ float red[4][LN_COLS];
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
    const float x = (ok && c < p.C) ? ib[(long)c * p.i_cs + t] : 0.f;
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
    if (c < p.C) p.out[(long)b * p.o_bs + (long)c * p.o_cs + t] = (v[k] - mean) * rstd * p.gamma[c] + p.beta[c];
  }
}

}  // namespace pe
