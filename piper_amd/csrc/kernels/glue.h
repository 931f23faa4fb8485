// Small element-wise kernels of the front half: noise scaling (un-fused duration-predictor path) and the speaker
// conditioning vectors.
// (gfx950 / CDNA4 device code; reference arithmetic cited per kernel, paths relative to
// /root/reference/src/python/piper_train/vits/.)
#pragma once
#include "../pe_rt.h"
#include "params.h"

namespace pe {

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

// Which XCD (accelerator complex) runs a workgroup: one wave per workgroup stores its XCC id. Engine creation launches 64
// workgroups of this once per device and derives the order in which the 4-column kernels hand out column tiles
// (col4.h c4_tile) from the observed dispatch pattern instead of assuming "linear workgroup id modulo 8".
__global__ void xcc_probe_kernel(int* out) {
#ifdef PE_EMU
  out[blockIdx.x] = (int)(blockIdx.x % 8);
#else
  // s_getreg_b32 HW_REG_XCC_ID (hardware register 20 on gfx940+), field XCC_ID = bits 3:0: simm16 = (size-1) << 11 | offset << 6 | id
  if (threadIdx.x == 0) out[blockIdx.x] = (int)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
#endif
}

}  // namespace pe
