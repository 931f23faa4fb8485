// ConvFlow pieces of the duration predictor: pre conv and the inverse rational-quadratic spline.
// (gfx950 / CDNA4 device code; reference arithmetic cited per kernel, paths relative to
// /root/reference/src/python/piper_train/vits/.)
#pragma once
#include "../pe_rt.h"
#include "params.h"

namespace pe {

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

}  // namespace pe
