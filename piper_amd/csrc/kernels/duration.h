// Durations (ElementwiseAffine reverse, exp, ceil, cumsum), the N(0,1) generator, the length regulator + prior sample.
// (gfx950 / CDNA4 device code; reference arithmetic cited per kernel, paths relative to
// /root/reference/src/python/piper_train/vits/.)
#pragma once
#include "../pe_rt.h"
#include "params.h"

namespace pe {

// ElementwiseAffine reverse + durations (modules.py:407-409; models.py:702-704):
//   logw = (z0 - m0) * exp(-logs0); w = exp(logw) * length_scale; d = ceil(w);
//   cum = inclusive prefix sum; frames = max(sum d, 1).   One block per utterance.
// Sums run in 64 bits and are clamped to MAX_FRAMES + 1 (a single duration to 1e6): an absurd length_scale cannot
// overflow `cum`, and the host rejects frames > MAX_FRAMES before sizing stage B from it.
__global__ __launch_bounds__(256) void duration_kernel(DurP p) {
  PE_KTRACE(13);
  __shared__ long long part[256];
  const int b = blockIdx.x;
  const int T = p.lens[b], tid = threadIdx.x;
  const int per = (T + 255) / 256;
  const int lo = tid * per, hi = (lo + per < T) ? lo + per : T;
  long long s = 0;
  if (tid < 256) {
    for (int t = lo; t < hi; ++t) {
      const float zv = p.z0[(long)b * p.z_bs + t];
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
// A site's stream is a logical 2-D array [row][RNG_PITCH] (row = utterance * channels + channel, column = phoneme id /
// frame): element (row, col) is draw number row * RNG_PITCH + col, whatever the physical row stride of the buffer it is
// written to -- so for a given (seed, run counter) the noise of frame f of channel c of utterance b does not depend on
// workspace capacities, shape buckets or whether the frame count was speculated. One thread = one Philox block = four
// consecutive columns of one row; `row0` = first logical row (test hook: any window of the stream).
__global__ void randn_kernel(float* out, long rows, int cols, long stride, long row0, const unsigned long long* state,
                             int site) {
  PE_KTRACE(15);
  const int nb = (cols + 1023) / 1024;                      // 256 threads x 4 columns per block
  const long row = (long)blockIdx.x / nb;
  const int c4 = (((int)((long)blockIdx.x - row * nb)) * 256 + (int)threadIdx.x) * 4;
  if (row >= rows || c4 >= cols) return;
  float g[4];
  randn4(((row0 + row) * RNG_PITCH + c4) >> 2, state, site, g);
  for (int k = 0; k < 4 && c4 + k < cols; ++k) out[row * stride + c4 + k] = g[k];
}

// ------------------------------------------------------------------------------------------------
// Length regulator + prior sample (models.py:705-718, commons.py:116-129). The reference multiplies
// by a one-hot path matrix; the same result is a gather: frame f takes id i with cum[i-1] <= f < cum[i].
//   z_p[c][f] = m_p[c][i] + noise[c][f] * exp(logs_p[c][i]) * noise_scale
// At batch 1 this launch is a latency chain, so: `cum` is copied to LDS once (the 7-step binary search then never
// leaves the CU) and the 3 x 16 operands of a thread's channels are requested together through row descriptors.
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

}  // namespace pe
