// Durations (ElementwiseAffine reverse, exp, ceil, cumsum), the N(0,1) generator, the length regulator + prior sample.
// (gfx950 / CDNA4 device code; reference arithmetic cited per kernel, paths relative to
// /root/reference/src/python/piper_train/vits/.)
#pragma once
#include "../pe_rt.h"
#include "params.h"
#include "rng.h"

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
// One thread = four consecutive frames of one channel = one Philox block of the prior-noise stream (site 1): with
// p.gen the N(0,1) draws are made here -- exactly the values randn_kernel would have written, which are stored to
// `noise` as well (pe_debug_tensor, tests) -- instead of by a launch of their own in front of this one. With p.fold
// (the whole utterance as one graph: no host read-back between the duration predictor and this kernel) every
// workgroup first computes the durations and their running sum itself -- duration_kernel's arithmetic, T <= REG_MAXT
// ids, integer results, so every workgroup gets the same table -- and workgroup (0, 0, b) publishes them (dur, cum,
// logw, the frame counts for the host and for the kernels behind this one): one launch instead of three at the only
// data-dependent point of the pipeline. `cum` lives in LDS, so the four 7-step searches never leave the CU.
__global__ __launch_bounds__(256) void regulate_kernel(RegP p) {
  PE_KTRACE(16);
  __shared__ int scum[REG_MAXT];
  __shared__ long long part[256];
  __shared__ int sF;
  const int b = blockIdx.z, tid = threadIdx.x;
  const bool lead = blockIdx.x == 0 && blockIdx.y == 0;
  if (p.absmax && lead && tid == 0) p.absmax[b] = 0u;
  const int T = p.tlens[b];
  int F;
  bool in_lds = T <= REG_MAXT;
  // the draws depend on nothing this kernel computes: made first, under the latency of the loads below
  const int f0 = ((int)blockIdx.x * 64 + (tid & 63)) * 4;
  const int c = (int)blockIdx.y * 4 + (tid >> 6);
  float nz[4] = {0.f, 0.f, 0.f, 0.f};
  if (p.gen && c < p.C) randn4((((long)b * p.C + c) * RNG_PITCH + f0) >> 2, p.rng, 1, nz);
  if (p.fold) {
    // ---- duration_kernel's arithmetic (modules.py:407-409; models.py:702-704), ids [lo, hi) per thread
    const DurP& d = p.dur;
    const int per = (T + 255) / 256;
    const int lo = tid * per < T ? tid * per : T, hi = (lo + per < T) ? lo + per : T;
    long long s = 0;
    for (int t = lo; t < hi; ++t) {
      const float zv = d.z0[(long)b * d.z_bs + t];
      const float logw = (zv - d.m0) * d.es0;
      const float w = expf(logw) * d.length_scale;
      float c = ceilf(w);
      c = c < 0.f ? 0.f : (c > 1.0e6f ? 1.0e6f : c);
      const int dv = (int)c;
      scum[t] = dv;
      if (lead) {
        d.dur[b * d.d_bs + t] = dv;
        if (d.logw_out) d.logw_out[(long)b * d.d_bs + t] = logw;
      }
      s += dv;
    }
    part[tid] = s;
    __syncthreads();
    long long run = 0;                       // exclusive prefix of this thread's ids (LDS broadcast reads)
    for (int i = 0; i < tid; ++i) run += part[i];
    if (tid == 255) {
      const long long tot = run + s;
      const int f = tot < 1 ? 1 : (tot > MAX_FRAMES ? MAX_FRAMES + 1 : (int)tot);
      sF = f < d.frame_cap ? f : d.frame_cap;
      if (lead) {
        d.frames[b] = f;
        if (d.frames_host) d.frames_host[b] = f;
        d.frames_clamped[b] = sF;
      }
    }
    for (int t = lo; t < hi; ++t) {
      run += scum[t];
      const int cv = run > MAX_FRAMES ? MAX_FRAMES + 1 : (int)run;
      scum[t] = cv;
      if (lead) d.cum[b * d.d_bs + t] = cv;
    }
    __syncthreads();
    F = sF;
  } else {
    F = p.frames[b];
    if ((int)(blockIdx.x * 256) >= F) return;
    if (in_lds) {
      const int* cum = p.cum + b * p.d_bs;
      for (int i = tid; i < T; i += 256) scum[i] = cum[i];
      __syncthreads();
    }
  }
  if (f0 >= F || c >= p.C) return;
  const int* cum = p.cum + b * p.d_bs;
  int id[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int f = f0 + k;
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
    id[k] = (lo < T && f < F) ? lo : -1;     // -1: beyond the utterance, or every duration is 0 (frames clamped to 1)
  }
  const pe_rowsrc sd = pe_make_row(p.stats + (long)b * p.s_bs, 2 * p.C * p.s_cs);
  float m[4], lg[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    m[k] = pe_row_load(sd, id[k] >= 0 ? c * p.s_cs + id[k] : -1);
    lg[k] = pe_row_load(sd, id[k] >= 0 ? (p.C + c) * p.s_cs + id[k] : -1);
  }
  float* nrow = p.noise ? p.noise + (long)b * p.n_bs + (long)c * p.n_cs + f0 : nullptr;
  if (p.gen) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (f0 + k < p.n_cs) nrow[k] = nz[k];
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) nz[k] = (nrow && f0 + k < F) ? nrow[k] : 0.f;
  }
  float* ob = p.out + (long)b * p.o_bs + (long)c * p.o_cs + f0;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (f0 + k < F) ob[k] = m[k] + nz[k] * expf(lg[k]) * p.noise_scale;
}

}  // namespace pe
