// The engine's counter-based N(0,1) generator: Philox-4x32-10 + Box-Muller, shared by every kernel that draws.
// (gfx950 / CDNA4 device code; the two sampling sites are models.py:111 and :718, paths relative to
// /root/reference/src/python/piper_train/vits/.)
#pragma once
#include "../pe_rt.h"
#include "params.h"

namespace pe {

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
// (seed, run counter) by value: embed_kernel draws with the state it is about to publish
__device__ __forceinline__ void randn4v(long q, unsigned long long seed, unsigned long long counter, int site, float (&g)[4]) {
  const unsigned long long stream = counter * 2ull + (unsigned long long)site;
  unsigned r[4];
  philox4x32((unsigned)q, (unsigned)((unsigned long long)q >> 32), (unsigned)stream, (unsigned)(stream >> 32),
             (unsigned)seed, (unsigned)(seed >> 32), r);
  for (int h = 0; h < 2; ++h) {
    const float u1 = ((float)r[2 * h] + 1.0f) * 2.3283064365386963e-10f;   // (0,1]
    const float u2 = (float)r[2 * h + 1] * 2.3283064365386963e-10f;
    // Box-Muller on the hardware's transcendental units: v_log_f32 (log2), v_sin_f32 / v_cos_f32 (argument in turns, i.e.
    // sin(2 pi u2) without a range reduction) -- a draw is a latency chain inside the kernels that consume it, and the
    // library forms of logf / sinf / cosf were 2/3 of it
    const float rad = sqrtf(-1.3862943611198906f * pe_log2(u1));           // -2 ln(u1) = -2 ln2 log2(u1)
    g[2 * h] = rad * pe_cos_turns(u2);
    g[2 * h + 1] = rad * pe_sin_turns(u2);
  }
}
__device__ __forceinline__ void randn4(long q, const unsigned long long* state, int site, float (&g)[4]) {
  randn4v(q, state[0], state[1], site, g);
}

}  // namespace pe
