// Microbenchmark (MI355X): what the f32 matrix pipe sustains in the instruction mixes of the engine's kernels.
//   (a) v_mfma_f32_32x32x2_f32 / 16x16x4_f32 with NACC independent accumulators, W waves per SIMD, operands in registers
//   (b) the same with one ds_read_b32 per MFMA (the tiled conv kernel's B operand) / per 2 MFMAs
//   (c) the same with a 16-byte buffer load per 4 MFMAs (the A fragments from L2)
// Prints TFLOP/s over the whole chip and the fraction of 157.3 (256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int LDSPER, int GLD>   // LDSPER: 0 none, 1 = one ds_read per MFMA, 2 = per two; GLD: 16-byte global load per 4 MFMAs
__global__ __launch_bounds__(256) void k32(const float* __restrict__ g, float* out, int iters) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 0.001f * i;
  __syncthreads();
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float av = threadIdx.x * 0.01f, bv = 1.0f;
  const float* lp = lds + (threadIdx.x & 63);
  const f32x4* gp = reinterpret_cast<const f32x4*>(g) + threadIdx.x;
  f32x4 gw = {1.f, 1.f, 1.f, 1.f};
  for (int it = 0; it < iters; ++it) {
    if (GLD) gw = gp[(it & 63) * 256];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (LDSPER == 1 || (LDSPER == 2 && (u & 1) == 0)) bv = lp[((it + u) & 31) * 64];
      const float a = GLD ? gw[u & 3] : av;
      acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[u % NACC], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  if (s == 12345.678f) out[0] = s;
}
template <int NACC, int LDSPER, int GLD>
__global__ __launch_bounds__(256) void k16(const float* __restrict__ g, float* out, int iters) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 0.001f * i;
  __syncthreads();
  f32x4 acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 4; ++r) acc[a][r] = 0.f;
  float av = threadIdx.x * 0.01f, bv = 1.0f;
  const float* lp = lds + (threadIdx.x & 63);
  const f32x4* gp = reinterpret_cast<const f32x4*>(g) + threadIdx.x;
  f32x4 gw = {1.f, 1.f, 1.f, 1.f};
  for (int it = 0; it < iters; ++it) {
    if (GLD) gw = gp[(it & 63) * 256];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (LDSPER == 1 || (LDSPER == 2 && (u & 1) == 0)) bv = lp[((it + u) & 31) * 64];
      const float a = GLD ? gw[u & 3] : av;
      acc[u % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc[u % NACC], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 4; ++r) s += acc[a][r];
  if (s == 12345.678f) out[0] = s;
}

template <class K>
static double run(K kern, int wgs_per_cu, int iters, double flop_per_mfma, const float* g, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * wgs_per_cu;
  kern<<<grid, 256>>>(g, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kern<<<grid, 256>>>(g, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * 4 * iters * 16 * flop_per_mfma;
  return flops / (ms * 1e-3) / 1e12;
}

int main() {
  float *g, *out;
  CK(hipMalloc(&g, 64 * 256 * 16 + 4096));
  CK(hipMemset(g, 0, 64 * 256 * 16 + 4096));
  CK(hipMalloc(&out, 64));
  const int it32 = 4000, it16 = 8000;
#define R32(NACC, L, G) for (int w = 1; w <= 4; ++w) { double t = run(k32<NACC, L, G>, w, it32, 4096.0, g, out); \
    printf("32x32x2  nacc %d lds/mfma %d gld %d  waves/SIMD %d: %7.1f TFLOP/s  %.3f of 157.3\n", NACC, L, G, w, t, t / 157.3); }
#define R16(NACC, L, G) for (int w = 1; w <= 4; ++w) { double t = run(k16<NACC, L, G>, w, it16, 2048.0, g, out); \
    printf("16x16x4  nacc %d lds/mfma %d gld %d  waves/SIMD %d: %7.1f TFLOP/s  %.3f of 157.3\n", NACC, L, G, w, t, t / 157.3); }
  R32(1, 0, 0) R32(2, 0, 0) R32(4, 0, 0)
  R32(1, 1, 0) R32(2, 1, 0) R32(2, 2, 0)
  R32(2, 1, 1) R32(1, 1, 1)
  R16(1, 0, 0) R16(2, 0, 0) R16(4, 0, 0) R16(8, 0, 0)
  R16(4, 1, 0) R16(8, 1, 0) R16(8, 2, 0) R16(8, 1, 1)
  return 0;
}
