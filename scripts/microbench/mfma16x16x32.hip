// Checks the operand / result layout of v_mfma_f32_16x16x32_{f16,bf16} (gfx950) assumed by kernels/mrf_split.h (and tests/emu):
//   A: lane l -> A[row l & 15][k = 8 * (l >> 4) + e], B: lane l -> B[k = 8 * (l >> 4) + e][col l & 15], e = 0..7
//   D: VGPR r of lane l -> D[row 4 * (l >> 4) + r][col l & 15]
// and that f16 SUBNORMAL inputs are not flushed (mode f16x3 relies on it for low terms below 6e-5).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__host__ __device__ inline float av(int i, int k) { return (float)((i * 7 + k * 3) % 11 - 5); }      // small integers: exact in f16 / bf16
__host__ __device__ inline float bvv(int k, int j) { return (float)((k * 5 + j * 2) % 13 - 6); }
template <bool BF>
__global__ void k(float* out, float sub) {
  const int l = threadIdx.x, i = l & 15, kb = 8 * (l >> 4);
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  if (BF) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)av(i, kb + e); b[e] = (__bf16)bvv(kb + e, i); }
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  } else {
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(av(i, kb + e) * sub); b[e] = (_Float16)bvv(kb + e, i); }
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
template <bool BF>
static int check(const char* name, float sub) {
  float* d; hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(k<BF>, dim3(1), dim3(64), 0, 0, d, sub);
  float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * (l >> 4) + r, col = l & 15;
      double e = 0;
      for (int kk = 0; kk < 32; ++kk) e += (double)((float)(_Float16)(av(row, kk) * sub)) * bvv(kk, col);
      if (BF) { e = 0; for (int kk = 0; kk < 32; ++kk) e += (double)av(row, kk) * bvv(kk, col); }
      if (std::fabs(h[l * 4 + r] - (float)e) > 1e-6 * std::fabs(e) + 1e-12) { if (bad < 6) printf("  lane %d r %d got %g expected %g\n", l, r, h[l * 4 + r], e); ++bad; }
    }
  printf("%s: %s (%d mismatches)\n", name, bad ? "DIFFERENT" : "as assumed", bad);
  hipFree(d);
  return bad;
}
int main() {
  int bad = check<false>("mfma_f32_16x16x32_f16 layout", 1.0f);
  bad += check<true>("mfma_f32_16x16x32_bf16 layout", 1.0f);
  bad += check<false>("mfma_f32_16x16x32_f16 with subnormal A inputs (x 2^-20)", 9.5367431640625e-07f);
  return bad != 0;
}
