// Checks the operand / result layout of v_mfma_f32_4x4x1_16B_f32 assumed by kernels/col4.h (and tests/emu):
//   A: lane l -> A[block l/4][row l%4], B: lane l -> B[block l/4][col l%4], D: VGPR r of lane l -> D[block l/4][row r][col l%4]
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
  const int l = threadIdx.x;
  const float a = 1.0f + l;            // A[block][row] = 1 + 4*block + row
  const float b = 100.0f + 3.0f * l;   // B[block][col] = 100 + 3*(4*block + col)
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main() {
  float* d; hipMalloc(&d, 256 * 4);
  k<<<1, 64>>>(d);
  float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const int blk = l / 4, col = l % 4;
      const float exp = (1.0f + 4 * blk + r) * (100.0f + 3.0f * (4 * blk + col));
      if (h[l * 4 + r] != exp) { if (bad < 8) printf("lane %d r %d got %g expected %g\n", l, r, h[l * 4 + r], exp); ++bad; }
    }
  printf("mfma4x4x1 layout: %s (%d mismatches)\n", bad ? "DIFFERENT" : "as assumed", bad);
  return bad != 0;
}
