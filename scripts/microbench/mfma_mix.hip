// Microbenchmark (MI355X), second part: how many matrix-pipe cycles ONE other instruction costs a SIMD that is otherwise
// issuing back-to-back f32 MFMAs -- ds_read_b32 / ds_read2_b32 / ds_read_b128 with immediate offsets (no address VALU),
// a 16-byte buffer load, a plain VALU op -- at R MFMAs per such instruction and W waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_mix mfma_mix.hip && ./mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// KIND: 0 none, 1 ds_read_b32, 2 ds_read2_b32 (8 bytes, two rows), 3 ds_read_b128, 4 global 16-byte load, 5 v_fma
// Software-pipelined like the engine's kernels: the loads of iteration i + 1 are issued (interleaved, one per R MFMAs, by
// sched_group_barrier) while the MFMAs of iteration i run on the operands loaded one iteration earlier.
template <int NL, int K, int R>
__device__ __forceinline__ void interleave() {
  if constexpr (NL > 0) {
    __builtin_amdgcn_sched_group_barrier(0x8, R, 0);
    __builtin_amdgcn_sched_group_barrier(K == 4 ? 0x20 : (K == 5 ? 0x2 : 0x100), 1, 0);
    interleave<NL - 1, K, R>();
  }
}
template <int M32, int KIND, int R, int NACC>
__global__ __launch_bounds__(256) void kmix(const float* __restrict__ g, float* out, int iters) {
  __shared__ float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 0.001f * i;
  __syncthreads();
  f32x16 a32[M32 ? NACC : 1];
  f32x4 a16[M32 ? 1 : NACC];
  for (int a = 0; a < (M32 ? NACC : 1); ++a) for (int r = 0; r < 16; ++r) a32[a][r] = 0.f;
  for (int a = 0; a < (M32 ? 1 : NACC); ++a) for (int r = 0; r < 4; ++r) a16[a][r] = 0.f;
  const float av = threadIdx.x * 0.01f;
  constexpr int U = 32, NL = KIND ? U / R : 0, NV = KIND == 0 ? 1 : (KIND == 1 ? 1 : (KIND == 2 ? 2 : (KIND == 5 ? 1 : 4)));
  float cur[NL ? NL : 1][NV], nxt[NL ? NL : 1][NV];
  for (int j = 0; j < (NL ? NL : 1); ++j) for (int v = 0; v < NV; ++v) cur[j][v] = 1.f + j + v;
  const float* lp = lds + (threadIdx.x & 63) * 4;
  const f32x4* gp = reinterpret_cast<const f32x4*>(g) + threadIdx.x;
  auto half = [&](int it, float (&use)[NL ? NL : 1][NV], float (&fill)[NL ? NL : 1][NV]) {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      if (KIND == 1) fill[j][0] = lp[(j * 64) & 4095];
      if (KIND == 2) { fill[j][0] = lp[(j * 64) & 2047]; fill[j][1] = lp[((j * 64) & 2047) + 2048]; }
      if (KIND == 3) { f32x4 v = *reinterpret_cast<const f32x4*>(lp + ((j * 256) & 4095)); for (int q = 0; q < 4; ++q) fill[j][q] = v[q]; }
      if (KIND == 4) { f32x4 v = gp[((it * NL + j) & 63) * 256]; for (int q = 0; q < 4; ++q) fill[j][q] = v[q]; }
      if (KIND == 5) { fill[j][0] = __builtin_fmaf(use[j][0], 1.0001f, av); }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float bv = NL ? use[u / R][u % NV] : 1.f;
      if (M32) a32[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, a32[u % NACC], 0, 0, 0);
      else a16[u % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, a16[u % NACC], 0, 0, 0);
    }
    interleave<NL, KIND, R>();
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int it = 0; it < iters; it += 2) {
    half(it, cur, nxt);
    half(it + 1, nxt, cur);
  }
  float s = 0.f;
  for (int a = 0; a < (M32 ? NACC : 1); ++a) for (int r = 0; r < 16; ++r) s += a32[a][r];
  for (int a = 0; a < (M32 ? 1 : NACC); ++a) for (int r = 0; r < 4; ++r) s += a16[a][r];
  if (s == 12345.678f) out[0] = s;
}

template <class K>
static double run(K kern, int w, int iters, double cyc_per_mfma, const float* g, float* out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int grid = 256 * w;
  kern<<<grid, 256>>>(g, out, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  kern<<<grid, 256>>>(g, out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  // matrix-pipe busy fraction = MFMAs per SIMD x cycles each / elapsed cycles at 2.4 GHz
  return (double)w * iters * 32 * cyc_per_mfma / (ms * 1e-3 * 2.4e9);
}
static const char* kname[] = {"none", "ds_read_b32", "ds_read2_b32", "ds_read_b128", "global_load_b128", "v_fma"};
template <int M32, int KIND, int R, int NACC>
static void one(const float* g, float* out) {
  printf("%-8s %-16s 1 per %d MFMAs:", M32 ? "32x32x2" : "16x16x4", kname[KIND], R);
  for (int w = 1; w <= 4; ++w) {
    const double f = run(kmix<M32, KIND, R, NACC>, w, M32 ? 1500 : 3000, M32 ? 64.0 : 32.0, g, out);
    // pipe cycles lost per inserted instruction = cyc_per_mfma * R * (1/f - 1/f0) with f0 ~ 0.988
    printf("  W%d %.3f (%.1f cyc)", w, f, (M32 ? 64.0 : 32.0) * R * (1.0 / f - 1.0 / 0.988));
  }
  printf("\n");
}
int main() {
  float *g, *out;
  (void)hipMalloc(&g, 64 * 256 * 16 + 4096);
  (void)hipMemset(g, 0, 64 * 256 * 16 + 4096);
  (void)hipMalloc(&out, 64);
  one<0, 0, 1, 8>(g, out);
  one<0, 1, 1, 8>(g, out); one<0, 1, 2, 8>(g, out); one<0, 1, 4, 8>(g, out); one<0, 1, 8, 8>(g, out);
  one<0, 2, 1, 8>(g, out); one<0, 2, 2, 8>(g, out); one<0, 2, 4, 8>(g, out); one<0, 2, 8, 8>(g, out);
  one<0, 3, 2, 8>(g, out); one<0, 3, 4, 8>(g, out); one<0, 3, 8, 8>(g, out);
  one<0, 4, 4, 8>(g, out); one<0, 4, 8, 8>(g, out); one<0, 4, 16, 8>(g, out);
  one<0, 5, 1, 8>(g, out); one<0, 5, 2, 8>(g, out); one<0, 5, 4, 8>(g, out);
  one<1, 0, 1, 2>(g, out);
  one<1, 1, 1, 2>(g, out); one<1, 1, 2, 2>(g, out); one<1, 1, 4, 2>(g, out);
  one<1, 2, 1, 2>(g, out); one<1, 2, 2, 2>(g, out);
  one<1, 3, 2, 2>(g, out); one<1, 3, 4, 2>(g, out);
  one<1, 4, 4, 2>(g, out); one<1, 4, 8, 2>(g, out);
  one<1, 5, 1, 2>(g, out); one<1, 5, 2, 2>(g, out);
  return 0;
}
