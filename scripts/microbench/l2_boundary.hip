// Microbenchmark (MI355X): does data in an XCD's L2 survive a kernel boundary inside a hipGraph, and does a prefetch of the
// NEXT kernel's weights by the current kernel pay? A chain of K dependent kernels; kernel k makes every workgroup read the
// same W-byte "weight" block number (k % NB) (all 256 workgroups read the whole block, like the 4-column chains), then
// writes one value that the next kernel reads (the dependency).
//   NB = 1     : the block was read by the previous kernel (L2-hot if the boundary keeps the L2)
//   NB = 512   : every kernel meets a block last touched 512 kernels ago (Infinity-Cache-hot at best: the real weights)
//   prefetch   : NB = 512, and kernel k touches one dword of every 128-byte line of block k + 1 before its own reads
//   hipcc --offload-arch=gfx950 -O3 -o l2_boundary l2_boundary.hip && ./l2_boundary
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_read(const float* __restrict__ w, int wfloats, const float* __restrict__ nextw,
                                              const float* dep_in, float* dep_out) {
  float acc = dep_in[blockIdx.x & 255];                 // depends on the previous kernel
  if (nextw) {                                          // prefetch: one dword per 128-byte line of the next block, spread over the workgroups of an XCD
    const int lines = wfloats / 32;
    const int per = (lines + 31) / 32;                  // 32 workgroups per XCD (blockIdx.x % 8 = XCD)
    const int l0 = (blockIdx.x >> 3) * per;
    for (int l = l0 + threadIdx.x; l < l0 + per && l < lines; l += 256) acc += nextw[l * 32] * 1e-30f;
  }
  const f32x4* p = reinterpret_cast<const f32x4*>(w);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < wfloats / 4; i += 256) s += p[i];
  acc += s[0] + s[1] + s[2] + s[3];
  if (threadIdx.x == 0) dep_out[blockIdx.x & 255] = acc * 1e-30f;
}

int main() {
  const int K = 256;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  float *dep0, *dep1, *w;
  const int NBMAX = 512;
  const int sizes[] = {16 * 1024 / 4, 147456 / 4, 589824 / 4};          // 16 KB, 144 KB (a 192x192 matrix), 576 KB
  CK(hipMalloc(&dep0, 1024)); CK(hipMalloc(&dep1, 1024));
  CK(hipMemset(dep0, 0, 1024)); CK(hipMemset(dep1, 0, 1024));
  CK(hipMalloc(&w, (size_t)NBMAX * sizes[2] * 4));
  CK(hipMemset(w, 0, (size_t)NBMAX * sizes[2] * 4));
  for (int grid : {32, 256}) for (int wf : sizes) for (int mode = 0; mode < 3; ++mode) {
    const int NB = mode == 0 ? 1 : NBMAX;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int k = 0; k < K; ++k) {
      const float* blk = w + (size_t)(k % NB) * wf;
      const float* nxt = mode == 2 ? w + (size_t)((k + 1) % NB) * wf : nullptr;
      k_read<<<grid, 256, 0, st>>>(blk, wf, nxt, (k & 1) ? dep1 : dep0, (k & 1) ? dep0 : dep1);
    }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    const auto t0 = std::chrono::steady_clock::now();
    const int reps = 10;
    for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (reps * K);
    printf("grid %3d  block %4d KB  %-28s %.2f us per kernel\n", grid, wf * 4 / 1024,
           mode == 0 ? "same block every kernel" : (mode == 1 ? "a new block every kernel" : "new block + prefetch of next"), us);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
