// Microbenchmark (MI355X): what one dependent "level" costs as (a) a kernel boundary in a hipGraph / eager stream
// and (b) an in-kernel hand-off between resident workgroups. Informs the batch-1 design:
// how much a fused launch saves, and whether persistent kernels with in-kernel synchronisation can beat launches.
//   hipcc --offload-arch=gfx950 -O3 -o launch_floor launch_floor.hip && ./launch_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct BigArgs { const float* in; float* out; const int* lens; int n; int pad[52]; };   // ~240 B like ConvP

__global__ void k_empty() {}
__global__ void k_copy(const float* in, float* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] * 1.0001f + 1.f;
}
__global__ void k_dep(const float* in, float* out, const int* lens) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < lens[0]) out[i] = in[i] * 1.0001f + 1.f;
}
__global__ void k_big(BigArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < a.lens[0] + a.pad[51]) a.out[i] = a.in[i] * 1.0001f + 1.f;
}

typedef unsigned long long u64;
// R2-style hand-off (guide G16): 8-byte {epoch, value} granules, relaxed agent-scope stores/loads, data is the flag.
// phase p: WG w publishes 256 granules, then reads the granules WG (w+1)%G published in the same phase.
__global__ void k_neighbor(u64* slots, int phases, float* sink) {
  const int G = gridDim.x, w = blockIdx.x, t = threadIdx.x;
  float acc = 0.f;
  for (int p = 1; p <= phases; ++p) {
    __hip_atomic_store(&slots[(size_t)w * 256 + t], ((u64)p << 32) | (unsigned)(w + p + t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u64* src = &slots[(size_t)((w + 1) % G) * 256 + t];
    u64 v;
    long spins = 0;
    do { v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((v >> 32) < (u64)p && ++spins < (1L << 24));
    acc += (float)(unsigned)v;
    __syncthreads();
  }
  if (t == 0) sink[w] = acc;
}
// full-grid barrier: one monotonic counter (relaxed agent atomics), data exchanged through granules as above, so no
// cache fences are needed; each phase every WG reads the granules of WG (w + p) % G (an all-to-all-like pattern)
__global__ void k_gridbar(u64* slots, unsigned* counter, int phases, float* sink) {
  const int G = gridDim.x, w = blockIdx.x, t = threadIdx.x;
  float acc = 0.f;
  for (int p = 1; p <= phases; ++p) {
    __hip_atomic_store(&slots[(size_t)w * 256 + t], ((u64)p << 32) | (unsigned)(w + t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      long spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(p * G) && ++spins < (1L << 24)) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    const u64 v = __hip_atomic_load(&slots[(size_t)((w + p) % G) * 256 + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    acc += (float)(unsigned)v + (float)(v >> 32);
  }
  if (t == 0) sink[w] = acc;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const int n = 256 * 256, N = 200, REP = 20;
  float *a, *b; int* lens;
  CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&lens, 64));
  CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
  int hn = n; CK(hipMemcpy(lens, &hn, 4, hipMemcpyHostToDevice));
  BigArgs ba{}; ba.lens = lens; ba.n = n;
  auto issue = [&](int kind, int grid) {
    for (int i = 0; i < N; ++i) {
      const float* in = (i & 1) ? b : a; float* out = (i & 1) ? a : b;
      switch (kind) {
        case 0: hipLaunchKernelGGL(k_empty, dim3(grid), dim3(64), 0, s); break;
        case 1: hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, s, in, out, n); break;
        case 2: hipLaunchKernelGGL(k_dep, dim3(grid), dim3(256), 0, s, in, out, lens); break;
        default: ba.in = in; ba.out = out; hipLaunchKernelGGL(k_big, dim3(grid), dim3(256), 0, s, ba); break;
      }
    }
  };
  const char* names[] = {"empty", "copy", "copy+dependent len load", "copy+len+240B kernarg"};
  for (int kind = 0; kind < 4; ++kind)
    for (int grid : {1, 16, 256}) {
      if (kind == 0 && grid != 1) continue;
      hipGraph_t g; hipGraphExec_t ex;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      issue(kind, grid);
      CK(hipStreamEndCapture(s, &g));
      CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
      CK(hipGraphLaunch(ex, s)); CK(hipStreamSynchronize(s));
      double t0 = now();
      for (int r = 0; r < REP; ++r) CK(hipGraphLaunch(ex, s));
      CK(hipStreamSynchronize(s));
      const double tg = (now() - t0) / (REP * N) * 1e6;
      issue(kind, grid); CK(hipStreamSynchronize(s));
      t0 = now();
      for (int r = 0; r < REP; ++r) issue(kind, grid);
      CK(hipStreamSynchronize(s));
      const double te = (now() - t0) / (REP * N) * 1e6;
      printf("chain of %d dependent kernels, %-28s grid %3d: graph %.2f us/kernel, eager %.2f us/kernel\n", N, names[kind], grid, tg, te);
      hipGraphExecDestroy(ex); hipGraphDestroy(g);
    }
  // in-kernel hand-offs
  u64* slots; unsigned* counter; float* sink;
  CK(hipMalloc(&slots, 256 * 256 * 8)); CK(hipMalloc(&counter, 64)); CK(hipMalloc(&sink, 256 * 4));
  const int P = 200;
  for (int G : {8, 32, 64, 128, 256}) {
    for (int kind = 0; kind < 2; ++kind) {
      double best = 1e9;
      for (int r = 0; r < 5; ++r) {
        CK(hipMemsetAsync(slots, 0, 256 * 256 * 8, s)); CK(hipMemsetAsync(counter, 0, 4, s));
        CK(hipStreamSynchronize(s));
        const double t0 = now();
        if (kind == 0) hipLaunchKernelGGL(k_neighbor, dim3(G), dim3(256), 0, s, slots, P, sink);
        else hipLaunchKernelGGL(k_gridbar, dim3(G), dim3(256), 0, s, slots, counter, P, sink);
        CK(hipStreamSynchronize(s));
        best = std::min(best, (now() - t0) * 1e6);
      }
      printf("persistent kernel, %3d workgroups, %-42s %.2f us/phase (launch included, %d phases)\n", G,
             kind == 0 ? "neighbour hand-off (2 KB granules):" : "grid barrier (counter) + granule read:", best / P, P);
    }
  }
  return 0;
}
