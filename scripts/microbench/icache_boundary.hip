// Microbenchmark (MI355X): what the FIRST use of a kernel's code costs inside a replayed hipGraph, and whether the
// previous kernel can take that cost away by touching the next kernel's code bytes (global loads bring the lines into the
// XCD's L2, which also serves the instruction caches).
// A chain of K dependent launches, each a small amount of straight-line code (CODE_KB of unrolled FMAs executed once):
//   same      : one instantiation K times (code hot after the first launch)
//   distinct  : K different instantiations, the L2s swept by a 96 MB copy in front of every replay (the engine's weight
//               stream does that between two steps)
//   prefetch  : distinct + launch k touches one dword per 128-byte line of launch k + 1's code (entry PC recorded by a
//               probe run with s_getpc_b64)
//   hipcc --offload-arch=gfx950 -O3 -o icache_boundary icache_boundary.hip && ./icache_boundary
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned long long u64;

template <int ID, int N>
__global__ __launch_bounds__(256) void k_code(const float* dep_in, float* dep_out, u64* pc_slot, const unsigned* next_code, int next_lines) {
  u64 pc;
  asm volatile("s_getpc_b64 %0" : "=s"(pc));
  if (pc_slot && threadIdx.x == 0 && blockIdx.x == 0) *pc_slot = pc;
  float tok = 0.f;
  if (next_code) {                       // 32 workgroups per XCD (blockIdx % 8), 256 threads each
    const int l = (blockIdx.x >> 3) * 256 + threadIdx.x;
    if (l < next_lines) tok = __builtin_bit_cast(float, next_code[l * 32]) * 1e-38f;
  }
  float a = dep_in[threadIdx.x & 63] + ID, b = 1.0001f;
#pragma unroll
  for (int i = 0; i < N; ++i) { a = __builtin_fmaf(a, b, 0.5f + i); b = __builtin_fmaf(b, 0.999f, 1e-3f * i); }   // 2 x 8-byte instructions per step
  if (threadIdx.x < 64 && blockIdx.x == 0) dep_out[threadIdx.x] = (a + b + tok) * 1e-30f;
}
__global__ void k_sweep(const float4* in, float4* out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

typedef void (*KFn)(const float*, float*, u64*, const unsigned*, int);
template <int N, int... I> static void fill(std::vector<KFn>& v, std::integer_sequence<int, I...>) { (v.push_back(k_code<I, N>), ...); }

template <int N>
static int run(const char* label, int grid) {
  constexpr int K = 48;
  std::vector<KFn> fns;
  fill<N>(fns, std::make_integer_sequence<int, K>{});
  hipStream_t st;
  CK(hipStreamCreate(&st));
  float *d0, *d1; u64* pcs; float4 *sa, *sb;
  const size_t sweep_n = (96u << 20) / 16;
  CK(hipMalloc(&d0, 1024)); CK(hipMalloc(&d1, 1024)); CK(hipMemset(d0, 0, 1024)); CK(hipMemset(d1, 0, 1024));
  CK(hipMalloc(&pcs, K * 8)); CK(hipMalloc(&sa, sweep_n * 16)); CK(hipMalloc(&sb, sweep_n * 16));
  CK(hipMemset(sa, 0, sweep_n * 16));
  for (int k = 0; k < K; ++k) fns[k]<<<grid, 256, 0, st>>>(d0, d1, pcs + k, nullptr, 0);     // probe: entry PCs
  CK(hipStreamSynchronize(st));
  std::vector<u64> pc(K);
  CK(hipMemcpy(pc.data(), pcs, K * 8, hipMemcpyDeviceToHost));
  const int code_bytes = N * 16 + 512, lines = (code_bytes + 127) / 128;
  double res[3];
  for (int mode = 0; mode < 3; ++mode) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int k = 0; k < K; ++k) {
      const int f = mode == 0 ? 0 : k;
      const unsigned* nc = (mode == 2 && k + 1 < K) ? reinterpret_cast<const unsigned*>(pc[k + 1] & ~127ull) : nullptr;
      fns[f]<<<grid, 256, 0, st>>>((k & 1) ? d1 : d0, (k & 1) ? d0 : d1, nullptr, nc, lines);
    }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    double tot = 0;
    const int reps = 12;
    for (int i = 0; i < reps + 2; ++i) {
      k_sweep<<<2048, 256, 0, st>>>(sa, sb, sweep_n);
      CK(hipStreamSynchronize(st));
      const auto t0 = std::chrono::steady_clock::now();
      CK(hipGraphLaunch(ge, st));
      CK(hipStreamSynchronize(st));
      if (i >= 2) tot += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    }
    res[mode] = tot / reps / K;
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  printf("%-26s grid %3d  code %5.1f KB: same kernel %.2f us/launch, distinct kernels %.2f, distinct + code prefetch %.2f\n",
         label, grid, code_bytes / 1024.0, res[0], res[1], res[2]);
  return 0;
}
int main() {
  for (int grid : {32, 256}) {
    if (run<64>("tiny", grid)) return 1;
    if (run<512>("8 KB straight line", grid)) return 1;
    if (run<2048>("32 KB straight line", grid)) return 1;
    if (run<6144>("96 KB straight line", grid)) return 1;
  }
  return 0;
}
