// Microbenchmark (MI355X): a persistent kernel whose workgroups all sit on ONE XCD and exchange data through that XCD's L2
// between phases -- the cost of a phase boundary without a kernel launch, and whether the exchange is coherent with
// nothing stronger than: stores drained (s_waitcnt vmcnt(0): the write-through L1 has handed them to the L2), an L2-level
// counter barrier (relaxed atomics, workgroup-scope encoding: performed at the L2, no cache maintenance), L1 invalidated
// (buffer_inv sc1) before the reads.
//   * 256 workgroups are launched; each reads HW_REG_XCC_ID and leaves unless it sits on the target XCD; the others take a
//     ticket (0 .. n-1) from an atomic counter and must number exactly `expect` (32 on a round-robin dispatch over 8 XCDs).
//   * phase p: participant i writes its 4 KB block = f(p, i, element) into the buffer of parity p, barrier, then reads the
//     blocks of ALL participants and checks every element; mismatches are counted.
//   * every spin is bounded: a participant that waits too long sets an error flag and everybody leaves (no hang).
//   hipcc --offload-arch=gfx950 -O3 -o xcd_persist xcd_persist.hip && ./xcd_persist
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Ctl { unsigned tickets; unsigned arrive; unsigned error; unsigned mism; unsigned nparts; unsigned pad[27]; };

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}
__device__ __forceinline__ unsigned l2_load(const unsigned* p) {      // a load that does not hit the L1
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ unsigned load_sc0(const unsigned* p) {     // workgroup-scope load: misses the L1, served by the XCD's L2
  unsigned v;
  asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// MODE 0: L2-local protocol (drain, L2 counter polled with an L2 atomic, agent-scope acquire = buffer_inv sc1); 1: agent-scope
// fences (__threadfence) as the reference; 2: mode 0, barriers only; 3: poll with an sc0 load, invalidate with buffer_inv sc0
// (L1 only); 4: mode 3, barriers only; 5: atomic poll + buffer_inv sc0; 7 / 8: mode 4 with s_sleep 2 / 8 between polls
template <int MODE>
__global__ __launch_bounds__(256) void k_persist(Ctl* ctl, float* buf, int phases, unsigned target, unsigned expect, int quiet) {
  __shared__ unsigned s_ticket, s_bail;
  if (xcc_id() != target) return;
  const int tid = threadIdx.x;
  if (tid == 0) {
    s_ticket = __hip_atomic_fetch_add(&ctl->tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    s_bail = 0;
  }
  __syncthreads();
  const unsigned me = s_ticket;
  if (me >= expect) { if (tid == 0 && !quiet) atomicOr(&ctl->error, 2u); return; }      // more participants than planned
  unsigned epoch = 0;
  auto barrier = [&]() -> bool {
    if (MODE == 1) __threadfence();
    else __builtin_amdgcn_s_waitcnt(0);               // vmcnt(0) lgkmcnt(0): this wave's stores have reached the L2
    __syncthreads();
    ++epoch;
    if (tid == 0) {
      __hip_atomic_fetch_add(&ctl->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const unsigned want = epoch * expect;
      long spins = 0;
      // the poll is an atomic performed AT the L2 (fetch-add of 0, no scope bits): an agent-scope load would go past the L2
      // to memory every time (first version of this file: 9 us per barrier)
      while ((MODE == 3 || MODE == 4 || MODE >= 7 ? load_sc0(&ctl->arrive)
                                     : __hip_atomic_fetch_add(&ctl->arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < want) {
        if (MODE == 7) __builtin_amdgcn_s_sleep(2);       // back off between polls: 128 / 512 clocks
        if (MODE == 8) __builtin_amdgcn_s_sleep(8);
        if (++spins > (1L << 20) || ((spins & 63) == 0 && l2_load(&ctl->error))) { atomicOr(&ctl->error, 1u); s_bail = 1; break; }
      }
    }
    __syncthreads();
    if (MODE == 1) __threadfence();
    else if (MODE >= 3) asm volatile("buffer_inv sc0" ::: "memory");     // the L1 alone
    else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // buffer_inv sc1: L1 invalidated, the reads below come from the L2
    return s_bail == 0;
  };
  if (!barrier()) return;                              // everybody is here (and exactly `expect` of us, or we time out)
  unsigned bad = 0;
  for (int p = 0; p < phases; ++p) {
    if (MODE == 2 || MODE == 4 || MODE >= 7) { if (!barrier() || !barrier()) return; continue; }
    float* wb = buf + (size_t)(p & 1) * expect * 1024 + (size_t)me * 1024;
    for (int e = tid; e < 1024; e += 256) wb[e] = (float)(p * 7 + (int)me * 3 + e);
    if (!barrier()) return;
    const float* rb = buf + (size_t)(p & 1) * expect * 1024;
    for (unsigned i = 0; i < expect; ++i)
      for (int e = tid; e < 1024; e += 256 * 8) {       // a sample of every block (128 of 1024 elements per reader thread set)
        const float v = rb[(size_t)i * 1024 + e];
        bad += v != (float)(p * 7 + (int)i * 3 + e);
      }
    if (!barrier()) return;                            // nobody overwrites a block that is still being read
  }
  if (bad) atomicAdd(&ctl->mism, bad);
  if (tid == 0 && me == 0) ctl->nparts = expect;
}

template <int MODE>
static int run(const char* label, int phases, unsigned target, unsigned expect, int quiet = 0) {
  Ctl* ctl; float* buf;
  CK(hipMalloc(&ctl, sizeof(Ctl)));
  CK(hipMalloc(&buf, (size_t)2 * expect * 1024 * 4));
  double best = 1e30;
  Ctl h{};
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(ctl, 0, sizeof(Ctl)));
    CK(hipMemset(buf, 0, (size_t)2 * expect * 1024 * 4));
    CK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    k_persist<MODE><<<256, 256>>>(ctl, buf, phases, target, expect, quiet);
    CK(hipDeviceSynchronize());
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    CK(hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
    if (h.error || h.mism) break;
    best = us < best ? us : best;
  }
  printf("%-34s XCD %u, %u participants (tickets %u): error %u, mismatches %u, %.2f us per phase (write 4 KB, barrier, read all, barrier)\n",
         label, target, expect, h.tickets, h.error, h.mism, best / phases);
  CK(hipFree(ctl)); CK(hipFree(buf));
  return 0;
}

int main() {
  for (unsigned target : {0u, 7u}) {
    if (run<0>("L2-local (drain, L2 counter, inv L1)", 20000, target, 32)) return 1;
    if (run<1>("agent-scope fences", 20000, target, 32)) return 1;
    if (run<2>("L2-local, two barriers only", 20000, target, 32)) return 1;
    if (run<3>("sc0 poll + buffer_inv sc0", 20000, target, 32)) return 1;
    if (run<4>("sc0 poll + buffer_inv sc0, barriers only", 20000, target, 32)) return 1;
    if (run<5>("atomic poll + buffer_inv sc0", 20000, target, 32)) return 1;
  }
  for (unsigned n : {2u, 8u, 16u, 18u, 20u, 24u, 30u, 32u}) {      // how the barrier scales with the participants (the other workgroups leave quietly)
    if (run<2>("atomic counter, barriers only", 20000, 0, n, 1)) return 1;
    if (run<4>("sc0-polled counter, barriers only", 20000, 0, n, 1)) return 1;
    if (run<7>("sc0-polled counter + s_sleep 2", 20000, 0, n, 1)) return 1;
    if (run<8>("sc0-polled counter + s_sleep 8", 20000, 0, n, 1)) return 1;
  }
  if (run<0>("L2-local, 8 participants wanted", 2000, 0, 8)) return 1;     // 24 of the 32 workgroups on the XCD report "too many"
  return 0;
}
