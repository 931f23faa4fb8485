// TEST INFRASTRUCTURE ONLY -- see hip_emu.h. Fiber scheduler for the HIP functional emulator.
#include "hip_emu.h"

#include <cstring>
#include <algorithm>
#include <vector>

emu_idx3 threadIdx, blockIdx, blockDim, gridDim;

extern "C" void emu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size emu_switch,.-emu_switch
)");

namespace emu {

char* dyn_smem = nullptr;

enum State { RUNNABLE, WAIT_WAVE, WAIT_BLOCK, DONE };

struct Fiber {
  void* sp;
  State st;
  emu_idx3 tid;
  unsigned useq;                  // PE_UNIFORM calls made so far (uniform_check)
  unsigned cseq;                  // wave collectives entered so far (collective_parity)
};

// Fibers of ALL blocks that are live at once: one block (ordinary launches) or the whole grid (launch_coop).
static constexpr size_t STACK = 96 * 1024;
static constexpr int MAXT = 1024;
static char* g_stacks = nullptr;
static size_t g_nstacks = 0;
static std::vector<Fiber> g_f;
static void* g_sched_sp;
static int g_cur = -1;            // global fiber index = block slot * nthreads + thread
static int g_nthreads = 0;
static const std::function<void()>* g_body = nullptr;
static std::vector<float> g_wave_scratch;      // [block slot][wave][8][64]
static std::vector<char> g_smem;
static size_t g_smem_stride = 0;
static std::vector<emu_idx3> g_block_of;       // block index of each live block slot
static std::vector<std::vector<long long>> g_uniform;   // [block slot * waves + wave]: values seen at the wave's PE_UNIFORM calls

int lane() { return (g_cur % g_nthreads) & 63; }
int wave() { return (g_cur % g_nthreads) >> 6; }
float* wave_f(int slot) {
  const int nw = (g_nthreads + 63) / 64;
  return &g_wave_scratch[(((size_t)(g_cur / g_nthreads) * nw + wave()) * 8 + slot) * 64];
}
int collective_parity() { return (int)(g_f[g_cur].cseq++ & 1u); }

static void yield_to_sched() { emu_switch(&g_f[g_cur].sp, g_sched_sp); }

void wave_sync() {
  g_f[g_cur].st = WAIT_WAVE;
  yield_to_sched();
}
void block_sync() {
  g_f[g_cur].st = WAIT_BLOCK;
  yield_to_sched();
}
void yield() { yield_to_sched(); }     // stays RUNNABLE: the scheduler comes back after everyone else had a turn

// PE_UNIFORM(x) is readfirstlane on the GPU: every lane silently gets lane 0's value. Here every lane of a wave must
// present the SAME value at its n-th PE_UNIFORM call, or the kernel would compute something else on hardware.
void uniform_check(long long v) {
  const int nw = (g_nthreads + 63) / 64;
  std::vector<long long>& seen = g_uniform[(size_t)(g_cur / g_nthreads) * nw + wave()];
  const unsigned s = g_f[g_cur].useq++;
  if (s == seen.size()) { seen.push_back(v); return; }
  if (s > seen.size() || seen[s] != v) {
    fprintf(stderr, "hip_emu: PE_UNIFORM call %u of wave %d is not wave-uniform (lane %d has %lld, an earlier lane %lld): "
                    "readfirstlane would change the result on the GPU\n", s, wave(), lane(), v,
            s < seen.size() ? seen[s] : -1LL);
    abort();
  }
}

static void fiber_main() {
  (*g_body)();
  g_f[g_cur].st = DONE;
  yield_to_sched();
  abort();
}

static void init_fiber(int i) {
  char* top = g_stacks + (size_t)(i + 1) * STACK;
  uintptr_t t = (uintptr_t)top & ~(uintptr_t)15;
  void** sp = (void**)t;
  *--sp = nullptr;                // fake return address: entry sees rsp % 16 == 8
  *--sp = (void*)&fiber_main;     // popped by 'ret'
  for (int r = 0; r < 6; ++r) *--sp = nullptr;
  g_f[i].sp = sp;
  g_f[i].st = RUNNABLE;
}

// Runs the fibers of `nblocks` live blocks (slots) to completion, round-robin; barriers release per block, wave
// collectives per wave.
static void run_blocks(unsigned nthreads, int nblocks) {
  g_nthreads = (int)nthreads;
  const int total = (int)nthreads * nblocks;
  if ((size_t)total > g_nstacks) {
    free(g_stacks);
    g_stacks = (char*)aligned_alloc(4096, STACK * (size_t)total);
    if (!g_stacks) { fprintf(stderr, "hip_emu: cannot allocate %d fiber stacks\n", total); abort(); }
    g_nstacks = (size_t)total;
  }
  g_f.assign(total, Fiber{});
  const int nw = ((int)nthreads + 63) / 64;
  g_wave_scratch.assign((size_t)nblocks * nw * 8 * 64, 0.f);
  g_uniform.assign((size_t)nblocks * nw, std::vector<long long>());
  for (int i = 0; i < total; ++i) {
    init_fiber(i);
    const unsigned l = (unsigned)i % nthreads, bx = blockDim.x, by = blockDim.y;
    g_f[i].tid = emu_idx3{l % bx, (l / bx) % by, l / (bx * by)};
  }
  // EMU_ORDER=reverse | shuffle: the order in which runnable fibers get their turn. On the GPU the waves of a workgroup
  // advance in no particular order between barriers; a kernel with a missing barrier (one wave reading LDS another has not
  // written yet) gives the same answer here on every run of the default ascending order, but not under another one --
  // tests/test_emu_engine.py runs the small-call kernels under all three and compares.
  static const int order = [] { const char* t = getenv("EMU_ORDER"); return !t ? 0 : (!strcmp(t, "reverse") ? 1 : (!strcmp(t, "shuffle") ? 2 : 0)); }();
  // Wave by wave: the 64 lanes of a wave take their turn back to back, and a wave whose live lanes have all reached a wave
  // collective is released on the spot and runs on -- until it parks at a block barrier or ends -- before the next wave
  // gets its turn (a wave's 64 stacks stay in cache; most rendezvous in the kernels are wave collectives). Between two
  // block barriers one wave may therefore run arbitrarily far ahead of another: a legal order on the GPU, and the order
  // in which the waves go (ascending, EMU_ORDER=reverse, =shuffle: another permutation every round) is what the
  // wave-order tests vary.
  unsigned sweep = 0;
  const int nwv_all = nblocks * nw;
  auto run_fiber = [&](int i) {
    g_cur = i;
    threadIdx = g_f[i].tid;
    blockIdx = g_block_of[i / (int)nthreads];
    dyn_smem = (char*)(((uintptr_t)g_smem.data() + 63) & ~(uintptr_t)63) + (size_t)(i / (int)nthreads) * g_smem_stride;
    emu_switch(&g_sched_sp, g_f[i].sp);
  };
  for (;;) {
    bool ran = false, released = false;
    ++sweep;
    for (int k = 0; k < nwv_all; ++k) {
      int wq = k;
      if (order == 1) wq = nwv_all - 1 - k;
      else if (order == 2) wq = (int)(((unsigned)k * 7u + sweep * 5u) % (unsigned)nwv_all), wq = (nwv_all % 7) ? wq : (int)((k + sweep) % (unsigned)nwv_all);
      const int blk = wq / nw, w = wq % nw;
      const int lo = blk * (int)nthreads + w * 64, hi = std::min(lo + 64, (blk + 1) * (int)nthreads);
      for (;;) {
        bool any = false;
        for (int j = 0; j < hi - lo; ++j) {
          const int i = order == 1 ? hi - 1 - j : (order == 2 ? lo + (int)((j + sweep) % (unsigned)(hi - lo)) : lo + j);
          if (g_f[i].st != RUNNABLE) continue;
          run_fiber(i);
          any = true;
        }
        ran = ran || any;
        int live = 0, ww = 0;
        for (int i = lo; i < hi; ++i) {
          if (g_f[i].st != DONE) ++live;
          if (g_f[i].st == WAIT_WAVE) ++ww;
        }
        if (ww > 0 && ww == live) {
          if (live != hi - lo) {
            fprintf(stderr, "hip_emu: wave collective with exited lanes (block slot %d wave %d)\n", blk, w);
            abort();
          }
          for (int i = lo; i < hi; ++i) g_f[i].st = RUNNABLE;
          released = true;
          continue;                      // the wave runs on
        }
        if (!any) break;                 // parked (block barrier, partial collective) or done
        bool runnable = false;
        for (int i = lo; i < hi; ++i) runnable = runnable || g_f[i].st == RUNNABLE;
        if (!runnable) break;
      }
    }
    int alive_all = 0;
    for (int blk = 0; blk < nblocks; ++blk) {
      const int base = blk * (int)nthreads;
      int alive = 0, at_block = 0;
      for (int i = base; i < base + (int)nthreads; ++i) {
        if (g_f[i].st != DONE) ++alive;
        if (g_f[i].st == WAIT_BLOCK) ++at_block;
      }
      alive_all += alive;
      if (at_block > 0 && at_block == alive) {
        for (int i = base; i < base + (int)nthreads; ++i)
          if (g_f[i].st == WAIT_BLOCK) g_f[i].st = RUNNABLE;
        released = true;
      }
    }
    if (alive_all == 0) break;
    if (!ran && !released) {
      fprintf(stderr, "hip_emu: deadlock (divergent barrier)\n");
      abort();
    }
  }
}

static bool plan_only() {
  // EMU_PLAN_ONLY=1: record-only mode for checking the host's launch decisions (which instantiation, which grid) on
  // full-size shapes that would take hours to emulate; nothing is executed
  static const bool v = getenv("EMU_PLAN_ONLY") && atoi(getenv("EMU_PLAN_ONLY")) != 0;
  return v;
}

static void setup(dim3 grid, dim3 block, size_t smem, int nslots, const std::function<void()>& body) {
  g_smem_stride = (smem + 127) / 64 * 64;
  if (g_smem.size() < g_smem_stride * nslots + 128) g_smem.resize(g_smem_stride * nslots + 128);
  g_body = &body;
  gridDim = emu_idx3{grid.x, grid.y, grid.z};
  blockDim = emu_idx3{block.x, block.y, block.z};
}

void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body) {
  if (plan_only()) return;
  unsigned nthreads = block.x * block.y * block.z;
  if (nthreads == 0 || nthreads > MAXT) {
    fprintf(stderr, "hip_emu: bad block size %u\n", nthreads);
    abort();
  }
  setup(grid, block, smem, 1, body);
  g_block_of.assign(1, emu_idx3{0, 0, 0});
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        g_block_of[0] = emu_idx3{bx, by, bz};
        run_blocks(nthreads, 1);
      }
  g_body = nullptr;
}

void launch_coop(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body) {
  if (plan_only()) return;
  unsigned nthreads = block.x * block.y * block.z;
  const size_t nblocks = (size_t)grid.x * grid.y * grid.z;
  if (nthreads == 0 || nthreads > MAXT || nblocks == 0 || nblocks * nthreads > 65536) {
    fprintf(stderr, "hip_emu: cooperative launch too large for the emulator (%zu blocks x %u threads)\n", nblocks, nthreads);
    abort();
  }
  setup(grid, block, smem, (int)nblocks, body);
  g_block_of.clear();
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) g_block_of.push_back(emu_idx3{bx, by, bz});
  run_blocks(nthreads, (int)nblocks);
  g_body = nullptr;
}

}  // namespace emu
