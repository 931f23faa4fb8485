// TEST INFRASTRUCTURE ONLY -- see hip_emu.h. Fiber scheduler for the HIP functional emulator.
#include "hip_emu.h"

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
};

static constexpr size_t STACK = 96 * 1024;
static constexpr int MAXT = 1024;
static char* g_stacks = nullptr;
static Fiber g_f[MAXT];
static void* g_sched_sp;
static int g_cur = -1;
static int g_nthreads = 0;
static const std::function<void()>* g_body = nullptr;
static float g_wave_scratch[MAXT / 64][4][64];
static std::vector<char> g_smem;

int lane() { return g_cur & 63; }
int wave() { return g_cur >> 6; }
float* wave_f(int slot) { return g_wave_scratch[g_cur >> 6][slot]; }

static void yield_to_sched() { emu_switch(&g_f[g_cur].sp, g_sched_sp); }

void wave_sync() {
  g_f[g_cur].st = WAIT_WAVE;
  yield_to_sched();
}
void block_sync() {
  g_f[g_cur].st = WAIT_BLOCK;
  yield_to_sched();
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

static void run_block(unsigned nthreads) {
  g_nthreads = (int)nthreads;
  for (unsigned i = 0; i < nthreads; ++i) {
    init_fiber((int)i);
    unsigned bx = blockDim.x, by = blockDim.y;
    g_f[i].tid = emu_idx3{i % bx, (i / bx) % by, i / (bx * by)};
  }
  int nw = ((int)nthreads + 63) / 64;
  for (;;) {
    bool ran = false;
    for (int i = 0; i < (int)nthreads; ++i) {
      if (g_f[i].st != RUNNABLE) continue;
      g_cur = i;
      threadIdx = g_f[i].tid;
      emu_switch(&g_sched_sp, g_f[i].sp);
      ran = true;
    }
    int alive = 0, at_block = 0;
    for (int i = 0; i < (int)nthreads; ++i) {
      if (g_f[i].st != DONE) ++alive;
      if (g_f[i].st == WAIT_BLOCK) ++at_block;
    }
    if (alive == 0) break;
    bool released = false;
    for (int w = 0; w < nw; ++w) {
      int lo = w * 64, hi = lo + 64 > (int)nthreads ? (int)nthreads : lo + 64;
      int live = 0, ww = 0;
      for (int i = lo; i < hi; ++i) {
        if (g_f[i].st != DONE) ++live;
        if (g_f[i].st == WAIT_WAVE) ++ww;
      }
      if (ww > 0 && ww == live) {
        if (live != hi - lo) {
          fprintf(stderr, "hip_emu: wave collective with exited lanes (block %u,%u,%u wave %d)\n",
                  blockIdx.x, blockIdx.y, blockIdx.z, w);
          abort();
        }
        for (int i = lo; i < hi; ++i) g_f[i].st = RUNNABLE;
        released = true;
      }
    }
    if (at_block > 0 && at_block == alive) {
      for (int i = 0; i < (int)nthreads; ++i)
        if (g_f[i].st == WAIT_BLOCK) g_f[i].st = RUNNABLE;
      released = true;
    }
    if (!ran && !released) {
      fprintf(stderr, "hip_emu: deadlock (divergent barrier) in block %u,%u,%u\n", blockIdx.x,
              blockIdx.y, blockIdx.z);
      abort();
    }
  }
}

void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body) {
  // EMU_PLAN_ONLY=1: record-only mode for checking the host's launch decisions (which instantiation, which grid) on
  // full-size shapes that would take hours to emulate; nothing is executed
  static const bool plan_only = getenv("EMU_PLAN_ONLY") && atoi(getenv("EMU_PLAN_ONLY")) != 0;
  if (plan_only) return;
  unsigned nthreads = block.x * block.y * block.z;
  if (nthreads == 0 || nthreads > MAXT) {
    fprintf(stderr, "hip_emu: bad block size %u\n", nthreads);
    abort();
  }
  if (!g_stacks) g_stacks = (char*)aligned_alloc(4096, STACK * MAXT);
  if (g_smem.size() < smem + 64) g_smem.resize(smem + 64);
  dyn_smem = (char*)(((uintptr_t)g_smem.data() + 63) & ~(uintptr_t)63);
  g_body = &body;
  gridDim = emu_idx3{grid.x, grid.y, grid.z};
  blockDim = emu_idx3{block.x, block.y, block.z};
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx = emu_idx3{bx, by, bz};
        run_block(nthreads);
      }
  g_body = nullptr;
}

}  // namespace emu
