// Internals shared by the engine's translation units (engine.cpp: life cycle, workspaces, graphs, the synthesis call;
// engine_pack.cpp: voice -> packed weight arena; engine_launch.cpp: one launcher per kernel family; engine_issue.cpp: the
// kernel sequence of the pipeline stages). Not part of any interface.
#pragma once
#include "engine.h"
#include "kernels/launch.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <shared_mutex>

namespace pe {

// a launch with a level-2 profile row of its own (the element-wise / integer glue kernels; the conv / attention /
// fused-stage launchers bracket themselves and also carry FLOP and byte counts)
#define PE_LAUNCH_KB(kname, bytes, call)                                         \
  do {                                                                           \
    const int kh_ = kbegin(prof_level_ >= 2 ? krow(kname) : 0, 0.0, (bytes));    \
    call;                                                                        \
    kend(kh_);                                                                   \
  } while (0)
#define PE_LAUNCH_K(kname, call) PE_LAUNCH_KB(kname, 0.0, call)

// Engines that share a process (pe_group_*: one per device, each on its own thread) must not be inside a HIP call while
// another one CAPTURES a graph: allocations / synchronising copies on a second thread invalidate a capture in progress on
// this runtime, whatever the capture mode. Every public entry holds this lock shared; a capture takes it exclusively.
// A single engine per process never contends. (Defined in engine.cpp.)
extern std::shared_mutex g_capture_mu;
extern thread_local int g_entry_depth;
struct EntryLock {
  EntryLock() { if (g_entry_depth++ == 0) g_capture_mu.lock_shared(); }
  ~EntryLock() { if (--g_entry_depth == 0) g_capture_mu.unlock_shared(); }
};

static inline int rup(int v, int m) { return (v + m - 1) / m * m; }

// tile configurations of conv_mfma_kernel: {WM, WN, MT, NT}
enum { CFG_A = 0, CFG_B = 1, CFG_C = 2, CFG_S = 3, CFG_G = 4, CFG_C2 = 5, CFG_B2 = 6 };
static const int CFG_BM[] = {128, 64, 32, 64, 128, 32, 64};
static const int CFG_BN[] = {128, 128, 128, 64, 64, 256, 256};

}  // namespace pe
