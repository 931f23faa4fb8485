// Runtime header for the engine: HIP on gfx950. (-DPE_EMU swaps in tests/emu/hip_emu.h, a
// test-only functional emulator used to check kernel index logic without a GPU; the shipped
// library is never built that way.)
#pragma once

namespace pe { extern thread_local long g_launches; }   // kernel launches issued by this thread (captured launches count once)
#ifdef PE_EMU
namespace pe { inline const char* g_emu_kernel = ""; }   // the kernel the emulator is running (diagnostics)
#include "hip_emu.h"
#define PE_LAUNCH(kernel, grid, block, smem, stream, ...) \
  (++pe::g_launches, pe::g_emu_kernel = #kernel, emu::launch((grid), (block), (smem), [=]() { kernel(__VA_ARGS__); }))
#define PE_DYN_SMEM(type, name) type* name = reinterpret_cast<type*>(emu::dyn_smem)
#define PE_STAMP(k, i) ((void)0)
#define PE_KTRACE(id) ((void)0)
#define pe_mfma_32x32x2(a, b, c) emu_mfma_32x32x2((a), (b), (c))
#define pe_mfma_16x16x4(a, b, c) emu_mfma_16x16x4((a), (b), (c))
#define pe_mfma_4x4x1(a, b, c) emu_mfma_4x4x1((a), (b), (c))
#define PE_WAVE_SYNC() emu::wave_sync()
#define PE_OPAQUE(x) ((void)0)
#define PE_UNIFORM(x) (emu::uniform_check((long long)(x)), (x))     // checked: readfirstlane on the GPU
#define PE_SCHED_FENCE() ((void)0)
#define PE_SCHED_GROUP(mask, n) ((void)0)
template <class T> inline T* pe_uniform_ptr(T* p) { return p; }
// bounds-checked row load: element idx of a row of n floats, 0 outside [0, n)
struct pe_rowsrc { const float* p; int n; };
inline pe_rowsrc pe_make_row(const float* row, int n) { return pe_rowsrc{row, n}; }
inline float pe_row_load(const pe_rowsrc& r, int idx) { return (idx >= 0 && idx < r.n) ? r.p[idx] : 0.f; }
inline pe_rowsrc pe_make_row_u(const float* row, int n) { return pe_rowsrc{row, n}; }
// element (vidx + sidx): vidx per lane, sidx wave-uniform (an SGPR offset on the GPU)
// The hardware's range check covers the VGPR offset only (gfx9 / CDNA: the SGPR offset of a raw buffer access is excluded
// from bounds checking): an access whose vidx passes while vidx + sidx lies outside the row would touch memory behind the
// tensor on the GPU. The emulator refuses it instead of returning the 0 a full check would give.
inline bool pe_so_in_row(const pe_rowsrc& r, int vidx, int sidx, int width) {
  if (vidx < 0 || vidx + width > r.n) return false;          // the hardware's check: reads give 0, writes are dropped
  const long i = (long)vidx + sidx;
  if (i < 0 || i + width > r.n) {
    fprintf(stderr, "hip_emu: buffer access with VGPR offset %d inside the row (%d elements) but VGPR + SGPR offset %ld outside: "
                    "the SGPR offset is not range-checked on the GPU [%s]\n", vidx, r.n, i, pe::g_emu_kernel);
    abort();
  }
  return true;
}
inline float pe_row_load_so(const pe_rowsrc& r, int vidx, int sidx) { return pe_so_in_row(r, vidx, sidx, 1) ? r.p[vidx + sidx] : 0.f; }
inline void pe_row_store_so(const pe_rowsrc& r, int vidx, int sidx, float v) {
  if (pe_so_in_row(r, vidx, sidx, 1)) const_cast<float*>(r.p)[vidx + sidx] = v;
}
// two / four consecutive floats starting at element idx (all of them inside the row, or none is written)
inline void pe_row_store2(const pe_rowsrc& r, int idx, float a, float b) {
  if (idx >= 0 && idx + 1 < r.n) { const_cast<float*>(r.p)[idx] = a; const_cast<float*>(r.p)[idx + 1] = b; }
}
inline void pe_row_store4(const pe_rowsrc& r, int idx, float a, float b, float c, float d) {
  if (idx >= 0 && idx + 3 < r.n) {
    float* q = const_cast<float*>(r.p) + idx;
    q[0] = a; q[1] = b; q[2] = c; q[3] = d;
  }
}
inline f32x4 pe_row_load4(const pe_rowsrc& r, int idx) {
  f32x4 v;
  for (int j = 0; j < 4; ++j) v[j] = (idx + j >= 0 && idx + j < r.n) ? r.p[idx + j] : 0.f;
  return v;
}
// element (vidx + sidx) .. + 3: vidx per lane, sidx wave-uniform (SGPR offset)
inline f32x4 pe_row_load4_so(const pe_rowsrc& r, int vidx, int sidx) {
  if (vidx >= 0 && vidx < r.n && vidx + 4 > r.n) return pe_row_load4(r, vidx + sidx);      // (a straddling group: element-wise)
  f32x4 z = {0.f, 0.f, 0.f, 0.f};
  return pe_so_in_row(r, vidx, sidx, 4) ? pe_row_load4(r, vidx + sidx) : z;
}
inline float pe_lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }
inline float pe_log2(float x) { return log2f(x); }
inline float pe_sin_turns(float x) { return sinf(6.283185307179586f * x); }
inline float pe_cos_turns(float x) { return cosf(6.283185307179586f * x); }
#else
#include <hip/hip_runtime.h>
#define PE_LAUNCH(kernel, grid, block, smem, stream, ...) \
  do { ++pe::g_launches; hipLaunchKernelGGL(kernel, (grid), (block), (smem), (stream), __VA_ARGS__); } while (0)
#define PE_DYN_SMEM(type, name) \
  extern __shared__ __attribute__((aligned(16))) unsigned char pe_dyn_smem_raw[]; \
  type* name = reinterpret_cast<type*>(pe_dyn_smem_raw)
// Phase timestamps for kernel tuning (`make stamps`, scripts/stamps.py; NOT in the shipped library): thread 0 of
// workgroup (0,0,0) records the 100 MHz wall clock at phase boundaries of the latency-critical small kernels.
#ifdef PE_STAMPS
#define PE_NSTAMP_K 8
#define PE_NSTAMP_I 24
static __device__ long long pe_stamps[PE_NSTAMP_K][PE_NSTAMP_I];   // one copy per translation unit; engine.cpp (kernels + reader) uses its own
#define PE_STAMP(k, i) \
  do { if ((threadIdx.x | blockIdx.x | blockIdx.y | blockIdx.z) == 0) pe_stamps[k][i] = (long long)wall_clock64(); } while (0)
// per-launch trace: workgroup (0,0,0) of EVERY launch takes the next slot and records kernel id, wall clock at entry and
// at exit (RAII: every return path) and the shader clock at both ends (clock frequency = d clock / d wall)
#define PE_NTRACE 2048
static __device__ unsigned pe_trace_seq;
static __device__ long long pe_trace[PE_NTRACE][5];
struct PeTrace {
  int slot;
  __device__ __forceinline__ explicit PeTrace(int id) : slot(-1) {
    if ((threadIdx.x | blockIdx.x | blockIdx.y | blockIdx.z) == 0) {
      slot = (int)(atomicAdd(&pe_trace_seq, 1u) % PE_NTRACE);
      pe_trace[slot][0] = id;
      pe_trace[slot][2] = 0;
      pe_trace[slot][3] = (long long)clock64();
      pe_trace[slot][1] = (long long)wall_clock64();
    }
  }
  __device__ __forceinline__ ~PeTrace() {
    if (slot >= 0) {
      pe_trace[slot][2] = (long long)wall_clock64();
      pe_trace[slot][4] = (long long)clock64();
    }
  }
};
#define PE_KTRACE(id) PeTrace pe_ktrace_obj(id)
// The stamp / trace arrays are one copy per translation unit (static __device__): every launch unit defines a fetcher of
// ITS copy with this macro, and engine.cpp's pe_debug_stamps / pe_debug_trace merge them (stamps: non-zero slots; trace:
// records appended, counters summed and reset).
#define PE_TRACE_FETCHER(tag) \
  extern "C" int pe_trace_fetch_##tag(long long* stamps, long long* trace, unsigned* count) { \
    hipDeviceSynchronize(); \
    int rc = (int)hipMemcpyFromSymbol(stamps, HIP_SYMBOL(pe_stamps), sizeof(long long) * PE_NSTAMP_K * PE_NSTAMP_I); \
    if (rc) return rc; \
    rc = (int)hipMemcpyFromSymbol(trace, HIP_SYMBOL(pe_trace), sizeof(long long) * PE_NTRACE * 5); \
    if (rc) return rc; \
    rc = (int)hipMemcpyFromSymbol(count, HIP_SYMBOL(pe_trace_seq), sizeof(unsigned)); \
    if (rc) return rc; \
    const unsigned zero = 0; \
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(pe_trace_seq), &zero, sizeof(unsigned)); \
  }
#else
#define PE_STAMP(k, i) do {} while (0)
#define PE_KTRACE(id) do {} while (0)
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define pe_mfma_32x32x2(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define pe_mfma_16x16x4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
// 16 independent 4x4 outer products (blocks): lane l gives A[block l/4][row l%4] and B[block l/4][col l%4], VGPR r of lane l
// holds D[block l/4][row r][col l%4] (checked on the hardware: scripts/microbench/mfma4x4.hip)
#define pe_mfma_4x4x1(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 0, 0, 0)
// lanes of a wave run in lockstep and LDS accesses of one wave complete in order: only the compiler
// must not reorder across this point
#define PE_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
// hides a loop-invariant value from LICM (keeps per-element epilogue addresses from being hoisted into
// hundreds of registers across the tile loop)
#define PE_OPAQUE(x) asm volatile("" : "+v"(x))
// makes a wave-uniform value provably uniform (SGPR) for the compiler
#define PE_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
// nothing is scheduled across this point: keeps a block of prefetch loads ahead of the MFMAs they overlap
// with (the machine scheduler otherwise sinks each load next to its use to save registers)
#define PE_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// the next `n` instructions of class `mask` (LLVM SchedGroupMask: MFMA 0x8, VMEM read 0x20, DS read 0x100) of the
// enclosing scheduling region, in the order these calls appear: a compile-time interleave of loads between MFMAs
#define PE_SCHED_GROUP(mask, n) __builtin_amdgcn_sched_group_barrier((mask), (n), 0)
template <class T> __device__ __forceinline__ T* pe_uniform_ptr(T* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (T*)(((unsigned long long)hi << 32) | lo);
}
// Bounds-checked row load through a buffer descriptor: the hardware range check returns 0 for any
// element outside [0, n) (negative indices wrap to huge unsigned offsets), so halo / tail / padded-channel
// reads need no clamps, selects or branches. `row` and `n` must be wave-uniform.
typedef __amdgpu_buffer_rsrc_t pe_rowsrc;
__device__ __forceinline__ pe_rowsrc pe_make_row(const float* row, int n) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(row), 0, n * 4, 0x00020000);
}
// same, for a base/length the compiler cannot prove wave-uniform although they are (values that depend on
// the wave index or live in vector registers): without this every access runs in a waterfall loop
__device__ __forceinline__ pe_rowsrc pe_make_row_u(const float* row, int n) {
  return pe_make_row(pe_uniform_ptr(row), __builtin_amdgcn_readfirstlane(n));
}
__device__ __forceinline__ float pe_row_load(pe_rowsrc r, int idx) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)((unsigned)idx * 4u), 0, 0));
}
// element (vidx + sidx): vidx per lane, sidx wave-uniform -> the uniform part rides in an SGPR, no VALU add
__device__ __forceinline__ float pe_row_load_so(pe_rowsrc r, int vidx, int sidx) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)((unsigned)vidx * 4u), sidx * 4, 0));
}
__device__ __forceinline__ void pe_row_store_so(pe_rowsrc r, int vidx, int sidx, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)((unsigned)vidx * 4u), sidx * 4, 0);
}
// two / four consecutive floats in one store instruction; idx must be a multiple of 2 / 4 elements past an aligned base
// (the callers check) and the whole group inside the row (a group that straddles the end is the caller's business)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void pe_row_store2(pe_rowsrc r, int idx, float a, float b) {
  f32x2 v = {a, b};
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned, v), r, (int)((unsigned)idx * 4u), 0, 0);
}
__device__ __forceinline__ void pe_row_store4(pe_rowsrc r, int idx, float a, float b, float c, float d) {
  f32x4 v = {a, b, c, d};
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), r, (int)((unsigned)idx * 4u), 0, 0);
}
// four consecutive floats (16-byte aligned index) in one instruction
__device__ __forceinline__ f32x4 pe_row_load4(pe_rowsrc r, int idx) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, idx * 4, 0, 0));
}
// element (vidx + sidx) .. + 3: vidx per lane, sidx wave-uniform (SGPR offset, outside the hardware's range check)
__device__ __forceinline__ f32x4 pe_row_load4_so(pe_rowsrc r, int vidx, int sidx) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, vidx * 4, sidx * 4, 0));
}
// the hardware's transcendental units: log2, and sine / cosine of an angle given in turns (x = 1 is a full circle)
__device__ __forceinline__ float pe_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float pe_sin_turns(float x) { return __builtin_amdgcn_sinf(x); }
__device__ __forceinline__ float pe_cos_turns(float x) { return __builtin_amdgcn_cosf(x); }
// leaky-relu for 0 < slope < 1 in two VALU ops: median(v, v*slope, +inf) = max(v, v*slope)
__device__ __forceinline__ float pe_lrelu(float v, float slope) {
  return __builtin_amdgcn_fmed3f(v, v * slope, __builtin_inff());
}
#endif

// ---- XCD-aware tile order. Workgroups go to the P XCDs (each with its own L2) round-robin by LINEAR workgroup id, so
// the workgroups of one residue class mod P share an L2. pe_xcd_tile maps workgroup `id` of `n` onto a tile such that
// every XCD owns one contiguous run of tiles (a bijection on [0, n); P <= 1: the identity).
__device__ __forceinline__ int pe_xcd_tile(int id, int n, int P) {
  if (P <= 1) return id;
  const int q = n / P, r = n - q * P, i = id / P, j = id - i * P;
  return j * q + (j < r ? j : r) + i;
}
// The same for a (column tile x, row part y) grid plane whose row parts read DIFFERENT weights: tiles are numbered row
// part-major, so an XCD's run covers one or two row parts and its L2 fetches only their weights. Used by ffn_kernel (16
// slices of the hidden dimension: 30.8 MB of fabric traffic per launch for 5.2 MB of operands with (x, y) = blockIdx,
// profiles/r04_pmc_traffic.json; 9.98 -> 8.81 us per launch with the map). Measured and NOT used (profiles/r04_notes.md):
// the split-K convs (the gate conv 10.03 -> 10.26 us: 27 workgroups asking ONE L2 for the same weight lines at the same
// moment cost more than eight L2s fetching them once each), and a column tile-major order for the tiled conv kernel
// (+1 % at one utterance, +0.5 % / +6.5 % on the medium / high voice at 64). The planes of a 3-D grid are mapped one by
// one: within a plane a residue class still sits on ONE XCD.
__device__ __forceinline__ void pe_xcd_xy(int P, int& bx, int& by) {
  if (P <= 1) return;
  const int nx = (int)gridDim.x;
  const int t = pe_xcd_tile(by * nx + bx, nx * (int)gridDim.y, P);
  by = t / nx;
  bx = t - by * nx;
}
#include <stdexcept>
#include <string>

#define PE_HIP(expr)                                                                         \
  do {                                                                                       \
    hipError_t pe_e_ = (expr);                                                               \
    if (pe_e_ != hipSuccess)                                                                 \
      throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(pe_e_));        \
  } while (0)
