// TEST INFRASTRUCTURE ONLY -- a single-threaded functional emulator of the small HIP subset the
// piper_amd kernels use, so that kernel index arithmetic, LDS staging, barriers, shuffles and the
// f32 MFMA fragment layouts can be checked on a machine without a GPU.
//
// It is NOT a CPU fallback of the product: the shipped library (libpiper_hip.so) is built by hipcc
// for gfx950 and never contains this file; only tests/ build and load the emulated variant
// (libpiper_hip_emu.so, -DPE_EMU). Nothing in piper_amd/ loads it.
//
// Model: one block at a time; every HIP thread of the block is a fiber (hand-rolled x86-64 context
// switch); __syncthreads() and the wave-collective operations (shuffles, MFMA) are rendezvous
// points handled by a round-robin scheduler. MFMA fragment layouts follow
// /opt/skills/guides/cdna_hip_programming.md section 3 (f32 32x32x2 and 16x16x4 forms).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <functional>
#include <chrono>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct emu_idx3 { unsigned x, y, z; };
extern emu_idx3 threadIdx, blockIdx, blockDim, gridDim;

typedef int hipError_t;
typedef void* hipStream_t;
struct emu_event { double t; };
typedef emu_event* hipEvent_t;
#define hipSuccess 0
#define hipMemcpyHostToDevice 1
#define hipMemcpyDeviceToHost 2
#define hipMemcpyDeviceToDevice 3
#define hipMemcpyDefault 4
#define hipStreamNonBlocking 1

namespace emu {
extern char* dyn_smem;
void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body);
// every block of the grid at once (kernels whose workgroups wait for each other); no static __shared__ inside
void launch_coop(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body);
void yield();        // a spinning thread lets the others run
void wave_sync();    // all live lanes of the calling fiber's wave
void uniform_check(long long v);   // PE_UNIFORM: aborts when the lanes of a wave disagree
void block_sync();
int lane();
int wave();
float* wave_f(int slot);   // 64-float scratch rows owned by the calling wave (slots 0..7)
// Parity of the calling fiber's n-th wave collective (0 / 1 alternating): collectives exchange their operands through
// TWO sets of scratch rows used in turn, so ONE rendezvous per collective is enough -- a lane that runs ahead writes the
// other set, and it cannot get two collectives ahead because the next rendezvous waits for everybody.
int collective_parity();
}  // namespace emu

inline void __syncthreads() { emu::block_sync(); }

// ---- wave collectives -------------------------------------------------------------------------
template <class T>
inline T emu_exchange(T v, int src_lane) {
  static_assert(sizeof(T) == 4, "emu shuffles are 32-bit");
  float* s = emu::wave_f(emu::collective_parity());
  memcpy(&s[emu::lane()], &v, 4);
  emu::wave_sync();
  T r;
  memcpy(&r, &s[src_lane & 63], 4);
  return r;
}
template <class T> inline T __shfl_xor(T v, int m, int = 64) { return emu_exchange(v, emu::lane() ^ m); }
template <class T> inline T __shfl_down(T v, int d, int = 64) {
  int s = emu::lane() + d;
  return emu_exchange(v, s > 63 ? emu::lane() : s);
}
template <class T> inline T __shfl(T v, int src, int = 64) { return emu_exchange(v, src); }

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_32x32x2_f32: lane l gives A[i=l&31][k=l>>5], B[k=l>>5][j=l&31];
// D col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5); k-ordered fmaf chain.
inline f32x16 emu_mfma_32x32x2(float a, float b, f32x16 c) {
  const int par = emu::collective_parity();
  float* sa = emu::wave_f(2 + par);
  float* sb = emu::wave_f(4 + par);
  int l = emu::lane();
  sa[l] = a;
  sb[l] = b;
  emu::wave_sync();
  int j = l & 31;
  for (int r = 0; r < 16; ++r) {
    int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    float acc = c[r];
    acc = fmaf(sa[i], sb[j], acc);
    acc = fmaf(sa[32 + i], sb[32 + j], acc);
    c[r] = acc;
  }
  return c;
}
// v_mfma_f32_16x16x4_f32: A[l&15][k=l>>4], B[k=l>>4][l&15]; D col = l&15, row = (l>>4)*4 + r.
inline f32x4 emu_mfma_16x16x4(float a, float b, f32x4 c) {
  const int par = emu::collective_parity();
  float* sa = emu::wave_f(2 + par);
  float* sb = emu::wave_f(4 + par);
  int l = emu::lane();
  sa[l] = a;
  sb[l] = b;
  emu::wave_sync();
  int j = l & 15;
  for (int r = 0; r < 4; ++r) {
    int i = (l >> 4) * 4 + r;
    float acc = c[r];
    for (int k = 0; k < 4; ++k) acc = fmaf(sa[k * 16 + i], sb[k * 16 + j], acc);
    c[r] = acc;
  }
  return c;
}
// v_mfma_f32_4x4x1_16B_f32: 16 blocks of a 4x4 outer product; lane l: A[block l/4][row l%4], B[block l/4][col l%4];
// D VGPR r = D[block l/4][row r][col l%4].
inline f32x4 emu_mfma_4x4x1(float a, float b, f32x4 c) {
  const int par = emu::collective_parity();
  float* sa = emu::wave_f(2 + par);
  float* sb = emu::wave_f(4 + par);
  const int l = emu::lane();
  sa[l] = a;
  sb[l] = b;
  emu::wave_sync();
  const int blk = l >> 2;
  for (int r = 0; r < 4; ++r) c[r] = fmaf(sa[4 * blk + r], sb[l], c[r]);
  return c;
}

// v_mfma_f32_32x32x16_bf16: lane l gives A[i=l&31][k = 8*(l>>5) + e], B[k = 8*(l>>5) + e][j=l&31], e = 0..7 (four dwords
// of two bf16 each); D as the f32 32x32 form. Products of bf16 values are exact in f32; f32 accumulation.
typedef __bf16 emu_bf16x8 __attribute__((ext_vector_type(8)));
inline f32x16 emu_mfma_bf16_32x32x16(emu_bf16x8 a, emu_bf16x8 b, f32x16 c) {
  const int l = emu::lane(), j = l & 31;
  float au[4], bu[4];
  memcpy(au, &a, 16);
  memcpy(bu, &b, 16);
  for (int d = 0; d < 4; ++d) {
    const int par = emu::collective_parity();
    float* sa = emu::wave_f(2 + par);
    float* sb = emu::wave_f(4 + par);
    sa[l] = au[d];
    sb[l] = bu[d];
    emu::wave_sync();
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
      float acc = c[r];
      for (int h = 0; h < 2; ++h) {
        unsigned ua, ub;
        memcpy(&ua, &sa[32 * h + i], 4);
        memcpy(&ub, &sb[32 * h + j], 4);
        for (int e = 0; e < 2; ++e) {
          const unsigned xa = (ua >> (16 * e)) << 16, xb = (ub >> (16 * e)) << 16;
          float fa, fb;
          memcpy(&fa, &xa, 4);
          memcpy(&fb, &xb, 4);
          acc = fmaf(fa, fb, acc);
        }
      }
      c[r] = acc;
    }
  }
  return c;
}

// v_mfma_f32_32x32x16_f16: the same operand / result layout with IEEE half terms (subnormal inputs are NOT flushed).
// Products of two halves are exact in f32 (11 x 11 significand bits); f32 accumulation.
typedef _Float16 emu_f16x8 __attribute__((ext_vector_type(8)));
inline f32x16 emu_mfma_f16_32x32x16(emu_f16x8 a, emu_f16x8 b, f32x16 c) {
  const int l = emu::lane(), j = l & 31;
  float au[4], bu[4];
  memcpy(au, &a, 16);
  memcpy(bu, &b, 16);
  for (int d = 0; d < 4; ++d) {
    const int par = emu::collective_parity();
    float* sa = emu::wave_f(2 + par);
    float* sb = emu::wave_f(4 + par);
    sa[l] = au[d];
    sb[l] = bu[d];
    emu::wave_sync();
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
      float acc = c[r];
      for (int h = 0; h < 2; ++h) {
        _Float16 ha[2], hb[2];
        memcpy(ha, &sa[32 * h + i], 4);
        memcpy(hb, &sb[32 * h + j], 4);
        for (int e = 0; e < 2; ++e) acc = fmaf((float)ha[e], (float)hb[e], acc);
      }
      c[r] = acc;
    }
  }
  return c;
}

// v_mfma_f32_16x16x32_{bf16,f16} (gfx950): lane l gives A[i = l & 15][k = 8 * (l >> 4) + e], B[k = 8 * (l >> 4) + e][j = l & 15],
// e = 0..7 (four dwords of two 16-bit terms each); D VGPR r = D[row 4 * (l >> 4) + r][col l & 15] (checked on the hardware:
// scripts/microbench/mfma16x16x32.hip, also that f16 subnormals are not flushed). Term products are exact in f32; f32
// accumulation in ascending k.
template <class T8, class T>
inline f32x4 emu_mfma_16x16x32_t(T8 a, T8 b, f32x4 c) {
  const int l = emu::lane(), j = l & 15;
  float au[4], bu[4];
  memcpy(au, &a, 16);
  memcpy(bu, &b, 16);
  float accs[4] = {c[0], c[1], c[2], c[3]};
  // dword d of a lane = terms 2d, 2d + 1 of its k block: gather all four dwords of every lane, then accumulate k-ascending
  float sa4[4][64], sb4[4][64];
  for (int d = 0; d < 4; ++d) {
    const int par = emu::collective_parity();
    float* sa = emu::wave_f(2 + par);
    float* sb = emu::wave_f(4 + par);
    sa[l] = au[d];
    sb[l] = bu[d];
    emu::wave_sync();
    memcpy(sa4[d], sa, sizeof(float) * 64);
    memcpy(sb4[d], sb, sizeof(float) * 64);
  }
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * (l >> 4) + r;
    float acc = accs[r];
    for (int kb = 0; kb < 4; ++kb)
      for (int d = 0; d < 4; ++d) {
        T ha[2], hb[2];
        memcpy(ha, &sa4[d][16 * kb + i], 4);
        memcpy(hb, &sb4[d][16 * kb + j], 4);
        for (int e = 0; e < 2; ++e) {
          float fa, fb;
          if constexpr (sizeof(T) == 2 && std::is_same<T, _Float16>::value) { fa = (float)ha[e]; fb = (float)hb[e]; }
          else {
            unsigned short ua, ub;
            memcpy(&ua, &ha[e], 2); memcpy(&ub, &hb[e], 2);
            const unsigned xa = (unsigned)ua << 16, xb = (unsigned)ub << 16;
            memcpy(&fa, &xa, 4); memcpy(&fb, &xb, 4);
          }
          acc = fmaf(fa, fb, acc);
        }
      }
    c[r] = acc;
  }
  return c;
}
inline f32x4 emu_mfma_bf16_16x16x32(emu_bf16x8 a, emu_bf16x8 b, f32x4 c) { return emu_mfma_16x16x32_t<emu_bf16x8, __bf16>(a, b, c); }
inline f32x4 emu_mfma_f16_16x16x32(emu_f16x8 a, emu_f16x8 b, f32x4 c) { return emu_mfma_16x16x32_t<emu_f16x8, _Float16>(a, b, c); }

// ---- atomics (single-threaded: plain read-modify-write) ------------------------------------------
template <class T> inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T> inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }

// ---- device math ---------------------------------------------------------------------------------
inline float __expf(float x) { return expf(x); }
inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline float __fdividef(float a, float b) { return a / b; }

// ---- host runtime --------------------------------------------------------------------------------
inline hipError_t hipMalloc(void** p, size_t n) {
  *p = aligned_alloc(256, (n + 255) / 256 * 256);
  return *p ? 0 : 2;
}
inline hipError_t hipFree(void* p) { free(p); return 0; }
inline hipError_t hipMemGetInfo(size_t* fr, size_t* tot) { *fr = *tot = (size_t)16 << 30; return 0; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
inline hipError_t hipHostFree(void* p) { free(p); return 0; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { memcpy(d, s, n); return 0; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
inline hipError_t hipMemcpyPeer(void* d, int, const void* s, int, size_t n) { memcpy(d, s, n); return 0; }
inline hipError_t hipDeviceCanAccessPeer(int* can, int, int) { *can = 1; return 0; }
inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return 0; }
inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return 0; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return 0; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return 0; }
inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
inline hipError_t hipDeviceSynchronize() { return 0; }
inline hipError_t hipSetDevice(int) { return 0; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return 0; }
inline hipError_t hipGetLastError() { return 0; }
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
inline double emu_now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new emu_event{0}; return 0; }
#define hipEventDisableTiming 2
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new emu_event{0}; return 0; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return 0; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = emu_now(); return 0; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  *ms = (float)((b->t - a->t) * 1e3);
  return 0;
}
