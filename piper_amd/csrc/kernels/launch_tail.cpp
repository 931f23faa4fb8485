// Launchers of the fused MRF stage kernel and the generator tail: one translation unit of the library build.
#include "launch.h"

#include "mrf.h"
#include "post.h"

namespace pe {
namespace launch {

void init_tail() {
#ifndef PE_EMU
  const int lim = 160 * 1024;
  const void* ks[] = {(const void*)mrf_kernel<32, 1, 1>, (const void*)mrf_kernel<32, 2, 1>, (const void*)mrf_kernel<32, 3, 1>, (const void*)mrf_kernel<32, 4, 1>,
                      (const void*)mrf_kernel<64, 1, 2>, (const void*)mrf_kernel<64, 2, 2>, (const void*)mrf_kernel<64, 3, 2>};
  for (const void* k : ks) PE_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
#endif
}

// cp = padded channels (32: one row group of 8 column groups, 1 halo unit per wave; 64: two row groups of 4 column
// groups, 2 halo units per wave); ou = output units per wave (N = 16 * column groups * ou)
void mrf(int cp, int ou, dim3 grid, hipStream_t stream, const MrfP& p) {
  const size_t smem = mrf_smem_bytes(cp, ou);
#define PE_MRF(CP_, OU_, HU_) PE_LAUNCH((mrf_kernel<CP_, OU_, HU_>), grid, dim3(64 * MRF_NW), smem, stream, p)
  if (cp == 32) {
    if (ou == 1) PE_MRF(32, 1, 1); else if (ou == 2) PE_MRF(32, 2, 1); else if (ou == 3) PE_MRF(32, 3, 1); else PE_MRF(32, 4, 1);
  } else {
    if (ou == 1) PE_MRF(64, 1, 2); else if (ou == 2) PE_MRF(64, 2, 2); else PE_MRF(64, 3, 2);
  }
#undef PE_MRF
}

void mrf_sum(dim3 grid, hipStream_t stream, const float* r0, const float* r1, const float* r2, float* out, long bs, int cs,
             const int* lens, int len_mul, float scale) {
  PE_LAUNCH(mrf_sum_kernel, grid, dim3(256), 0, stream, r0, r1, r2, out, bs, cs, lens, len_mul, scale);
}

void conv_post(dim3 grid, hipStream_t stream, const float* x, long x_bs, int x_cs, const float* w, int Cin, float slope,
               const int* lens, int len_mul, float* audio, long a_bs, unsigned* absmax) {
  PE_LAUNCH(conv_post_kernel, grid, dim3(256), 0, stream, x, x_bs, x_cs, w, Cin, slope, lens, len_mul, audio, a_bs, absmax);
}

void pcm16(dim3 grid, hipStream_t stream, const float* audio, long a_bs, const unsigned* absmax, const int* lens,
           int len_mul, short* pcm, long p_bs, short* host) {
  PE_LAUNCH(pcm16_kernel, grid, dim3(256), 0, stream, audio, a_bs, absmax, lens, len_mul, pcm, p_bs, host);
}

void window_copy(dim3 grid, hipStream_t stream, const float* z, int zs, const int* win, float* out, int ws, int C) {
  PE_LAUNCH(window_copy_kernel, grid, dim3(64), 0, stream, z, zs, win, out, ws, C);
}

}  // namespace launch
}  // namespace pe

#ifdef PE_STAMPS
PE_TRACE_FETCHER(tail)
#endif
