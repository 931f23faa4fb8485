// Launchers of the split-operand conv GEMM kernel (conv_bf3.h: conv_split_kernel), split modes bf16x3 / f16x3 / bf16x6:
// one translation unit of the library build.
#include "launch.h"

#include "conv_bf3.h"
#include "mrf_split.h"

namespace pe {
namespace launch {

#define PE_B2(SM, WM, WN, MT, NT, G) (const void*)conv_split_kernel<SM, WM, WN, MT, NT, G, 64>, (const void*)conv_split_kernel<SM, WM, WN, MT, NT, G, 128>
#define PE_B10(SM) PE_B2(SM, 2, 2, 2, 2, false), PE_B2(SM, 2, 2, 1, 2, false), PE_B2(SM, 1, 4, 1, 2, false), PE_B2(SM, 2, 2, 2, 2, true), \
                   PE_B2(SM, 1, 4, 2, 1, true)

void init_bf3() {
#ifndef PE_EMU
  const int lim = 160 * 1024;
  const void* ks[] = {PE_B10(0), PE_B10(1), PE_B10(2),
#define PE_MS7(SM) (const void*)mrf_split_kernel<SM, 32, 1, 1>, (const void*)mrf_split_kernel<SM, 32, 2, 1>, (const void*)mrf_split_kernel<SM, 32, 3, 1>, \
                   (const void*)mrf_split_kernel<SM, 32, 4, 1>, (const void*)mrf_split_kernel<SM, 64, 1, 2>, (const void*)mrf_split_kernel<SM, 64, 2, 2>, \
                   (const void*)mrf_split_kernel<SM, 64, 3, 2>
                      PE_MS7(0), PE_MS7(1)};
#undef PE_MS7
  for (const void* k : ks) PE_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
#endif
}
#undef PE_B10
#undef PE_B2

template <int SM>
static void conv_split_mode(int cfg, bool gate, int halo, dim3 grid, size_t smem, hipStream_t stream, const ConvP& p) {
#define PE_BF3_LAUNCH(WM, WN, MT, NT, G)                                                                       \
  do {                                                                                                         \
    if (halo == 64) PE_LAUNCH((conv_split_kernel<SM, WM, WN, MT, NT, G, 64>), grid, dim3(256), smem, stream, p); \
    else PE_LAUNCH((conv_split_kernel<SM, WM, WN, MT, NT, G, 128>), grid, dim3(256), smem, stream, p);         \
  } while (0)
  if (gate) {
    if (cfg == 0) PE_BF3_LAUNCH(2, 2, 2, 2, true);
    else PE_BF3_LAUNCH(1, 4, 2, 1, true);
  } else {
    switch (cfg) {
      case 0: PE_BF3_LAUNCH(2, 2, 2, 2, false); break;
      case 1: PE_BF3_LAUNCH(2, 2, 1, 2, false); break;
      default: PE_BF3_LAUNCH(1, 4, 1, 2, false); break;
    }
  }
#undef PE_BF3_LAUNCH
}

void conv_bf3(int sm, int cfg, bool gate, int halo, dim3 grid, size_t smem, hipStream_t stream, const ConvP& p) {
  if (sm == 1) conv_split_mode<1>(cfg, gate, halo, grid, smem, stream, p);
  else if (sm == 2) conv_split_mode<2>(cfg, gate, halo, grid, smem, stream, p);
  else conv_split_mode<0>(cfg, gate, halo, grid, smem, stream, p);
}

template <int SM>
static void mrf_split_mode(int cp, int ou, dim3 grid, hipStream_t stream, const MrfP& p) {
  const size_t smem = mrf_smem_bytes(cp, ou);
#define PE_MRFS(CP_, OU_, HU_) PE_LAUNCH((mrf_split_kernel<SM, CP_, OU_, HU_>), grid, dim3(64 * MRF_NW), smem, stream, p)
  if (cp == 32) {
    if (ou == 1) PE_MRFS(32, 1, 1); else if (ou == 2) PE_MRFS(32, 2, 1); else if (ou == 3) PE_MRFS(32, 3, 1); else PE_MRFS(32, 4, 1);
  } else {
    if (ou == 1) PE_MRFS(64, 1, 2); else if (ou == 2) PE_MRFS(64, 2, 2); else PE_MRFS(64, 3, 2);
  }
#undef PE_MRFS
}

void mrf_split(int sm, int cp, int ou, dim3 grid, hipStream_t stream, const MrfP& p) {
  if (sm == 1) mrf_split_mode<1>(cp, ou, grid, stream, p);
  else mrf_split_mode<0>(cp, ou, grid, stream, p);
}

}  // namespace launch
}  // namespace pe

#ifdef PE_STAMPS
PE_TRACE_FETCHER(bf3)
#endif
