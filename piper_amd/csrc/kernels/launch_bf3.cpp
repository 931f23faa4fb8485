// Launchers of the split-bf16 conv GEMM kernel (conv_bf3.h): one translation unit of the library build.
#include "launch.h"

#include "conv_bf3.h"

namespace pe {
namespace launch {

#define PE_B2(WM, WN, MT, NT, G) (const void*)conv_bf3_kernel<WM, WN, MT, NT, G, 64>, (const void*)conv_bf3_kernel<WM, WN, MT, NT, G, 128>

void init_bf3() {
#ifndef PE_EMU
  const int lim = 160 * 1024;
  const void* ks[] = {PE_B2(2, 2, 2, 2, false), PE_B2(2, 2, 1, 2, false), PE_B2(1, 4, 1, 2, false),
                      PE_B2(2, 2, 2, 2, true), PE_B2(1, 4, 2, 1, true)};
  for (const void* k : ks) PE_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
#endif
}
#undef PE_B2

void conv_bf3(int cfg, bool gate, int halo, dim3 grid, size_t smem, hipStream_t stream, const ConvP& p) {
#define PE_BF3_LAUNCH(WM, WN, MT, NT, G)                                                                 \
  do {                                                                                                   \
    if (halo == 64) PE_LAUNCH((conv_bf3_kernel<WM, WN, MT, NT, G, 64>), grid, dim3(256), smem, stream, p); \
    else PE_LAUNCH((conv_bf3_kernel<WM, WN, MT, NT, G, 128>), grid, dim3(256), smem, stream, p);         \
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

}  // namespace launch
}  // namespace pe

#ifdef PE_STAMPS
PE_TRACE_FETCHER(bf3)
#endif
