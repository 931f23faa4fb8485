// Launchers of the conv GEMM kernels (conv_mfma.h, conv_splitk.h): one translation unit of the library build.
#include "launch.h"

#include "conv_mfma.h"
#include "conv1x1.h"
#include "conv_splitk.h"
#include "gate4.h"

namespace pe {
namespace launch {

void init_conv() {
#ifndef PE_EMU
  const int lim = 160 * 1024;       // > 64 KiB of dynamic LDS needs the attribute (gfx950: 160 KiB per workgroup)
#define PE_K2(WM, WN, MT, NT, KS, G) (const void*)conv_mfma_kernel<WM, WN, MT, NT, KS, G, 64>, (const void*)conv_mfma_kernel<WM, WN, MT, NT, KS, G, 128>
  const void* ks[] = {PE_K2(1, 4, 1, 1, 16, false), PE_K2(2, 2, 1, 1, 16, false), PE_K2(1, 4, 2, 1, 16, true),
                      PE_K2(2, 2, 2, 1, 16, true),
                      (const void*)conv_mfma_group_kernel<2, 2, 1, 1, 16, 64>, (const void*)conv_mfma_group_kernel<2, 2, 1, 1, 16, 128>,
                      (const void*)conv_mfma_group_kernel<1, 4, 1, 1, 16, 64>, (const void*)conv_mfma_group_kernel<1, 4, 1, 1, 16, 128>,
                      (const void*)conv_splitk_kernel<2, true, 8, 3>, (const void*)conv_splitk_kernel<2, true, 4, 3>,
                      (const void*)conv_splitk_kernel<1, false, 8, 4>, (const void*)conv_splitk_kernel<1, false, 4, 4>,
                      (const void*)conv_splitk_kernel<2, true, 12, 2>, (const void*)conv_splitk_kernel<1, false, 12, 4>,
                      (const void*)conv_splitk16_kernel<true, 12, 2>, (const void*)conv_splitk16_kernel<false, 8, 4>,
                      (const void*)conv_splitk_group_kernel<4, 2, 64>, (const void*)conv_splitk_group_kernel<4, 2, 128>,
                      (const void*)conv_splitk_sum_kernel<4, 2>};
#undef PE_K2
  for (const void* k : ks) PE_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
#endif
}

// tile configurations {WM, WN, MT, NT, KS}: ids as engine_internal.h's CFG_* -- B (gate: 64 x 128), C (32 x 128), S (64 x 64), G (gate: 128 x 64)
void conv_tile(int cfg, bool gate, int halo, dim3 grid, size_t smem, hipStream_t stream, const ConvP& p) {
#define PE_CONV_LAUNCH(WM, WN, MT, NT, KS, G)                                                                  \
  do {                                                                                                         \
    if (halo == 64) PE_LAUNCH((conv_mfma_kernel<WM, WN, MT, NT, KS, G, 64>), grid, dim3(256), smem, stream, p); \
    else PE_LAUNCH((conv_mfma_kernel<WM, WN, MT, NT, KS, G, 128>), grid, dim3(256), smem, stream, p);          \
  } while (0)
  // (the 128x128, 64x128 non-gate and 256-column configurations lost every A/B of rounds 1-3 and are no longer compiled)
  if (gate) {
    if (cfg == 1) PE_CONV_LAUNCH(1, 4, 2, 1, 16, true);
    else PE_CONV_LAUNCH(2, 2, 2, 1, 16, true);
  } else {
    if (cfg == 2) PE_CONV_LAUNCH(1, 4, 1, 1, 16, false);
    else PE_CONV_LAUNCH(2, 2, 1, 1, 16, false);
  }
#undef PE_CONV_LAUNCH
}

// sibling convs of one tile configuration (non-gate: cfg 2 = 32 x 128 tiles, else 64 x 64) in one launch
void conv_tile_group(int cfg, int halo, dim3 grid, size_t smem, hipStream_t stream, const ConvG& g) {
  if (cfg == 2) {
    if (halo == 64) PE_LAUNCH((conv_mfma_group_kernel<1, 4, 1, 1, 16, 64>), grid, dim3(256), smem, stream, g);
    else PE_LAUNCH((conv_mfma_group_kernel<1, 4, 1, 1, 16, 128>), grid, dim3(256), smem, stream, g);
  } else {
    if (halo == 64) PE_LAUNCH((conv_mfma_group_kernel<2, 2, 1, 1, 16, 64>), grid, dim3(256), smem, stream, g);
    else PE_LAUNCH((conv_mfma_group_kernel<2, 2, 1, 1, 16, 128>), grid, dim3(256), smem, stream, g);
  }
}

void conv1x1(dim3 grid, hipStream_t stream, const ConvP& p) { PE_LAUNCH((conv1x1_kernel<1>), grid, dim3(256), 0, stream, p); }

void conv_splitk(bool gate, int nw, dim3 grid, size_t smem, hipStream_t stream, const ConvP& p) {
  if (gate) {
    if (nw == 12) PE_LAUNCH((conv_splitk_kernel<2, true, 12, 2>), grid, dim3(768), smem, stream, p);
    else if (nw == 8) PE_LAUNCH((conv_splitk_kernel<2, true, 8, 3>), grid, dim3(512), smem, stream, p);
    else PE_LAUNCH((conv_splitk_kernel<2, true, 4, 3>), grid, dim3(256), smem, stream, p);
  } else {
    if (nw == 12) PE_LAUNCH((conv_splitk_kernel<1, false, 12, 4>), grid, dim3(768), smem, stream, p);
    else if (nw == 8) PE_LAUNCH((conv_splitk_kernel<1, false, 8, 4>), grid, dim3(512), smem, stream, p);
    else PE_LAUNCH((conv_splitk_kernel<1, false, 4, 4>), grid, dim3(256), smem, stream, p);
  }
}

void conv_splitk16(bool gate, dim3 grid, size_t smem, hipStream_t stream, const ConvP& p, bool half) {
  if (gate && half) PE_LAUNCH((conv_splitk16_kernel<true, 6, 5, 2>), grid, dim3(64 * 6), smem, stream, p);
  else if (gate) PE_LAUNCH((conv_splitk16_kernel<true, 12, 2>), grid, dim3(64 * 12), smem, stream, p);
  else PE_LAUNCH((conv_splitk16_kernel<false, 8, 4>), grid, dim3(64 * 8), smem, stream, p);
}

void gate4(dim3 grid, size_t smem, hipStream_t stream, const ConvP& p) {
  PE_LAUNCH(gate4_kernel, grid, dim3(64 * G4_NW), smem, stream, p);
}

void conv_group(bool wide, dim3 grid, size_t smem, hipStream_t stream, const ConvG& g) {
  if (wide) PE_LAUNCH((conv_splitk_group_kernel<4, 2, 128>), grid, dim3(256), smem, stream, g);
  else PE_LAUNCH((conv_splitk_group_kernel<4, 2, 64>), grid, dim3(256), smem, stream, g);
}

void conv_group_sum(dim3 grid, size_t smem, hipStream_t stream, const ConvP& p) {
  PE_LAUNCH((conv_splitk_sum_kernel<4, 2>), grid, dim3(256), smem, stream, p);
}

}  // namespace launch
}  // namespace pe

#ifdef PE_STAMPS
PE_TRACE_FETCHER(conv)
#endif
