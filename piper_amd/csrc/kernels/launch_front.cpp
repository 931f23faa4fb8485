// Launchers of the text-encoder / duration-predictor / flow-glue kernels: one translation unit of the library build.
#include "launch.h"

#include "attention.h"
#include "attno.h"
#include "attn4.h"
#include "colchain.h"
#include "dds.h"
#include "dds4.h"
#include "ffn.h"
#include "duration.h"
#include "layernorm.h"
#include "glue.h"
#include "spline.h"

namespace pe {
namespace launch {

void init_front() {
#ifndef PE_EMU
  const int lim = 160 * 1024;
  const void* ks[] = {(const void*)attn_kernel<0>, (const void*)attn_kernel<48>, (const void*)attn_kernel<96>,
                      (const void*)attno_kernel<96>, (const void*)attn4_kernel<96, false>, (const void*)attn4_kernel<96, true>};
  for (const void* k : ks) PE_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
#endif
}

void embed(dim3 grid, hipStream_t stream, const EmbedP& p) { PE_LAUNCH(embed_kernel, grid, dim3(64), 0, stream, p); }

// compiled per head width (96 / 48); <0> = any even width <= 128 with guarded loops
void attention(int dk, dim3 grid, size_t smem, hipStream_t stream, const AttnP& p) {
  if (p.sglobal) {
    if (dk == 96) PE_LAUNCH(attn_long_kernel<96>, grid, dim3(256), smem, stream, p);
    else if (dk == 48) PE_LAUNCH(attn_long_kernel<48>, grid, dim3(256), smem, stream, p);
    else PE_LAUNCH(attn_long_kernel<0>, grid, dim3(256), smem, stream, p);
    return;
  }
  if (dk == 96) PE_LAUNCH(attn_kernel<96>, grid, dim3(256), smem, stream, p);
  else if (dk == 48) PE_LAUNCH(attn_kernel<48>, grid, dim3(256), smem, stream, p);
  else PE_LAUNCH(attn_kernel<0>, grid, dim3(256), smem, stream, p);
}

void attno(dim3 grid, size_t smem, hipStream_t stream, const AttnOP& p) {
  PE_LAUNCH(attno_kernel<96>, grid, dim3(512), smem, stream, p);
}

void attn4(bool long_rows, dim3 grid, size_t smem, hipStream_t stream, const AttnOP& p) {
  if (long_rows) PE_LAUNCH((attn4_kernel<96, true>), grid, dim3(256), smem, stream, p);
  else PE_LAUNCH((attn4_kernel<96, false>), grid, dim3(256), smem, stream, p);
}

void layer_norm(dim3 grid, hipStream_t stream, const LnP& p) { PE_LAUNCH(ln_kernel, grid, dim3(256), 0, stream, p); }

// <3> / <6> are compiled for exactly 96 / 192 padded channels; <8> takes any width up to 256
void dds_layer(int nchunks, dim3 grid, size_t smem, hipStream_t stream, const DdsP& p) {
  if (nchunks == 3) PE_LAUNCH(dds_layer16_kernel<3>, grid, dim3(512), smem, stream, p);
  else if (nchunks == 6) PE_LAUNCH(dds_layer16_kernel<6>, grid, dim3(512), smem, stream, p);
  else PE_LAUNCH(dds_layer16_kernel<8>, grid, dim3(512), smem, stream, p);
}

void dds_layer4(dim3 grid, size_t smem, hipStream_t stream, const DdsP& p) {
  PE_LAUNCH(dds_layer4_kernel, grid, dim3(256), smem, stream, p);
}

void colchain(dim3 grid, size_t smem, hipStream_t stream, const ColP& p) {
  PE_LAUNCH(colchain_kernel<6>, grid, dim3(512), smem, stream, p);
}

void lngemm(dim3 grid, size_t smem, hipStream_t stream, const LnGemmP& p) {
  PE_LAUNCH(lngemm_kernel<6>, grid, dim3(512), smem, stream, p);
}

void colchain4(dim3 grid, size_t smem, hipStream_t stream, const ColP& p) {
  if (p.w0) PE_LAUNCH(colchain4_kernel<true>, grid, dim3(256), smem, stream, p);
  else PE_LAUNCH(colchain4_kernel<false>, grid, dim3(256), smem, stream, p);
}

void lngemm4(dim3 grid, size_t smem, hipStream_t stream, const LnGemmP& p) {
  PE_LAUNCH(lngemm4_kernel, grid, dim3(256), smem, stream, p);
}

void ffn(dim3 grid, size_t smem, hipStream_t stream, const FfnP& p) {
  PE_LAUNCH(ffn_kernel, grid, dim3(256), smem, stream, p);
}

void xcc_probe(hipStream_t stream, int* out64) { PE_LAUNCH(xcc_probe_kernel, dim3(64), dim3(64), 0, stream, out64); }

void cf_pre(dim3 grid, hipStream_t stream, const float* z0, long z_bs, const float* w, const float* bias, const float* xg,
            long g_bs, int g_cs, float* out, long o_bs, int o_cs, const int* lens, int H) {
  PE_LAUNCH(cf_pre_kernel, grid, dim3(64), 0, stream, z0, z_bs, w, bias, xg, g_bs, g_cs, out, o_bs, o_cs, lens, H);
}

void spline_inverse(dim3 grid, hipStream_t stream, const float* hproj, long h_bs, int h_cs, float* z1, long z_bs,
                    const int* lens, float inv_sqrt_h) {
  PE_LAUNCH(spline_inverse_kernel, grid, dim3(64), 0, stream, hproj, h_bs, h_cs, z1, z_bs, lens, inv_sqrt_h);
}

void scale(dim3 grid, hipStream_t stream, const float* in, float* out, long n, float s) {
  PE_LAUNCH(scale_kernel, grid, dim3(256), 0, stream, in, out, n, s);
}

void duration(dim3 grid, hipStream_t stream, const DurP& p) { PE_LAUNCH(duration_kernel, grid, dim3(256), 0, stream, p); }

void randn(hipStream_t stream, float* out, long rows, int cols, long stride, long row0, const unsigned long long* state,
           int site) {
  PE_LAUNCH(randn_kernel, dim3(randn_blocks(rows, cols)), dim3(256), 0, stream, out, rows, cols, stride, row0, state, site);
}

void regulate(dim3 grid, hipStream_t stream, const RegP& p) { PE_LAUNCH(regulate_kernel, grid, dim3(256), 0, stream, p); }

void cond(dim3 grid, hipStream_t stream, const float* emb_g, int gin, const int* sids, const float* w, const float* bias,
          int rows, float* out, int o_bs) {
  PE_LAUNCH(cond_kernel, grid, dim3(128), 0, stream, emb_g, gin, sids, w, bias, rows, out, o_bs);
}

}  // namespace launch
}  // namespace pe

#ifdef PE_STAMPS
PE_TRACE_FETCHER(front)
#endif
