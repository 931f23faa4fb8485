// Host-side launchers of the kernels, one translation unit per kernel family (launch_conv.cpp, launch_front.cpp,
// launch_tail.cpp) so that the library builds in parallel: the engine (engine_launch.cpp) sees only the parameter structs (params.h) and
// these prototypes. A launcher picks the template instantiation for its runtime arguments and issues ONE launch on
// `stream` (counted in pe::g_launches). `init()` of a family raises the dynamic-LDS limit of its kernels (160 KiB per
// workgroup on gfx950) and must run once per process before the first launch.
#pragma once
#include "../pe_rt.h"
#include "params.h"

namespace pe {
namespace launch {

// ---- conv GEMM family (launch_conv.cpp)
void init_conv();
// tiled implicit GEMM: cfg = tile configuration id (engine_internal.h CFG_*), halo = 64 | 128 columns of staging slack
void conv_tile(int cfg, bool gate, int halo, dim3 grid, size_t smem, hipStream_t stream, const ConvP& p);
// one-tap convs of a batched call, B operand straight from global memory (conv1x1.h):
// grid = (64-column tiles, blocks of 64 rows, utterances)
void conv1x1(dim3 grid, hipStream_t stream, const ConvP& p);
// split-K forms: nw = 4 | 8 | 12 waves
void conv_splitk(bool gate, int nw, dim3 grid, size_t smem, hipStream_t stream, const ConvP& p);
// half (gate only): half a 32-channel group per workgroup on six waves with the whole K range in flight (conv_splitk.h GT = 2)
void conv_splitk16(bool gate, dim3 grid, size_t smem, hipStream_t stream, const ConvP& p, bool half = false);
// WN gate conv of short calls on 64-row x 12-column workgroups (gate4.h): grid = (12-column tiles, 32-channel groups, utterances)
void gate4(dim3 grid, size_t smem, hipStream_t stream, const ConvP& p);
void conv_group(bool wide, dim3 grid, size_t smem, hipStream_t stream, const ConvG& g);
// the tiled kernel on up to three sibling convs of one tile configuration (conv_mfma.h: conv_mfma_group_kernel)
void conv_tile_group(int cfg, int halo, dim3 grid, size_t smem, hipStream_t stream, const ConvG& g);
void conv_group_sum(dim3 grid, size_t smem, hipStream_t stream, const ConvP& p);

// ---- split-bf16 tiled conv GEMM (launch_bf3.cpp; opt-in matrix mode PIPER_HIP_MATRIX=bf16x3)
void init_bf3();
// cfg: 0 = 128 x 128 tile, 1 = 64 x 128, 2 = 32 x 256 (engine_launch.cpp BF3_BM / BF3_BN); gate needs cfg 0 or 1
// sm: split mode of conv_split_kernel (0 bf16x3, 1 f16x3, 2 bf16x6)
void conv_bf3(int sm, int cfg, bool gate, int halo, dim3 grid, size_t smem, hipStream_t stream, const ConvP& p);
// the fused MRF stage on the 16-bit pipe (mrf_split.h): sm = 0 (bf16x3) or 1 (f16x3); cp / ou as launch::mrf
void mrf_split(int sm, int cp, int ou, dim3 grid, hipStream_t stream, const MrfP& p);

// ---- text encoder / duration predictor / flow glue (launch_front.cpp)
void init_front();
void embed(dim3 grid, hipStream_t stream, const EmbedP& p);
void attention(int dk, dim3 grid, size_t smem, hipStream_t stream, const AttnP& p);
void attno(dim3 grid, size_t smem, hipStream_t stream, const AttnOP& p);        // attention + conv_o + LN, dk = 96 x 2 heads (attno.h)
void attn4(bool long_rows, dim3 grid, size_t smem, hipStream_t stream, const AttnOP& p);   // the same on 4-query workgroups (attn4.h); long_rows: more than 128 ids per utterance
void layer_norm(dim3 grid, hipStream_t stream, const LnP& p);
void dds_layer(int nchunks, dim3 grid, size_t smem, hipStream_t stream, const DdsP& p);
void dds_layer4(dim3 grid, size_t smem, hipStream_t stream, const DdsP& p);      // 4-column form, 192 channels (dds4.h)
void colchain(dim3 grid, size_t smem, hipStream_t stream, const ColP& p);
void colchain4(dim3 grid, size_t smem, hipStream_t stream, const ColP& p);     // 4-column forms (col4.h): weights in pack4 order
void lngemm(dim3 grid, size_t smem, hipStream_t stream, const LnGemmP& p);
void lngemm4(dim3 grid, size_t smem, hipStream_t stream, const LnGemmP& p);
void ffn(dim3 grid, size_t smem, hipStream_t stream, const FfnP& p);            // fused small-call FFN (ffn.h)
void cf_pre(dim3 grid, hipStream_t stream, const float* z0, long z_bs, const float* w, const float* bias, const float* xg,
            long g_bs, int g_cs, float* out, long o_bs, int o_cs, const int* lens, int H);
void spline_inverse(dim3 grid, hipStream_t stream, const float* hproj, long h_bs, int h_cs, float* z1, long z_bs,
                    const int* lens, float inv_sqrt_h);
void scale(dim3 grid, hipStream_t stream, const float* in, float* out, long n, float s);
void duration(dim3 grid, hipStream_t stream, const DurP& p);
void randn(hipStream_t stream, float* out, long rows, int cols, long stride, long row0, const unsigned long long* state,
           int site);
void regulate(dim3 grid, hipStream_t stream, const RegP& p);
void xcc_probe(hipStream_t stream, int* out64);       // 64 workgroups, out64[i] = XCC id of workgroup i
void cond(dim3 grid, hipStream_t stream, const float* emb_g, int gin, const int* sids, const float* w, const float* bias,
          int rows, float* out, int o_bs);

// ---- vocoder stage kernels and the generator tail (launch_tail.cpp)
void init_tail();
void mrf(int cp, int ou, dim3 grid, hipStream_t stream, const MrfP& p);
void mrf_sum(dim3 grid, hipStream_t stream, const float* r0, const float* r1, const float* r2, float* out, long bs, int cs,
             const int* lens, int len_mul, float scale);
void conv_post(dim3 grid, hipStream_t stream, const float* x, long x_bs, int x_cs, const float* w, int Cin, float slope,
               const int* lens, int len_mul, float* audio, long a_bs, unsigned* absmax);
void pcm16(dim3 grid, hipStream_t stream, const float* audio, long a_bs, const unsigned* absmax, const int* lens,
           int len_mul, short* pcm, long p_bs, short* host);
void window_copy(dim3 grid, hipStream_t stream, const float* z, int zs, const int* win, float* out, int ws, int C);

}  // namespace launch
}  // namespace pe
