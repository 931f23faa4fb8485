// Launch parameters and host-visible constants of every kernel (what engine_launch.cpp fills in and the launch translation
// units kernels/launch_*.cpp pass on). Device code lives in the per-kernel headers next to this file; each struct's
// fields are documented there where the kernel uses them.
#pragma once
#include <cstddef>

namespace pe {

// ---- conv GEMM kernels (conv_common.h)
static constexpr int KC = 32;           // input channels staged per K-chunk of the conv GEMM
enum Epi { EPI_STORE = 0, EPI_RESADD = 1, EPI_GATE = 2, EPI_WNRS = 3, EPI_SUBFROM = 4,
           EPI_ACCUM = 5, EPI_CONVT = 6 };
enum Act { ACT_NONE = 0, ACT_RELU = 1 };
struct ConvP {
  const float* x; long x_bs; int x_cs;          // input  x[b][ci][t]
  const float* wp;                              // packed weights (engine_pack.cpp: pack_conv)
  const float* wp16;                            // conv_splitk16_kernel: the same in 16x16x4 fragment order, or null
  const float* wpb;                             // conv_split_kernel: 16-bit split-term fragments of the matrix mode (engine_pack.cpp pack_matrix), or null
  const float* wunscale;                        // conv_split_kernel, mode f16x3: the weights were packed times 1 / *wunscale (a power of two)
  const float* wpg4;                            // gate4_kernel: the gate conv in [group][tap][k quad][lane][4] order (192 input channels), or null
  const float* bias;                            // per output channel or null
  const float* bias2; int bias2_bs;             // per-utterance extra bias (speaker cond) or null
  float* out; long o_bs; int o_cs;
  const float* res; long r_bs; int r_cs;        // residual input (may alias out)
  float* out2; long o2_bs; int o2_cs;           // second output (WN skip accumulator)
  const int* lens; int len_mul;                 // valid input length = lens[b]*len_mul
  int Cin, rows;                                // real input channels; GEMM rows (Cout, or Cout*up)
  int nchunks;                                  // ceil(Cin/KC)
  int ntaps, dil, padl;                         // tap k reads x[t + k*dil - padl]
  int xhalo;                                    // (ntaps-1)*dil
  float in_slope;                               // leaky-relu slope applied to x while staging (1 = none)
  int epi, act;
  int split;                                    // GATE: H ; WNRS: rows < split go to h, rest to skip
  int up, padT;                                 // CONVT: stride and padding
  unsigned up_magic;                            // CONVT: ceil(2^32 / up): row / up == (row * up_magic) >> 32 for row < 2^16
  int up_vec;                                   // CONVT, conv_mfma_kernel: a lane's four accumulator rows are consecutive output samples of ONE
                                                // channel when up is a multiple of 4 (4: one 16-byte store) or both phases of two channels when
                                                // up == 2 (2: two 8-byte stores); 0: one by one
  int mode;                                     // ACCUM: 0 first,1 middle,2 last,3 only ; WNRS: 1 = first layer
  float alpha;                                  // ACCUM last/only: scale
  int tpb;                                      // conv_mfma_kernel: column tiles walked by one workgroup
  int tgroups;                                  // conv_splitk_kernel: 1, or 2 = two halves of the waves split the taps
  // conv_splitk_body<..., MS = true> only: K = the concatenation of nseg convs of one shape whose outputs are summed
  // (segment 0 repeats x / wp / ntaps / dil / padl); res2 / res3 = the residual tensors of segments 1 / 2
  int nseg;
  const float* seg_x[3]; const float* seg_wp[3];
  int seg_ntaps[3], seg_dil[3], seg_padl[3];
  const float* res2; const float* res3;
};

// ---- grouped split-K launches (conv_splitk.h)
struct ConvG {
  ConvP c[3];
  int n, B;
};

// ---- relative-position attention (attention.h)
struct AttnP {
  const float* qkv; long q_bs; int q_cs;
  const float* relk; const float* relv;     // [2w+1][dk]
  float* out; long o_bs; int o_cs;
  const int* lens;
  int H, dk, window;
  int SP;                                   // score row stride in LDS: odd, >= round_up(max len, 64)
  float qscale;
  float* sglobal;                           // attn_long_kernel: [utterance][head][query block][32][SP] score slabs (null: LDS)
};
// attno_kernel (attno.h): attention + conv_o + residual + norm_layers_1 of an encoder layer in one launch
struct AttnOP {
  const float* qkv; long q_bs; int q_cs;
  const float* relk; const float* relv;     // [2w+1][dk]
  const int* lens;
  int window;
  int SP;                                   // score row stride in LDS: round_up(max len, 64) + 2 (== 2 mod 32)
  float qscale;
  const float* wo16; const float* bo;       // conv_o in pack16 order, bias [H]
  const float* gamma; const float* beta;    // norm_layers_1
  float* x; long x_bs; int x_cs;            // residual in, LayerNorm output (in place)
  const float* wo4;                         // attn4_kernel: conv_o in pack4 order; SP = round_up(max len, 64) + 4 there
  const float* kT; long kt_bs;              // attn4_kernel: K as [utterance][channel quad][column stride q_cs][4] (written by the q/k/v launch)
  const float* vQ;                          // attn4_kernel: V as [utterance][column quad][192][4] (same batch stride)
  int xcd;                                  // attn4_kernel: XCD-contiguous column tiles (col4.h c4_tile)
};
static constexpr int ATT_QB = 32;           // queries per workgroup (one MFMA tile)
static constexpr int ATT_KCH = 64;          // keys staged per V chunk
static constexpr int ATT_MAXDK = 128;

// ---- LayerNorm (layernorm.h)
struct LnP {
  const float* in; long i_bs; int i_cs;
  float* out; long o_bs; int o_cs;
  const float* gamma; const float* beta;
  const int* lens;
  int C;
};
static constexpr int LN_COLS = 8, LN_NV = 8;

// ---- spline (spline.h)
static constexpr int SPL_NB = 10;

// ---- DDSConv layer (dds.h)
struct DdsP {
  const float* x; long x_bs; int x_cs;
  float* out; long o_bs; int o_cs;
  const float* dw_w; const float* dw_b; int dw_k, dw_dil;
  const float* g1; const float* b1; const float* g2; const float* b2;
  const float* bias;                        // 1x1 conv bias
  const float* wp16;                        // 1x1 conv weights in the 16x16x4 fragment order (engine_pack.cpp)
  const float* wp4;                         // the same in the 4x4x1 fragment order (dds_layer4_kernel; null: not packed)
  int nchunks;                              // ceil(H / 32)
  const int* lens;
  int H;
  // Optional fold of ConvFlow.pre + DDSConv's "x = x + g" into the layer input (modules.py:504-505, 118-119), first
  // layer of a ConvFlow: the input is  pre_w[c] * (z0[t] * z_scale) + pre_b[c] + x[c][t]  with x = the conditioning g.
  const float* pre_z; long pre_z_bs;        // z0 row of utterance b (null: no fold)
  const float* pre_w; const float* pre_b;
  float z_scale;                            // noise_scale_w on the first flow (z is still the raw N(0,1) draw), else 1
  // Optional second 1x1 conv on the layer's output columns (last layer of a DDSConv: dp.proj / ConvFlow.proj,
  // models.py:65, modules.py:507), weights in the 16x16x4 fragment order; the layer output itself is then not stored.
  const float* post_w16; const float* post_w4; const float* post_bias; int post_rows;
  float* post_out; long po_bs; int po_cs;   // plain store of the post conv (dp.proj), or null
  // Optional spline epilogue (ConvFlow, modules.py:508-526): the post conv's 29 rows are the per-position parameters;
  // z1 <- rq_spline_inverse(z1 * z_scale), z0 <- z0 * z_scale (pass-through), both [2][Ts] tensors may alias.
  const float* zin; long zin_bs; int z_cs; int c0, c1;
  float* zout; long zout_bs;
  float inv_sqrt_h;
  int xcd;                                  // dds_layer4_kernel: XCDs the dispatch round-robins over (0: unknown), col4.h c4_tile
};

// ---- 1x1 conv chains (colchain.h)
struct ColP {
  const float* in1; long in1_bs; int in1_cs; int K1;
  const float* w1; const float* b1; int rows1;
  int mode;
  const float* res; long res_bs; int res_cs;            // mode 0
  const float* gamma; const float* beta;
  float* out; long out_bs; int out_cs;
  float* x1; long x1_bs; int x1_cs;                     // mode 1 (updated in place)
  const float* w2; const float* b2; int rows2;          // w2 == null: no second GEMM (last coupling layer)
  float* out2; long o2_bs; int o2_cs;
  const int* lens;
  int first;                                            // mode 2 (colchain4_kernel): first WN layer -- the skip sum is not read
  int xcd;                                              // colchain4_kernel: XCDs the dispatch round-robins over (0: unknown)
  // colchain4_kernel<true> (mode 1): the last WN layer's res/skip conv in front -- GEMM 1 reads (w0.in0 + b0) + in1
  const float* in0; long in0_bs; int in0_cs;
  const float* w0; const float* b0;
  float* kT; long kt_bs;                                // mode 3, optional: row part 1 also as [channel quad][column][4] (see LnGemmP)
  float* vQ;                                            // ... and row part 2 as [column quad][192][4]
};
// ---- fused FFN (ffn.h): partial outputs per 48-row slice of the hidden dimension
struct FfnP {
  const float* x; long x_bs; int x_cs;           // norm_layers_1 output [192][T]
  const float* w1p; const float* b1;             // conv_1: pack_ffn order per slice, bias [FC]
  const float* w2p;                              // conv_2: pack_ffn order per slice (its bias is added by the consumer)
  float* parts; long p_bs; int nslices;          // [utterance][4-column tile][slice][192][4], p_bs floats per utterance
  const int* lens;
  int xcd;                                       // XCDs the dispatch round-robins over -> (column tile, slice) by pe_xcd_xy; 0 / 1: blockIdx
};
struct LnGemmP {
  const float* in; long in_bs; int in_cs;        // y = x + ffn(x)  (parts != null: the residual x alone)
  // lngemm4_kernel only: y = in + pbias + sum of the nparts partial FFN outputs (ffn_kernel), in slice order
  const float* parts; long p_bs; int nparts;     // [utterance][4-column tile][slice][192][4]
  const float* pbias;
  int xcd;                                       // lngemm4_kernel: XCDs the dispatch round-robins over (0: unknown)
  const float* gamma; const float* beta;
  float* xout; long x_bs; int x_cs;              // LN(y)
  const float* w16; const float* bias; int rows; // pack16 order, all parts; part z owns rows [32 NVT z, 32 NVT (z + 1))
  float* out; long o_bs; int o_cs;
  const int* lens;
  // lngemm4_kernel, optional: rows >= split belong to a SECOND conv over the same LN(y) (enc_p.proj stacked with dp.pre):
  // they go to out2 (row 0 = GEMM row split) and take the per-utterance bias vector bias2 (speaker conditioning) as well
  int split; float* out2; long o2_bs; int o2_cs; const float* bias2; long bias2_bs;
  // lngemm4_kernel, optional (q/k/v conv in front of attn4_kernel): row part 1 (the K rows) is ALSO written as
  // kT[utterance][channel quad 48][column stride][4], so that the attention kernel takes four channel steps of a key as ONE
  // 16-byte load that is contiguous across the lanes (= keys) of a wave
  float* kT; long kt_bs;
  // ... and row part 2 (the V rows) as vQ[utterance][column quad][192][4]: four keys of a channel per 16-byte load,
  // contiguous across the lanes (= channels) of a wave (same batch stride kt_bs)
  float* vQ;
};

// ---- text embedding = the first kernel of every run (attention.h: embed_kernel)
struct EmbedP {
  const int* ids; int ids_bs;                  // [B][Ts] phoneme ids
  const int* lens;                             // [B]
  const float* emb; int H; float scale;
  float* out; long o_bs; int o_cs;
  unsigned long long* rng;                     // device {seed, runs so far, last ingested upload, -}
  // Zero-copy inputs (null: upload() copied the input block to the device). The pinned host mirror of the block is read
  // in place: ids and lengths by every workgroup that needs them, lengths / speaker ids / {seed, counter} published to
  // device memory for the kernels behind this one.
  const unsigned long long* h_rng; const int* h_lens; const int* h_sids; const int* h_ids;
  int* d_lens; int* d_sids;
  // Optional (small calls): the duration noise of models.py:111 -- rows 2 b, 2 b + 1 of the site-0 stream, draw_cols columns,
  // exactly the values randn_kernel writes -- drawn by workgroup (0, 0, 0), the one that advances the generator state, with
  // the state it publishes; null: randn_kernel draws it in a launch of its own.
  float* draw_out; long draw_stride; int draw_rows, draw_cols;
};

// ---- durations, N(0,1) generator, length regulator (duration.h)
static constexpr int MAX_FRAMES = 60000;      // per-utterance activations stay below the 2 GiB descriptor range
struct DurP {
  const float* z0; long z_bs; float m0, es0, length_scale;
  const int* lens; int* dur; int* cum; int d_bs; int* frames; float* logw_out;
  int* frames_host; int* frames_clamped; int frame_cap;
};
static constexpr int RNG_PITCH = 65536;       // >= MAX_FRAMES and >= the longest id sequence
static inline unsigned randn_blocks(long rows, int cols) { return (unsigned)(rows * ((cols + 1023) / 1024)); }
struct RegP {
  const float* stats; long s_bs; int s_cs;     // [B][2C][Ts]: m_p rows [0,C), logs_p rows [C,2C)
  const int* cum; int d_bs;
  const int* tlens; const int* frames;
  float* noise; long n_bs; int n_cs;           // [B][C][>=F]: read (injected noise), or written by the kernel's own draws (gen)
  float noise_scale;
  float* out; long o_bs; int o_cs;
  int C;
  unsigned* absmax;                            // per-utterance peak accumulator of conv_post_kernel: zeroed here
  const unsigned long long* rng; int gen;      // gen: the prior noise (site 1) is drawn here and stored to `noise`
  DurP dur; int fold;                          // fold: the durations are computed here too (duration_kernel's fields)
};
static constexpr int REG_MAXT = 4096;          // ids whose cumulative durations fit the LDS copy; longer: search in global memory

// ---- generator tail (post.h)
static constexpr int POST_K = 7, POST_OPT = 4, POST_CU = 8, POST_CG = 4, POST_SPB = 64 * POST_OPT;

// ---- fused MRF stage (mrf.h)
enum { MRF_RES = 1, MRF_KEEP = 2, MRF_FINAL = 4, MRF_INIT = 8, MRF_RESTAGE = 16 };
struct MrfPhase {        // one conv of one resblock chain; 12 ints wide (the kernel copies the table to LDS as ints)
  const float* bias;
  int ntaps, dil;
  int e;                 // columns of halo its OUTPUT still needs (0 for the last conv of a resblock)
  int src, dst;          // LDS activation buffers (0 = stage input window, 1 = chain buffer); dst < 0: none
  int flags;             // RES: + running x (registers); KEEP: result becomes the running x; FINAL: add to the MRF sum;
                         // INIT: running x = stage input (first conv of a resblock); RESTAGE: reload buffer 0 first
  int pad[4];
};
static_assert(sizeof(MrfPhase) == 48, "MrfPhase is read as 12 ints");
struct MrfP {
  const float* x; long x_bs; int x_cs;
  float* out; long o_bs; int o_cs;
  const int* lens; int len_mul;
  const MrfPhase* phases; int nphases;
  const float* wstream; int wfloats;
  const float* wunscale; // mrf_split_kernel, mode f16x3: per phase, the power of two that undoes the packing scale of its weights
  int C;                 // real channels (<= CP)
  int N;                 // output columns per workgroup (16 * NCG * OU)
  int wcols;             // window columns in use: hxa + N + the stage's halo (<= the row stride)
  int hxa;               // window column of the first output column (halo rounded up to 16)
  int cu_lo, cu_hi;      // 16-column units any phase needs: [cu_lo, cu_hi)
  int nleft, nhalo;      // halo units left of the output columns / in total
  float slope, alpha;
  int stride, n0off;     // workgroup bx owns window output columns [bx * stride - n0off, ... + N): (N, 0) without the tail
  // generator tail fused into the last stage (post_w != nullptr; out is not written): conv_post weights [C][7], waveform,
  // per-utterance peak (post.h: conv_post_kernel); stride = N - 6, n0off = 3
  const float* post_w; float* audio; long a_bs; unsigned* absmax; float post_slope;
};
static constexpr int MRF_NW = 8, MRF_PAD = 128, MRF_MAXPH = 24;
// row stride of the LDS window (== 16 mod 32). 32 channels with 4 output units per wave (N = 512, one utterance's last
// stage in ONE round over the chip) take the whole 160 KB: 2 x 32 x 624 floats + pads + table = 161 920 B
static constexpr int mrf_ws(int cp, int ou = 1) { return cp == 32 ? (ou >= 4 ? 624 : 528) : 304; }
static constexpr size_t mrf_smem_bytes(int cp, int ou = 1) {
  return ((size_t)2 * MRF_PAD + (size_t)2 * cp * mrf_ws(cp, ou) + MRF_MAXPH * 12) * sizeof(float);
}

}  // namespace pe
