// Launch policy of the engine: every threshold and A/B knob that decides WHICH kernel form a launch takes, in ONE place.
//
// The engine (engine*.cpp) asks the questions below with plain numbers -- workgroup counts, columns per call, halo
// widths -- and never reads the environment or compares against a tuning constant itself. The knobs are one table
// (environment variable, field, range, meaning), read once at engine creation; their defaults are the measured choices
// (profiles/r01..r04_notes.md carry the A/B behind each one). They exist so that every kernel variant can be forced on any
// shape by the parity tests and A/B-ed on one box: the product needs none of them. A value that is not an integer is an
// error (engine creation fails with the variable's name); a value outside the range is clamped to it.
#pragma once
#include <cstddef>

namespace pe {

struct LaunchPolicy {
  // ---- knobs (field = default)
  long mrf = 1;               // fused MRF stage kernel: 0 off (conv by conv), 1 by the measured policy, 2 wherever it applies
  long mrf_maxf = 1100;       // ResBlock1 stages on 32 channels: batch frames up to which the fused kernel is used in mode 1
  long mrf_ou = 0;            // force the output units per wave of mrf_kernel (1..4); 0 = cost model
  long mrf_tail = 1;          // generator tail (conv_post, tanh, peak) inside the last stage's mrf_kernel
  long mrf_split = 1;         // split matrix modes bf16x3 / f16x3: the fused MRF stage on the 16-bit pipe (mrf_split_kernel); 0 = the f32 fused kernel / conv by conv
  long bf3_minf = 1100;       // split matrix modes without a fused 16-bit stage kernel: batch frames from which the <= 64-channel stages run conv by conv on the 16-bit pipe
  long splitk_max = 96;       // tile-kernel workgroups below which a conv goes to the split-K kernels (0: always the tiled kernel)
  long splitk16 = 2;          // 16-column split-K: 0 off, 1 WN gate conv, 2 + long-K plain convs, 3 everywhere (tests)
  long wide_splitk = 1;       // 12-wave split-K workgroups: 0 off, 1 WN gate conv, 2 always (tests)
  long tpb = 0;               // tiled kernel: column tiles walked by one workgroup (0 = 1, the measured choice)
  long group_mrf = 1;         // sibling resblock convs of a wider one-utterance stage as grouped launches: 0 off, 2 without the K-concatenated last step
  long group_maxb = 160;      // ... as split-K launches while one conv of the stage is fewer than this many 64 x 64 tiles, as tiled launches above
  long group_tiled = 1;       // sibling resblock convs of a stage that runs the TILED kernel as grouped launches while one conv is only a few tiles per CU
  long colchain = 1;          // colchain_kernel / lngemm_kernel: 0 off, 1 by batch size, 2 always
  long col4 = 1;              // 4-column forms of the 192-channel chains: 0 off, 1 by batch size, 2 always
  long col4_maxc = 1024;      // ... up to this many ids per call (text encoder, duration predictor)
  long ffn = 1;               // encoder FFN as one launch (ffn_kernel) wherever the 4-column chains run
  long attno = 1;             // attention + conv_o + norm_layers_1 as one launch wherever the 4-column chains run
  long attn4 = 1;             // ... on 4-query workgroups (attn4_kernel): 0 off, 1 for utterances up to attn4_maxc ids, 2 wherever attno applies
  long attn4_maxc = 512;      // ids per UTTERANCE (the call's longest) up to which attention runs on 4-query workgroups: each reads all of K and V of its utterance, 4x attno's L2 traffic (measured up to 512 ids: 25.0 against 35.8 us; any batch the 4-column chains take)
  long fuse_dp = 1;           // ConvFlow.pre / proj / spline fused into the DDSConv layer launches
  long spec = 1;              // speculative stage-B sizing: the whole utterance as one graph for <= spec_max_batch utterances
  long spec_expect = 1;       // speculative graphs planned for the expected frame counts (0: for the bucket capacity)
  long pcm_zc = 1;            // int16 PCM written straight into pinned host memory by pcm16_kernel
  long ids_zc = 1;            // phoneme ids / lengths / speaker ids read straight from the pinned host block by embed_kernel (no copy in front of the graph)
  long no_graph = 0;          // launch kernels directly instead of replaying hipGraphs
  long graphs = 256;          // hipGraphs kept per engine (least recently used evicted one at a time)
  long convt_vec = 1;         // polyphase up-conv tiles stored as 16- / 8-byte pieces straight from the accumulators (0: one 4-byte store per phase)
  long xcd = -1;              // XCDs the dispatch round-robins over: -1 = probed at engine creation, 0 = tiles in workgroup order
  long xcd_ffn = 1;           // fused FFN: (column tile, slice) dealt to the XCDs slice-major (0: blockIdx order)
  long stack_pre = 1;         // small calls: enc_p.proj and dp.pre as one lngemm4_kernel launch over the stacked matrix (0: two launches)
  long chain_rs = 1;          // small calls: the last WN layer's res/skip conv in front of the post + pre chain launch (0: a launch of its own)
  long gate4 = 1;             // short calls: the WN gate conv over 192 channels on 64-row x 12-column workgroups with the 4x4x1 MFMA (gate4_kernel) up to 640 workgroups
  long gate_half = 1;         // short one-utterance calls: the WN gate conv on half a 32-channel group per workgroup (6 waves) while twice the workgroups still fit one per CU
  long conv1x1 = 1;           // batched one-tap convs through conv1x1_kernel (B operand straight from global memory): 0 = the tiled kernel
  long ws_budget_mb = 0;      // MiB a stage's workspace may take: 0 = a third of the device's memory
  long attn_long = 0;         // attention score slabs in global memory at every length (tests; by default only where they do not fit LDS)
  long prof_sites = 0;        // level-2 profile rows of the tiled conv kernel per conv SHAPE (tuning aid)
  long debug_keep = 0;        // test hook: keep z_p for pe_debug_tensor
  long debug_poison = 0;      // test hook: every activation workspace is filled with NaN bit patterns when it is allocated

  // ---- thresholds without a knob (measured once, profiles/r02_notes.md / r03_notes.md)
  static constexpr long colchain_max_ids = 4096, colchain_max_frames = 8192;   // colchain / lngemm replace conv + LN pairs up to here
  static constexpr long col4_max_frames = 2048;   // 4-column WN res/skip conv and coupling pre: frames per call
  static constexpr long ffn_max_cols = 2048;      // ffn_kernel's partial-output buffer is allocated for this many columns
  static constexpr int spec_max_batch = 4;        // utterances per call up to which stage B is sized speculatively
  static constexpr long ids_zc_max = 65536;       // padded ids per call up to which embed_kernel reads them from host memory (beyond: one H2D copy)
  static constexpr long group_max_blocks64 = 700; // grouped sibling launches: 64-column tiles of the stage up to which they pay
  static constexpr long group_tiled_max_blocks = 2048;   // grouped TILED sibling launches: tile workgroups of one conv (8 per CU) up to which grouping pays

  struct Knob { const char* env; long LaunchPolicy::*field; long lo, hi; const char* doc; };
  static const Knob* knobs(int* n);
  // reads every knob that is set; throws std::runtime_error naming the variable when its value is not an integer
  void read_env();
  // the table as a JSON array (name, default, lo, hi, doc): pe_policy_describe() of the C ABI, DESIGN.md section 4.1
  static const char* describe();
  // the one string-valued knob: PIPER_HIP_MATRIX = f32 (default) | bf16x3 | f16x3 | bf16x6 (opt-in split-operand matrix
  // modes, kernels/conv_bf3.h); anything else throws. matrix_mode_env: -1 = f32, else the kernel's split mode SM (0 / 1 / 2)
  static int matrix_mode_env();
  static bool matrix_bf3_env() { return matrix_mode_env() >= 0; }

  // ---- decisions ------------------------------------------------------------------------------------------------
  // conv routing (engine_launch.cpp Engine::conv): `blocks` = workgroups the tiled kernel would launch, `halo` =
  // (taps - 1) * dilation, `units` = 32-channel chunks x taps of the K loop
  // (`cin`: the split-K kernels and conv1x1_kernel put a channel row's offset into the SGPR offset of their buffer loads,
  // which the hardware's range check of gfx9 / CDNA does not include: only convs of whole 32-channel chunks go there,
  // the tiled kernel masks channels explicitly)
  bool splitk(long blocks, int halo, int cin) const { return blocks < splitk_max && halo <= 32 && cin % 32 == 0; }
  bool groupable(bool gate, bool convt, long blocks, int halo, int cin) const {
    return !gate && !convt && blocks < splitk_max && halo <= 96 && cin % 32 == 0;
  }
  bool splitk_16col(bool packed16, bool convt, bool gate, int units) const {
    return packed16 && !convt && ((gate && splitk16 >= 1) || (!gate && splitk16 >= 2)) && (units >= 24 || splitk16 >= 3);
  }
  bool splitk_12wave(bool gate, int units, int nchunks, int ntaps) const {
    return (wide_splitk == 1 && gate && units >= 24 && nchunks <= 6 && ntaps >= 4) || wide_splitk == 2;
  }
  bool one_tap_direct(bool gate, bool convt, int ntaps, int cin) const {
    return conv1x1 && !gate && !convt && ntaps == 1 && cin % 32 == 0;
  }
  bool gate_half_groups(bool gate, int nchunks, int ntaps, long workgroups_whole) const {
    return gate_half && gate && nchunks == 6 && ntaps <= 5 && 2 * workgroups_whole <= 256;      // one workgroup per CU at most
  }
  // the 12-column form: more, smaller workgroups than the 16-column split-K form. Measured (profiles/r05_notes.md, call 12):
  // equal at 210 workgroups (128 ids: 9.4 against 10.6 us per launch, step +-0.1 %), -12 % per launch at 420 (256 ids or two
  // utterances: step -3.1 %), +4 % at 840 (four utterances)
  static constexpr long gate4_max_workgroups = 640;
  bool gate_12col(bool gate, bool packed4, int ntaps, int dil, long workgroups) const {
    return gate4 && gate && packed4 && ntaps <= 5 && dil == 1 && (gate4 == 2 || workgroups <= gate4_max_workgroups);
  }
  int tiles_per_workgroup() const { return tpb > 0 ? (int)tpb : 1; }
  // 192-channel chains
  bool chain16(double cols, bool frames, int k1, int half) const {     // colchain_kernel<6> / lngemm_kernel<6>
    return colchain && k1 == 192 && half == 96 && (colchain == 2 || cols <= (double)(frames ? colchain_max_frames : colchain_max_ids));
  }
  bool chain4(long cols, long limit = 0) const { return col4 && (col4 == 2 || cols <= (limit ? limit : col4_maxc)); }
  bool attn4_ids(long longest_utterance) const { return attn4 == 2 || (attn4 == 1 && longest_utterance <= attn4_maxc); }
  bool chain4_frames(long cols) const { return chain4(cols, col4_max_frames); }
  bool chain_rs_front(long cols) const { return chain_rs && chain4_frames(cols); }
  // fused MRF stage
  bool mrf_build(int channels) const { return mrf != 0 && channels <= 64 && channels % 4 == 0; }
  // `split_fused`: the engine runs a two-term split matrix mode AND holds the stage's split weight stream (mrf_split_kernel):
  // the fused stage then runs on the 16-bit pipe at every batch size (ResBlock2 stages; ResBlock1 stages on 32 channels).
  // Without it (mode bf16x6, or PIPER_HIP_MRF_SPLIT=0) a split mode keeps the f32 fused kernel below bf3_minf frames and goes
  // conv by conv on the 16-bit pipe above.
  bool mrf_stage(bool built, bool resblock1, int padded_channels, double frames, bool matrix_bf3, bool split_fused = false) const {
    if (mrf && built && split_fused) return mrf == 2 || !resblock1 || padded_channels == 32;
    return mrf && built && !(matrix_bf3 && mrf != 2 && frames >= (double)bf3_minf) &&
           (mrf == 2 || !resblock1 || (padded_channels == 32 && frames <= (double)mrf_maxf));
  }
  bool group_stage(int B, int nk, long blocks64, bool buffers_fit) const {
    return group_mrf && B == 1 && nk >= 2 && nk <= 3 && blocks64 < group_max_blocks64 && blocks64 < group_maxb && buffers_fit;
  }
  // ... and past that size every conv of the stage keeps the tiled kernel (its siblings must take ONE route to be grouped)
  bool stage_all_tiled(int B, int nk, long blocks64) const {
    return group_mrf && group_tiled && B == 1 && nk >= 2 && nk <= 3 && blocks64 >= group_maxb;
  }
  bool group_sum() const { return group_mrf != 2; }
  // the tiled kernel's form of the same idea: one conv of the stage is 1..8 tiles per CU, so the CUs that draw one tile
  // more than the rest set the launch time; three siblings in one launch even that out
  bool group_stage_tiled(int nk, long tile_blocks, bool buffers_fit) const {
    return group_tiled && nk >= 2 && nk <= 3 && tile_blocks <= group_tiled_max_blocks && buffers_fit;
  }
  bool speculate(int B) const { return spec && !no_graph && B <= spec_max_batch; }
  bool ids_from_host(long padded_ids) const { return ids_zc && padded_ids <= ids_zc_max; }
};

}  // namespace pe
