// Host engine: owns the packed voice weights in HBM, the workspaces, one HIP stream, and issues the
// kernel sequence that replaces Ort::Session::Run() inside piper::synthesize
// (reference src/cpp/piper.cpp:386-388).
#pragma once
#include <cstdint>
#include <deque>
#include <list>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "pe_rt.h"
#include "kernels/params.h"
#include "policy.h"
#include "weights.h"

namespace pe {

struct PackedConv {
  float* wp = nullptr;      // device, packed for conv_mfma_kernel
  float* wp16 = nullptr;    // device, the same in 16x16x4 fragment order (conv_splitk16_kernel), long-K convs only
  float* wpb = nullptr;     // device, 16-bit split-term fragments (conv_split_kernel), split matrix modes only
  const float* wunscale = nullptr;   // device: wpb holds the weights times 1 / *wunscale (a power of two; 1 except in mode f16x3)
  float* wpg4 = nullptr;    // device, gate convs over 192 channels in the 4x4x1 MFMA's order (gate4_kernel), or null
  float* bias = nullptr;    // device or null
  int rows = 0;             // GEMM rows (real)
  int mtiles = 0;           // packed 32-row tiles (padded to the block tile)
  int Cin = 0, nchunks = 0, ntaps = 1, dil = 1, padl = 0;
  int cfg = 0;              // tile configuration id
  bool gate = false;
  int split = 0;
  int up = 0, padT = 0;     // conv-transpose
  double macs_per_col = 0;  // algorithmic MACs per output column (for roofline accounting)
};

struct DdsW {               // one DDSConv (modules.py:81-129)
  std::vector<float*> dw_w, dw_b, g1, b1, g2, b2;
  std::vector<PackedConv> c1x1;
  std::vector<float*> w16;           // 1x1 weights in the 16x16x4 fragment order (dds_layer16_kernel)
};

struct ProfileRow {
  const char* name;
  double ms = 0;
  double flops = 0;
  long launches = 0;
  double bytes = 0;         // algorithmic HBM bytes (activations in + out + residual operands + weights), level 2 rows
};

struct NoiseIn {            // optional injected N(0,1) draws (parity tests); host pointers
  const float* noise_w = nullptr;  // [B][2][w_stride]
  int64_t w_stride = 0;
  const float* noise_z = nullptr;  // [B][inter][z_stride]
  int64_t z_stride = 0;
};

// Where the packed voice weights live. Default: one device allocation owned by the engine. A caller-provided arena
// (multi-GPU: a device buffer the host framework can hand to RCCL) receives them instead; with `skeleton` the engine only
// lays the arena out (same offsets as on the packing rank, derived from tensor shapes alone) and waits for its content
// to be broadcast into it -- no parsing of weight data, no packing, no upload on that rank.
struct ArenaSpec {
  void* base = nullptr;
  size_t bytes = 0;
  bool skeleton = false;
};

class Engine {
 public:
  Engine(const WeightSet& ws, int device, ArenaSpec arena = ArenaSpec{});
  ~Engine();
  // upper bound of the packed-weight arena for a voice (from tensor shapes), and what the engine actually used
  static size_t arena_bound(const WeightSet& ws);
  size_t arena_used() const { return arena_off_; }
  const void* arena_base() const { return arena_; }

  // Phase 1: copy inputs to HBM. ids: concatenated phoneme ids, offsets[B+1].
  void upload(const int64_t* ids, const int64_t* offsets, int B, const float scales[3],
              const int64_t* sids, const NoiseIn* noise);
  // Phase 2: the whole device pipeline (one host read-back of B frame counts in the middle).
  void run();
  // Phase 3: results to host (pinned buffers owned by the engine, valid until the next upload()).
  void download(bool want_audio, bool want_pcm);

  // Streaming decode of ONE utterance (BASELINE configs[4]; reference infer_onnx_streaming.py:76-124):
  // stream_begin runs everything up to the latent z (encoder, durations, flow) and returns the frame
  // count; every stream_next decodes the next `chunk_frames` frames through HiFiGAN on a window padded
  // with the generator's exact receptive half-width, so the concatenated chunks equal the unchunked
  // waveform (the reference pads by a heuristic 5-10 frames and does not). Returns false when done.
  int stream_begin(const int64_t* ids, int64_t n, const float scales[3], int64_t sid, const NoiseIn* noise);
  bool stream_next(int chunk_frames, const float** audio, const int16_t** pcm, int64_t* nsamples);
  int decoder_halo_frames() const { return halo_frames_; }

  int batch() const { return B_; }
  const std::vector<int64_t>& sample_offsets() { finish_run(); return sample_off_; }
  const float* audio_host() const { return h_audio_; }
  const int16_t* pcm_host() const { return pcm_zc_live_ ? h_pcm_zc_ : h_pcm_; }
  const std::vector<int32_t>& durations_host();   // concatenated per id, same offsets as ids
  const std::vector<int32_t>& frames_host() { finish_run(); return frames_h_; }
  void debug_tensor(const std::string& name, int b, std::vector<float>& out, int* rows, int* cols);
  // test hook: n draws of the engine's N(0,1) generator (randn_kernel) at sampling site 0/1 with the engine's
  // seed and the given call counter, starting at logical row `row` of the site's stream: exactly what run() would draw
  // at that counter for (utterance, channel) = row, columns 0 .. n-1
  void debug_randn(int site, uint64_t call, int64_t row, int64_t n, float* out);
  uint64_t rng_call() const { return call_; }
  long run_launches() const { return run_launches_; }          // kernel launches (graph nodes) of the last run()
  // speculative stage-B sizing (see below): runs issued with a guessed frame bucket / guesses that were too small and
  // cost a second pass of stage B, since the engine was created
  long speculation_runs() const { return spec_runs_; }
  long speculation_misses() const { return spec_misses_; }
  // XCC id of workgroups 0..63 of a 1-D launch as seen at engine creation, and the round-robin period (0: not a round-robin)
  const int* xcc_pattern(int* period) const { if (period) *period = xcd_period_; return xcc_of_; }

  void set_seed(uint64_t s) { seed_ = s; }
  void set_use_graphs(bool on) { use_graphs_ = on; }
  // level 0 off; 1 = one HIP-event pair per pipeline stage; 2 = additionally one pair around every
  // conv/attention/layer-norm launch (per-kernel rows after the five stage rows)
  void set_profile(int level);
  const std::vector<ProfileRow>& profile();
  void reset_profile();
  hipStream_t stream() const { return stream_; }
  const int32_t* arch() const { return arch_; }
  int sample_rate() const { return arch_[A_SR]; }
  int hop() const { return hop_; }
  size_t weight_bytes() const { return weight_bytes_; }

 private:
  // every threshold and A/B knob that picks a kernel form (policy.h; read from the environment at engine creation)
  LaunchPolicy pol_;
  // ---- setup
  void init(const WeightSet& ws);
  float* dev_copy(const std::vector<float>& v);
  float* dev_alloc(size_t nfloats, const float* src);   // bump allocation in the weight arena (+ upload unless skeleton)
  char* arena_ = nullptr; size_t arena_bytes_ = 0, arena_off_ = 0;
  bool arena_owned_ = false, skeleton_ = false;
  float* dev_tensor(const WeightSet& ws, const std::string& name);
  PackedConv pack_conv(const WeightSet& ws, const std::string& wname, const std::string& bname, int dil,
                       int padl_override, bool gate, int in_rev, int out_rev);
  PackedConv pack_qkv(const WeightSet& ws, const std::string& prefix, float** out16 = nullptr);
  PackedConv pack_convT(const WeightSet& ws, const std::string& prefix, int stride);
  PackedConv pack_matrix(const std::vector<float>& W, int rows, int Cin, int ntaps,
                         const std::vector<float>* bias, int nbias, int dil, int padl, bool gate, int split);
  DdsW load_dds(const WeightSet& ws, const std::string& prefix);
  void ensure_stage_a(int B, int Tmax);
  void ensure_stage_b(int Fmax, int batch = 0);

  // ---- launches
  struct View { float* p; long bs; int cs; };
  void conv(const PackedConv& pc, View x, View out, const int* lens, int len_mul, int Lmax, int epi,
            float in_slope = 1.f, int act = 0, View res = View{nullptr, 0, 0},
            View out2 = View{nullptr, 0, 0}, int mode = 0, float alpha = 1.f, const float* bias2 = nullptr,
            int bias2_bs = 0);
  // Grouped launches (conv_splitk_group_kernel): between group_begin() and group_end() conv() records the launch instead
  // of issuing it; group_end() issues all of them (<= 3 independent convs of one launch shape) as one launch.
  bool grouping_ = false;
  bool stage_tiled_ = false;            // the MRF stage being issued takes the tiled kernel for every conv (a single utterance's stage past policy.h: group_maxb)
  bool group_tiled_ = false;            // the open group goes to the TILED kernel (conv_mfma_group_kernel), cfg = group_cfg_
  int group_cfg_ = 0;
  std::vector<struct ConvP> group_;
  double group_flops_ = 0, group_bytes_ = 0;
  int group_ncols_ = 0;
  bool can_group(const PackedConv& pc, int ncols) const;
  bool can_group_tiled(const PackedConv& pc, int ncols) const;
  void group_begin(bool tiled = false);
  void group_end();
  // the recorded convs as ONE GEMM over their concatenated K, summed: out = (sum_j (res_j + conv_j)) * alpha
  bool can_group_sum() const;
  void group_end_sum(View out, const float* bias_sum, float alpha);
  enum { ROUTE_TILE = 0, ROUTE_SPLITK = 1, ROUTE_SPLITK16 = 2 };
  int route(const PackedConv& pc, int ncols, int epi) const;
  void layer_norm(View in, View out, const float* g, const float* b, int C, const int* lens, int Lmax);
  // options of one DDSConv run: ConvFlow.pre folded into the first layer, a 1x1 conv (+ spline) fused after the last
  struct DdsOpt {
    const float* pre_z = nullptr; long pre_z_bs = 0; const float* pre_w = nullptr; const float* pre_b = nullptr;
    float z_scale = 1.f;
    const float* post_w16 = nullptr; const float* post_bias = nullptr; int post_rows = 0;
    View post_out{nullptr, 0, 0};
    const float* zin = nullptr; long zin_bs = 0; int z_cs = 0, c0 = 0, c1 = 1; float* zout = nullptr; long zout_bs = 0;
  };
  void dds(const DdsW& d, View in, View out, View tmp, const DdsOpt* opt = nullptr);
  double dds_bytes(const struct DdsP& p) const;
  double cols_ids_ = 0, cols_frames_ = 0;     // ids / frames of the call being issued (algorithmic byte counts of the profile rows)
  void dds_params(const DdsW& d, View in, View out, View tmp, const DdsOpt* opt, std::vector<struct DdsP>& list);
  float* pack16(const std::vector<float>& W, int rows, int K);    // [16-row tile][q][lane][4] (dds_layer16_kernel)
  // every pack16 matrix with K = 96 / 192 is also packed for the 4x4x1 MFMA of the 4-column kernels (kernels/col4.h):
  // [64-row tile][k quad][lane][4]; looked up by its pack16 pointer (null: not packed)
  float* pack4(const std::vector<float>& W, int rows, int K);
  std::unordered_map<const float*, const float*> w4_of_;
  // fused FFN for small calls (kernels/ffn.h): per-slice weight orders, the partial-output buffer [b][slice][192][Tp]
  // (allocated once for LaunchPolicy::ffn_max_cols columns), and the buffer that holds the encoder output after the last layer (the
  // fused path ping-pongs x_ / y_: lngemm4_kernel must not write LN(y) over the residual other parts still read)
  const float* pack_ffn1(const WeightSet& ws, const std::string& wname);
  const float* pack_ffn2(const WeightSet& ws, const std::string& wname);
  float* ffn_parts_ = nullptr;
  bool stage_a_ffn_fused() const;
  float* stage_a_enc_out() const;
  const float* w4_of(const float* w16) const { auto it = w4_of_.find(w16); return it == w4_of_.end() ? nullptr : it->second; }
  float* dp_proj16_ = nullptr;
  float* dp_pre16_ = nullptr;
  // enc_p.proj and dp.pre stacked for lngemm4_kernel (small calls): pack4 matrix, stacked bias, rows of proj
  float* projpre4_ = nullptr; float* projpre_bias_ = nullptr; int projpre_split_ = 0;
  void colchain(const struct ColP& p, int B, int Lmax, double flops);
  bool conv1x1_col4(const float* w16, const float* bias, int rows, View in, View out, const int* lens, int B, int Lmax,
                    double flops, const float* bias2 = nullptr, long bias2_bs = 0, const float* w4direct = nullptr, int kin = 192,
                    long max_cols = 0);
  const float* pack4_conv_pad192(const WeightSet& ws, const std::string& wname, int in_rev, int out_rev);
  // second: rows >= split of a STACKED pack4 matrix (w4) are another conv over the same LN(y), written to out2 with the
  // per-utterance bias vector bias2 on top (enc_p.proj + dp.pre of a small call in one launch)
  struct LnSecond { const float* w4 = nullptr; int split = 0; View out2{nullptr, 0, 0}; const float* bias2 = nullptr; long bias2_bs = 0; };
  void lngemm(View y, const float* g, const float* b, View x, const float* w16, const float* bias, int rows, View out,
              int T, double flops, const float* parts = nullptr, int nparts = 0, const float* pbias = nullptr,
              const LnSecond* second = nullptr);
  bool proj_pre_stacked() const;     // this call runs enc_p.proj and dp.pre as one lngemm4_kernel launch
  float* pack16_conv(const WeightSet& ws, const std::string& wname, int in_rev, int out_rev);
  // XCD dispatch pattern of this device (xcc_probe_kernel at engine creation): XCC id of workgroups 0..63 of a 1-D launch,
  // and the period P when it is a round-robin over P XCDs (0: anything else -- the 4-column kernels then use tile = id)
  int xcc_of_[64] = {0};
  int xcd_period_ = 0;
  void probe_xcds();
  static size_t col4_smem() { return ((size_t)4 * 196 + 4 * 192 * 4 + 32 + 64 * 4) * sizeof(float); }   // YT | P | red | ZL (kernels/col4.h, dds4.h)
  void issue_stage_a();
  void issue_stage_b();
  void issue_flow();
  void issue_window();
  void issue_decoder(const float* zsrc, const int* lens, int Fmax, double fsum, bool zero_absmax);
  void run_stage(char which, const std::string& key);
  void dispatch_stage(char which);
  void drop_graphs();
  // Speculative stage B (one to LaunchPolicy::spec_max_batch utterances): the frame count F is the path's only data-dependent
  // shape and normally costs a host round trip in the middle of the pipeline. When a previous run of this engine gives
  // a frames-per-id estimate, stage B is launched right behind stage A for a guessed bucket (kernels read the true
  // lengths from device memory, clamped to the allocated capacity), and the guess is verified when the results are
  // fetched; a wrong guess re-runs stage B with the right size.
  bool finish_run();                 // completes a speculative run; false if stage B had to be re-run
  void finish_stage_b_sizes();
  bool spec_pending_ = false;
  int spec_fg_ = 0;
  int spec_fg_force_ = 0;            // warm-up only: the frame bucket the next speculative run is issued for
  // The guess = (slowly decaying maximum of the frames-per-id ratios seen so far) x margin. The margin adapts: a miss
  // widens it (x 1.15, up to 1.5), 32 hits in a row narrow it again (down to 1.10); four misses within 16 speculative
  // runs switch speculation off for the next 64 calls (texts whose lengths vary too much for the estimate to hold).
  float last_ratio_ = 0.f, spec_margin_ = 1.10f;
  long spec_misses_ = 0, spec_runs_ = 0;
  int spec_hit_streak_ = 0, spec_recent_misses_ = 0, spec_recent_runs_ = 0, spec_cooldown_ = 0;
  int* d_framesc_ = nullptr;
  const int* lens_b_ = nullptr;      // frame counts stage B reads: d_frames_, or d_framesc_ on a speculative run
  void prof_begin();
  void prof_end(int row, double flops);

  int32_t arch_[ARCH_INTS];
  int device_ = 0;
  hipStream_t stream_ = nullptr;
  int H_, C_, FC_, nh_, dk_, nlayers_, ksz_, window_, hop_, U_;
  std::vector<void*> owned_;          // device allocations to free
  size_t weight_bytes_ = 0;

  // weights
  float* emb_ = nullptr;
  float* emb_g_ = nullptr;
  struct EncLayer {
    PackedConv qkv, o, f1, f2;
    float* o16 = nullptr;                        // conv_o in pack16 order (colchain_kernel)
    float* qkv16 = nullptr;                      // the fused q/k/v matrix in pack16 order (lngemm_kernel)
    const float *f1p = nullptr, *f2p = nullptr;   // conv_1 / conv_2 in ffn_kernel's per-slice order (kernels/ffn.h; null: not packed)
    float *relk, *relv, *g1, *b1, *g2, *b2;
  };
  std::vector<EncLayer> enc_;
  PackedConv enc_proj_;
  float* enc_proj16_ = nullptr;                  // pack16 order (lngemm_kernel)
  PackedConv dp_pre_, dp_proj_;
  DdsW dp_dds_;
  struct CFlow {
    float *pre_w, *pre_b;
    DdsW dds;
    PackedConv proj;
    float* proj16 = nullptr;             // proj in the 16x16x4 fragment order (fused after the last DDSConv layer)
  };
  std::vector<CFlow> cflows_;
  float ea_m0_ = 0, ea_es0_ = 1;
  float *ea_dev_m_ = nullptr, *ea_dev_logs_ = nullptr;
 public:
  // skeleton engines: call once the arena content has been broadcast into place (fetches the few host-side scalars)
  void arena_ready();
 private:
  struct Rcl {
    PackedConv pre, post;
    float *pre16 = nullptr, *post16 = nullptr;   // the same two 1x1 convs in pack16 order (colchain_kernel)
    const float* pre4pad = nullptr;               // the first layer's pre in pack4 order, K padded to 192 (colchain4_kernel mode 3)
    std::vector<PackedConv> in, rs;
    std::vector<const float*> rs4;   // the res/skip 1x1 convs in pack4 order (colchain4_kernel mode 2; null: not packed)
    int in_off, out_off;             // channel offsets of x0 / x1 in the physical (unflipped) layout
  };
  std::vector<Rcl> rcls_;            // in execution order
  PackedConv dec_pre_;
  struct UpStage {
    PackedConv up;
    int rate, ch;
    std::vector<std::vector<PackedConv>> rb;   // [resblock][conv] (ResBlock1: c1_0,c2_0,c1_1,...)
    float* last_bias_sum = nullptr;            // sum over the resblocks of their LAST conv's bias (conv_splitk_sum_kernel)
    // fused MRF stage (mrf_kernel, kernels/mrf.h): device phase table + weight stream, or not ok: the stage runs conv by conv
    struct HostConv { std::vector<float> w; int co = 0, ci = 0, k = 0, dil = 1; const float* bias = nullptr; };
    std::vector<std::vector<HostConv>> rb_host;   // host copies of the resblock convs, dropped after build_mrf
    void* mrf_phases = nullptr; float* mrf_w = nullptr;
    // the same stage for mrf_split_kernel (matrix modes bf16x3 / f16x3): the weight stream as 16-bit term fragments and,
    // per phase, the factor that undoes the f16 packing scale (both in the arena: they travel with the broadcast)
    float* mrf_wsplit = nullptr; float* mrf_unscale = nullptr; int mrf_wsplit_floats = 0;
    int mrf_wfloats = 0, mrf_cp = 0, mrf_hx = 0;  // padded channels (32 / 64), halo of the stage (widest resblock chain)
    std::vector<struct MrfPhase> mrf_ph;          // host copy of the phases (cost model of the geometry choice)
    bool mrf_ok = false, mrf_rb1 = false;
  };
  struct MrfGeo { int ou = 0, N = 0, cu_lo = 0, cu_hi = 0, nleft = 0, nhalo = 0; };
  bool mrf_geo(const UpStage& st, int len_mul, bool tail, MrfGeo& best) const;      // window geometry by the cost model
  void build_mrf(UpStage& st);
  void mrf(const UpStage& st, View x, View out, const int* lens, int len_mul, int Lmax, bool tail = false);
  // Opt-in split-operand matrix modes PIPER_HIP_MATRIX = bf16x3 | f16x3 | bf16x6 (read at engine creation): the tiled conv
  // GEMMs of the coupling flow and the generator -- and, in the two-term modes, the fused MRF stages -- run on the 16-bit matrix
  // pipe with both f32 operands split into 16-bit terms (kernels/conv_bf3.h, mrf_split.h; 16 / 22 / 24 significand bits per
  // operand, f32 accumulate). The text encoder and the duration predictor stay f32 (the integer durations are those of the
  // f32 path), and so does every latency-bound split-K launch. Default: off, all f32.
  bool matrix_bf3_ = false, pack_bf3_now_ = false;      // a split matrix mode is on / the convs being packed take part in it
  int matrix_sm_ = 0;                                   // ... which: conv_split_kernel's SM (0 bf16x3, 1 f16x3, 2 bf16x6)
  static bool env_bf3();
 public:
  bool matrix_bf3() const { return matrix_bf3_; }
  int matrix_split_mode() const { return matrix_bf3_ ? matrix_sm_ : -1; }
 private:
  std::vector<UpStage> ups_;
  float* post_w_ = nullptr;
  int post_cin_ = 0;
  // speaker conditioning (all per-utterance bias vectors, see cond_kernel)
  struct CondW { float* w; float* b; int rows; };
  CondW cond_dp_{nullptr, nullptr, 0}, cond_dec_{nullptr, nullptr, 0};
  std::vector<CondW> cond_wn_;
  int gin_ = 0, nspk_ = 1;

  // per-call state
  int B_ = 0, Tmax_ = 0, Ts_ = 0, Fmax_ = 0, Fs_ = 0, Tg_ = 0, Fg_ = 0;
  bool use_graphs_ = true;
  // hipGraphExec_t per (stage, shape bucket, scales), least recently used first: a new key beyond graph_cap_ entries
  // evicts ONE graph (the coldest), never the whole cache
  struct GraphEntry { std::string key; void* exec; long launches; };
  std::list<GraphEntry> graphs_;
  std::unordered_map<std::string, std::list<GraphEntry>::iterator> graph_of_;
  long graph_captures_ = 0;                   // captures since the engine was created (pe_graph_stats)
 public:
  long graph_captures() const { return graph_captures_; }
  size_t graphs_cached() const { return graphs_.size(); }
  // Shape buckets of the captured graphs: ids in steps of 32 up to 512, beyond that 8 steps per octave; frames in steps
  // of 64 up to 1024, then 16 per octave. A text of any length maps onto a bounded set of graphs.
  static int id_bucket(int T);
  static int frame_bucket(int F);
  // Pre-size the workspaces (so that no later call grows them -- growth re-creates every graph) and, given a sample
  // utterance, capture the single-utterance graphs of every id bucket up to max_ids by synthesising the sample cut /
  // tiled to each bucket length
  void warmup(int max_batch, int max_ids, float frames_per_id, const float* scales, const int64_t* sample, int64_t n_sample);
 private:
  long run_launches_ = 0;
  unsigned long long* d_rng_ = nullptr;       // {seed, call counter} read by randn_kernel
  float scales_[3] = {0.667f, 1.0f, 0.8f};
  bool have_noise_w_ = false, have_noise_z_ = false;
  bool drew_w_ = false;                       // this call's duration noise is drawn by embed_kernel (small calls), not randn_kernel
  bool fold_dur_ = false;                  // the stage being issued is the one-graph form: regulate_kernel computes the durations
  DurP fold_dp_{};                         // ... from these fields (filled by issue_stage_a)
  std::vector<int64_t> id_off_;
  std::vector<int32_t> tlens_h_, frames_h_, dur_h_;
  std::vector<int64_t> sample_off_;
  uint64_t seed_ = 1234, call_ = 0;
  const float* h_noise_z_ = nullptr;
  int64_t h_noise_z_stride_ = 0;

  // workspaces
  size_t capA_B_ = 0, capA_T_ = 0, capB_B_ = 0, capB_F_ = 0;      // utterances / padded ids of stage A, utterances / frames of stage B
  bool capB_exact_ = false;      // stage B was last sized for exactly one call (memory pressure): it stays while calls fit
  size_t ws_budget_ = 0;             // bytes a stage's workspace may take (a third of the device's memory)
  size_t ws_budget();
  char* wsA_ = nullptr; size_t wsA_bytes_ = 0;
  char* wsB_ = nullptr; size_t wsB_bytes_ = 0;
  int *d_ids_ = nullptr, *d_tlens_ = nullptr, *d_sids_ = nullptr, *d_dur_ = nullptr, *d_cum_ = nullptr,
      *d_frames_ = nullptr;
  // The small inputs of a call -- {seed, call}, lengths, speaker ids, phoneme ids -- are one contiguous block in the stage-A
  // workspace (d_in_) mirrored by a pinned host block (h_in_): upload() fills the host block and enqueues ONE copy, no
  // synchronisation (the pinned block outlives the copy; the call's final synchronisation covers it)
  char* d_in_ = nullptr; char* h_in_ = nullptr; size_t in_bytes_ = 0, h_in_cap_ = 0;
  // short calls: no copy at all -- embed_kernel reads the pinned block in place (policy.h: ids_from_host)
  bool ids_zc_ = false;
  unsigned long long upload_serial_ = 0;
  float* att_s_ = nullptr;           // attention score slabs of long utterances (attn_long_kernel); null while every slab fits LDS
  size_t attn_smem(int T, bool global_scores) const;
  bool attn_scores_global(int T) const;
  float* vQ_ = nullptr;
  float* kT_ = nullptr;              // K transposed [utterance][column][H] for attn4_kernel; written by the q/k/v launch while kt_on_
  bool kt_on_ = false, kt_valid_ = false;      // kt_valid_: the last q/k/v launch did write kT
  float *x_ = nullptr, *y_ = nullptr, *qkv_ = nullptr, *att_ = nullptr, *ffh_ = nullptr, *stats_ = nullptr,
        *xg_ = nullptr, *dh_ = nullptr, *dy_ = nullptr, *dy2_ = nullptr, *hproj_ = nullptr, *z2_ = nullptr,
        *logw_ = nullptr, *noise_w_ = nullptr, *cond_ = nullptr;
  int cond_bs_ = 0;
  std::vector<int> cond_off_wn_;
  int cond_off_dp_ = 0, cond_off_dec_ = 0;
  float *zp_ = nullptr, *fh_ = nullptr, *facts_ = nullptr, *fskip_ = nullptr, *noise_z_ = nullptr;
  float* hb_[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  hipStream_t ls_ = nullptr;       // stream conv()/layer launches go to
  // per-resblock buffers of the grouped sibling schedule (one-utterance calls)
  float* side_[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t side_floats_ = 0;
  float* zwin_ = nullptr;          // streaming: current window of z, [C][Fs]
  int* d_win_ = nullptr;           // streaming: {start, length} of the window in frames
  int halo_frames_ = 0, s_frames_ = 0, s_pos_ = 0, s_wg_ = 0;
  bool s_active_ = false;
  std::vector<int16_t> s_pcm_;
  float* audio_ = nullptr;
  int16_t* pcm_ = nullptr;
  unsigned* absmax_ = nullptr;
  long Ss_ = 0;
  // host pinned
  float* h_audio_ = nullptr; size_t h_audio_cap_ = 0;
  int16_t* h_pcm_ = nullptr; size_t h_pcm_cap_ = 0;
  // pcm16_kernel also writes the samples straight into this pinned host buffer (zero-copy), utterances packed back to
  // back: delivering the PCM costs no copy launches behind the graph (one per utterance before: 1.3 ms at B=64) -- the
  // call ends with one stream synchronisation.
  int16_t* h_pcm_zc_ = nullptr; size_t h_pcm_zc_cap_ = 0;
  bool pcm_zc_live_ = false;                // the last run's PCM is in h_pcm_zc_
  int* h_frames_ = nullptr;

  // profiling
  bool prof_on_ = false;
  int prof_level_ = 0;
  std::vector<ProfileRow> prof_;
  hipEvent_t ev0_ = nullptr, ev1_ = nullptr;
  // level-2 per-launch timing: event pairs recorded without host syncs, resolved in profile()
  struct KEvent { int row; double flops; double bytes; hipEvent_t a, b; };
  std::vector<KEvent> kev_;
  std::vector<hipEvent_t> ev_pool_;
  int kbegin(int row, double flops, double bytes = 0);
  void kend(int h);
  int krow(const char* name);
  int krow(const std::string& name);
  std::deque<std::string> names_;            // storage of generated profile row names (stable pointers)
  float* zp_keep_ = nullptr;
  void free_all();
};

}  // namespace pe
