// The kernel sequence of the pipeline stages (what a hipGraph captures): text encoder + duration predictor (stage A),
// length regulator + coupling flow + HiFiGAN generator + int16 conversion (stage B), the streaming window.
#include "engine_internal.h"

namespace pe {

// Whether stage A of the CURRENT call (B_, Tg_, tlens_h_) runs the encoder FFNs as ffn_kernel launches, and which of the
// two ping-pong buffers then holds the encoder output: functions of the call alone, so that a replayed graph and the code
// that captured it agree (the fused path swaps x / y once per layer; debug_tensor("x_enc") reads the result).
bool Engine::stage_a_ffn_fused() const {
  double tsum = 0;
  for (int b = 0; b < B_; ++b) tsum += tlens_h_[b];
  const bool chain_q = pol_.chain16(tsum, false, H_, 96);
  bool f = pol_.ffn && chain_q && pol_.chain4((long)B_ * Tg_) && ffn_parts_ && (long)B_ * rup(Tg_, 4) <= LaunchPolicy::ffn_max_cols &&
           FC_ % 48 == 0 && FC_ / 48 <= 16 && w4_of(enc_proj16_);
  for (auto& e : enc_) f = f && e.f1p && e.f2p && w4_of(e.qkv16);
  return f;
}
// Whether enc_p.proj and dp.pre of the current call go as ONE lngemm4_kernel launch over the stacked matrix (small calls of
// the 192-channel voices, where both would take 4-column launches anyway): a function of the call alone.
bool Engine::proj_pre_stacked() const {
  double tsum = 0;
  for (int b = 0; b < B_; ++b) tsum += tlens_h_[b];
  return pol_.stack_pre && projpre4_ && projpre_split_ == enc_proj_.rows && dp_pre_.rows == 192 && H_ == 192 &&
         pol_.chain16(tsum, false, H_, 96) && pol_.chain4((long)B_ * Tg_);
}
float* Engine::stage_a_enc_out() const { return (stage_a_ffn_fused() && (enc_.size() & 1)) ? y_ : x_; }

// Everything up to the frame counts: speaker vectors, text encoder, duration predictor, durations.
// Grids are sized by the bucketed maximum length Tg_; kernels bound themselves by the device-side
// per-utterance lengths, so the same captured graph serves every batch of that bucket.
void Engine::issue_stage_a() {
  stage_tiled_ = false;         // (a call that threw inside a generator stage must not leave it set)
  const int B = B_, Ts = Ts_, T = Tg_;
  const long bsH = (long)H_ * Ts;
  auto V = [&](float* p, int ch) { return View{p, (long)ch * Ts, Ts}; };
  View x = V(x_, H_), y = V(y_, H_);
  const View qkv = V(qkv_, 3 * H_), att = V(att_, H_), ffh = V(ffh_, FC_),
             stats = V(stats_, 2 * C_), xg = V(xg_, H_), dh = V(dh_, H_), dy = V(dy_, H_), dy2 = V(dy2_, H_),
             hproj = V(hproj_, 32);
  const View none{nullptr, 0, 0};
  (void)bsH;
  double tsum = 0;
  for (int b = 0; b < B; ++b) tsum += tlens_h_[b];
  cols_ids_ = tsum;

  // ================= text encoder (models.py:198-209, attentions.py:60-74)
  prof_begin();
  double fl = 0;
  {
    // the first kernel of the run: embedding lookup; also ingests the call's inputs when they stay in pinned host memory
    // (zero-copy ids: lengths / speaker ids / generator state are published to device memory for everything behind it)
    EmbedP ep{};
    ep.ids = d_ids_; ep.ids_bs = Ts; ep.lens = d_tlens_; ep.emb = emb_; ep.H = H_; ep.scale = std::sqrt((float)H_);
    ep.out = x_; ep.o_bs = (long)H_ * Ts; ep.o_cs = Ts; ep.rng = d_rng_;
    if (ids_zc_) {
      const size_t Bc = capA_B_;
      ep.h_rng = reinterpret_cast<const unsigned long long*>(h_in_);
      ep.h_lens = reinterpret_cast<const int*>(h_in_ + 32);
      ep.h_sids = ep.h_lens + Bc;
      ep.h_ids = ep.h_sids + Bc;
      ep.d_lens = d_tlens_; ep.d_sids = d_sids_;
    }
    // small calls: the duration noise is drawn here too (one workgroup, <= 4 Philox blocks per thread), not by a launch of
    // its own in front of the first ConvFlow
    drew_w_ = !have_noise_w_ && (long)B * 2 * ((T + 3) / 4) <= 256;
    if (drew_w_) { ep.draw_out = noise_w_; ep.draw_stride = Ts; ep.draw_rows = B * 2; ep.draw_cols = T; }
    PE_LAUNCH_KB("embed_kernel", 4.0 * tsum * (1.0 + H_), launch::embed(dim3((T + 63) / 64, (H_ + 15) / 16, B), stream_, ep));
  }
  // ================= speaker conditioning vectors
  const float* cb_dp = nullptr;
  if (nspk_ > 1) {
    auto cond = [&](const CondW& c, int off) {
      PE_LAUNCH_K("cond_kernel", launch::cond(dim3((c.rows + 127) / 128, B), stream_, emb_g_, gin_, d_sids_, c.w, c.b, c.rows, cond_ + off, cond_bs_));
    };
    cond(cond_dp_, cond_off_dp_);
    for (size_t i = 0; i < cond_wn_.size(); ++i) cond(cond_wn_[i], cond_off_wn_[i]);
    cond(cond_dec_, cond_off_dec_);
    cb_dp = cond_ + cond_off_dp_;
  }

  // norm_layers_2 of a layer feeds only the next layer's q/k/v conv (or, after the last layer, proj) + the residual of
  // conv_o. Small batches with the 192-channel encoder run norm_layers_2 + that conv as one launch (lngemm_kernel), and
  // conv_o + residual + norm_layers_1 as another (colchain_kernel); otherwise conv, then ln_kernel.
  const float *pg = nullptr, *pb = nullptr;        // pending norm_layers_2 of the previous layer (input still in y)
  const bool chain_q = pol_.chain16(tsum, false, H_, 96);
  // Small calls: the FFN as ONE launch that leaves FC/48 partial outputs for lngemm4_kernel to sum (kernels/ffn.h). That
  // consumer then reads the residual from x and writes LN(y) to the other buffer (its parts read x concurrently): x / y
  // swap roles per layer.
  const bool ffn_fused = stage_a_ffn_fused();
  const int nsl = FC_ / 48;
  const float* pend_bias = nullptr;                // conv_2 bias of the layer whose partial outputs are pending
  // short calls run attention on 4-query workgroups (kernels/attn4.h), which take K transposed: the q/k/v launches then
  // write kT beside qkv (a function of the call alone, like the flags above)
  const auto attn4_for = [&](const EncLayer& e) -> const float* {
    const bool attno_ok = pol_.attno && !pol_.attn_long && chain_q && pol_.chain4((long)B * T) && H_ == 192 && nh_ == 2 && dk_ == 96 &&
                          window_ <= 4 && e.o16;
    // (its LDS need grows with the id bucket: 8 score slabs of SP floats + the fixed part -- a forced policy must fall
    // back to attno_kernel / attn_kernel past 160 KB instead of failing the launch)
    const size_t smem4 = ((size_t)8 * (rup(T, 64) + 4) + 8 * (dk_ + 4) + 2 * 9 * dk_ + 3 * 72 + 8 * 12 + 4 * 196 + 4 * 192 * 4 + 32) * sizeof(float);
    return (attno_ok && pol_.attn4_ids(T) && smem4 <= (size_t)160 * 1024) ? w4_of(e.o16) : nullptr;
  };
  for (auto& e : enc_) {
    kt_on_ = attn4_for(e) != nullptr;
    kt_valid_ = false;
    if (pg && pend_bias) {
      lngemm(x, pg, pb, y, e.qkv16, e.qkv.bias, 3 * H_, qkv, T, 2.0 * tsum * e.qkv.macs_per_col, ffn_parts_, nsl, pend_bias);
      std::swap(x, y);
    } else if (pg) lngemm(y, pg, pb, x, e.qkv16, e.qkv.bias, 3 * H_, qkv, T, 2.0 * tsum * e.qkv.macs_per_col);
    else if (!(chain_q && conv1x1_col4(e.qkv16, e.qkv.bias, 3 * H_, x, qkv, d_tlens_, B, T, 2.0 * tsum * e.qkv.macs_per_col)))
      conv(e.qkv, x, qkv, d_tlens_, 1, T, EPI_STORE);
    pg = pb = nullptr;
    pend_bias = nullptr;
    kt_on_ = false;
    // Small calls of the 192-channel voices: attention + conv_o + residual + norm_layers_1 as ONE launch (kernels/attno.h:
    // 16 queries of both heads per workgroup)
    const int ao_sp = rup(T, 64) + 2;
    const size_t ao_smem = ((size_t)2 * 16 * ao_sp + 2 * 64 * (dk_ + 1) + 2 * dk_ * 16 + (size_t)2 * (2 * window_ + 1) * dk_ + 8 * 256 + 256) * sizeof(float);
    const bool attno = pol_.attno && !pol_.attn_long && chain_q && pol_.chain4((long)B * T) && H_ == 192 && nh_ == 2 && dk_ == 96 && window_ <= 4 && e.o16 &&
                       ao_smem <= (size_t)160 * 1024;
    const float* wo4 = kt_valid_ ? attn4_for(e) : nullptr;        // (the q/k/v launch of this layer wrote kT / vQ)
    if (attno || wo4) {
      AttnOP ap{};
      ap.qkv = qkv_; ap.q_bs = (long)3 * H_ * Ts; ap.q_cs = Ts;
      ap.relk = e.relk; ap.relv = e.relv;
      ap.lens = d_tlens_; ap.window = window_; ap.SP = ao_sp;
      ap.qscale = 1.0f / std::sqrt((float)dk_);
      ap.wo16 = e.o16; ap.bo = e.o.bias; ap.gamma = e.g1; ap.beta = e.b1;
      ap.x = x.p; ap.x_bs = x.bs; ap.x_cs = x.cs;
      double afl = 0;
      for (int b = 0; b < B; ++b) afl += 4.0 * (double)tlens_h_[b] * tlens_h_[b] * H_;
      // utterances up to PIPER_HIP_ATTN4_MAXC ids: 4-query workgroups (kernels/attn4.h: 32 workgroups for 128 ids instead of 8;
      // its LDS need is a fraction of attno_kernel's, whose 16 x T score slabs of both heads end near 830 ids)
      if (wo4) {
        ap.wo4 = wo4; ap.xcd = xcd_period_; ap.SP = rup(T, 64) + 4;
        ap.kT = kT_; ap.vQ = vQ_; ap.kt_bs = (long)Ts * H_;
        const size_t smem4 = ((size_t)8 * ap.SP + 8 * (dk_ + 4) + 2 * 9 * dk_ + 3 * 72 + 8 * 12 + 4 * 196 + 4 * 192 * 4 + 32) * sizeof(float);
        const bool long_rows = T > 128;           // more than one K unit / V chunk per wave: double-buffered fragments
        const int kh = kbegin(prof_level_ >= 2 ? krow(long_rows ? "attn4_kernel<96,true>" : "attn4_kernel<96,false>") : 0,
                              afl + 2.0 * tsum * e.o.macs_per_col, 4.0 * (tsum * 5.0 * H_ + e.o.macs_per_col));
        launch::attn4(long_rows, dim3((T + 3) / 4, B), smem4, stream_, ap);
        kend(kh);
      } else {
      const int kh = kbegin(prof_level_ >= 2 ? krow("attno_kernel<96>") : 0, afl + 2.0 * tsum * e.o.macs_per_col,
                            4.0 * (tsum * 5.0 * H_ + e.o.macs_per_col));
      launch::attno(dim3((T + 15) / 16, B), ao_smem, stream_, ap);
      kend(kh);
      }
    } else {
    AttnP ap;
    ap.qkv = qkv_; ap.q_bs = (long)3 * H_ * Ts; ap.q_cs = Ts;
    ap.relk = e.relk; ap.relv = e.relv;
    ap.out = att_; ap.o_bs = (long)H_ * Ts; ap.o_cs = Ts;
    ap.lens = d_tlens_; ap.H = H_; ap.dk = dk_; ap.window = window_;
    ap.SP = rup(T, 64) + 1;
    ap.qscale = 1.0f / std::sqrt((float)dk_);
    // the 32 x T score slab of a workgroup in LDS, or -- utterances of more than ~830 ids -- in global memory (same kernel body)
    const bool sg = attn_scores_global(T);
    if (sg && !att_s_) throw std::runtime_error("internal: attention score scratch not allocated");
    ap.sglobal = sg ? att_s_ : nullptr;
    const size_t smem = attn_smem(T, sg);
    double afl = 0;
    for (int b = 0; b < B; ++b) afl += 4.0 * (double)tlens_h_[b] * tlens_h_[b] * H_;
    const int kh = kbegin(prof_level_ >= 2 ? krow(std::string(sg ? "attn_long_kernel" : "attn_kernel") + (ap.dk == 96 ? "<96>" : ap.dk == 48 ? "<48>" : "<0>")) : 0, afl, 4.0 * 4.0 * H_ * tsum);
    const dim3 agrid((T + ATT_QB - 1) / ATT_QB, nh_, B);
    launch::attention(ap.dk, agrid, smem, stream_, ap);
    kend(kh);
    const bool chain_o = chain_q;
    if (chain_o) {
      // conv_o + residual + norm_layers_1 in one launch (the 192 x 192 GEMM fits one workgroup per 16 columns)
      ColP cp{};
      cp.in1 = att.p; cp.in1_bs = att.bs; cp.in1_cs = att.cs; cp.K1 = H_;
      cp.w1 = e.o16; cp.b1 = e.o.bias; cp.rows1 = H_;
      cp.mode = 0;
      cp.res = x.p; cp.res_bs = x.bs; cp.res_cs = x.cs;
      cp.gamma = e.g1; cp.beta = e.b1;
      cp.out = x.p; cp.out_bs = x.bs; cp.out_cs = x.cs;
      cp.lens = d_tlens_;
      colchain(cp, B, T, 2.0 * tsum * e.o.macs_per_col);
    } else {
      conv(e.o, att, y, d_tlens_, 1, T, EPI_RESADD, 1.f, ACT_NONE, x);
    }
    if (!chain_o) layer_norm(y, x, e.g1, e.b1, H_, d_tlens_, T);
    }      // !attno
    if (ffn_fused) {
      FfnP fp{};
      fp.xcd = pol_.xcd_ffn ? xcd_period_ : 0;          // (column tile, slice) dealt to the XCDs slice-major
      const int Tp = rup(T, 4);
      fp.x = x.p; fp.x_bs = x.bs; fp.x_cs = x.cs;
      fp.w1p = e.f1p; fp.b1 = e.f1.bias; fp.w2p = e.f2p;
      fp.parts = ffn_parts_; fp.nslices = nsl; fp.p_bs = (long)nsl * H_ * Tp;
      fp.lens = d_tlens_;
      const int khf = kbegin(prof_level_ >= 2 ? krow("ffn_kernel") : 0, 2.0 * tsum * (e.f1.macs_per_col + e.f2.macs_per_col),
                             4.0 * (tsum * (1.0 + nsl) * H_ + e.f1.macs_per_col + e.f2.macs_per_col));
      const size_t smemf = ((size_t)192 * 48 + 4 * 48 * 16 + 48 * 48) * sizeof(float);
      launch::ffn(dim3((T + 11) / 12, nsl, B), smemf, stream_, fp);
      kend(khf);
      pend_bias = e.f2.bias;
    } else {
      conv(e.f1, x, ffh, d_tlens_, 1, T, EPI_STORE, 1.f, ACT_RELU);
      conv(e.f2, ffh, y, d_tlens_, 1, T, EPI_RESADD, 1.f, ACT_NONE, x);
    }
    if (chain_q) { pg = e.g2; pb = e.b2; }
    else layer_norm(y, x, e.g2, e.b2, H_, d_tlens_, T);
    fl += 2.0 * tsum * (e.qkv.macs_per_col + e.o.macs_per_col + e.f1.macs_per_col + e.f2.macs_per_col);
    for (int b = 0; b < B; ++b) fl += 2.0 * 2.0 * (double)tlens_h_[b] * tlens_h_[b] * H_;
  }
  // small calls: the duration predictor's first conv reads the same LN(y) as proj -- one launch over the stacked matrix
  // (three row parts of 192: m_p, logs_p, dp.pre) instead of two launches in a row
  const bool stacked = pg && proj_pre_stacked();
  LnSecond sec;
  if (stacked) {
    sec.w4 = projpre4_; sec.split = projpre_split_; sec.out2 = dy;
    sec.bias2 = cb_dp; sec.bias2_bs = cond_bs_;
  }
  const float* pj_bias = stacked ? projpre_bias_ : enc_proj_.bias;
  const int pj_rows = stacked ? projpre_split_ + dp_pre_.rows : enc_proj_.rows;
  const double pj_flops = 2.0 * tsum * (enc_proj_.macs_per_col + (stacked ? dp_pre_.macs_per_col : 0));
  if (pg && pend_bias) {
    lngemm(x, pg, pb, y, enc_proj16_, pj_bias, pj_rows, stats, T, pj_flops, ffn_parts_, nsl, pend_bias, stacked ? &sec : nullptr);
    std::swap(x, y);
  } else if (pg) lngemm(y, pg, pb, x, enc_proj16_, pj_bias, pj_rows, stats, T, pj_flops, nullptr, 0, nullptr, stacked ? &sec : nullptr);
  else conv(enc_proj_, x, stats, d_tlens_, 1, T, EPI_STORE);
  if (x.p != stage_a_enc_out()) throw std::runtime_error("internal: encoder output buffer bookkeeping");
  fl += 2.0 * tsum * enc_proj_.macs_per_col;
  prof_end(0, fl);

  // ================= stochastic duration predictor, reverse (models.py:63-71,108-117)
  prof_begin();
  fl = 0;
  if (!stacked &&
      !(chain_q && dp_pre16_ && conv1x1_col4(dp_pre16_, dp_pre_.bias, dp_pre_.rows, x, dy, d_tlens_, B, T, 2.0 * tsum * dp_pre_.macs_per_col,
                                             cb_dp, cond_bs_)))
    conv(dp_pre_, x, dy, d_tlens_, 1, T, EPI_STORE, 1.f, ACT_NONE, none, none, 0, 1.f, cb_dp, cond_bs_);
  if (pol_.fuse_dp) {
    DdsOpt o;                      // dp.proj fused after the last DDSConv layer (models.py:65)
    o.post_w16 = dp_proj16_; o.post_bias = dp_proj_.bias; o.post_rows = dp_proj_.rows; o.post_out = xg;
    dds(dp_dds_, dy, dh, dy2, &o);
  } else {
    dds(dp_dds_, dy, dh, dy2);
    conv(dp_proj_, dh, xg, d_tlens_, 1, T, EPI_STORE);
  }
  fl += 2.0 * tsum * (2 + arch_[A_DDSLAYERS]) * dp_pre_.macs_per_col;
  // z = noise * noise_scale_w   [B][2][Ts]
  if (!have_noise_w_ && !drew_w_)
    PE_LAUNCH_KB("randn_kernel", 4.0 * 2.0 * tsum, launch::randn(stream_, noise_w_, (long)B * 2, T, (long)Ts, 0L, d_rng_, 0));
  if (!pol_.fuse_dp) {
    const long n = (long)B * 2 * Ts;
    PE_LAUNCH_K("scale_kernel", launch::scale(dim3((unsigned)((n + 255) / 256)), stream_, noise_w_, z2_, n, scales_[2]));
  }
  // Flip is folded into which physical channel is x0 (conditioning) and which is x1 (transformed):
  // logical = physical when an even number of flips has been applied.
  int flips = 0;
  for (size_t fi = 0; fi < cflows_.size(); ++fi) {
    auto& cf = cflows_[fi];
    ++flips;
    const int c0 = (flips & 1) ? 1 : 0;     // physical channel holding logical x0
    const int c1 = 1 - c0;
    if (pol_.fuse_dp) {
      // One launch per DDSConv layer and nothing else: ConvFlow.pre (+ g) is folded into the first layer's input,
      // proj and the spline run on the last layer's columns. The first flow reads the raw N(0,1) draw and applies
      // noise_scale_w itself; its spline epilogue also moves the pass-through channel into z2_.
      const float* zin = fi == 0 ? noise_w_ : z2_;
      DdsOpt o;
      o.pre_z = zin + (long)c0 * Ts; o.pre_z_bs = (long)2 * Ts; o.pre_w = cf.pre_w; o.pre_b = cf.pre_b;
      o.z_scale = fi == 0 ? scales_[2] : 1.f;
      o.post_w16 = cf.proj16; o.post_bias = cf.proj.bias; o.post_rows = cf.proj.rows;
      o.zin = zin; o.zin_bs = (long)2 * Ts; o.z_cs = Ts; o.c0 = c0; o.c1 = c1; o.zout = z2_; o.zout_bs = (long)2 * Ts;
      dds(cf.dds, xg, dh, dy2, &o);
    } else {
      PE_LAUNCH_K("cf_pre_kernel", launch::cf_pre(dim3((T + 63) / 64, H_, B), stream_, z2_ + (long)c0 * Ts, (long)2 * Ts, cf.pre_w, cf.pre_b, xg_, (long)H_ * Ts, Ts, dy_, (long)H_ * Ts, Ts, d_tlens_, H_));
      dds(cf.dds, dy, dh, dy2);
      conv(cf.proj, dh, hproj, d_tlens_, 1, T, EPI_STORE);
      PE_LAUNCH_K("spline_inverse_kernel", launch::spline_inverse(dim3((T + 63) / 64, B), stream_, hproj_, (long)32 * Ts, Ts, z2_ + (long)c1 * Ts, (long)2 * Ts, d_tlens_, 1.0f / std::sqrt((float)H_)));
    }
    fl += 2.0 * tsum * (arch_[A_DDSLAYERS] * dp_pre_.macs_per_col + cf.proj.macs_per_col);
  }
  ++flips;   // the Flip before ElementwiseAffine
  {
    const int c0 = (flips & 1) ? 1 : 0;     // physical channel holding logical channel 0 = logw
    DurP dp{};
    dp.z0 = z2_ + (long)c0 * Ts; dp.z_bs = (long)2 * Ts; dp.m0 = ea_m0_; dp.es0 = ea_es0_; dp.length_scale = scales_[1];
    dp.lens = d_tlens_; dp.dur = d_dur_; dp.cum = d_cum_; dp.d_bs = Ts; dp.frames = d_frames_; dp.logw_out = logw_;
    dp.frames_host = h_frames_; dp.frames_clamped = d_framesc_; dp.frame_cap = std::max(Fs_, 1);
    // the whole utterance as one graph: regulate_kernel, first launch of stage B, computes the durations itself
    fold_dp_ = dp;
    if (!fold_dur_) PE_LAUNCH_KB("duration_kernel", 4.0 * 4.0 * tsum, launch::duration(dim3(B), stream_, dp));
  }
  prof_end(1, fl);
}

// Length regulator, prior sample, coupling flow, HiFiGAN, int16 conversion -- sized by the bucketed
// maximum frame count Fg_.
void Engine::issue_flow() {
  const int B = B_, Ts = Ts_, Fmax = Fg_, Fs = Fs_;
  const View none{nullptr, 0, 0};
  double fsum = 0;
  for (int b = 0; b < B; ++b) fsum += frames_h_[b];
  cols_frames_ = fsum;
  double fl = 0;

  // ================= length regulator + prior noise + coupling flow (models.py:705-719)
  prof_begin();
  if (have_noise_z_) {
    // rows of the caller's [B][C][z_stride] buffer -> [B][C][Fs]
    if (h_noise_z_stride_ < Fmax_) throw std::runtime_error("noise_z stride shorter than the frame count");
    for (int b = 0; b < B; ++b)
      for (int c = 0; c < C_; ++c)
        PE_HIP(hipMemcpyAsync(noise_z_ + ((size_t)b * C_ + c) * Fs,
                              h_noise_z_ + ((size_t)b * C_ + c) * h_noise_z_stride_,
                              frames_h_[b] * sizeof(float), hipMemcpyHostToDevice, stream_));
  }
  {
    // The prior noise is drawn inside regulate_kernel: one thread = four frames of one channel = one Philox block, the
    // mapping of randn_kernel (round 2 had tried it in a 16-channels-per-thread loop: 21 us against a 5 us launch). In the
    // one-graph form the kernel also computes the durations (no duration_kernel in front of it): three launches -> one.
    RegP rp{};
    rp.stats = stats_; rp.s_bs = (long)2 * C_ * Ts; rp.s_cs = Ts;
    rp.cum = d_cum_; rp.d_bs = Ts; rp.tlens = d_tlens_; rp.frames = lens_b_;
    rp.noise = noise_z_; rp.n_bs = (long)C_ * Fs; rp.n_cs = Fs;
    rp.noise_scale = scales_[0];
    rp.out = zp_; rp.o_bs = (long)C_ * Fs; rp.o_cs = Fs; rp.C = C_;
    rp.absmax = absmax_;
    rp.rng = d_rng_; rp.gen = have_noise_z_ ? 0 : 1;
    rp.fold = fold_dur_ ? 1 : 0;
    if (fold_dur_) rp.dur = fold_dp_;
    PE_LAUNCH_KB("regulate_kernel", 4.0 * (2.0 * C_ * cols_ids_ + 3.0 * C_ * fsum), launch::regulate(dim3((Fmax + 255) / 256, (C_ + 3) / 4, B), stream_, rp));
    if (zp_keep_)     // tests: z_p, the flow's input (the flow transforms zp_ in place)
      PE_HIP(hipMemcpyAsync(zp_keep_, zp_, (size_t)B * C_ * Fs * sizeof(float), hipMemcpyDeviceToDevice, stream_));
  }
  auto VF = [&](float* p, int ch) { return View{p, (long)ch * Fs, Fs}; };
  const View fh = VF(fh_, H_), facts = VF(facts_, H_), fskip = VF(fskip_, H_);
  const int half = C_ / 2;
  const bool chain = pol_.chain16(fsum, true, H_, half);
  for (size_t ri = 0; ri < rcls_.size(); ++ri) {
    Rcl& r = rcls_[ri];
    const View x0{zp_ + (long)r.in_off * Fs, (long)C_ * Fs, Fs};
    const View x1{zp_ + (long)r.out_off * Fs, (long)C_ * Fs, Fs};
    if (!(chain && ri > 0)) {                                                     // else: written by the previous layer's chain
      if (!(chain && r.pre4pad && conv1x1_col4(nullptr, r.pre.bias, r.pre.rows, x0, fh, lens_b_, B, Fmax, 2.0 * fsum * r.pre.macs_per_col,
                                               nullptr, 0, r.pre4pad, half, LaunchPolicy::col4_max_frames)))
        conv(r.pre, x0, fh, lens_b_, 1, Fmax, EPI_STORE);
    }
    const int nl = (int)r.in.size();
    // small calls: the LAST layer's res/skip conv (skip rows only) rides in front of the post + pre chain launch
    const bool rs_front = chain && nl >= 1 && r.rs4[nl - 1] && r.rs[nl - 1].rows == H_ && H_ == 192 && half == 96 &&
                          pol_.chain_rs_front((long)B * Fmax) && w4_of(r.post16) &&
                          (ri + 1 >= rcls_.size() || (rcls_[ri + 1].pre.rows <= 192 && w4_of(rcls_[ri + 1].pre16)));
    for (int i = 0; i < nl; ++i) {
      const float* b2 = nspk_ > 1 ? cond_ + cond_off_wn_[ri] + (long)i * 2 * H_ : nullptr;
      conv(r.in[i], fh, facts, lens_b_, 1, Fmax, EPI_GATE, 1.f, ACT_NONE, none, none, 0, 1.f, b2, cond_bs_);
      if (rs_front && i == nl - 1) {
        fl += 2.0 * fsum * (r.in[i].macs_per_col + r.rs[i].macs_per_col);
        continue;
      }
      if (r.rs4[i] && pol_.chain4_frames((long)B * Fmax) && H_ == 192 && r.rs[i].rows <= 2 * H_) {
        // small calls: the res/skip 1x1 conv on 4-column workgroups (colchain4_kernel mode 2), one part per 192 rows
        ColP cp{};
        cp.in1 = facts.p; cp.in1_bs = facts.bs; cp.in1_cs = facts.cs; cp.K1 = H_;
        cp.w1 = r.rs4[i]; cp.b1 = r.rs[i].bias; cp.rows1 = r.rs[i].rows;
        cp.mode = 2; cp.first = i == 0 ? 1 : 0;
        cp.xcd = xcd_period_;
        cp.x1 = fh.p; cp.x1_bs = fh.bs; cp.x1_cs = fh.cs;
        cp.out = fskip.p; cp.out_bs = fskip.bs; cp.out_cs = fskip.cs;
        cp.lens = lens_b_;
        const int kh4 = kbegin(prof_level_ >= 2 ? krow("colchain4_kernel<false>") : 0, 2.0 * fsum * r.rs[i].macs_per_col,
                               4.0 * (fsum * (H_ + 2.0 * cp.rows1) + (double)cp.rows1 * H_));
        launch::colchain4(dim3((Fmax + 3) / 4, B, (cp.rows1 + 191) / 192), col4_smem(), stream_, cp);
        kend(kh4);
      } else {
        conv(r.rs[i], facts, fh, lens_b_, 1, Fmax, EPI_WNRS, 1.f, ACT_NONE, none, fskip, i == 0 ? 1 : 0);
      }
      fl += 2.0 * fsum * (r.in[i].macs_per_col + r.rs[i].macs_per_col);
    }
    if (chain) {
      // post + "x1 -= m" + the next coupling layer's pre over the updated half, one launch
      ColP cp{};
      cp.in1 = fskip.p; cp.in1_bs = fskip.bs; cp.in1_cs = fskip.cs; cp.K1 = H_;
      cp.w1 = r.post16; cp.b1 = r.post.bias; cp.rows1 = half;
      cp.mode = 1;
      cp.x1 = x1.p; cp.x1_bs = x1.bs; cp.x1_cs = x1.cs;
      if (ri + 1 < rcls_.size()) {
        const Rcl& nx = rcls_[ri + 1];
        if (nx.in_off != r.out_off) throw std::runtime_error("coupling layers do not alternate halves");
        cp.w2 = nx.pre16; cp.b2 = nx.pre.bias; cp.rows2 = H_;
        cp.out2 = fh.p; cp.o2_bs = fh.bs; cp.o2_cs = fh.cs;
      }
      cp.lens = lens_b_;
      if (rs_front) {
        cp.in0 = facts.p; cp.in0_bs = facts.bs; cp.in0_cs = facts.cs;
        cp.w0 = r.rs4[nl - 1]; cp.b0 = r.rs[nl - 1].bias; cp.first = nl == 1 ? 1 : 0;
      }
      colchain(cp, B, Fmax, 2.0 * fsum * (r.post.macs_per_col + (cp.w2 ? rcls_[ri + 1].pre.macs_per_col : 0) +
                                          (rs_front ? r.rs[nl - 1].macs_per_col : 0)));
    } else {
      conv(r.post, fskip, x1, lens_b_, 1, Fmax, EPI_SUBFROM);
    }
    fl += 2.0 * fsum * (r.pre.macs_per_col + r.post.macs_per_col);
  }
  prof_end(2, fl);

}

void Engine::issue_stage_b() {
  issue_flow();
  double fsum = 0;
  for (int b = 0; b < B_; ++b) fsum += frames_h_[b];
  issue_decoder(zp_, lens_b_, Fg_, fsum, false);     // regulate_kernel zeroed the peak accumulators
}

// streaming: window of z -> window buffer -> generator (lens = window length, in device memory)
void Engine::issue_window() {
  PE_LAUNCH_K("window_copy_kernel", launch::window_copy(dim3((s_wg_ + 63) / 64, C_), stream_, zp_, Fs_, d_win_, zwin_, Fs_, C_));
  issue_decoder(zwin_, d_win_ + 1, s_wg_, (double)s_wg_, true);
}

// HiFiGAN generator + conv_post + int16 on z (already masked by its length semantics). `zsrc` is
// [B][C][Fs_]; `lens` the per-utterance frame counts in device memory; Fmax the grid bound.
void Engine::issue_decoder(const float* zsrc, const int* lens, int Fmax, double fsum, bool zero_absmax) {
  stage_tiled_ = false;
  const int B = B_, Fs = Fs_;
  const View none{nullptr, 0, 0};
  const float* cb_dec = nspk_ > 1 ? cond_ + cond_off_dec_ : nullptr;
  double fl = 0;
  bool tail_done = false;      // conv_post + tanh + peak computed inside the last stage's mrf_kernel
  // (zero_absmax marks the streaming window path; the whole-utterance path clears the peaks in regulate_kernel)
  if (zero_absmax) PE_HIP(hipMemsetAsync(absmax_, 0, B * sizeof(unsigned), stream_));
  // ================= HiFiGAN generator (models.py:348-368)
  prof_begin();
  fl = 0;
  {
    View cur{hb_[0], (long)U_ * Fs, Fs};
    conv(dec_pre_, View{const_cast<float*>(zsrc), (long)C_ * Fs, Fs}, cur, lens, 1, Fmax, EPI_STORE, 1.f, ACT_NONE, none, none, 0, 1.f,
         cb_dec, cond_bs_);
    fl += 2.0 * fsum * dec_pre_.macs_per_col;
    int mult = 1;
    int cur_buf = 0;
    const int nk = arch_[A_NRB];
    const float inv_nk = 1.0f / (float)nk;
    for (auto& st : ups_) {
      // pick the five working buffers for this stage: u, ta, tb, tc, xs (all != cur)
      int ids[5], n = 0;
      for (int i = 0; i < 5 && n < 4; ++i)
        if (i != cur_buf) ids[n++] = i;
      const int Lin = mult;            // length multiplier of the input
      mult *= st.rate;
      const long Ls = (long)Fs * mult;
      auto VS = [&](int bi) { return View{hb_[bi], (long)st.ch * Ls, (int)Ls}; };
      const View u = VS(ids[0]), ta = VS(ids[1]), tb = VS(ids[2]), tc = VS(ids[3]);
      const int Lmax = Fmax * mult;
      // One launch per stage (mrf_kernel). Measured (profiles/r03_notes.md): ResBlock2 stages (medium / x-low) win at every
      // batch size (B=1 -3 %, B=16 / 64 +4.5 % end to end over the conv-by-conv schedule); ResBlock1 stages (high) tie at
      // one utterance and lose at batch (its 64-channel stage: 86 vs ~110 TFLOP/s for the conv GEMM kernel on K = 64 * 11
      // convs), so those are fused for one or two utterances and on 32 channels only.
      // (split matrix modes: the two-term modes run the fused stage on the 16-bit pipe, mrf_split_kernel; mode bf16x6 keeps the f32
      // fused kernel for a few utterances and goes conv by conv on the 16-bit pipe from PIPER_HIP_BF3_MINF frames up)
      const bool fuse = pol_.mrf_stage(st.mrf_ok, st.mrf_rb1, st.mrf_cp, fsum, matrix_bf3_,
                                       matrix_bf3_ && pol_.mrf_split && st.mrf_wsplit != nullptr);
      // the last stage also runs the generator tail (conv_post, tanh, peak) on its MRF mean while it is still on chip
      const bool tail = fuse && pol_.mrf_tail && &st == &ups_.back() && st.mrf_cp == 32 && st.ch == post_cin_ && mult == hop_;
      // leaky_relu(0.1) -> ConvTranspose1d
      // (folding the up-conv into the stage kernel's prologue was built and measured: the window GEMM with its halo
      // recompute on the 209 workgroups of a single round costs what the launch costs -- profiles/r04_notes.md)
      conv(st.up, cur, u, lens, Lin, Fmax * Lin, EPI_CONVT, 0.1f);
      fl += 2.0 * fsum * Lin * st.up.macs_per_col;
      // xs accumulates into the buffer that held the stage input (free once the up-conv is done)
      const View xs{hb_[cur_buf], (long)st.ch * Ls, (int)Ls};
      // One resblock chain, accumulated into xs with the MRF mode. `t` = {c1 output, ping, pong}.
      auto chain = [&](int j, const View (&t)[3], View dst, int accmode) {
        auto& cv = st.rb[j];
        const int last_epi = EPI_ACCUM;
        View xin = u;
        if (arch_[A_RESBLOCK] == 1) {
          // ResBlock1 (modules.py:301-314): x = x + c2(lrelu(c1(lrelu(x)))) per dilation
          const int np = (int)cv.size() / 2;
          for (int d = 0; d < np; ++d) {
            conv(cv[2 * d], xin, t[0], lens, mult, Lmax, EPI_STORE, 0.1f);
            if (d < np - 1) {
              const View nxt = (d & 1) ? t[2] : t[1];
              conv(cv[2 * d + 1], t[0], nxt, lens, mult, Lmax, EPI_RESADD, 0.1f, ACT_NONE, xin);
              xin = nxt;
            } else {
              conv(cv[2 * d + 1], t[0], dst, lens, mult, Lmax, last_epi, 0.1f, ACT_NONE, xin, none, accmode, inv_nk);
            }
            fl += 2.0 * fsum * mult * (cv[2 * d].macs_per_col + cv[2 * d + 1].macs_per_col);
          }
        } else {
          // ResBlock2 (modules.py:355-364): x = x + c(lrelu(x)) per dilation
          const int nc = (int)cv.size();
          for (int d = 0; d < nc; ++d) {
            if (d < nc - 1) {
              const View nxt = (d & 1) ? t[2] : t[1];
              conv(cv[d], xin, nxt, lens, mult, Lmax, EPI_RESADD, 0.1f, ACT_NONE, xin);
              xin = nxt;
            } else {
              conv(cv[d], xin, dst, lens, mult, Lmax, last_epi, 0.1f, ACT_NONE, xin, none, accmode, inv_nk);
            }
            fl += 2.0 * fsum * mult * cv[d].macs_per_col;
          }
        }
      };
      const size_t need = (size_t)B * st.ch * Ls;
      const long blocks64 = (long)((Lmax + 63) / 64) * ((st.ch + 63) / 64) * B;
      // grouped sibling launches are a single-utterance latency measure: measured -24 us (medium) / -4 % (high) at
      // B=1, but +1..2 % at B=2 and B=4, where every conv already fills the chip on its own
      stage_tiled_ = !fuse && pol_.stage_all_tiled(B, nk, blocks64);
      bool grp = pol_.group_stage(B, nk, blocks64, need <= side_floats_);
      for (auto& cv : st.rb) {
        if (cv.size() != st.rb[0].size()) grp = false;
        for (auto& c : cv) grp = grp && can_group(c, Lmax);
      }
      // the same schedule through the TILED kernel where one conv of the stage is only a few tiles per CU (the high voice's
      // 128- / 64-channel stages of a single utterance: 834 tiles = 3.26 per CU)
      bool grp_t = false;
      if (!fuse && !grp) {
        const PackedConv& c0 = st.rb[0][0];
        const long tblocks = (long)((Lmax + CFG_BN[c0.cfg == CFG_C ? CFG_C : CFG_S] - 1) / CFG_BN[c0.cfg == CFG_C ? CFG_C : CFG_S]) *
                             ((c0.rows + CFG_BM[c0.cfg == CFG_C ? CFG_C : CFG_S] - 1) / CFG_BM[c0.cfg == CFG_C ? CFG_C : CFG_S]) * B;
        grp_t = pol_.group_stage_tiled(nk, tblocks, need <= side_floats_);
        for (auto& cv : st.rb) {
          if (cv.size() != st.rb[0].size()) grp_t = false;
          for (auto& c : cv) grp_t = grp_t && can_group_tiled(c, Lmax) && (c.cfg == CFG_C) == (c0.cfg == CFG_C);
        }
      }
      if (fuse) {
        mrf(st, u, xs, lens, mult, Lmax, tail);
        for (auto& cv : st.rb)
          for (auto& c : cv) fl += 2.0 * fsum * mult * c.macs_per_col;
        if (tail) {
          tail_done = true;
          fl += 2.0 * fsum * hop_ * post_cin_ * POST_K;
        }
      } else if (grp || grp_t) {
        // step d of every resblock in one grouped launch; each resblock keeps its own buffers, one pass sums them
        auto SV = [&](int k) { return View{side_[k], (long)st.ch * Ls, (int)Ls}; };
        View xin[3] = {u, u, u};
        bool summed = false;
        const int nsteps = (int)st.rb[0].size();
        const bool rb1 = arch_[A_RESBLOCK] == 1;
        for (int d = 0; d < nsteps; ++d) {
          group_begin(grp_t);
          for (int j = 0; j < nk; ++j) {
            auto& cv = st.rb[j];
            const View t0 = j == 0 ? tb : SV(4 * (j - 1)), t1 = j == 0 ? ta : SV(4 * (j - 1) + 1),
                       t2 = j == 0 ? tc : SV(4 * (j - 1) + 2), dst = j == 0 ? SV(8) : SV(4 * (j - 1) + 3);
            if (rb1 && !(d & 1)) {
              conv(cv[d], xin[j], t0, lens, mult, Lmax, EPI_STORE, 0.1f);
            } else {
              const int dd = rb1 ? d / 2 : d, nd = rb1 ? nsteps / 2 : nsteps;
              const View o = dd < nd - 1 ? ((dd & 1) ? t2 : t1) : dst;
              conv(cv[d], rb1 ? t0 : xin[j], o, lens, mult, Lmax, EPI_RESADD, 0.1f, ACT_NONE, xin[j]);
              xin[j] = o;
            }
            fl += 2.0 * fsum * mult * cv[d].macs_per_col;
          }
          // the last step's outputs are only ever summed: one GEMM over the concatenated K writes the mean directly
          if (d == nsteps - 1 && !grp_t && pol_.group_sum() && can_group_sum()) {
            group_end_sum(xs, st.last_bias_sum, inv_nk);
            summed = true;
          } else {
            group_end();
          }
        }
        if (!summed)
          PE_LAUNCH_K("mrf_sum_kernel", launch::mrf_sum(dim3((Lmax + 255) / 256, st.ch, B), stream_, side_[8], side_[3], nk == 3 ? side_[7] : (const float*)nullptr, xs.p, xs.bs, xs.cs, lens, mult, inv_nk));
      } else {
        for (int j = 0; j < nk; ++j) {
          const int accmode = nk == 1 ? 3 : (j == 0 ? 0 : (j == nk - 1 ? 2 : 1));
          const View t[3] = {tb, ta, tc};
          chain(j, t, xs, accmode);
        }
      }
      stage_tiled_ = false;
      cur = xs;      // same buffer index cur_buf, new shape
    }
    prof_end(3, fl);

    // ================= conv_post + tanh + peak, int16 (models.py:364-366; piper.cpp:410-431)
    prof_begin();
    const int K = 7, Lmax = Fmax * hop_;
    if (!tail_done)
      PE_LAUNCH_KB("conv_post_kernel", 4.0 * fsum * hop_ * (post_cin_ + 1.0), launch::conv_post(dim3((Lmax + POST_SPB - 1) / POST_SPB, B), stream_, cur.p, cur.bs, cur.cs, post_w_, post_cin_, 0.01f, lens, hop_, audio_, Ss_, absmax_));
    // (the streaming window path delivers per chunk from the device buffer)
    int16_t* zc = (pol_.pcm_zc && !zero_absmax && h_pcm_zc_cap_ >= (size_t)B * (size_t)Ss_) ? h_pcm_zc_ : nullptr;
    PE_LAUNCH_KB("pcm16_kernel", fsum * hop_ * (4.0 + 2.0 + (zc ? 2.0 : 0.0)), launch::pcm16(dim3((Lmax + 255) / 256, B), stream_, audio_, Ss_, absmax_, lens, hop_, pcm_, Ss_, zc));
    prof_end(4, tail_done ? 0.0 : 2.0 * fsum * hop_ * post_cin_ * K);
  }
}

}  // namespace pe
