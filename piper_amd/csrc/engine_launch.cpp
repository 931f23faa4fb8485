// One launcher per kernel family: parameter structs, grids, shared-memory sizes, profile rows; which form a launch takes is
// asked of the launch policy (policy.h).
#include "engine_internal.h"

namespace pe {

// ------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------

// A conv that may ride in a grouped split-K launch: few enough column tiles that the launch is latency- rather than
// throughput-bound, and a halo the 128-column slab covers.
bool Engine::can_group(const PackedConv& pc, int ncols) const {
  if (stage_tiled_) return false;
  const long blocks = (long)((ncols + CFG_BN[pc.cfg] - 1) / CFG_BN[pc.cfg]) * (pc.mtiles * 32 / CFG_BM[pc.cfg]) * B_;
  return pol_.groupable(pc.gate, pc.up != 0, blocks, (pc.ntaps - 1) * pc.dil, pc.Cin);
}
// The tiled kernel's grouped form: a plain conv (no gate, no ConvTranspose) that takes the tiled route anyway.
bool Engine::can_group_tiled(const PackedConv& pc, int ncols) const {
  return !pc.gate && pc.up == 0 && route(pc, ncols, EPI_STORE) == ROUTE_TILE && !(matrix_bf3_ && pc.wpb) &&
         !pol_.one_tap_direct(pc.gate, false, pc.ntaps, pc.Cin) && (pc.ntaps - 1) * pc.dil <= 128;
}
static int tile_cfg_of(int cfg) { return cfg == CFG_C ? CFG_C : CFG_S; }      // non-gate configurations: 32 x 128 or 64 x 64 tiles
void Engine::group_begin(bool tiled) {
  grouping_ = true;
  group_tiled_ = tiled;
  group_cfg_ = -1;
  group_.clear();
  group_flops_ = group_bytes_ = 0;
}
void Engine::group_end() {
  grouping_ = false;
  if (group_.empty()) return;
  if (group_tiled_) {
    group_tiled_ = false;
    // longest kernel first: workgroups are dispatched in grid order (z slowest)
    std::stable_sort(group_.begin(), group_.end(), [](const ConvP& a, const ConvP& b) { return a.ntaps > b.ntaps; });
    ConvG g{};
    int n = 0, rows = 0, xhalo = 0;
    for (const ConvP& c : group_) {
      g.c[n++] = c;
      rows = std::max(rows, c.rows);
      xhalo = std::max(xhalo, c.xhalo);
    }
    g.n = n;
    g.B = B_;
    const int cfg = group_cfg_, BM = CFG_BM[cfg], BN = CFG_BN[cfg];
    const int HALO = xhalo <= 64 ? 64 : 128;
    const size_t smem = (size_t)2 * KC * ((BN + HALO + 63) / 64 * 64) * sizeof(float);
    const dim3 grid((group_ncols_ + BN - 1) / BN, (rows + BM - 1) / BM, n * B_);
    int kh = -1;
    if (prof_level_ >= 2) {
      char nm[96];
      snprintf(nm, sizeof(nm), "conv_mfma_group_kernel<%s,%d>", cfg == CFG_C ? "1,4,1,1,16" : "2,2,1,1,16", HALO);
      kh = kbegin(krow(std::string(nm)), group_flops_, group_bytes_);
    }
    launch::conv_tile_group(cfg, HALO, grid, smem, ls_, g);
    kend(kh);
    group_.clear();
    return;
  }
  // 4 waves per workgroup: 32 / 64 KB of slabs, so 4 / 2 workgroups share a CU and the <= 3 x ~420 workgroups of a
  // group run in one or two rounds (8 waves: 64 / 128 KB, five rounds, slower than one launch per conv). The convs that
  // need the 128-column slab go in a launch of their own: two resident workgroups per CU carry ~420 of them, not 1260.
  constexpr int NW = 4;
  // workgroups are dispatched in grid order (z slowest): the conv with the most taps goes first so that the longest
  // workgroups do not form the tail of the launch
  std::stable_sort(group_.begin(), group_.end(), [](const ConvP& a, const ConvP& b) { return a.ntaps > b.ntaps; });
  for (int wide = 0; wide < 2; ++wide) {
    ConvG g{};
    int n = 0, mt = 0;
    for (const ConvP& c : group_)
      if ((c.xhalo > 32) == (wide == 1)) {
        g.c[n++] = c;
        mt = std::max(mt, (c.rows + 31) / 32);
      }
    if (!n) continue;
    g.n = n;
    g.B = B_;
    const int XW = wide ? 128 : 64;
    const size_t smem = std::max<size_t>((size_t)NW * KC * XW, (size_t)NW * 16 * 64) * sizeof(float);
    const dim3 grid((group_ncols_ + 31) / 32, mt, n * B_);
    const double share = (double)n / (double)group_.size();
    const int kh = kbegin(prof_level_ >= 2 ? krow(wide ? "conv_splitk_group_kernel<4,2,128>" : "conv_splitk_group_kernel<4,2,64>") : 0,
                          group_flops_ * share, group_bytes_ * share);
    launch::conv_group(wide != 0, grid, smem, ls_, g);
    kend(kh);
  }
  group_.clear();
}

bool Engine::can_group_sum() const {
  if (group_.size() < 2 || group_.size() > 3) return false;
  const ConvP& a = group_[0];
  for (const ConvP& c : group_)
    if (c.rows != a.rows || c.Cin != a.Cin || c.nchunks != a.nchunks || c.x_bs != a.x_bs || c.x_cs != a.x_cs ||
        c.r_bs != a.r_bs || c.r_cs != a.r_cs || c.in_slope != a.in_slope || c.epi != EPI_RESADD || c.xhalo > 96)
      return false;
  return true;
}
void Engine::group_end_sum(View out, const float* bias_sum, float alpha) {
  grouping_ = false;
  if (!can_group_sum()) throw std::runtime_error("internal: convs do not fit a K-concatenated launch");
  ConvP q = group_[0];
  q.nseg = (int)group_.size();
  for (int i = 0; i < q.nseg; ++i) {
    q.seg_x[i] = group_[i].x; q.seg_wp[i] = group_[i].wp;
    q.seg_ntaps[i] = group_[i].ntaps; q.seg_dil[i] = group_[i].dil; q.seg_padl[i] = group_[i].padl;
  }
  q.res2 = group_[1].res;
  q.res3 = q.nseg > 2 ? group_[2].res : nullptr;
  q.bias = bias_sum;
  q.out = out.p; q.o_bs = out.bs; q.o_cs = out.cs;
  q.epi = EPI_ACCUM; q.mode = 3; q.alpha = alpha;      // (acc + bias + residuals) * alpha, nothing read back
  q.tgroups = 1;
  constexpr int NW = 4;
  const size_t smem = std::max<size_t>((size_t)NW * KC * 128, (size_t)NW * 16 * 64) * sizeof(float);
  const dim3 grid((group_ncols_ + 31) / 32, (q.rows + 31) / 32, B_);
  // (a 4-deep weight ring measured slower than 2: hifigan stage 0.345 vs 0.338 ms)
  // (a 16-deep ring -- a wave's whole K range in flight at entry, 256 registers, one workgroup per CU -- measured 59 us
  // against 30: profiles/r04_notes.md)
  const int kh = kbegin(prof_level_ >= 2 ? krow("conv_splitk_sum_kernel<4,2>") : 0, group_flops_, group_bytes_);
  launch::conv_group_sum(grid, smem, ls_, q);
  kend(kh);
  group_.clear();
}

// Which kernel family a conv launch goes to (the policy conv() applies).
int Engine::route(const PackedConv& pc, int ncols, int epi) const {
  const int cfg = pc.cfg;
  const long blocks = (long)((ncols + CFG_BN[cfg] - 1) / CFG_BN[cfg]) * (pc.mtiles * 32 / CFG_BM[cfg]) * B_;
  if (stage_tiled_ || !pol_.splitk(blocks, (pc.ntaps - 1) * pc.dil, pc.Cin)) return ROUTE_TILE;
  return pol_.splitk_16col(pc.wp16 != nullptr, epi == EPI_CONVT, pc.gate, pc.nchunks * pc.ntaps) ? ROUTE_SPLITK16 : ROUTE_SPLITK;
}
void Engine::conv(const PackedConv& pc, View x, View out, const int* lens, int len_mul, int Lmax, int epi,
                  float in_slope, int act, View res, View out2, int mode, float alpha, const float* bias2,
                  int bias2_bs) {
  ConvP p;
  p.x = x.p; p.x_bs = x.bs; p.x_cs = x.cs;
  p.wp = pc.wp; p.wp16 = pc.wp16; p.wpb = pc.wpb; p.wunscale = pc.wunscale; p.wpg4 = nullptr; p.bias = pc.bias;
  p.bias2 = bias2; p.bias2_bs = bias2_bs;
  p.out = out.p; p.o_bs = out.bs; p.o_cs = out.cs;
  p.res = res.p; p.r_bs = res.bs; p.r_cs = res.cs;
  p.out2 = out2.p; p.o2_bs = out2.bs; p.o2_cs = out2.cs;
  p.lens = lens; p.len_mul = len_mul;
  p.Cin = pc.Cin; p.rows = pc.rows; p.nchunks = pc.nchunks;
  p.ntaps = pc.ntaps; p.dil = pc.dil; p.padl = pc.padl;
  p.xhalo = (pc.ntaps - 1) * pc.dil;
  p.in_slope = in_slope;
  p.epi = epi; p.act = act;
  p.split = (epi == EPI_GATE) ? pc.split : (epi == EPI_WNRS ? (pc.rows > H_ ? H_ : 0) : 0);
  p.up = pc.up; p.padT = pc.padT;
  p.up_magic = pc.up ? (unsigned)((0x100000000ULL + pc.up - 1) / pc.up) : 0u;
  p.up_vec = 0;
  p.mode = mode; p.alpha = alpha;
  p.tpb = 1;
  p.tgroups = 1;
  if ((epi == EPI_GATE) != pc.gate) throw std::runtime_error("internal: gate epilogue/packing mismatch");

  const int ncols = (epi == EPI_CONVT) ? Lmax + 1 : Lmax;
  int cfg = pc.cfg;
  double kflops = 0, kbytes = 0;
  if (prof_level_ >= 2) {
    // algorithmic FLOPs of this launch: 2 * MACs per output column * valid columns over the batch
    const std::vector<int32_t>& lh = (lens == d_tlens_) ? tlens_h_ : frames_h_;
    double cols = 0;
    for (int b = 0; b < B_; ++b) cols += (double)lh[b] * len_mul;
    kflops = 2.0 * pc.macs_per_col * cols;
    // algorithmic bytes: every input channel and every output row once per column, residual / read-modify-write
    // operands once more each, the weights once per launch
    const bool rd_res = epi == EPI_RESADD || epi == EPI_ACCUM;
    const bool rd_old = epi == EPI_SUBFROM || epi == EPI_WNRS || (epi == EPI_ACCUM && (mode == 1 || mode == 2));
    const double out_rows = epi == EPI_GATE ? pc.split : pc.rows;
    kbytes = 4.0 * (cols * (pc.Cin + out_rows * (1 + (rd_res ? 1 : 0) + (rd_old ? 1 : 0))) +
                    (double)pc.rows * pc.Cin * pc.ntaps);
  }
  const long blocks = (long)((ncols + CFG_BN[cfg] - 1) / CFG_BN[cfg]) * (pc.mtiles * 32 / CFG_BM[cfg]) * B_;
  if (grouping_ && group_tiled_) {
    const int tc = tile_cfg_of(cfg);
    if (!can_group_tiled(pc, ncols) || epi == EPI_CONVT || epi == EPI_GATE || group_.size() >= 3 ||
        (!group_.empty() && (group_ncols_ != ncols || group_cfg_ != tc)))
      throw std::runtime_error("internal: conv does not fit a grouped tiled launch");
    group_cfg_ = tc;
    p.tpb = 1;
    group_.push_back(p);
    group_ncols_ = ncols;
    group_flops_ += kflops;
    group_bytes_ += kbytes;
    return;
  }
  if (grouping_) {
    if (!can_group(pc, ncols) || epi == EPI_CONVT || epi == EPI_GATE || group_.size() >= 3 ||
        (!group_.empty() && group_ncols_ != ncols))
      throw std::runtime_error("internal: conv does not fit a grouped launch");
    p.tgroups = 1;
    group_.push_back(p);
    group_ncols_ = ncols;
    group_flops_ += kflops;
    group_bytes_ += kbytes;
    return;
  }
  const bool to_splitk = !stage_tiled_ && pol_.splitk(blocks, p.xhalo, pc.Cin);
  if (to_splitk && epi == EPI_GATE &&
      pol_.gate_12col(pc.gate, pc.wpg4 != nullptr && pc.Cin == 192, pc.ntaps, pc.dil, (long)((ncols + 11) / 12) * (pc.split / 32) * B_)) {
    // the WN gate conv of a short call: 64 rows x 12 columns per workgroup on the 4x4x1 MFMA (kernels/gate4.h)
    p.wpg4 = pc.wpg4;
    const dim3 grid((ncols + 11) / 12, pc.split / 32, B_);
    const size_t smem = ((size_t)16 * 196 + (size_t)12 * 64 * 12) * sizeof(float);
    const int kh = kbegin(prof_level_ >= 2 ? krow("gate4_kernel") : 0, kflops, kbytes);
    launch::gate4(grid, smem, ls_, p);
    kend(kh);
    return;
  }
  if (to_splitk) {
    // few columns (one utterance through encoder / duration predictor / flow): split K across the waves
    const int MT = pc.gate ? 2 : 1;
    // waves per workgroup: 4 / 8 take whole chunks; the WN gate conv (6 chunks x 5 taps, two M tiles per wave)
    // goes to 12 waves whose two halves split the taps: 15.8 -> 14.2 us per launch at B=1. Measured and not used:
    // the same 12 waves for conv_pre (6 x 7) and FFN conv_2 (24 chunks) are slower than 8.
    const int units = pc.nchunks * pc.ntaps;
    int NW = pc.nchunks >= 5 ? 8 : 4;
    p.tgroups = 1;
    if (pol_.splitk_12wave(pc.gate, units, pc.nchunks, pc.ntaps)) {
      NW = 12;
      p.tgroups = pc.nchunks <= 6 ? 2 : 1;
    }
    dim3 grid((ncols + 31) / 32, pc.mtiles / MT, B_);
    const size_t smem = std::max<size_t>((size_t)NW * KC * 64, (size_t)NW * MT * 16 * 64) * sizeof(float);
    const bool k16 = pol_.splitk_16col(pc.wp16 != nullptr, epi == EPI_CONVT, pc.gate, units);
    // profile rows carry the instantiation exactly as rocprofv3 prints it (minus spaces)
    int kh = -1;
    if (prof_level_ >= 2) {
      char nm[96];
      if (k16) snprintf(nm, sizeof(nm), "conv_splitk16_kernel<%s>", pc.gate ? (pol_.gate_half_groups(true, pc.nchunks, pc.ntaps, (long)((ncols + 15) / 16) * (pc.mtiles / MT) * B_) ? "true,6,5,2" : "true,12,2,4") : "false,8,4,4");      // (as rocprofv3 prints them: the default GT = 4 included)
      else snprintf(nm, sizeof(nm), "conv_splitk_kernel<%d,%s,%d,%d>", MT, pc.gate ? "true" : "false", NW,
                    pc.gate ? (NW == 12 ? 2 : 3) : 4);
      kh = kbegin(krow(std::string(nm)), kflops, kbytes);
    }
    // MFMA-pipe bound inside the workgroup (>= 24 chunk-tap units) although most CUs idle: 16 output columns
    if (k16) {
      dim3 grid16((ncols + 15) / 16, pc.mtiles / MT, B_);
      // the gate conv of a SHORT utterance (<= 128 workgroups of 12 waves: half the CUs idle): half a channel group per
      // workgroup on 6 waves (one chunk and every tap each, all five weight steps in flight at entry) is twice the
      // workgroups with half the matrix time each -- 64 ids: 9.64 -> 8.37 us per launch; beyond one workgroup per CU it
      // loses (128 ids, 324 workgroups: 9.77 -> 12.6 us; profiles/r04_notes.md, call 32)
      const bool half = pol_.gate_half_groups(pc.gate, pc.nchunks, pc.ntaps, (long)grid16.x * grid16.y * grid16.z);
      if (half) grid16.y *= 2;
      const int nw = pc.gate ? (half ? 6 : 12) : 8;
      p.tgroups = pc.gate ? (half ? 1 : (pc.nchunks <= 6 ? 2 : 1)) : 1;
      launch::conv_splitk16(pc.gate, grid16, (size_t)nw * KC * 64 * sizeof(float), ls_, p, half);
      kend(kh);
      return;
    }
    launch::conv_splitk(pc.gate, NW, grid, smem, ls_, p);
    kend(kh);
    return;
  }
  // polyphase up-conv: a lane's four accumulator rows are consecutive output samples of one channel (stride a multiple of
  // 4) or both phases of two channels (stride 2): stored as 16- / 8-byte pieces straight from the accumulators (the 32x32
  // result layout of the tiled f32 kernel and of the split-operand kernel is the same). Measured against one 4-byte store
  // per phase and against the tile transposed through LDS (profiles/r04_notes.md, calls 7 / 11).
  p.up_vec = (epi == EPI_CONVT && pol_.convt_vec) ? (pc.up % 4 == 0 ? 4 : (pc.up == 2 ? 2 : 0)) : 0;
  if (matrix_bf3_ && pc.wpb && p.xhalo <= 128) {
    // split matrix modes (bf16x3 / f16x3 / bf16x6): 128 x 128 / 64 x 128 / 32 x 256 tiles (two 32x32 MFMA tiles per wave at least: the bf16 pipe
    // is fast enough that operand traffic per MFMA matters more than workgroup count)
    static const int BF3_BM[] = {128, 64, 32}, BF3_BN[] = {128, 128, 256};
    static const char* bnames[] = {"2,2,2,2", "2,2,1,2", "1,4,1,2"};
    const int bc = cfg == CFG_A ? 0 : (cfg == CFG_B ? 1 : 2);
    if (pc.gate && bc == 2) throw std::runtime_error("internal: gate conv packed for 32-row blocks");
    const int BM = BF3_BM[bc], BN = BF3_BN[bc];
    const int HALO = p.xhalo <= 64 ? 64 : 128;
    const int nbuf = pc.nchunks == 1 ? 1 : 2;
    const int nterm = matrix_sm_ == 2 ? 3 : 2;
    const size_t smem = (size_t)nbuf * nterm * 64 * ((BN + HALO + 63) / 64 * 64);       // [terms][4 k groups][XS] x 16 B
    dim3 grid((ncols + BN - 1) / BN, pc.mtiles * 32 / BM, B_);
    int kh = -1;
    if (prof_level_ >= 2) {
      char nm[96];
      snprintf(nm, sizeof(nm), "conv_split_kernel<%d,%s,%s,%d>", matrix_sm_, (pc.gate && bc == 1) ? "1,4,2,1" : bnames[bc],
               pc.gate ? "true" : "false", HALO);
      kh = kbegin(krow(std::string(nm)), kflops, kbytes);
    }
    launch::conv_bf3(matrix_sm_, bc, pc.gate, HALO, grid, smem, ls_, p);
    kend(kh);
    return;
  }
  if (pol_.one_tap_direct(pc.gate, epi == EPI_CONVT, pc.ntaps, pc.Cin)) {
    // one tap: nothing to share between the MFMA's k rows, so the B operand skips LDS (kernels/conv1x1.h)
    const dim3 grid((ncols + 63) / 64, (pc.mtiles + 1) / 2, B_);
    int kh = -1;
    if (prof_level_ >= 2) {
      char nm[96];
      int n = snprintf(nm, sizeof(nm), "conv1x1_kernel<1>");
      if (pol_.prof_sites) snprintf(nm + n, sizeof(nm) - n, "|%dx%dx%d e%d L%d", pc.rows, pc.Cin, pc.ntaps, epi, len_mul);
      kh = kbegin(krow(std::string(nm)), kflops, kbytes);
    }
    launch::conv1x1(grid, ls_, p);
    kend(kh);
    return;
  }
  // 32x32 wave tiles everywhere (64x64 / 32x128 workgroup tiles; the gate form pairs two row tiles per wave): measured in
  // rounds 1-3 against 128x128, 64x128 and 256-column tiles at every batch size -- latency here is hidden across
  // workgroups, occupancy beats register reuse (profiles/r01_ablation.txt, r02_notes.md); the larger instantiations are gone
  if (cfg == CFG_A) cfg = pc.gate ? CFG_G : CFG_S;
  else if (cfg == CFG_B && !pc.gate) cfg = CFG_S;
  const int BM = CFG_BM[cfg], BN = CFG_BN[cfg];
  const int ntile = (ncols + BN - 1) / BN, mblocks = pc.mtiles * 32 / BM;
  // Column tiles walked by one workgroup. Measured on MI355X (profiles/r01_tpb_sweep.txt): with 2-3
  // workgroups resident per CU, one tile per workgroup (latency hidden across workgroups) beats walking
  // several tiles with the in-kernel prefetch pipeline at every batch size, so the default is 1; the
  // multi-tile path stays available through PIPER_HIP_TPB.
  const int tpb = pol_.tiles_per_workgroup();
  p.tpb = tpb;
  dim3 grid((ntile + tpb - 1) / tpb, mblocks, B_);
  if (p.xhalo > 128) throw std::runtime_error("conv halo (kernel-1)*dilation > 128 is not supported");
  // one x slab when the workgroup only ever stages one (single chunk, single tile): more workgroups per CU
  const int nbuf = (tpb == 1 && pc.nchunks == 1) ? 1 : 2;
  const int HALO = p.xhalo <= 64 ? 64 : 128;
  const size_t smem = (size_t)nbuf * KC * ((BN + HALO + 63) / 64 * 64) * sizeof(float);
  static const char* knames[] = {"2,2,2,2,8", "1,4,2,1,16", "1,4,1,1,16", "2,2,1,1,16", "2,2,2,1,16", "1,4,1,2,16", "1,4,2,2,8"};
  int kh = -1;
  if (prof_level_ >= 2) {
    char nm[96];
    int n = snprintf(nm, sizeof(nm), "conv_mfma_kernel<%s,%s,%d>", knames[cfg], pc.gate ? "true" : "false", HALO);
    // tuning aid (PIPER_HIP_PROF_SITES=1): one profile row per conv SHAPE instead of per instantiation
    if (pol_.prof_sites) snprintf(nm + n, sizeof(nm) - n, "|%dx%dx%d d%d e%d L%d", pc.rows, pc.Cin, pc.ntaps, pc.dil, epi, len_mul);
    kh = kbegin(krow(std::string(nm)), kflops, kbytes);
  }
  launch::conv_tile(cfg, pc.gate, HALO, grid, smem, ls_, p);
  kend(kh);
}

// mrf_kernel launch: the window geometry -- output columns per workgroup N = 16 * NCG * OU -- is chosen here. Few
// utterances: the launch is one or two rounds of workgroups over the 256 CUs, so the workgroup count should sit just under
// a multiple of 256 and a workgroup should be short; batches: many rounds, so large N (less halo recompute, fewer
// prologues) wins. Cost model: rounds x (MFMA columns of one workgroup incl. recompute + a fixed prologue / epilogue).
bool Engine::mrf_geo(const UpStage& st, int len_mul, bool tail, MrfGeo& best) const {
  const int CP = st.mrf_cp, NCG = CP == 32 ? 8 : 4, HU = CP == 32 ? 1 : 2;
  const int OUMAX = CP == 32 ? 4 : 3;      // 32 channels: 4 units per wave (N = 512) fill the 160 KB of LDS
  const int hx = st.mrf_hx, hxa = rup(hx, 16);
  auto geo = [&](int ou, MrfGeo& g) {
    g.ou = ou; g.N = 16 * NCG * ou;
    if (hxa + g.N + hx > mrf_ws(CP, ou)) return false;
    g.cu_lo = (hxa - hx) / 16; g.cu_hi = (hxa + g.N + hx + 15) / 16;
    g.nleft = hxa / 16 - g.cu_lo; g.nhalo = g.cu_hi - g.cu_lo - g.N / 16;
    return g.nhalo <= NCG * HU;
  };
  best = MrfGeo{};
  double best_cost = 0;
  MrfGeo forced;
  const int force = (pol_.mrf_ou >= 1 && pol_.mrf_ou <= OUMAX && geo((int)pol_.mrf_ou, forced)) ? (int)pol_.mrf_ou : 0;     // (tests / A-B; ignored when it does not fit)
  for (int ou = 1; ou <= OUMAX; ++ou) {
    MrfGeo g;
    if ((force && ou != force) || !geo(ou, g)) continue;
    double wgs = 0;
    const int stride = tail ? g.N - (POST_K - 1) : g.N;
    for (int b = 0; b < B_; ++b) wgs += (double)(((long)frames_h_[b] * len_mul + stride - 1) / stride);
    double work = 0, taps = 0;
    for (auto& P : st.mrf_ph) {
      const int lo = (hxa - P.e) / 16, hi = (hxa + g.N + P.e + 15) / 16;
      work += (double)(hi - lo) * 16 * P.ntaps;
      taps += P.ntaps;
    }
    double cost = std::ceil(wgs / 256.0) * (work + 16.0 * taps);      // prologue + epilogue ~ 16 columns' worth
    // 4 units per wave ran at the register limit (12 spilled VGPRs until the half-step B buffers of call 48): measured 3-5 % slower per column than 3 units at
    // batch (B=16: 1108 vs 1082 us, B=64: 4.29 vs 4.25 ms), but one round instead of two for a single utterance's last
    // stage (B=1: 84.6 vs 93.4 us) -- profiles/r03_notes.md
    if (ou == 4) cost *= 1.06;
    if (!best.ou || cost < best_cost) { best = g; best_cost = cost; }
  }
  return best.ou != 0;
}

void Engine::mrf(const UpStage& st, View x, View out, const int* lens, int len_mul, int Lmax, bool tail) {
  const int CP = st.mrf_cp, HU = CP == 32 ? 1 : 2;
  const int hx = st.mrf_hx, hxa = rup(hx, 16);
  MrfGeo best;
  if (!mrf_geo(st, len_mul, tail, best)) throw std::runtime_error("internal: no mrf_kernel geometry for this stage");
  MrfP p{};
  p.x = x.p; p.x_bs = x.bs; p.x_cs = x.cs;
  p.out = out.p; p.o_bs = out.bs; p.o_cs = out.cs;
  p.lens = lens; p.len_mul = len_mul;
  p.phases = static_cast<const MrfPhase*>(st.mrf_phases); p.nphases = (int)st.mrf_ph.size();
  p.wstream = st.mrf_w; p.wfloats = st.mrf_wfloats;
  p.C = st.ch; p.N = best.N; p.wcols = hxa + best.N + hx; p.hxa = hxa; p.cu_lo = best.cu_lo; p.cu_hi = best.cu_hi;
  p.nleft = best.nleft; p.nhalo = best.nhalo;
  p.slope = 0.1f;                            // modules.py LRELU_SLOPE
  p.alpha = 1.0f / (float)st.rb.size();
  p.stride = best.N; p.n0off = 0;
  p.post_w = nullptr; p.audio = nullptr; p.a_bs = 0; p.absmax = nullptr; p.post_slope = 0.01f;
  if (tail) {       // generator tail inside the stage kernel: windows overlap by the conv_post taps
    p.stride = best.N - (POST_K - 1); p.n0off = (POST_K - 1) / 2;
    p.post_w = post_w_; p.audio = audio_; p.a_bs = Ss_; p.absmax = absmax_;
  }
  double kflops = 0, kbytes = 0;
  if (prof_level_ >= 2) {
    double cols = 0;
    for (int b = 0; b < B_; ++b) cols += (double)frames_h_[b] * len_mul;
    double macs = 0;
    for (auto& cv : st.rb)
      for (auto& c : cv) macs += c.macs_per_col;
    kflops = 2.0 * macs * cols;
    kbytes = 8.0 * st.ch * cols + 4.0 * st.mrf_wfloats;      // one read of x, one write of the mean, the weights once
    if (tail) {
      kflops += 2.0 * cols * st.ch * POST_K;
      kbytes = 4.0 * (st.ch + 1) * cols + 4.0 * st.mrf_wfloats;   // one read of x, one write of the waveform
    }
  }
  dim3 grid((Lmax + p.stride - 1) / p.stride, B_);
  char nm[64];
  const bool split = matrix_bf3_ && pol_.mrf_split && st.mrf_wsplit != nullptr && matrix_sm_ < 2;
  if (split) {          // the same stage on the 16-bit pipe (kernels/mrf_split.h): same window geometry, the split weight stream
    p.wstream = st.mrf_wsplit; p.wfloats = st.mrf_wsplit_floats; p.wunscale = st.mrf_unscale;
    snprintf(nm, sizeof(nm), "mrf_split_kernel<%d,%d,%d,%d>", matrix_sm_, CP, best.ou, HU);
  } else {
    snprintf(nm, sizeof(nm), "mrf_kernel<%d,%d,%d>", CP, best.ou, HU);
  }
  const int kh = prof_level_ >= 2 ? kbegin(krow(std::string(nm)), kflops, kbytes) : -1;
  if (split) launch::mrf_split(matrix_sm_, CP, best.ou, grid, ls_, p);
  else launch::mrf(CP, best.ou, grid, ls_, p);
  kend(kh);
}

// LDS of attn_kernel / attn_long_kernel for utterances padded to T ids (kernels/attention.h: S | Vt | Qs | RK, RV | part)
size_t Engine::attn_smem(int T, bool global_scores) const {
  const int SP = rup(T, 64) + 1, VS = dk_ + 1 + (dk_ & 1);
  return ((global_scores ? 0 : (size_t)ATT_QB * SP) + (size_t)ATT_KCH * VS + (size_t)dk_ * ATT_QB +
          (size_t)2 * (2 * window_ + 1) * dk_ + 4 * ATT_QB * 16) * sizeof(float);
}
bool Engine::attn_scores_global(int T) const { return pol_.attn_long || attn_smem(T, false) > (size_t)160 * 1024; }

void Engine::layer_norm(View in, View out, const float* g, const float* b, int C, const int* lens, int Lmax) {
  LnP p;
  p.in = in.p; p.i_bs = in.bs; p.i_cs = in.cs;
  p.out = out.p; p.o_bs = out.bs; p.o_cs = out.cs;
  p.gamma = g; p.beta = b;
  p.lens = lens; p.C = C;
  if (C > LN_COLS * 32) throw std::runtime_error("LayerNorm over more than 256 channels is not supported");
  dim3 grid((Lmax + LN_COLS - 1) / LN_COLS, B_);
  const int kh = kbegin(prof_level_ >= 2 ? krow("ln_kernel") : 0, 0.0, 4.0 * 2.0 * C * (lens == d_tlens_ ? cols_ids_ : cols_frames_));
  launch::layer_norm(grid, stream_, p);
  kend(kh);
}

// DDSConv.forward (modules.py:117-129): one fused launch per layer (dds_layer16_kernel), ping-ponging between
// `out` and `tmp` so that the last layer lands in `out`; `in` must not alias the first layer's target.
void Engine::dds_params(const DdsW& d, View in, View out, View tmp, const DdsOpt* opt, std::vector<DdsP>& list) {
  int dil = 1;
  const int n = (int)d.c1x1.size();
  View cur = in;
  for (int i = 0; i < n; ++i) {
    const View dst = ((n - 1 - i) & 1) ? tmp : out;
    if (dst.p == cur.p) throw std::runtime_error("internal: DDSConv buffer aliasing");
    DdsP p{};
    if (opt && i == 0 && opt->pre_z) {
      p.pre_z = opt->pre_z; p.pre_z_bs = opt->pre_z_bs; p.pre_w = opt->pre_w; p.pre_b = opt->pre_b;
    }
    p.z_scale = opt ? opt->z_scale : 1.f;
    if (opt && i == n - 1 && opt->post_w16) {
      p.post_w16 = opt->post_w16; p.post_w4 = w4_of(opt->post_w16); p.post_bias = opt->post_bias; p.post_rows = opt->post_rows;
      p.post_out = opt->post_out.p; p.po_bs = opt->post_out.bs; p.po_cs = opt->post_out.cs;
      p.zin = opt->zin; p.zin_bs = opt->zin_bs; p.z_cs = opt->z_cs; p.c0 = opt->c0; p.c1 = opt->c1;
      p.zout = opt->zout; p.zout_bs = opt->zout_bs;
      p.inv_sqrt_h = 1.0f / std::sqrt((float)H_);
    }
    p.x = cur.p; p.x_bs = cur.bs; p.x_cs = cur.cs;
    p.out = dst.p; p.o_bs = dst.bs; p.o_cs = dst.cs;
    p.dw_w = d.dw_w[i]; p.dw_b = d.dw_b[i]; p.dw_k = ksz_; p.dw_dil = dil;
    p.g1 = d.g1[i]; p.b1 = d.b1[i]; p.g2 = d.g2[i]; p.b2 = d.b2[i];
    p.bias = d.c1x1[i].bias;
    p.wp16 = d.w16[i];
    p.wp4 = w4_of(d.w16[i]);
    p.nchunks = d.c1x1[i].nchunks;
    p.lens = d_tlens_; p.H = H_;
    list.push_back(p);
    dil *= ksz_;
    cur = dst;
  }
}

// algorithmic bytes of one DDSConv layer launch: x in, out (+ the fused 1x1 conv's rows), the 1x1 matrix (+ the fused one)
double Engine::dds_bytes(const DdsP& p) const {
  return 4.0 * (cols_ids_ * (2.0 * H_ + p.post_rows) + (double)H_ * H_ + (double)p.post_rows * H_);
}

void Engine::dds(const DdsW& d, View in, View out, View tmp, const DdsOpt* opt) {
  std::vector<DdsP> list;
  dds_params(d, in, out, tmp, opt, list);
  // Small calls of the 192-channel voices: 4-column workgroups on 4x the CUs (kernels/dds4.h). Every layer of the chain
  // needs its matrices in the 4x4x1 order; the form reads 4x the weight bytes, hence the column limit.
  bool four = H_ == 192 && ksz_ <= 3 && pol_.chain4((long)B_ * Tg_);
  for (const DdsP& p : list) four = four && p.wp4 && (!p.post_w16 || p.post_w4);
  for (const DdsP& p : list) {
    if (four) {
      const int kh4 = kbegin(prof_level_ >= 2 ? krow("dds_layer4_kernel") : 0, 0.0, dds_bytes(p));
      DdsP p4 = p;
      p4.xcd = xcd_period_;
      launch::dds_layer4(dim3((Tg_ + 3) / 4, B_), col4_smem(), stream_, p4);
      kend(kh4);
      continue;
    }
    const int kh = kbegin(prof_level_ >= 2 ? krow(p.nchunks == 3 ? "dds_layer16_kernel<3>" : p.nchunks == 6 ? "dds_layer16_kernel<6>"
                                                                                       : "dds_layer16_kernel<8>") : 0, 0.0, dds_bytes(p));
    const dim3 grid16((Tg_ + 15) / 16, B_);
    const size_t smem16 = ((size_t)2 * p.nchunks * 32 * 16 + 16 * 16 + 16 * 3 * 16) * sizeof(float);     // Y, Z, red, the spline tail's S
    // <3> / <6> are compiled for exactly 96 / 192 padded channels; <8> takes any width up to 256
    launch::dds_layer(p.nchunks, grid16, smem16, stream_, p);
    kend(kh);
  }
}

void Engine::set_profile(int level) {
  prof_level_ = level;
  prof_on_ = level > 0;
}
int Engine::krow(const char* name) {
  for (size_t i = 5; i < prof_.size(); ++i)
    if (!strcmp(prof_[i].name, name)) return (int)i;
  prof_.push_back(ProfileRow{name});
  return (int)prof_.size() - 1;
}
int Engine::krow(const std::string& name) {
  for (size_t i = 5; i < prof_.size(); ++i)
    if (name == prof_[i].name) return (int)i;
  names_.push_back(name);
  prof_.push_back(ProfileRow{names_.back().c_str()});
  return (int)prof_.size() - 1;
}
void Engine::lngemm(View y, const float* g, const float* b, View x, const float* w16, const float* bias, int rows,
                    View out, int T, double flops, const float* parts, int nparts, const float* pbias, const LnSecond* second) {
  // algorithmic bytes: y in, LN(y) out, the conv's rows out, the FFN's partial outputs in; weights once
  const double kbytes = 4.0 * (cols_ids_ * (2.0 * H_ + rows + (parts ? (double)nparts * H_ : 0.0)) + (double)rows * H_);
  LnGemmP p{};
  p.in = y.p; p.in_bs = y.bs; p.in_cs = y.cs;
  p.gamma = g; p.beta = b;
  p.xout = x.p; p.x_bs = x.bs; p.x_cs = x.cs;
  p.w16 = w16; p.bias = bias; p.rows = rows;
  p.out = out.p; p.o_bs = out.bs; p.o_cs = out.cs;
  p.lens = d_tlens_;
  // small calls: 4-column workgroups on the 4x4x1 MFMA (kernels/col4.h), like Engine::dds
  if (parts) {        // y = (View y: the residual) + pbias + the fused FFN's partial outputs (ffn_kernel)
    const int Tp = rup(T, 4);
    p.parts = parts; p.nparts = nparts; p.pbias = pbias;
    p.p_bs = (long)nparts * H_ * Tp;
    if (!(pol_.chain4((long)B_ * T) && w4_of(w16))) throw std::runtime_error("internal: FFN partials without the 4-column consumer");
  }
  if (second && !(pol_.chain4((long)B_ * T) && second->w4)) throw std::runtime_error("internal: stacked convs without the 4-column form");
  if (const float* w4 = pol_.chain4((long)B_ * T) ? (second ? second->w4 : w4_of(w16)) : nullptr) {
    p.w16 = w4;
    p.xcd = xcd_period_;
    if (kt_on_ && out.p == qkv_ && rows == 3 * H_) { p.kT = kT_; p.vQ = vQ_; p.kt_bs = (long)Ts_ * H_; kt_valid_ = true; }      // attn4_kernel follows
    if (second) {
      p.split = second->split; p.out2 = second->out2.p; p.o2_bs = second->out2.bs; p.o2_cs = second->out2.cs;
      p.bias2 = second->bias2; p.bias2_bs = second->bias2_bs;
    }
    const int kh4 = kbegin(prof_level_ >= 2 ? krow("lngemm4_kernel") : 0, flops, kbytes);
    launch::lngemm4(dim3((T + 3) / 4, B_, (rows + 191) / 192), col4_smem(), stream_, p);
    kend(kh4);
    return;
  }
  const int kh = kbegin(prof_level_ >= 2 ? krow("lngemm_kernel<6>") : 0, flops, kbytes);
  const size_t smem = ((size_t)192 * 16 + 16 * 16) * sizeof(float);
  launch::lngemm(dim3((T + 15) / 16, B_, (rows + 191) / 192), smem, stream_, p);
  kend(kh);
}

// A plain 1x1 conv over 192 input channels of a small call on 4-column workgroups (colchain4_kernel mode 3); false: the
// caller launches the conv kernel instead. `w16`: the conv's pack16 matrix (its pack4 twin is looked up).
bool Engine::conv1x1_col4(const float* w16, const float* bias, int rows, View in, View out, const int* lens, int B, int Lmax,
                          double flops, const float* bias2, long bias2_bs, const float* w4direct, int kin, long max_cols) {
  const bool small = pol_.chain4((long)B * Lmax, max_cols);
  const float* w4 = (H_ == 192 && small) ? (w4direct ? w4direct : w4_of(w16)) : nullptr;
  if (!w4) return false;
  ColP cp{};
  cp.in1 = in.p; cp.in1_bs = in.bs; cp.in1_cs = in.cs; cp.K1 = kin;
  cp.w1 = w4; cp.b1 = bias; cp.rows1 = rows;
  cp.mode = 3;
  cp.xcd = xcd_period_;
  cp.res = bias2; cp.res_bs = bias2_bs;
  cp.out = out.p; cp.out_bs = out.bs; cp.out_cs = out.cs;
  cp.lens = lens;
  if (kt_on_ && out.p == qkv_ && rows == 3 * H_) { cp.kT = kT_; cp.vQ = vQ_; cp.kt_bs = (long)Ts_ * H_; kt_valid_ = true; }      // attn4_kernel follows
  const double cols = lens == d_tlens_ ? cols_ids_ : cols_frames_;
  const int kh4 = kbegin(prof_level_ >= 2 ? krow("colchain4_kernel<false>") : 0, flops, 4.0 * (cols * (kin + rows) + (double)rows * kin));
  launch::colchain4(dim3((Lmax + 3) / 4, B, (rows + 191) / 192), col4_smem(), stream_, cp);
  kend(kh4);
  return true;
}

void Engine::colchain(const ColP& p, int B, int Lmax, double flops) {
  // algorithmic bytes: GEMM input, the residual / x1 read and written, the second GEMM's output; weights once
  const double cols = p.lens == d_tlens_ ? cols_ids_ : cols_frames_;
  const double kbytes = 4.0 * (cols * (p.K1 + 2.0 * p.rows1 + (p.w2 ? p.rows2 : 0)) + (double)p.rows1 * p.K1 +
                               (p.w2 ? (double)p.rows2 * p.rows1 : 0.0));
  // mode 1 runs on frames (coupling post + pre), mode 0 on ids: separate column limits (profiles/r03_notes.md)
  if ((p.mode == 1 ? pol_.chain4_frames((long)B * Lmax) : pol_.chain4((long)B * Lmax)) && p.K1 == 192 && (p.mode == 0 ? p.rows1 == 192 : (p.rows1 == 96 && (!p.w2 || p.rows2 <= 192)))) {
    const float* w1 = w4_of(p.w1);
    const float* w2 = p.w2 ? w4_of(p.w2) : nullptr;
    if (w1 && (!p.w2 || w2)) {
      ColP q = p;
      q.w1 = w1; q.w2 = w2;
      q.xcd = xcd_period_;
      const double kb0 = q.w0 ? 4.0 * (cols * 192.0 + 192.0 * 192.0) : 0.0;      // the front conv: its input, its weights
      const int kh4 = kbegin(prof_level_ >= 2 ? krow(q.w0 ? "colchain4_kernel<true>" : "colchain4_kernel<false>") : 0, flops, kbytes + kb0);
      launch::colchain4(dim3((Lmax + 3) / 4, B), col4_smem(), stream_, q);
      kend(kh4);
      return;
    }
  }
  if (p.w0) throw std::runtime_error("internal: a res/skip conv in front of a chain that does not take the 4-column form");
  const int kh = kbegin(prof_level_ >= 2 ? krow("colchain_kernel<6>") : 0, flops, kbytes);
  const size_t smem = ((size_t)2 * 6 * 32 * 16 + 16 * 16) * sizeof(float);
  launch::colchain(dim3((Lmax + 15) / 16, B), smem, stream_, p);
  kend(kh);
}

int Engine::kbegin(int row, double flops, double bytes) {
  if (prof_level_ < 2) return -1;
  hipEvent_t a, b;
  if (ev_pool_.size() >= 2) {
    a = ev_pool_.back(); ev_pool_.pop_back();
    b = ev_pool_.back(); ev_pool_.pop_back();
  } else {
    PE_HIP(hipEventCreate(&a));
    PE_HIP(hipEventCreate(&b));
  }
  PE_HIP(hipEventRecord(a, ls_));
  kev_.push_back(KEvent{row, flops, bytes, a, b});
  return (int)kev_.size() - 1;
}
void Engine::kend(int h) {
  if (h >= 0) PE_HIP(hipEventRecord(kev_[h].b, ls_));
}
const std::vector<ProfileRow>& Engine::profile() {
  if (!kev_.empty()) {
    PE_HIP(hipStreamSynchronize(stream_));
    for (auto& k : kev_) {
      float ms = 0;
      PE_HIP(hipEventElapsedTime(&ms, k.a, k.b));
      prof_[k.row].ms += ms;
      prof_[k.row].flops += k.flops;
      prof_[k.row].bytes += k.bytes;
      prof_[k.row].launches += 1;
      ev_pool_.push_back(k.a);
      ev_pool_.push_back(k.b);
    }
    kev_.clear();
  }
  return prof_;
}
void Engine::reset_profile() {
  profile();
  for (auto& r : prof_) { r.ms = 0; r.flops = 0; r.launches = 0; r.bytes = 0; }
}
void Engine::prof_begin() {
  if (prof_on_) PE_HIP(hipEventRecord(ev0_, stream_));
}
void Engine::prof_end(int row, double flops) {
  if (!prof_on_) return;
  PE_HIP(hipEventRecord(ev1_, stream_));
  PE_HIP(hipEventSynchronize(ev1_));
  float ms = 0;
  PE_HIP(hipEventElapsedTime(&ms, ev0_, ev1_));
  prof_[row].ms += ms;
  prof_[row].flops += flops;
  prof_[row].launches += 1;
}

}  // namespace pe
