// attn4_kernel: an encoder layer's self-attention + conv_o + residual + norm_layers_1 on FOUR-query workgroups
// (attentions.py:60-72: y = attn(x); x = norm_layers_1(x + conv_o(y)); :225-272, 292-348 for the attention itself).
// (gfx950 / CDNA4 device code; reference arithmetic cited per kernel, paths relative to
// /root/reference/src/python/piper_train/vits/.)
#pragma once
#include "col4.h"

namespace pe {

// attno_kernel (attno.h) gives a 128-id utterance 8 workgroups of 16 queries: 98 us per step on 8 of 256 CUs, a chain of
// ~12 barrier-separated phases whose matrix time (16x16x4 MFMA, two waves per SIMD) is a third of the workgroup's life.
// Here a workgroup owns 4 queries of BOTH heads -- 32 workgroups for 128 ids, 256 threads = one wave per SIMD -- and every
// product runs on v_mfma_f32_4x4x1 (64 rows x 4 columns per instruction, the f32 MAC rate of the 16x16x4 form on a quarter
// of the columns; col4.h):
//   1. S^T[key][q] = K^T (q / sqrt(dk)): a wave takes (head, 64-key block) units; A = K[d][key0 + lane] straight from
//      global memory (one dword per channel step, a 256-byte row piece per wave), B = Q[d][q] from LDS. The relative-key
//      logits q . rel_k[r] (9 offsets) are VALU partial sums over three channel slices, added inside the softmax pass.
//   2. softmax over the valid keys: 32 lanes per (head, query) row.
//   3. O^T[c][q] = sum_key V[c][key] P[q][key] over the 192 channels of both heads as three 64-row blocks (96 is a multiple
//      of 4, so a 4-row MFMA block never straddles the heads): A = V[64 m + lane][key .. key + 3] as 16-byte loads, B =
//      P[head(c)][q][key]; the keys are dealt to the four waves in 32-key chunks and the four partial tiles meet in LDS
//      in wave order -- col_gemm4_run's layout, so col_gemm4_get sums them. The relative-value band (9 offsets) is added
//      on the VALU when the tile is read back.
//   4. conv_o + residual + norm_layers_1 on the workgroup's 4 columns: colchain4_kernel's mode 0 (col_gemm4, pe_col_sum4).
// Every workgroup reads all of K and V of its utterance (2 x 192 x T x 4 B) + conv_o's matrix: 343 KB at 128 ids, the
// same as a 16-query workgroup -- 4x the L2 traffic per launch, which is why the launcher (policy.h: attn4) takes this form
// only for short calls. Masked keys (>= len) get weight exactly 0, like the reference's -1e4 fill in fp32. The sums run in
// another order than attno_kernel's (keys in 32-key chunks across the waves): a different rounding of the same values.
// DB: utterances of more than 128 ids (more than one K unit / V chunk per wave) request the next unit's fragments into a
// second register set before the current unit's MFMAs (T = 384: 28.9 -> 20.9 us per launch); up to 128 ids there is no next
// unit and the second set only costs registers (10.9 -> 12.4 us), so the launcher picks <DK, false> there.
template <int DK, bool DB>
__global__ __launch_bounds__(256) void attn4_kernel(AttnOP p) {
  PE_KTRACE(14);
  constexpr int NH = 2, H = NH * DK, NC = 4, NVT = 3, KS1 = Col4W<H>::KS, QS = DK + 4, NREL = 9;
  static_assert(H == C4_H && DK % 16 == 0, "compiled for the 192-channel voices (two heads of 96)");
  PE_DYN_SMEM(float, sm);
  const int b = blockIdx.y;
  PE_STAMP(0, 0);
  // The utterance length lives in device memory: nothing below touches it until Q, the tables, the first K unit and the
  // first V chunk are requested (against the row stride; what lies beyond the length is masked where it is used), so its
  // latency overlaps theirs instead of preceding them.
  const int T = p.lens[b];
  const int i0 = c4_tile(blockIdx.x, gridDim.x, p.xcd) * NC;
  const int tid = threadIdx.x, lane = tid & 63, wv = PE_UNIFORM(tid >> 6);
  const int l3 = lane & 3, lb = lane >> 2;
  const int SP = p.SP, nrel = 2 * p.window + 1;
  float* Sc = sm;                                  // [NH * 4 rows (head, query)][SP] scores, then probabilities
  float* Qs = Sc + NH * NC * SP;                   // [NH][4][QS], scaled by 1/sqrt(dk)
  float* RK = Qs + NH * NC * QS;                   // [NREL][DK]
  float* RV = RK + NREL * DK;                      // [NREL][DK]
  float* BP = RV + NREL * DK;                      // [3 channel slices][NH * 4 * NREL] relative-key partial logits
  float* BV = BP + 3 * NH * NC * NREL;             // [NH * 4][12] probabilities on the relative-value band, 0 outside the utterance
  float* YT = BV + NH * NC * 12;                   // [4][KS1]: attention output of both heads = conv_o's B operand
  float* P = YT + NC * KS1;                        // [4 waves][192][4]
  float* red = P + 4 * C4_H * NC;                  // [2][4][4]
  const float* qb = p.qkv + (long)b * p.q_bs;
  const pe_rowsrc qd = pe_make_row(qb, H * p.q_cs);
  const int Lb = p.q_cs;                           // row stride: the bound of every request made before the length is known

  // ---- requests that depend on nothing computed here, in the order they are waited for: Q + the relative tables, this
  // wave's first K unit, its first V chunk
  // K comes as kT[channel quad][key][4] (the q/k/v launch writes it beside qkv): lane = key takes four channel steps per
  // 16-byte load and the 64 lanes of a wave read 1 KB in one piece, 24 loads per unit. From the [channel][key] tensor the
  // same fragments are 96 dword loads per lane, and with the V chunk behind them a wave has 131 loads to issue against a
  // memory counter of 63: the first barrier then stands two memory round trips from kernel entry (phase stamps: 4.4 us).
  // (A plain transpose [key][192] was measured too: 16 bytes from each of 64 cache lines per instruction, 6.0 us.)
  f32x4 kfA[DK / 4], kfB[DK / 4];                  // two sets: the next unit's fragments fly under the current unit's MFMAs
  const pe_rowsrc ktd = pe_make_row(p.kT + (long)b * p.kt_bs, (H / 4) * p.q_cs * 4);
  auto load_k = [&](int u, bool live, f32x4 (&kf)[DK / 4]) {       // A[row = key][k = channel]: kT[head * DK / 4 + d4][64 kb + lane][0..3]
    const int h = u & 1, kbk = u >> 1, j = kbk * 64 + lane;
    const int o = (live && j < Lb) ? (h * (DK / 4) * p.q_cs + j) * 4 : -4;      // keys in [len, stride): finite or not, their scores are never read
#pragma unroll
    for (int d4 = 0; d4 < DK / 4; ++d4) kf[d4] = pe_row_load4_so(ktd, o, d4 * p.q_cs * 4);      // (every quad row exists: the SGPR offset stays inside)
  };
  f32x4 vfA[NVT][8], vfB[NVT][8];
  // V comes as vQ[key quad][192][4] (written by the q/k/v launch like kT): lane = channel takes four keys per 16-byte
  // load and a wave reads 1 KB in one piece. (From the [channel][key] tensor every lane's 16 bytes sit in another cache
  // line: 64 lines per instruction, and the 96 KB a workgroup touches that way do not fit its L1.)
  const pe_rowsrc vqd = pe_make_row(p.vQ + (long)b * p.kt_bs, (Lb / 4) * H * 4);
  auto load_v = [&](int kc, bool live, f32x4 (&vf)[NVT][8]) {      // A[row = channel 64 m + lane][k = key]: four keys per 16-byte load
#pragma unroll
    for (int m = 0; m < NVT; ++m)
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const int key = 32 * kc + 4 * g;
        vf[m][g] = pe_row_load4(vqd, (live && key < Lb) ? ((key >> 2) * H + 64 * m + lane) * 4 : -4);      // (keys >= len are masked where the fragment is used)
      }
  };
  {
    float qv[3], rk[4], rv[4];
    const pe_rowsrc rkd = pe_make_row(p.relk, nrel * DK), rvd = pe_make_row(p.relv, nrel * DK);
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int e = tid + 256 * u, c = e >> 2, q = e & 3;          // channel c of both heads' stacked q rows
      qv[u] = pe_row_load(qd, (i0 + q < Lb) ? c * p.q_cs + i0 + q : -1);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      rk[u] = pe_row_load(rkd, tid + 256 * u);
      rv[u] = pe_row_load(rvd, tid + 256 * u);
    }
    PE_SCHED_FENCE();
    load_k(wv, true, kfA);
    PE_SCHED_FENCE();
    load_v(wv, true, vfA);
    PE_SCHED_FENCE();
    if (i0 >= T) return;                           // first use of the length
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int e = tid + 256 * u, c = e >> 2, q = e & 3, h = c >= DK ? 1 : 0;
      Qs[(h * NC + q) * QS + (c - h * DK)] = (i0 + q < T) ? qv[u] * p.qscale : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + 256 * u;
      if (e < NREL * DK) { RK[e] = e < nrel * DK ? rk[u] : 0.f; RV[e] = e < nrel * DK ? rv[u] : 0.f; }
    }
  }
  const int nkb = (T + 63) / 64;                   // 64-key blocks; units u = (head u & 1, block u >> 1), wave w takes u = w, w + 4, ..
  const int nkc = (T + 31) / 32;                   // 32-key chunks of phase 3: wave w takes chunks w, w + 4, ..
  __syncthreads();
  PE_STAMP(0, 1);

  // ---- 1. score units of this wave; relative-key partial logits on the VALU (216 threads: (head, query, offset) x 3 slices)
  {
    auto score = [&](int u, const f32x4 (&kf)[DK / 4]) {
      const int h = u & 1, kbk = u >> 1;
      f32x4 acc;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = 0.f;
      const float* qp = Qs + (h * NC + l3) * QS;
#pragma unroll
      for (int d4 = 0; d4 < DK / 4; ++d4) {
        const f32x4 q4 = *reinterpret_cast<const f32x4*>(qp + 4 * d4);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = pe_mfma_4x4x1(kf[d4][j], q4[j], acc);
      }
      // D[r] of lane l = S[key 64 kb + 4 (l >> 2) + r][query l & 3]: four consecutive keys of one row
      *reinterpret_cast<f32x4*>(Sc + (h * NC + l3) * SP + 64 * kbk + 4 * lb) = acc;
    };
    // units wv, wv + 4, ..: the fetch of unit u + 4 is requested (unconditionally: zero-length reads behind the last unit)
    // before unit u's MFMAs, into the other register set
    const int nu = 2 * nkb;
    if constexpr (DB) {
      for (int u = wv; u < nu; u += 8) {
        PE_SCHED_FENCE();
        load_k(u + 4, u + 4 < nu, kfB);
        PE_SCHED_FENCE();
        score(u, kfA);
        if (u + 4 < nu) {
          PE_SCHED_FENCE();
          load_k(u + 8, u + 8 < nu, kfA);
          PE_SCHED_FENCE();
          score(u + 4, kfB);
        }
      }
    } else {
      for (int u = wv; u < nu; u += 4) {
        if (u != wv) load_k(u, true, kfA);          // (never taken up to 128 ids)
        score(u, kfA);
      }
    }
  }
  if (tid < NH * NC * 12) BV[tid] = 0.f;            // band slots outside the utterance stay 0 (softmax fills the others)
  if (tid < 3 * NH * NC * NREL) {
    const int sl = tid / (NH * NC * NREL), it = tid - sl * (NH * NC * NREL);
    const int hq = it / NREL, r = it - hq * NREL;
    const float* qp = Qs + hq * QS + sl * (DK / 3);
    const float* rp = RK + r * DK + sl * (DK / 3);
    float a = 0.f;
#pragma unroll
    for (int d = 0; d < DK / 3; ++d) a = fmaf(qp[d], rp[d], a);
    BP[sl * (NH * NC * NREL) + it] = a;
  }
  // conv_o's weight fragments and the operands of the LayerNorm tail: in flight under the softmax and phase 3 (the K
  // fragments' registers are free now; the first V chunk was requested before them and is waited for first)
  const int col = tid & 3, rl = tid >> 2, t = i0 + col;
  const bool ok = t < T;
  Col4W<H> gw;
  float ov[NVT], gg[NVT], bb[NVT], b1v[NVT];
  {
    const pe_rowsrc od = pe_make_row(p.x + (long)b * p.x_bs, H * p.x_cs);
    const pe_rowsrc gd = pe_make_row(p.gamma, H), bd = pe_make_row(p.beta, H), b1d = pe_make_row(p.bo, H);
#pragma unroll
    for (int k = 0; k < NVT; ++k) {
      const int c = rl + 64 * k;
      ov[k] = pe_row_load(od, ok ? c * p.x_cs + t : -1);
      gg[k] = pe_row_load(gd, c);
      bb[k] = pe_row_load(bd, c);
      b1v[k] = pe_row_load(b1d, c);
    }
    PE_SCHED_FENCE();
    col_gemm4_fetch<H>(gw, p.wo4, C4_NT, wv, lane);
  }
  PE_STAMP(0, 2);
  __syncthreads();
  PE_STAMP(0, 3);

  // ---- 2. relative-key band + softmax over the valid keys: row = tid >> 5 (head, query), 32 lanes per row
  {
    const int row = tid >> 5, sj = tid & 31, q = row & 3;
    float* Sr = Sc + row * SP;
    const float* bp = BP + row * NREL;
    const int jlo = i0 + q - p.window;             // key of relative offset r = 0
    const int Tpad = (T + 31) / 32 * 32;
    auto band = [&](int j) -> float {
      const int r = j - jlo;
      return (r >= 0 && r < nrel) ? (bp[r] + bp[NH * NC * NREL + r]) + bp[2 * NH * NC * NREL + r] : 0.f;
    };
    if (T <= 128) {
      float ev[4];
      float mx = -3.0e38f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int j = sj + 32 * k;
        ev[k] = j < T ? Sr[j] + band(j) : -3.0e38f;
        mx = fmaxf(mx, ev[k]);
      }
      for (int m = 16; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m));
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int j = sj + 32 * k;
        ev[k] = j < T ? expf(ev[k] - mx) : 0.f;
        sum += ev[k];
      }
      for (int m = 16; m >= 1; m >>= 1) sum += __shfl_xor(sum, m);
      const float inv = 1.f / sum;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int j = sj + 32 * k;
        if (j < Tpad) Sr[j] = ev[k] * inv;
        const int r = j - jlo;
        if (j < T && r >= 0 && r < nrel) BV[row * 12 + r] = ev[k] * inv;
      }
    } else {
      float mx = -3.0e38f;
      for (int j = sj; j < T; j += 32) {
        const float s = Sr[j] + band(j);
        Sr[j] = s;
        mx = fmaxf(mx, s);
      }
      for (int m = 16; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m));
      float sum = 0.f;
      for (int j = sj; j < T; j += 32) {
        const float e = expf(Sr[j] - mx);
        Sr[j] = e;
        sum += e;
      }
      for (int m = 16; m >= 1; m >>= 1) sum += __shfl_xor(sum, m);
      const float inv = 1.f / sum;
      for (int j = sj; j < Tpad; j += 32) {
        const float pv = (j < T) ? Sr[j] * inv : 0.f;
        Sr[j] = pv;
        const int r = j - jlo;
        if (j < T && r >= 0 && r < nrel) BV[row * 12 + r] = pv;
      }
    }
  }
  __syncthreads();
  PE_STAMP(0, 4);

  // ---- 3. O^T partial tiles of this wave's key chunks (every wave writes its tile: zeros without a chunk)
  {
    f32x4 acc[NVT];
#pragma unroll
    for (int m = 0; m < NVT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[m][r] = 0.f;
    // B operand: P[head of the lane's 4-channel block][query l & 3][key]; block lb of 64-row tile m holds channels 64 m + 4 lb ..
    const float* pr[NVT];
#pragma unroll
    for (int m = 0; m < NVT; ++m) pr[m] = Sc + (((64 * m + 4 * lb) >= DK ? NC : 0) + l3) * SP;
    auto chunk = [&](int kc, const f32x4 (&vf)[NVT][8]) {
      const bool whole = 32 * kc + 32 <= T;        // a chunk that straddles the length zeroes the stale columns behind it
#pragma unroll
      for (int m = 0; m < NVT; ++m)
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          const int key = 32 * kc + 4 * g;
          const f32x4 p4 = *reinterpret_cast<const f32x4*>(pr[m] + key);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float a = (whole || key + e < T) ? vf[m][g][e] : 0.f;
            acc[m] = pe_mfma_4x4x1(a, p4[e], acc[m]);
          }
        }
    };
    if constexpr (DB) {
      for (int kc = wv; kc < nkc; kc += 8) {        // (chunk kc + 4's fragments requested before chunk kc's MFMAs, like the K units)
        PE_SCHED_FENCE();
        load_v(kc + 4, kc + 4 < nkc, vfB);
        PE_SCHED_FENCE();
        chunk(kc, vfA);
        if (kc + 4 < nkc) {
          PE_SCHED_FENCE();
          load_v(kc + 8, kc + 8 < nkc, vfA);
          PE_SCHED_FENCE();
          chunk(kc + 4, vfB);
        }
      }
    } else {
      for (int kc = wv; kc < nkc; kc += 4) {
        if (kc != wv) load_v(kc, true, vfA);
        chunk(kc, vfA);
      }
    }
#pragma unroll
    for (int m = 0; m < NVT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) P[((wv * C4_NT + m) * 64 + 4 * lb + r) * 4 + l3] = acc[m][r];
  }
  __syncthreads();
  PE_STAMP(0, 5);
  // ---- the four partial tiles + the relative-value band -> YT[col][channel] (attentions.py:255-262)
#pragma unroll
  for (int k = 0; k < NVT; ++k) {
    const int c = rl + 64 * k, h = c >= DK ? 1 : 0, d = c - h * DK;
    float o = col_gemm4_get(P, c, col);
    const float* bv = BV + (h * NC + col) * 12;      // (rows of RV beyond the window are zero)
#pragma unroll
    for (int r = 0; r < NREL; ++r) o = fmaf(bv[r], RV[r * DK + d], o);
    YT[col * KS1 + c] = ok ? o : 0.f;
  }
  __syncthreads();
  PE_STAMP(0, 6);
  // ---- 4. conv_o + residual + norm_layers_1 (colchain4_kernel mode 0)
  col_gemm4_run<H>(gw, YT, P, wv, lane);
  __syncthreads();
  PE_STAMP(0, 7);
  int red_flip = 0;
  auto col_sum = [&](float x) -> float { return pe_col_sum4(x, red, red_flip, wv, lane, col); };
  float v[NVT];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NVT; ++k) {
    v[k] = ok ? (col_gemm4_get(P, rl + 64 * k, col) + b1v[k]) + ov[k] : 0.f;
    s += v[k];
  }
  const float mean = col_sum(s) / (float)H;
  float qq = 0.f;
#pragma unroll
  for (int k = 0; k < NVT; ++k) { const float dd = v[k] - mean; qq = fmaf(dd, dd, qq); }
  const float rstd = 1.f / sqrtf(col_sum(qq) / (float)H + 1e-5f);
  if (!ok) return;
  float* ob = p.x + (long)b * p.x_bs;
#pragma unroll
  for (int k = 0; k < NVT; ++k) {
    const int c = rl + 64 * k;
    ob[(long)c * p.x_cs + t] = (v[k] - mean) * rstd * gg[k] + bb[k];
  }
  PE_STAMP(0, 8);
}

}  // namespace pe
