// Embedding lookup and windowed relative-position self-attention of the text encoder.
// (gfx950 / CDNA4 device code; reference arithmetic cited per kernel, paths relative to
// /root/reference/src/python/piper_train/vits/.)
#pragma once
#include "../pe_rt.h"
#include "params.h"
#include "rng.h"

namespace pe {

// ------------------------------------------------------------------------------------------------
// Text-encoder embedding: x[b][h][t] = emb[id][h] * sqrt(H)  (models.py:199-200)
// The first kernel of every pipeline run also advances the RNG call counter (state[1]) that both randn sites of
// the run read afterwards, so a replayed graph draws fresh noise on every run without a host copy.
// Zero-copy inputs (p.h_ids != null): the call's ids / lengths / speaker ids / {seed, counter} are read straight from
// the pinned host block upload() filled (the counterpart of pcm16_kernel writing the PCM straight into pinned host
// memory): no copy is enqueued in front of the graph. state[2] remembers the upload that was ingested last, so a replay
// without a new upload keeps counting on the device.
__global__ void embed_kernel(EmbedP p) {
  PE_KTRACE(10);
  const bool zc = p.h_ids != nullptr;
  const int b = blockIdx.z;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  // (both loads requested before either is used; t < ids_bs: the grid covers the id bucket, which is <= the row stride)
  const int len = (zc ? p.h_lens : p.lens)[b];
  const int id = (zc ? p.h_ids : p.ids)[(long)b * p.ids_bs + (t < p.ids_bs ? t : 0)];
  if (blockIdx.x == 0 && blockIdx.y == 0) {
    if (threadIdx.x == 0 && zc) { p.d_lens[b] = len; p.d_sids[b] = p.h_sids[b]; }
    if (b == 0) {
      // the state of this run, read by every thread of the workgroup BEFORE thread 0 replaces it (no other workgroup reads it)
      const bool fresh = zc && p.h_rng[2] != p.rng[2];
      const unsigned long long seed = fresh ? p.h_rng[0] : p.rng[0];
      const unsigned long long counter = (fresh ? p.h_rng[1] : p.rng[1]) + 1ull;
      const unsigned long long serial = fresh ? p.h_rng[2] : p.rng[2];
      __syncthreads();
      if (threadIdx.x == 0) { p.rng[0] = seed; p.rng[1] = counter; p.rng[2] = serial; }
      if (p.draw_out) {
        // the duration noise, one Philox block (four columns of one row) per thread and round: under the latency of the
        // embedding gather below instead of a launch of its own (4.5 us for 2 x 128 values)
        const int per_row = (p.draw_cols + 3) >> 2, nblk = p.draw_rows * per_row;
        for (int q = threadIdx.x; q < nblk; q += blockDim.x) {
          const int row = q / per_row, c4 = (q - row * per_row) * 4;
          float g[4];
          randn4v(((long)row * RNG_PITCH + c4) >> 2, seed, counter, 0, g);
          for (int k = 0; k < 4 && c4 + k < p.draw_cols; ++k) p.draw_out[(long)row * p.draw_stride + c4 + k] = g[k];
        }
      }
    }
  }
  if (t >= len) return;
  const float* e = p.emb + (long)id * p.H;
  float* o = p.out + (long)b * p.o_bs + t;
  const int h0 = blockIdx.y * 16;
#pragma unroll
  for (int k = 0; k < 16; ++k)
    if (h0 + k < p.H) o[(long)(h0 + k) * p.o_cs] = e[h0 + k] * p.scale;
}

// ------------------------------------------------------------------------------------------------
// Windowed relative-position multi-head self-attention (attentions.py:225-272, 292-348).
// qkv: [B][3H][Ts] rows [0,H)=q, [H,2H)=k, [2H,3H)=v. The reference's pad/reshape "relative to
// absolute" trick is evaluated directly as a band: logits[i][j] += q_i . rel_k[j-i+w] and
// out_i += sum_r p[i][i+r] rel_v[r+w] for |r| <= w. Masked keys (>= len) get weight exactly 0, which is
// what the reference's -1e4 fill yields in fp32.

// One workgroup = 32 queries of one (utterance, head); 4 waves.
//   1. S = (q/sqrt(dk)) k^T on the f32 MFMAs: wave w owns key tiles w, w+4, ...; both operands are read
//      from global memory directly in fragment order (q and k rows are contiguous along time).
//   2. banded relative-key logits, softmax over the valid keys (8 lanes per query row).
//   3. O^T = V P^T on the MFMAs (V chunk transposed through LDS so the A fragment is contiguous; P read
//      from the score slab with an odd stride), wave w owns channel tiles w, w+4, ...; banded
//      relative-value term added before the coalesced store.
// DKT: channels per head known at compile time (96 for the 192-channel voices, 48 for x-low): every unrolled loop has its
// exact trip count. DKT = 0: any even dk <= ATT_MAXDK, loops sized for the maximum and guarded per step (on the common
// shapes those guards were ~190 scalar branches per workgroup, a third of the kernel's time).
// SG: the score slab [32][SP] of the workgroup lives in a global scratch buffer instead of LDS -- utterances whose slab
// does not fit the 160 KB (more than ~830 ids; the reference has no such limit: attentions.py builds the full T x T
// matrix). The same code on another address space: the waves of a workgroup share their CU's L1, so the workgroup
// barriers publish the slab exactly as they do for LDS, and every sum runs in the same order (bit-identical results,
// tests: forced on short utterances against the LDS form).
template <int DKT, bool SG>
__device__ __forceinline__ void attn_body(const AttnP& p) {
  constexpr int MAXDK = DKT ? DKT : ATT_MAXDK;
  PE_KTRACE(0);
  PE_DYN_SMEM(float, sm);
  PE_STAMP(0, 0);
  const int b = blockIdx.z, h = blockIdx.y, i0 = blockIdx.x * ATT_QB;
  const int T = p.lens[b];
  if (i0 >= T) return;
  PE_STAMP(0, 1);
  const int dk = DKT ? DKT : p.dk, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int SP = p.SP, VS = dk + 1 + (dk & 1);       // odd strides -> conflict-free column reads
  float* S;                                           // [32][SP]
  float* Vt;                                          // [KCH][VS]
  if constexpr (SG) {
    S = p.sglobal + ((size_t)((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * ((size_t)ATT_QB * SP);
    Vt = sm;
  } else {
    S = sm;
    Vt = S + ATT_QB * SP;
  }
  float* Qs = Vt + ATT_KCH * VS;                      // [dk][32], scaled by 1/sqrt(dk)
  const int nrel = 2 * p.window + 1;
  float* RK = Qs + dk * ATT_QB;                       // [nrel][dk] relative-key embeddings
  float* RV = RK + nrel * dk;                         // [nrel][dk] relative-value embeddings
  const float* qb = p.qkv + (long)b * p.q_bs + (long)(h * dk) * p.q_cs;
  const float* kb = qb + (long)p.H * p.q_cs;
  const float* vb = kb + (long)p.H * p.q_cs;
  const int nkt = (T + 31) / 32;
  const int nk2 = dk / 2;                             // MFMA k-steps over channels (dk even)

  // ---- 1. scores
  // all global reads go through buffer descriptors with index -1 for masked elements (hardware returns 0), so
  // each staging step issues its loads back to back: one memory latency per step instead of one per element
  const pe_rowsrc qd = pe_make_row(qb, dk * p.q_cs), kd = pe_make_row(kb, dk * p.q_cs), vd = pe_make_row(vb, dk * p.q_cs);
  constexpr int NKF = MAXDK / 2;
  float kf[NKF];
  auto load_k = [&](int kt) {
    // one per-lane base (channel parity, key) + a wave-uniform 2*u*stride in an SGPR: no VALU per load
    const int j = kt * 32 + l31;
    const int base = (kt < nkt && j < T) ? lhi * p.q_cs + j : 0x3fffffff;
#pragma unroll
    for (int u = 0; u < NKF; ++u)          // (k-steps beyond dk exist only in the guarded <0> form: poisoned per step)
      kf[u] = pe_row_load_so(kd, (DKT || 2 * u < dk) ? base : 0x3fffffff, 2 * u * p.q_cs);
  };
  // Every global operand of the kernel that does not depend on earlier phases is requested NOW, together: this wave's
  // first key tile, the first V chunk, then Q and the relative-position tables -- one memory latency instead of three
  // serialised ones (the barriers below wait for all of them anyway).
  // V chunk staging: thread -> key jj = tid&63, channel group tid>>6
  float vv[(MAXDK + 31) / 32][8];
  auto load_v = [&](int j0) {
    const int jj = tid & 63;
    const int base = (j0 + jj < T) ? (tid >> 6) * 8 * p.q_cs + j0 + jj : 0x3fffffff;
#pragma unroll
    for (int g = 0; g < (MAXDK + 31) / 32; ++g)
#pragma unroll
      for (int u = 0; u < 8; ++u)          // (the SGPR offset is outside the hardware's range check: rows >= dk are poisoned here)
        vv[g][u] = pe_row_load_so(vd, ((int)(tid >> 6) * 8 + 32 * g + u < dk) ? base : 0x3fffffff, (32 * g + u) * p.q_cs);
  };
  auto store_v = [&]() {
    const int jj = tid & 63;
#pragma unroll
    for (int g = 0; g < (MAXDK + 31) / 32; ++g)
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int d = (tid >> 6) * 8 + 32 * g + u;
        if (d < dk) Vt[jj * VS + d] = vv[g][u];
      }
  };
  load_k(wv);
  load_v(0);
  {
    constexpr int NQ = MAXDK * ATT_QB / 256;     // 16 elements per thread at dk = 128
    float qv[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
      const int e = tid + 256 * u, d = e >> 5, i = e & 31;
      qv[u] = pe_row_load(qd, (d < dk && i0 + i < T) ? d * p.q_cs + i0 + i : -1);
    }
    // the two small relative-position tables ride along: the band loops below then never touch global memory
    constexpr int NR = 5;                              // (2*4+1) * 128 / 256 rounded up
    const pe_rowsrc rkd = pe_make_row(p.relk, nrel * dk), rvd = pe_make_row(p.relv, nrel * dk);
    float rk[NR], rv[NR];
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      rk[u] = pe_row_load(rkd, tid + 256 * u);
      rv[u] = pe_row_load(rvd, tid + 256 * u);
    }
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
      const int e = tid + 256 * u;
      if (e < dk * ATT_QB) Qs[e] = qv[u] * p.qscale;
    }
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      const int e = tid + 256 * u;
      if (e < nrel * dk) { RK[e] = rk[u]; RV[e] = rv[u]; }
    }
  }
  PE_STAMP(0, 2);
  __syncthreads();
  PE_STAMP(0, 3);
  // relative-key partial products R[q][r] = Q . rel_k^T (a 32 x (2w+1) GEMM over dk, a quarter of the channel steps
  // per wave): computed here, next to the score tiles -- both only need Q and the tables in LDS -- so that one barrier
  // publishes the scores and the partials together; they are added onto the band in step 2a
  float* part = RV + nrel * dk;                        // [4 waves][32 queries][16 offsets]
  {
    f32x16 racc;
#pragma unroll
    for (int r = 0; r < 16; ++r) racc[r] = 0.f;
    for (int s2 = wv; s2 < nk2; s2 += 4) {
      const int d = 2 * s2 + lhi;
      racc = pe_mfma_32x32x2(Qs[d * ATT_QB + l31], l31 < nrel ? RK[l31 * dk + d] : 0.f, racc);
    }
    if (l31 < 16) {
#pragma unroll
      for (int r = 0; r < 16; ++r) part[(wv * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * 16 + l31] = racc[r];
    }
  }
  {
    for (int kt = wv; kt < nkt; kt += 4) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      // all K fragments of a key tile are requested at once (<= 64 loads; the wave's first tile already at kernel
      // start), then each batch of 32 steps reads its Q operands from LDS in one go and issues its MFMAs back to back
      if (kt != wv) load_k(kt);
#pragma unroll
      for (int s0 = 0; s0 < NKF; s0 += 32) {
        if (s0 < nk2) {
          float qf[32];
#pragma unroll
          for (int u = 0; u < 32; ++u) qf[u] = (s0 + u < nk2) ? Qs[(2 * (s0 + u) + lhi) * ATT_QB + l31] : 0.f;
          PE_SCHED_FENCE();
#pragma unroll
          for (int u = 0; u < 32; ++u)
            if (s0 + u < nk2) acc = pe_mfma_32x32x2(qf[u], kf[s0 + u], acc);
          PE_SCHED_FENCE();
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) S[((r & 3) + 8 * (r >> 2) + 4 * lhi) * SP + kt * 32 + l31] = acc[r];
    }
  }
  PE_STAMP(0, 4);
  __syncthreads();
  PE_STAMP(0, 5);
  // ---- 2a. relative-key band: S[i][i+r-w] += (q_i/sqrt(dk)) . rel_k[r]: the four waves' partial tiles (above) meet
  // here and are scattered onto the band.
  {
    PE_STAMP(0, 6);
    for (int e = tid; e < ATT_QB * nrel; e += 256) {
      const int i = e % ATT_QB, r = e / ATT_QB;
      const int j = i0 + i + r - p.window;
      if (i0 + i < T && j >= 0 && j < T)
        S[i * SP + j] += (part[i * 16 + r] + part[(32 + i) * 16 + r]) + (part[(64 + i) * 16 + r] + part[(96 + i) * 16 + r]);
    }
  }
  __syncthreads();
  PE_STAMP(0, 7);
  // ---- 2b. softmax over valid keys: row = tid/8, 8 adjacent lanes per row (values stay in registers for the
  // common T <= 128)
  {
    const int i = tid >> 3, sj = tid & 7;
    float* Sr = S + i * SP;
    const int Tpad = (T + ATT_KCH - 1) / ATT_KCH * ATT_KCH;
    if (T <= 128) {
      float ev[16];
      float mx = -3.0e38f;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int j = sj + 8 * k;
        ev[k] = j < T ? Sr[j] : -3.0e38f;
        mx = fmaxf(mx, ev[k]);
      }
      for (int m = 4; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m));
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int j = sj + 8 * k;
        ev[k] = j < T ? expf(ev[k] - mx) : 0.f;
        sum += ev[k];
      }
      for (int m = 4; m >= 1; m >>= 1) sum += __shfl_xor(sum, m);
      const float inv = 1.f / sum;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int j = sj + 8 * k;
        if (j < Tpad) Sr[j] = ev[k] * inv;
      }
    } else {
      float mx = -3.0e38f;
      for (int j = sj; j < T; j += 8) mx = fmaxf(mx, Sr[j]);
      for (int m = 4; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m));
      float sum = 0.f;
      for (int j = sj; j < T; j += 8) {
        const float e = expf(Sr[j] - mx);
        Sr[j] = e;
        sum += e;
      }
      for (int m = 4; m >= 1; m >>= 1) sum += __shfl_xor(sum, m);
      const float inv = 1.f / sum;
      for (int j = sj; j < Tpad; j += 8) Sr[j] = (j < T) ? Sr[j] * inv : 0.f;
    }
  }
  __syncthreads();
  PE_STAMP(0, 8);
  // ---- 3. O^T[d][q] = sum_key V[d][key] P[q][key]
  const int ndt = (dk + 31) / 32;
  f32x16 oacc;                                        // this wave's channel tile (wv < ndt), one tile per wave pass
  for (int dt0 = 0; dt0 < ndt; dt0 += 4) {
    const int dt = dt0 + wv;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
    for (int j0 = 0; j0 < T; j0 += ATT_KCH) {
      __syncthreads();                                // previous chunk consumed / softmax finished
      PE_STAMP(0, 9 + 3 * (j0 / ATT_KCH));
      store_v();
      // next chunk (of this pass, or the first one of the next channel pass) in flight under the MFMAs
      if (j0 + ATT_KCH < T) load_v(j0 + ATT_KCH);
      else if (dt0 + 4 < ndt) load_v(0);
      __syncthreads();
      PE_STAMP(0, 10 + 3 * (j0 / ATT_KCH));
      if (dt < ndt) {
        const int d = dt * 32 + l31;
        float af[ATT_KCH / 2], pf[ATT_KCH / 2];
#pragma unroll
        for (int s2 = 0; s2 < ATT_KCH / 2; ++s2) {
          const int key = 2 * s2 + lhi;
          af[s2] = d < dk ? Vt[key * VS + d] : 0.f;
          pf[s2] = S[l31 * SP + j0 + key];
        }
        PE_SCHED_FENCE();
#pragma unroll
        for (int s2 = 0; s2 < ATT_KCH / 2; ++s2) oacc = pe_mfma_32x32x2(af[s2], pf[s2], oacc);
        PE_SCHED_FENCE();
      }
      PE_STAMP(0, 11 + 3 * (j0 / ATT_KCH));
    }
    if (dt < ndt) {
      // relative-value band as five more k-steps of the same accumulation: key index -> relative offset rr,
      // A = rel_v[rr][d], B = p[q][q + rr - w] (zero outside the band / the utterance)
      const int q = i0 + l31;
      const int d0 = dt * 32 + l31;
      constexpr int MAXREL = 9;
#pragma unroll
      for (int s2 = 0; s2 < (MAXREL + 1) / 2; ++s2) {
        const int rr = 2 * s2 + lhi;
        const int j = q + rr - p.window;
        const float av = (rr < nrel && d0 < dk) ? RV[rr * dk + d0] : 0.f;
        const float bvv = (rr < nrel && q < T && j >= 0 && j < T) ? S[l31 * SP + j] : 0.f;
        oacc = pe_mfma_32x32x2(av, bvv, oacc);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (d < dk && q < T) p.out[(long)b * p.o_bs + (long)(h * dk + d) * p.o_cs + q] = oacc[r];
      }
    }
  }
  PE_STAMP(0, 20);
}

template <int DKT>
__global__ __launch_bounds__(256) void attn_kernel(AttnP p) { attn_body<DKT, false>(p); }
// long utterances: the score slab in global memory (AttnP::sglobal)
template <int DKT>
__global__ __launch_bounds__(256) void attn_long_kernel(AttnP p) { attn_body<DKT, true>(p); }

}  // namespace pe
