// gate4_kernel: the WN gate conv of SHORT calls on 64-row x 12-column workgroups with the 4x4x1 MFMA.
// (gfx950 / CDNA4 device code; reference arithmetic: modules.py:196-199 -- x_in = in_layers[i](x); acts =
// fused_add_tanh_sigmoid_multiply(x_in, g_l) (commons.py:99-106); paths relative to /root/reference/src/python/piper_train/vits/.)
#pragma once
#include "conv_common.h"

namespace pe {

// conv_splitk16_kernel<true, 12, 2> gives one utterance's gate conv (384 x 960 x 417) 27 column tiles x 6 channel groups =
// 162 workgroups -- 162 of 256 CUs, each matrix-pipe bound for 3.8 us (960 16x16x4 MFMAs on SIMDs that three waves share).
// The 16-column tile is the 16x16x4 MFMA's; v_mfma_f32_4x4x1 (64 rows x 4 columns per instruction at the same MAC rate)
// allows 12: 35 x 6 = 210 workgroups with 0.75x the matrix time each.
//   * workgroup = one 32-channel group = 64 GEMM rows (lane l < 32: the tanh row of channel 32 g + l, l >= 32: the sigmoid
//     row of channel 32 g + l - 32) x 12 output columns x K = 192 channels x ntaps (<= 5, dilation 1);
//   * 12 waves: wave w owns input channels [16 w, 16 w + 16) for every tap and all three 4-column groups: 20 16-byte
//     weight loads per lane (all in flight at kernel entry, behind the 4 window loads) and 240 MFMAs, three independent
//     accumulator chains; B operand = the x window, transposed in LDS ([16 columns][192 + 4]: a lane reads four channel
//     steps of its column as one 16-byte LDS read, a tap is the next row);
//   * the twelve partial tiles meet in LDS in wave order (deterministic), then bias + speaker bias + tanh * sigmoid.
// Weights: engine_pack.cpp pack_gate4 -- [group][tap][k quad 48][lane][4].
constexpr int G4_NC = 12, G4_WC = 16, G4_NW = 12, G4_K = 192, G4_XS = G4_K + 4;
__global__ __launch_bounds__(64 * G4_NW) void gate4_kernel(ConvP p) {
  PE_KTRACE(4);
  PE_DYN_SMEM(float, sm);                         // XT[16][196] | P[12 waves][64][12]
  float* XT = sm;
  float* P = XT + G4_WC * G4_XS;
  const int b = blockIdx.z, grp = blockIdx.y;
  PE_STAMP(5, 0);
  const int L = p.lens[b] * p.len_mul;            // first used after every load below is requested
  const int n0 = blockIdx.x * G4_NC;
  const int tid = threadIdx.x, lane = tid & 63, wv = PE_UNIFORM(tid >> 6);
  const int l3 = lane & 3, lb = lane >> 2;
  const int ntaps = p.ntaps;
  // ---- the x window (192 channels x 16 columns from n0 - padl), then this wave's fragments
  const pe_rowsrc xd = pe_make_row(p.x + (long)b * p.x_bs, p.Cin * p.x_cs);
  float xv[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int idx = tid + 64 * G4_NW * u, ch = idx >> 4, c = idx & 15, t = n0 - p.padl + c;
    xv[u] = pe_row_load(xd, (t >= 0 && t < p.x_cs) ? ch * p.x_cs + t : -1);
  }
  PE_SCHED_FENCE();
  f32x4 wf[5][4];
  {
    const int tap_floats = (G4_K / 4) * 256;       // one tap of one group: 48 quads x 64 lanes x 4
    const pe_rowsrc wd = pe_make_row_u(p.wpg4 + (long)grp * ntaps * tap_floats, ntaps * tap_floats);
#pragma unroll
    for (int tp = 0; tp < 5; ++tp)
#pragma unroll
      for (int q = 0; q < 4; ++q) wf[tp][q] = pe_row_load4(wd, tp < ntaps ? ((tp * (G4_K / 4) + 4 * wv + q) * 64 + lane) * 4 : -4);
  }
  PE_SCHED_FENCE();
  if (n0 >= L) return;
  PE_STAMP(5, 1);
  {
    const float slope = p.in_slope;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = tid + 64 * G4_NW * u, ch = idx >> 4, c = idx & 15, t = n0 - p.padl + c;
      float v = (t < L) ? xv[u] : 0.f;               // (t < 0: the load was poisoned)
      if (slope != 1.f) v = pe_lrelu(v, slope);
      XT[c * G4_XS + ch] = v;
    }
  }
  __syncthreads();
  PE_STAMP(5, 2);
  // ---- this wave's partial tile over its 16 channels x ntaps
  {
    f32x4 acc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[g][r] = 0.f;
    const float* xp = XT + l3 * G4_XS + 16 * wv;
#pragma unroll
    for (int tp = 0; tp < 5; ++tp) {
      if (tp < ntaps) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 b4[3];
#pragma unroll
          for (int g = 0; g < 3; ++g) b4[g] = *reinterpret_cast<const f32x4*>(xp + (4 * g + tp) * G4_XS + 4 * q);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int g = 0; g < 3; ++g) acc[g] = pe_mfma_4x4x1(wf[tp][q][j], b4[g][j], acc[g]);
        }
      }
    }
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) P[(wv * 64 + 4 * lb + r) * G4_NC + 4 * g + l3] = acc[g][r];
  }
  PE_STAMP(5, 3);
  __syncthreads();
  PE_STAMP(5, 4);
  // ---- the twelve partial tiles in wave order, biases, tanh * sigmoid (commons.py:99-106)
  if (tid < 32 * G4_NC) {
    const int c = tid / G4_NC, col = tid - c * G4_NC;
    const int ch = grp * 32 + c, t = n0 + col;
    if (ch < p.split && t < L) {
      float ta = 0.f, sa = 0.f;
#pragma unroll
      for (int w = 0; w < G4_NW; ++w) {
        ta += P[(w * 64 + c) * G4_NC + col];
        sa += P[(w * 64 + 32 + c) * G4_NC + col];
      }
      conv_store_gate(p, b, ch, t, ta, sa);
    }
  }
  PE_STAMP(5, 5);
}

}  // namespace pe
