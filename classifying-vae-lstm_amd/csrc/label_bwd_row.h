// label_bwd_row.h -- the cl_vrnn label path's backward of one batch row (shared by label_head.hip, whose kernel runs it
// for a row per workgroup, and lstm_pair.hip, whose backward kernel runs it as its epilogue).
// cl_vrnn/model.py:174-191, 244-252 under K.gradients: dW = dzsum_dec.K_dec_w^T + dzsum_enc.K_enc_w^T, the label backward
// through the logistic-normal sample, dWargs, dhW = (dWargs.K_a^T) * (hW > 0), and the Wargs layer's own gradient as the
// row's outer product.
#pragma once
#include "common.h"

namespace clv {

constexpr int LH_T = 384;     // threads (6 waves) >= 4H = 352
constexpr int LH_MAXC = 32;
constexpr float LEPS_K = 1e-7f, LW2 = 1e-10f;

struct LabelBwdArgs {
  int B, D, C, G4;
  const float* dzsum_enc;   // [B,G4]
  const float* dzsum_dec;
  const float* Kenc_w;      // [C,G4]
  const float* Kdec_w;
  const float* wargs;       // [B,2(C-1)]
  const float* eps;
  const float* onehot;
  const float* W;           // [B,C]
  const float* hW;          // [B,D]
  const float* Ka;          // [D,2(C-1)]
  float prior, class_weight, w_kl_weight, inv_b;
  float* dwargs;            // [B,2(C-1)]
  float* dhW;               // [B,D]
  float* wa_slab;           // optional [B][D+1][2(C-1)]: this row's share of the Wargs layer's kernel / bias gradient
};

// The label path's backward of batch row b.  The first LH_T threads of the workgroup work, all NT threads pass the
// barriers: the routine is the body of vrnn_label_bwd_kernel (NT = LH_T) and, since round 3, the epilogue of the pair
// backward kernel (NT = its 768 threads: row b's sum_t dz has just been written by the same workgroup, and a launch of its
// own was 10.7 us of a 0.4 ms step).
template <int NT>
__device__ __forceinline__ void label_bwd_row(const LabelBwdArgs& a, const int b, const int tid) {
  __shared__ float s_part[LH_T / 64][LH_MAXC], s_dw[LH_MAXC], s_dwa[2 * LH_MAXC];
  __shared__ float s_wv[LH_MAXC], s_oh[LH_MAXC], s_wa[2 * LH_MAXC], s_eps[LH_MAXC];
  const int lane = tid & 63, wave = tid >> 6;
  const bool on = NT == LH_T || tid < LH_T;
  const int C1 = a.C - 1, NA = 2 * C1;
  // the row's small vectors -> LDS in one round trip (the serial label backward below reads them element by element)
  if (tid < a.C) { s_wv[tid] = a.W[(size_t)b * a.C + tid]; s_oh[tid] = a.onehot[(size_t)b * a.C + tid]; }
  if (tid >= 64 && tid - 64 < NA) s_wa[tid - 64] = a.wargs[(size_t)b * NA + tid - 64];
  if (tid >= 128 && tid - 128 < C1) s_eps[tid - 128] = a.eps[(size_t)b * C1 + tid - 128];
  // dW[j] = sum_c dzsum_dec[c] K_dec_w[j,c] + dzsum_enc[c] K_enc_w[j,c]: classes in batches of 8 (8 accumulators and 16
  // kernel values live at a time: as the pair backward kernel's epilogue the routine must not raise that kernel's
  // register count); the loads of a batch are unconditional (clamped), so that 16 are in flight -- a load under
  // `if (j < C)` is waited for where it is issued: one L2 round trip per class
#pragma unroll 1
  for (int j0 = 0; j0 < a.C; j0 += 8) {
    float part[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) part[q] = 0.f;
    for (int c = on ? tid : a.G4; c < a.G4; c += LH_T) {
      const float de = a.dzsum_enc[(size_t)b * a.G4 + c], dd = a.dzsum_dec[(size_t)b * a.G4 + c];
      float ke[8], kd[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int j = min(j0 + q, a.C - 1);
        ke[q] = a.Kenc_w[(size_t)j * a.G4 + c];
        kd[q] = a.Kdec_w[(size_t)j * a.G4 + c];
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float m = j0 + q < a.C ? 1.f : 0.f;
        part[q] = fmaf(dd * m, kd[q], fmaf(de * m, ke[q], part[q]));
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float v = wave_sum(part[q]);
      if (on && lane == 0 && j0 + q < a.C) s_part[wave][j0 + q] = v;
    }
  }
  __syncthreads();
  if (tid < a.C) {
    float v = 0.f;
    for (int w = 0; w < LH_T / 64; ++w) v += s_part[w][tid];
    s_dw[tid] = v;
  }
  __syncthreads();
  if (wave == 0) {
    // label backward of the row, one class per lane (cl_vrnn/model.py:244-252 through the logistic-normal sample):
    // three wave sums instead of a serial walk over the classes by one thread
    const float ep = __expf(a.prior);
    const int j = lane;
    const bool in = j < a.C;
    const float w = in ? s_wv[j] : 0.f;
    const float qs = wave_sum(in ? w + LW2 : 0.f);
    const float n = (w + LW2) / qs;
    const bool inside = (n >= LEPS_K) && (n <= 1.f - LEPS_K);
    const float nc = fminf(fmaxf(n, LEPS_K), 1.f - LEPS_K);
    const float dn = (in && inside) ? -(float)C1 * s_oh[j] / nc : 0.f;
    const float dot = wave_sum(dn * n);
    const float d = in ? s_dw[j] + a.class_weight * a.inv_b * ((dn - dot) / qs) : 0.f;
    const float dsum = wave_sum(d * w);
    if (j < C1) {
      const float ds = w * (d - dsum);
      const float m = s_wa[j], lv = s_wa[C1 + j];
      const float sd = expf(0.5f * lv);
      const float dm = ds + a.w_kl_weight * a.inv_b * (m / ep);
      const float dl = ds * s_eps[j] * 0.5f * sd + a.w_kl_weight * a.inv_b * (-0.5f * (1.f - sd * sd / ep));
      s_dwa[j] = dm; s_dwa[C1 + j] = dl;
      a.dwargs[(size_t)b * NA + j] = dm;
      a.dwargs[(size_t)b * NA + C1 + j] = dl;
    }
  }
  __syncthreads();
  if (tid < a.D) {
    float acc = 0.f;
    const float hv = a.hW[(size_t)b * a.D + tid];
    for (int j0 = 0; j0 < NA; j0 += 32) {           // 32 loads in flight: one round trip for up to 17 classes
      float kv[32];
#pragma unroll
      for (int q = 0; q < 32; ++q) kv[q] = a.Ka[(size_t)tid * NA + min(j0 + q, NA - 1)];
#pragma unroll
      for (int q = 0; q < 32; ++q) acc = fmaf(j0 + q < NA ? s_dwa[j0 + q] : 0.f, kv[q], acc);
    }
    a.dhW[(size_t)b * a.D + tid] = hv > 0.f ? acc : 0.f;
  }
  if (a.wa_slab && on) {
    // [hW_b | 1]^T . dwargs_b: the row's outer product; the rows are summed by the backward pass's pending reductions
    // (a GEMM of its own over K = batch was a 10 us launch for 0.4 MFLOP)
    float* slab = a.wa_slab + (size_t)b * (a.D + 1) * NA;
    for (int e = tid; e < (a.D + 1) * NA; e += LH_T) {
      const int r = e / NA, j = e - r * NA;
      slab[e] = (r < a.D ? a.hW[(size_t)b * a.D + r] : 1.f) * s_dwa[j];
    }
  }
}

}  // namespace clv
