// lstm_any.hip -- the LSTM sequence kernels for ANY number of hidden units (gfx950).
//
// cl_vrnn/train.py:90 takes --intermediate_dim and cl_vrnn/model.py:196-199, 225-228 build LSTM(intermediate_dim) for
// whatever it is; lstm.hip / lstm_pair.hip / lstm_mx.hip are laid out for the default of 88 units (register-resident
// recurrent kernel, 22 k values per slice).  These two kernels take the number of units at run time, with the contract of
// clv_lstm_seq_fwd / clv_lstm_seq_bwd (include/clvae.h), which dispatch here for H != 88:
//   z_t = xproj[b,t,:] + rowbias[b,:] + h_{t-1} . U,   gate blocks i, f, c, o along 4H (Keras order)
//   forward stores hs, cs and gates = (z_i, z_f, tanh z_c, z_o) in place of xproj; backward turns gates into dz in place.
//
// One workgroup of 256 threads per batch row, all T steps.  The recurrent kernel U [H,4H] does not fit registers for a
// run-time H and not the LDS beyond 100 units, so it is streamed from L2 every step (256 KB at 128 units, shared by all
// workgroups): thread (ks, u) sums its k-slice of h . U[:, g*H + u] for the four gates of unit u -- lanes along u, so
// every load is a contiguous row piece -- the slices meet in LDS, the unit's owner (ks == 0) does the cell.  Backward:
// dh_{t-1} = dz_t . U^T needs ROWS of U: a wave per row, lanes along the 4H columns (contiguous again), wave reduction.
// Latency-bound by design (one row per workgroup, like the reference's K.rnn loop per sample); the fast paths stay where
// the reference's own default lives.
#include "lstm_common.h"

namespace clv {

constexpr int LA_NT = 256;
constexpr int LA_MAXH = 1024;                 // units (4 per thread at most)
constexpr int LA_UPT = LA_MAXH / LA_NT;       // units per owner thread
constexpr int LA_KB = 8;                      // rows of U a thread keeps in flight (forward), rows of U per wave round (backward)

struct LstmAnyFwdArgs {
  int B, T, H;
  const float *xproj, *rowbias, *U, *h0, *c0;
  float *hs, *cs, *gates, *hT, *cT;
};

// k-slices per unit: as many as 256 threads allow (a power of two, at most 8)
__host__ __device__ inline int la_slices(int H) {
  int ks = 1;
  while (ks < 8 && 2 * ks * H <= LA_NT) ks *= 2;
  return ks;
}

template <int GATE>
__global__ __launch_bounds__(LA_NT) void lstm_any_fwd_kernel(LstmAnyFwdArgs a) {
  extern __shared__ float la_lds[];
  const int H = a.H, G4 = 4 * a.H, T = a.T, b = blockIdx.x, tid = threadIdx.x;
  const int KS = la_slices(H);                 // uniform
  const int UT = H < LA_NT ? H : LA_NT;        // owner threads per slice
  const int ks = tid / UT, ul = tid % UT;
  const bool active = ks < KS;
  const bool owner = active && ks == 0;
  const int KL = (H + KS - 1) / KS, k0 = ks * KL, k1 = min(H, k0 + KL);
  float* hbuf = la_lds;                        // [2][H]
  float* part = la_lds + 2 * H;                // [KS][4][H] partial gate sums
  const size_t bt0 = (size_t)b * T;
  float c[LA_UPT];
#pragma unroll
  for (int i = 0; i < LA_UPT; ++i) {
    const int u = ul + i * LA_NT;
    c[i] = (owner && u < H && a.c0) ? a.c0[(size_t)b * H + u] : 0.f;
    if (owner && u < H) hbuf[u] = a.h0 ? a.h0[(size_t)b * H + u] : 0.f;
  }
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    const int cur = t & 1;
    const float* hv = hbuf + cur * H;
    const float* xp = a.xproj + (bt0 + t) * G4;
    float x[LA_UPT][4];
    if (owner) {                               // requested first: lands under the product below
#pragma unroll
      for (int i = 0; i < LA_UPT; ++i) {
        const int u = ul + i * LA_NT;
#pragma unroll
        for (int g = 0; g < 4; ++g)
          x[i][g] = u < H ? xp[g * H + u] + (a.rowbias ? a.rowbias[(size_t)b * G4 + g * H + u] : 0.f) : 0.f;
      }
    }
    if (active) {
#pragma unroll
      for (int i = 0; i < LA_UPT; ++i) {
        const int u = ul + i * LA_NT;
        if (u >= H) break;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const float* Up = a.U + u;
        // LA_KB rows of U in flight per round (4 LA_KB loads requested before the first FMA): a k loop that waits for every
        // row's four loads pays one L2 round trip per k (17 us per step at 128 units: tools/any_width_bench.py)
        int k = k0;
        for (; k + LA_KB <= k1; k += LA_KB) {
          float uv[LA_KB][4];
#pragma unroll
          for (int j = 0; j < LA_KB; ++j) {
            const float* row = Up + (size_t)(k + j) * G4;
#pragma unroll
            for (int g = 0; g < 4; ++g) uv[j][g] = row[g * H];
          }
#pragma unroll
          for (int j = 0; j < LA_KB; ++j) {
            const float hk = hv[k + j];
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = fmaf(hk, uv[j][g], acc[g]);
          }
        }
        for (; k < k1; ++k) {
          const float hk = hv[k];
          const float* row = Up + (size_t)k * G4;
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[g] = fmaf(hk, row[g * H], acc[g]);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) part[(ks * 4 + g) * H + u] = acc[g];
      }
    }
    __syncthreads();
    if (owner) {
#pragma unroll
      for (int i = 0; i < LA_UPT; ++i) {
        const int u = ul + i * LA_NT;
        if (u >= H) break;
        float z[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float s = x[i][g];
          for (int q = 0; q < KS; ++q) s += part[(q * 4 + g) * H + u];
          z[g] = s;
        }
        float h, gg;
        lstm_cell<GATE>(z, c[i], h, gg);
        hbuf[(cur ^ 1) * H + u] = h;
        a.hs[(bt0 + t) * H + u] = h;
        if (a.cs) a.cs[(bt0 + t) * H + u] = c[i];
        if (a.gates) {
          float* gp = a.gates + (bt0 + t) * G4 + u;
          gp[0] = z[0]; gp[H] = z[1]; gp[2 * H] = gg; gp[3 * H] = z[3];
        }
      }
    }
    __syncthreads();
  }
  if (owner) {
#pragma unroll
    for (int i = 0; i < LA_UPT; ++i) {
      const int u = ul + i * LA_NT;
      if (u >= H) break;
      if (a.hT) a.hT[(size_t)b * H + u] = hbuf[(T & 1) * H + u];
      if (a.cT) a.cT[(size_t)b * H + u] = c[i];
    }
  }
}

struct LstmAnyBwdArgs {
  int B, T, H;
  const float *U, *dhs, *cs, *c0;
  float *gates, *dzsum;
};

template <int GATE>
__global__ __launch_bounds__(LA_NT) void lstm_any_bwd_kernel(LstmAnyBwdArgs a) {
  extern __shared__ float la_lds[];
  const int H = a.H, G4 = 4 * a.H, T = a.T, b = blockIdx.x, tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  float* dzb = la_lds;                         // [4H] dz_t
  float* dhr = la_lds + G4;                    // [H]  dh_t from step t+1 (recurrent part)
  const size_t bt0 = (size_t)b * T;
  float dc[LA_UPT], zs[LA_UPT][4];
#pragma unroll
  for (int i = 0; i < LA_UPT; ++i) {
    dc[i] = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) zs[i][g] = 0.f;
  }
  for (int u = tid; u < H; u += LA_NT) dhr[u] = 0.f;
  __syncthreads();
  for (int t = T - 1; t >= 0; --t) {
#pragma unroll
    for (int i = 0; i < LA_UPT; ++i) {
      const int u = tid + i * LA_NT;
      if (u >= H) break;
      float* gp = a.gates + (bt0 + t) * G4 + u;
      const float zi = gp[0], zf = gp[H], g = gp[2 * H], zo = gp[3 * H];
      const float ct = a.cs[(bt0 + t) * H + u];
      const float cp = t > 0 ? a.cs[(bt0 + t - 1) * H + u] : (a.c0 ? a.c0[(size_t)b * H + u] : 0.f);
      const float dh = a.dhs[(bt0 + t) * H + u] + dhr[u];
      const float ig = gate_fn<GATE>(zi), fg = gate_fn<GATE>(zf), og = gate_fn<GATE>(zo);
      const float tc = fast_tanh(ct);
      const float dct = fmaf(dh, og * (1.f - tc * tc), dc[i]);
      const float dz[4] = {dct * g * gate_grad<GATE>(zi, ig), dct * cp * gate_grad<GATE>(zf, fg), dct * ig * (1.f - g * g),
                           dh * tc * gate_grad<GATE>(zo, og)};
      dc[i] = dct * fg;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        gp[q * H] = dz[q];
        dzb[q * H + u] = dz[q];
        zs[i][q] += dz[q];
      }
    }
    __syncthreads();
    if (t > 0) {                               // dh_{t-1}[k] += sum_j dz_t[j] U[k][j]: a wave per row of U
      // LA_KB / 2 rows of U per round, their loads requested together, their wave reductions interleaved
      constexpr int RB = LA_KB / 2, NWV = LA_NT / 64;
      for (int kb = wave * RB; kb < H; kb += NWV * RB) {
        float s[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) s[r] = 0.f;
        for (int j = lane; j < G4; j += 64) {
          const float dv = dzb[j];
          float uv[RB];
#pragma unroll
          for (int r = 0; r < RB; ++r) uv[r] = a.U[(size_t)min(kb + r, H - 1) * G4 + j];
#pragma unroll
          for (int r = 0; r < RB; ++r) s[r] = fmaf(dv, uv[r], s[r]);
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) s[r] = wave_sum(s[r]);
        if (lane == 0) {
#pragma unroll
          for (int r = 0; r < RB; ++r)
            if (kb + r < H) dhr[kb + r] = s[r];
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < LA_UPT; ++i) {
    const int u = tid + i * LA_NT;
    if (u >= H) break;
#pragma unroll
    for (int g = 0; g < 4; ++g) a.dzsum[(size_t)b * G4 + g * H + u] = zs[i][g];
  }
}

int launch_lstm_any_fwd(int B, int T, int H, int gate_act, const float* xproj, const float* rowbias, const float* U,
                        const float* h0, const float* c0, float* hs, float* cs, float* gates, float* hT, float* cT,
                        hipStream_t s) {
  if (H < 1 || H > LA_MAXH) return CLV_EINVAL;
  LstmAnyFwdArgs a{B, T, H, xproj, rowbias, U, h0, c0, hs, cs, gates, hT, cT};
  const size_t lds = (size_t)(2 + 4 * la_slices(H)) * H * sizeof(float);
  ProfScope p("lstm_any_fwd", s);
  if (gate_act == CLV_GATE_HARD_SIGMOID)
    hipLaunchKernelGGL(lstm_any_fwd_kernel<CLV_GATE_HARD_SIGMOID>, dim3(B), dim3(LA_NT), lds, s, a);
  else
    hipLaunchKernelGGL(lstm_any_fwd_kernel<CLV_GATE_SIGMOID>, dim3(B), dim3(LA_NT), lds, s, a);
  return launch_status();
}

int launch_lstm_any_bwd(int B, int T, int H, int gate_act, const float* U, const float* dhs, const float* cs,
                        const float* c0, float* gates_inout_dz, float* dzsum, hipStream_t s) {
  if (H < 1 || H > LA_MAXH) return CLV_EINVAL;
  LstmAnyBwdArgs a{B, T, H, U, dhs, cs, c0, gates_inout_dz, dzsum};
  const size_t lds = (size_t)5 * H * sizeof(float);
  ProfScope p("lstm_any_bwd", s);
  if (gate_act == CLV_GATE_HARD_SIGMOID)
    hipLaunchKernelGGL(lstm_any_bwd_kernel<CLV_GATE_HARD_SIGMOID>, dim3(B), dim3(LA_NT), lds, s, a);
  else
    hipLaunchKernelGGL(lstm_any_bwd_kernel<CLV_GATE_SIGMOID>, dim3(B), dim3(LA_NT), lds, s, a);
  return launch_status();
}

}  // namespace clv
