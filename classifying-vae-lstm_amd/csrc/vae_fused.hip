// vae_fused.hip -- one whole cl_vae training step (forward, sampling, four losses, backward) in ONE launch.
//
// cl_vae is 33 k parameters: as separate layers it is ~45 launches of a few microseconds each, i.e. launch-
// bound.  Here a workgroup owns RB batch rows and walks the entire graph of cl_vae/model.py:136-219 with its
// activations resident in LDS (8 wide + 8 narrow [RB x .] buffers, < 140 KB) and the weights streamed from
// L2 straight into MFMA B operands (v_mfma_f32_16x16x4_f32, exact fp32); per-row sampling / loss math runs
// on the first RB threads.  Weight gradients are produced per workgroup (K = RB rows of the batch) into a
// slab laid out like the flat parameter buffer and summed by a second tiny launch, so the result is
// deterministic (no float atomics).  Dims: D, H, Hc <= 96; C, L <= 16.
#include "common.h"

namespace clv {

typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int VLW = 98;    // wide LDS row stride  (== 2 mod 32: MFMA A reads of 16 rows x 2 k are conflict-free)
constexpr int VLS = 34;    // narrow LDS row stride
constexpr int VNW = 8;     // waves per workgroup: one 16-column tile of an 88-wide layer per wave
constexpr int VNT = VNW * 64;
constexpr float VEPS_K = 1e-7f, VW2 = 1e-10f, VLOGIT_CLIP = 16.11809555f;

struct VaeArgs {
  int B, D, H, Hc, C, L, use_xp;
  const float* x; const float* xp; const float* onehot;    // [B,D] [B,D] [B,C]
  const float* y;                                          // [B,D] reconstruction target (NULL: x itself)
  const float* eps_w; const float* eps_z;                    // [B,C-1] [B,L]
  const float* P;                                            // flat parameters
  long o_hw_k, o_hw_b, o_wa_k, o_wa_b, o_h_k, o_h_b, o_za_k, o_za_b, o_d_k, o_d_b, o_x_k, o_x_b;
  float prior, class_weight, kl_weight, w_kl_weight;
  int need_grads;
  float* slab; long n_params;                                // [n_wg][n_params] partial gradients
  float* logits; float* w_out; float* wargs_out; float* zargs_out;   // [B,D] [B,C] [B,2(C-1)] [B,2L] (for predict/tests)
  float* rownll; float* rowkl; float* rowloss;               // [B] [B] [B,3]
};

struct MSrc { const float* a; int lda; int K; const float* w; };

// out[RB x N] = act(sum_s A_s[RB x K_s] . W_s[K_s x N] + bias) (* mask>0); A in LDS, W in global (row stride ldw)
template <int RB>
__device__ void dense(const MSrc* src, int nsrc, int ldw, int N, const float* bias, bool relu, float* out, int ldo) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  for (int nt = wave; nt * 16 < N; nt += VNW) {
    const int n0 = nt * 16;
    f32x4v acc[RB / 16];
#pragma unroll
    for (int mt = 0; mt < RB / 16; ++mt) acc[mt] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < nsrc; ++s) {
      const MSrc sr = src[s];
      // 8 k-steps per chunk: the 8 weight loads (L2 latency ~0.3 us each) are issued together, then consumed
      for (int kc = 0; kc < sr.K; kc += 32) {
        float bv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int k = kc + 4 * i + q;
          bv[i] = (k < sr.K && n0 + r < N) ? sr.w[(size_t)k * ldw + n0 + r] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int k = kc + 4 * i + q;
#pragma unroll
          for (int mt = 0; mt < RB / 16; ++mt) {
            const float a = k < sr.K ? sr.a[(mt * 16 + r) * sr.lda + k] : 0.f;
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[i], acc[mt], 0, 0, 0);
          }
        }
      }
    }
    const int col = n0 + r;
    if (col < N) {
      const float bv = bias ? bias[col] : 0.f;
#pragma unroll
      for (int mt = 0; mt < RB / 16; ++mt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          float v = acc[mt][reg] + bv;
          if (relu) v = fmaxf(v, 0.f);
          out[(mt * 16 + q * 4 + reg) * ldo + col] = v;
        }
    }
  }
}

// out[RB x Kr] (+)= (DY[RB x N] . W[Kr x N]^T) (* (mask > 0)); DY, out, mask in LDS
template <int RB>
__device__ void dense_t(const float* dy, int ldy, int N, const float* w, int ldw, int Kr, float* out, int ldo,
                        const float* mask, int ldm, bool accumulate) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  for (int kt = wave; kt * 16 < Kr; kt += VNW) {
    const int c0 = kt * 16;
    f32x4v acc[RB / 16];
#pragma unroll
    for (int mt = 0; mt < RB / 16; ++mt) acc[mt] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    for (int nc = 0; nc < N; nc += 32) {
      float bv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int n = nc + 4 * i + q;
        bv[i] = (n < N && c0 + r < Kr) ? w[(size_t)(c0 + r) * ldw + n] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int n = nc + 4 * i + q;
#pragma unroll
        for (int mt = 0; mt < RB / 16; ++mt) {
          const float a = n < N ? dy[(mt * 16 + r) * ldy + n] : 0.f;
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[i], acc[mt], 0, 0, 0);
        }
      }
    }
    const int col = c0 + r;
    if (col < Kr) {
#pragma unroll
      for (int mt = 0; mt < RB / 16; ++mt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int row = mt * 16 + q * 4 + reg;
          float v = acc[mt][reg];
          if (accumulate) v += out[row * ldo + col];
          if (mask) v = mask[row * ldm + col] > 0.f ? v : 0.f;
          out[row * ldo + col] = v;
        }
    }
  }
}

// gout[Kin x N] = A[RB x Kin]^T . DY[RB x N]  (this workgroup's share of a kernel gradient); A, DY in LDS
template <int RB>
__device__ void wgrad(const float* a_lds, int lda, int Kin, const float* dy, int ldy, int N, float* gout, int ldo) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int kts = (Kin + 15) / 16, nts = (N + 15) / 16;
  for (int tile = wave; tile < kts * nts; tile += VNW) {
    const int k0 = (tile / nts) * 16, n0 = (tile % nts) * 16;
    f32x4v acc = (f32x4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r0 = 0; r0 < RB; r0 += 4) {
      const float av = k0 + r < Kin ? a_lds[(r0 + q) * lda + k0 + r] : 0.f;
      const float bv = n0 + r < N ? dy[(r0 + q) * ldy + n0 + r] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
    }
    const int col = n0 + r;
    if (col < N)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int row = k0 + q * 4 + reg;
        if (row < Kin) gout[(size_t)row * ldo + col] = acc[reg];
      }
  }
}

template <int RB>
__device__ void bgrad(const float* dy, int ldy, int N, float* gout) {
  for (int c = threadIdx.x; c < N; c += VNT) {
    float s = 0.f;
#pragma unroll 8
    for (int r = 0; r < RB; ++r) s += dy[r * ldy + c];
    gout[c] = s;
  }
}

template <int RB>
__global__ __launch_bounds__(VNT) void vae_fused_kernel(VaeArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* X = lds;                 float* XP = X + RB * VLW;    float* HW = XP + RB * VLW;  float* Hh = HW + RB * VLW;
  float* HD = Hh + RB * VLW;      float* LG = HD + RB * VLW;   float* G1 = LG + RB * VLW;  float* G2 = G1 + RB * VLW;
  float* WARGS = G2 + RB * VLW;   float* WV = WARGS + RB * VLS; float* ZARGS = WV + RB * VLS; float* ZV = ZARGS + RB * VLS;
  float* DWV = ZV + RB * VLS;     float* DWARGS = DWV + RB * VLS; float* DZARGS = DWARGS + RB * VLS; float* DZ = DZARGS + RB * VLS;
  const int tid = threadIdx.x;
  const int row0 = blockIdx.x * RB;
  const int nvalid = min(RB, a.B - row0);
  const int D = a.D, H = a.H, Hc = a.Hc, C = a.C, L = a.L, C1 = a.C - 1, NA = 2 * (a.C - 1), NZ = 2 * a.L;
  const float* P = a.P;
  const float inv_b = 1.f / (float)a.B;

  // ---- inputs -> LDS (rows beyond the batch are zero) -----------------------------------------------------
  for (int i = tid; i < RB * D; i += VNT) {
    const int r = i / D, c = i % D;
    X[r * VLW + c] = r < nvalid ? a.x[(size_t)(row0 + r) * D + c] : 0.f;
    XP[r * VLW + c] = (a.use_xp && r < nvalid) ? a.xp[(size_t)(row0 + r) * D + c] : 0.f;
  }
  __syncthreads();
  // ---- label encoder (:141-143) ---------------------------------------------------------------------------
  { MSrc s[1] = {{X, VLW, D, P + a.o_hw_k}}; dense<RB>(s, 1, Hc, Hc, P + a.o_hw_b, true, HW, VLW); }
  __syncthreads();
  { MSrc s[1] = {{HW, VLW, Hc, P + a.o_wa_k}}; dense<RB>(s, 1, NA, NA, P + a.o_wa_b, false, WARGS, VLS); }
  __syncthreads();
  // ---- w ~ logistic-normal, label losses (:146-157,198-206) -----------------------------------------------
  if (tid < RB) {
    const int r = tid;
    float* wv = WV + r * VLS;
    float e[16];
    float S = 1.f, klw = 0.f;
    const float ep = __expf(a.prior);
    for (int j = 0; j < C1; ++j) {
      const float m = WARGS[r * VLS + j], lv = WARGS[r * VLS + C1 + j];
      const float sd = expf(0.5f * lv);
      const float ew = r < nvalid ? a.eps_w[(size_t)(row0 + r) * C1 + j] : 0.f;
      e[j] = expf(m + sd * ew);
      S += e[j];
      klw += 1.f - a.prior + lv - sd * sd / ep - m * m / ep;
    }
    e[C1] = 1.f;
    const float invS = 1.f / S;
    float qs = 0.f, wbest = -1.f, tbest = -1.f;
    int amax = 0, tmax = 0;
    for (int j = 0; j < C; ++j) {
      wv[j] = e[j] * invS;
      qs += wv[j] + VW2;
      if (wv[j] > wbest) { wbest = wv[j]; amax = j; }
      const float tj = (a.onehot && r < nvalid) ? a.onehot[(size_t)(row0 + r) * C + j] : 0.f;
      if (tj > tbest) { tbest = tj; tmax = j; }
    }
    for (int j = C; j < ((C + 3) & ~3); ++j) wv[j] = 0.f;      // zero padding for the MFMA k-steps
    if (r < nvalid) {
      float wrec = 0.f;
      if (a.onehot)
        for (int j = 0; j < C; ++j)
          wrec -= a.onehot[(size_t)(row0 + r) * C + j] * logf(fminf(fmaxf((wv[j] + VW2) / qs, VEPS_K), 1.f - VEPS_K));
      a.rowloss[(size_t)(row0 + r) * 3 + 0] = -0.5f * klw;
      a.rowloss[(size_t)(row0 + r) * 3 + 1] = (float)C1 * wrec;
      a.rowloss[(size_t)(row0 + r) * 3 + 2] = (a.onehot && amax == tmax) ? 1.f : 0.f;
      for (int j = 0; j < C; ++j) a.w_out[(size_t)(row0 + r) * C + j] = wv[j];
      for (int j = 0; j < NA; ++j) a.wargs_out[(size_t)(row0 + r) * NA + j] = WARGS[r * VLS + j];
    }
  }
  __syncthreads();
  // ---- latent encoder (:160-174) --------------------------------------------------------------------------
  { MSrc s[2] = {{X, VLW, D, P + a.o_h_k}, {WV, VLS, C, P + a.o_h_k + (long)D * H}};
    dense<RB>(s, 2, H, H, P + a.o_h_b, true, Hh, VLW); }
  __syncthreads();
  { MSrc s[1] = {{Hh, VLW, H, P + a.o_za_k}}; dense<RB>(s, 1, NZ, NZ, P + a.o_za_b, false, ZARGS, VLS); }
  __syncthreads();
  if (tid < RB) {
    const int r = tid;
    float kl = 0.f;
    for (int j = 0; j < L; ++j) {
      const float m = ZARGS[r * VLS + j], lv = ZARGS[r * VLS + L + j];
      const float sd = expf(0.5f * lv);
      const float ez = r < nvalid ? a.eps_z[(size_t)(row0 + r) * L + j] : 0.f;
      ZV[r * VLS + j] = m + sd * ez;
      kl += 1.f + lv - m * m - sd * sd;
    }
    for (int j = L; j < ((L + 3) & ~3); ++j) ZV[r * VLS + j] = 0.f;
    if (r < nvalid) {
      a.rowkl[row0 + r] = -0.5f * kl;
      for (int j = 0; j < NZ; ++j) a.zargs_out[(size_t)(row0 + r) * NZ + j] = ZARGS[r * VLS + j];
    }
  }
  __syncthreads();
  // ---- decoder on [w, xp, z] (:177-188) -------------------------------------------------------------------
  const long xo = a.use_xp ? D : 0;
  { MSrc s[3] = {{WV, VLS, C, P + a.o_d_k}, {XP, VLW, a.use_xp ? D : 0, P + a.o_d_k + (long)C * H},
                 {ZV, VLS, L, P + a.o_d_k + (long)(C + xo) * H}};
    dense<RB>(s, 3, H, H, P + a.o_d_b, true, HD, VLW); }
  __syncthreads();
  { MSrc s[1] = {{HD, VLW, H, P + a.o_x_k}}; dense<RB>(s, 1, D, D, P + a.o_x_b, false, LG, VLW); }
  __syncthreads();
  // ---- Bernoulli NLL on the logits (Keras clip semantics) and its gradient, in place ----------------------
  {
    const int lane = tid & 63, wave = tid >> 6;
    for (int r = wave; r < RB; r += VNW) {
      float acc = 0.f;
      for (int j = lane; j < D; j += 64) {
        const float av = LG[r * VLW + j];
        const float t = a.y ? (r < nvalid ? a.y[(size_t)(row0 + r) * D + j] : 0.f) : X[r * VLW + j];
        if (r < nvalid && a.logits) a.logits[(size_t)(row0 + r) * D + j] = av;
        const float l = fminf(fmaxf(av, -VLOGIT_CLIP), VLOGIT_CLIP);
        const float e = __expf(-fabsf(l));
        acc += fmaxf(l, 0.f) + __logf(1.f + e) - l * t;
        const float r1 = fast_rcp(1.f + e);
        const float sg = l >= 0.f ? r1 : e * r1;
        const bool inside = (av >= -VLOGIT_CLIP) && (av <= VLOGIT_CLIP);
        LG[r * VLW + j] = (inside && r < nvalid) ? inv_b * (sg - t) : 0.f;
      }
      acc = wave_sum(acc);
      if (lane == 0 && r < nvalid) a.rownll[row0 + r] = acc;
    }
  }
  __syncthreads();
  if (!a.need_grads) return;
  float* G = a.slab + (size_t)blockIdx.x * a.n_params;

  // ---- output layer ---------------------------------------------------------------------------------------
  wgrad<RB>(HD, VLW, H, LG, VLW, D, G + a.o_x_k, D);
  bgrad<RB>(LG, VLW, D, G + a.o_x_b);
  dense_t<RB>(LG, VLW, D, P + a.o_x_k, D, H, G1, VLW, HD, VLW, false);          // G1 = d h_dec
  __syncthreads();
  // ---- decoder hidden layer -------------------------------------------------------------------------------
  wgrad<RB>(WV, VLS, C, G1, VLW, H, G + a.o_d_k, H);
  if (a.use_xp) wgrad<RB>(XP, VLW, D, G1, VLW, H, G + a.o_d_k + (long)C * H, H);
  wgrad<RB>(ZV, VLS, L, G1, VLW, H, G + a.o_d_k + (long)(C + xo) * H, H);
  bgrad<RB>(G1, VLW, H, G + a.o_d_b);
  dense_t<RB>(G1, VLW, H, P + a.o_d_k, H, C, DWV, VLS, nullptr, 0, false);                         // dL/dw (decoder part)
  dense_t<RB>(G1, VLW, H, P + a.o_d_k + (long)(C + xo) * H, H, L, DZ, VLS, nullptr, 0, false);    // dL/dz
  __syncthreads();
  // ---- gaussian head backward -----------------------------------------------------------------------------
  if (tid < RB) {
    const int r = tid;
    const float ks = a.kl_weight * inv_b;
    for (int j = 0; j < L; ++j) {
      const float m = ZARGS[r * VLS + j], lv = ZARGS[r * VLS + L + j];
      const float sd = expf(0.5f * lv);
      const float d = DZ[r * VLS + j];
      const float ez = r < nvalid ? a.eps_z[(size_t)(row0 + r) * L + j] : 0.f;
      DZARGS[r * VLS + j] = r < nvalid ? d + ks * m : 0.f;
      DZARGS[r * VLS + L + j] = r < nvalid ? d * ez * 0.5f * sd - 0.5f * ks * (1.f - sd * sd) : 0.f;
    }
  }
  __syncthreads();
  wgrad<RB>(Hh, VLW, H, DZARGS, VLS, NZ, G + a.o_za_k, NZ);
  bgrad<RB>(DZARGS, VLS, NZ, G + a.o_za_b);
  dense_t<RB>(DZARGS, VLS, NZ, P + a.o_za_k, NZ, H, G2, VLW, Hh, VLW, false);    // G2 = d h
  __syncthreads();
  wgrad<RB>(X, VLW, D, G2, VLW, H, G + a.o_h_k, H);
  wgrad<RB>(WV, VLS, C, G2, VLW, H, G + a.o_h_k + (long)D * H, H);
  bgrad<RB>(G2, VLW, H, G + a.o_h_b);
  dense_t<RB>(G2, VLW, H, P + a.o_h_k + (long)D * H, H, C, DWV, VLS, nullptr, 0, true);    // dL/dw += encoder part
  __syncthreads();
  // ---- label head backward --------------------------------------------------------------------------------
  if (tid < RB) {
    const int r = tid;
    const float ep = __expf(a.prior);
    const float* wv = WV + r * VLS;
    float dn[16], d[16];
    float qs = 0.f, dot = 0.f, dsum = 0.f;
    for (int j = 0; j < C; ++j) qs += wv[j] + VW2;
    for (int j = 0; j < C; ++j) {
      const float n = (wv[j] + VW2) / qs;
      const bool inside = (n >= VEPS_K) && (n <= 1.f - VEPS_K);
      const float nc = fminf(fmaxf(n, VEPS_K), 1.f - VEPS_K);
      const float oh = r < nvalid ? a.onehot[(size_t)(row0 + r) * C + j] : 0.f;
      dn[j] = inside ? -(float)C1 * oh / nc : 0.f;
      dot += dn[j] * n;
    }
    for (int j = 0; j < C; ++j) {
      d[j] = DWV[r * VLS + j] + a.class_weight * inv_b * ((dn[j] - dot) / qs);
      dsum += d[j] * wv[j];
    }
    for (int j = 0; j < C1; ++j) {
      const float ds = wv[j] * (d[j] - dsum);
      const float m = WARGS[r * VLS + j], lv = WARGS[r * VLS + C1 + j];
      const float sd = expf(0.5f * lv);
      const float ew = r < nvalid ? a.eps_w[(size_t)(row0 + r) * C1 + j] : 0.f;
      DWARGS[r * VLS + j] = r < nvalid ? ds + a.w_kl_weight * inv_b * (m / ep) : 0.f;
      DWARGS[r * VLS + C1 + j] =
          r < nvalid ? ds * ew * 0.5f * sd + a.w_kl_weight * inv_b * (-0.5f * (1.f - sd * sd / ep)) : 0.f;
    }
  }
  __syncthreads();
  wgrad<RB>(HW, VLW, Hc, DWARGS, VLS, NA, G + a.o_wa_k, NA);
  bgrad<RB>(DWARGS, VLS, NA, G + a.o_wa_b);
  dense_t<RB>(DWARGS, VLS, NA, P + a.o_wa_k, NA, Hc, G1, VLW, HW, VLW, false);   // G1 = d h_w
  __syncthreads();
  wgrad<RB>(X, VLW, D, G1, VLW, Hc, G + a.o_hw_k, Hc);
  bgrad<RB>(G1, VLW, Hc, G + a.o_hw_b);
}

// grads[i] = sum_s slab[s][i]
__global__ __launch_bounds__(256) void slab_sum_kernel(const float* slab, int nslabs, long n, float* out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int s = 0;
  for (; s + 3 < nslabs; s += 4) {
    a0 += slab[(size_t)s * n + i]; a1 += slab[(size_t)(s + 1) * n + i];
    a2 += slab[(size_t)(s + 2) * n + i]; a3 += slab[(size_t)(s + 3) * n + i];
  }
  for (; s < nslabs; ++s) a0 += slab[(size_t)s * n + i];
  out[i] = (a0 + a1) + (a2 + a3);
}

template <int RB>
static size_t vae_lds_bytes() { return (size_t)(8 * RB * VLW + 8 * RB * VLS) * sizeof(float); }

}  // namespace clv

using namespace clv;

static int vae_rb(int B) { return B <= 2048 ? 16 : 32; }

extern "C" int clv_vae_fused_supported(int D, int H, int Hc, int C, int L) {
  return D > 0 && D <= 96 && H > 0 && H <= 96 && Hc > 0 && Hc <= 96 && C >= 2 && C <= 16 && L > 0 && L <= 16;
}

extern "C" size_t clv_vae_fused_workspace_bytes(int B, long n_params) {
  const int rb = vae_rb(B);
  return (size_t)((B + rb - 1) / rb) * n_params * sizeof(float);
}

extern "C" int clv_vae_fused_step(int B, int D, int H, int Hc, int C, int L, int use_x_prev,
                                  const float* x, const float* xp, const float* target, const float* onehot,
                                  const float* eps_w, const float* eps_z,
                                  const float* params, const int64_t* host_offsets12, long n_params,
                                  float prior_logvar, float class_weight, float kl_weight, float w_kl_weight,
                                  int need_grads, float* grads, void* ws, size_t ws_bytes,
                                  float* logits, float* w_out, float* wargs_out, float* zargs_out,
                                  float* rownll, float* rowkl, float* rowloss, void* stream) {
  if (!clv_vae_fused_supported(D, H, Hc, C, L) || B <= 0) return CLV_EINVAL;
  if (!x || !eps_w || !eps_z || !params || !host_offsets12 || !w_out || !wargs_out || !zargs_out || !rownll || !rowkl || !rowloss)
    return CLV_EINVAL;
  if (use_x_prev && !xp) return CLV_EINVAL;
  if (need_grads && (!grads || !onehot)) return CLV_EINVAL;
  const int rb = vae_rb(B);
  const int nwg = (B + rb - 1) / rb;
  if (need_grads && (!ws || ws_bytes < (size_t)nwg * n_params * sizeof(float))) return CLV_EWORKSPACE;
  const int64_t* o = host_offsets12;
  VaeArgs a{B, D, H, Hc, C, L, use_x_prev, x, xp, onehot, target == x ? nullptr : target, eps_w, eps_z, params,
            (long)o[0], (long)o[1], (long)o[2], (long)o[3], (long)o[4], (long)o[5], (long)o[6], (long)o[7], (long)o[8],
            (long)o[9], (long)o[10], (long)o[11], prior_logvar, class_weight, kl_weight, w_kl_weight, need_grads,
            (float*)ws, n_params, logits, w_out, wargs_out, zargs_out, rownll, rowkl, rowloss};
  hipStream_t s = (hipStream_t)stream;
  {
    ProfScope p("vae_fused_step", s);
    if (rb == 16) {
      auto k = vae_fused_kernel<16>;
      if (int e = allow_dynamic_lds(reinterpret_cast<const void*>(k), (int)vae_lds_bytes<16>())) return e;
      hipLaunchKernelGGL(k, dim3(nwg), dim3(VNT), vae_lds_bytes<16>(), s, a);
    } else {
      auto k = vae_fused_kernel<32>;
      if (int e = allow_dynamic_lds(reinterpret_cast<const void*>(k), (int)vae_lds_bytes<32>())) return e;
      hipLaunchKernelGGL(k, dim3(nwg), dim3(VNT), vae_lds_bytes<32>(), s, a);
    }
  }
  int st = launch_status();
  if (st || !need_grads) return st;
  {
    ProfScope p("vae_slab_sum", s);
    hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)((n_params + 255) / 256)), dim3(256), 0, s, (const float*)ws, nwg,
                       n_params, grads);
  }
  return launch_status();
}
