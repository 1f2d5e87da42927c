// generate.hip -- cl_vrnn autoregressive generation as ONE persistent kernel per batch of sequences (gfx950).
//
// cl_vrnn/model.py:9-60 generates a sequence frame by frame: encoder LSTM step on [x_{t-1}, w], z ~ N(mean,
// exp(log_var)) from the latent head, decoder LSTM step on [x_{t-1}, z, w], x_hat = sigmoid(head), x_t ~
// Bernoulli(x_hat), with teacher forcing over the seed frames.  Every frame depends on the previous sample, so
// per-frame launches (14 kernels, even as one hipGraph replay ~100 us) are pure latency.  Here a workgroup owns a
// sequence for its whole length: both recurrent kernels live in registers (the 4-lane k-slice layout of
// lstm_pair.hip), the encoder's input kernel in LDS (the input frame is a handful of notes: its projection is a
// gather of kernel rows), the decoder's input-kernel rows are prefetched from L2 while the encoder runs, the
// noise comes from Philox in place (same values as clv_philox_normal/uniform give for (seed, frame, stream, index)),
// and a frame is four LDS barriers:
//     encoder cell | latent head + z | decoder cell | output head + Bernoulli sample.
#include "lstm_common.h"
#include "philox.h"

namespace clv {

constexpr int GN_NW = 6;                 // waves per role
constexpr int GN_NT = 2 * GN_NW * 64;    // 768 threads: waves 0-5 encoder + head, waves 6-11 decoder
constexpr int GN_LMAX = 16;              // latent dims carried by the encoder's surplus lane groups (2 per group)
constexpr int GN_LWIDE = 32;             // wide-latent variant: one head column per encoder lane group, one more barrier
constexpr int GN_CMAX = 32;

struct GenArgs {
  int N, S, nsteps, L, C, z_prior, has_xp;
  uint32_t k0, k1;              // Philox key (seed)
  const float* x_seed;          // [N,S,88]
  const float* w;               // [N,C]
  const float* Kx_enc;          // [88,352]  rows of encoder_h/kernel that multiply x_{t-1}
  const float* Kw_enc;          // [C,352]
  const float* b_enc;           // [352]
  const float* U_enc;           // [88,352]
  const float* Wz;              // [88,2L]
  const float* bz;              // [2L]
  const float* Kx_dec;          // [88,352] or unused
  const float* Kz;              // [L,352]
  const float* Kw_dec;          // [C,352]
  const float* b_dec;
  const float* U_dec;
  const float* Wo;              // [88,88]
  const float* bo;              // [88]
  float* Xs;                    // [N,nsteps,88]
  float* xhat;                  // [N,S+nsteps,88] or null
};

// slice_matvec with half the live registers: the h slice is consumed in two halves of 12 (the kernel is at its
// 168-register budget; a spill inside the frame loop costs far more than the second LDS wait)
__device__ __forceinline__ void slice_matvec_lr(const float* hslice, const f2 (&Ur)[PKK][2], f2 (&acc2)[2]) {
  const float4* hp = reinterpret_cast<const float4*>(hslice);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    float hv[12];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const float4 v = hp[3 * half + q];
      hv[4 * q] = v.x; hv[4 * q + 1] = v.y; hv[4 * q + 2] = v.z; hv[4 * q + 3] = v.w;
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const int kk = 12 * half + j;
      if (kk < PKK) {
        const f2 hh = {hv[j], hv[j]};
        acc2[0] = __builtin_elementwise_fma(hh, Ur[kk][0], acc2[0]);
        acc2[1] = __builtin_elementwise_fma(hh, Ur[kk][1], acc2[1]);
      }
    }
  }
}

// nonzero inputs of the frame in xbuf as two scalar masks (inputs 0..63 / 64..87), values in (x0, x1)
__device__ __forceinline__ void frame_masks(const float* xbuf, int lane, float& x0, float& x1,
                                            unsigned long long& m0, unsigned long long& m1) {
  x0 = xbuf[lane];
  x1 = lane + 64 < LH ? xbuf[lane + 64] : 0.f;
  m0 = __ballot(x0 != 0.f);
  m1 = __ballot(x1 != 0.f);
}

// ZW = false: latent_dim <= 16, the head rides in the surplus lane groups of the encoder's last wave (4 barriers per
// frame).  ZW = true: latent_dim <= 32, every encoder lane group < 2L owns one head column (22 more registers), the
// head's outputs meet in LDS and L lanes draw z (5 barriers per frame).
template <int GATE, bool ZW>
__global__ __launch_bounds__(GN_NT) void vrnn_generate_kernel(GenArgs a) {
  constexpr int GN_LQ = (ZW ? GN_LWIDE : GN_LMAX) / PK;
  extern __shared__ __attribute__((aligned(16))) float Kxl[];            // encoder input kernel [88][352], then Wo [88][88]
  float* Wol = Kxl + LH * LG;
  __shared__ __attribute__((aligned(16))) float hb[2][2][PK * PKP];       // [chain][parity][sliced h]
  __shared__ __attribute__((aligned(16))) float zbuf[GN_LWIDE];
  __shared__ float zargs_l[2 * GN_LWIDE];
  __shared__ float xbuf[128];
  __shared__ float wbuf[GN_CMAX];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool enc = wave < GN_NW;
  const int rw = enc ? wave : wave - GN_NW;
  const int s = lane & 3;
  const int u_raw = rw * 16 + (lane >> 2);
  const int u = min(u_raw, LH - 1);
  const int L = a.L, T = a.S + a.nsteps;
  const int n = blockIdx.x;

  // ---- one-time staging ----------------------------------------------------------------------------------------
  {
    const int nv = LH * LG / 4;
    const float4* src = reinterpret_cast<const float4*>(a.Kx_enc);
    float4* dst = reinterpret_cast<float4*>(Kxl);
    for (int i0 = tid; i0 < nv; i0 += 4 * GN_NT) {
      float4 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = src[min(i0 + q * GN_NT, nv - 1)];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (i0 + q * GN_NT < nv) dst[i0 + q * GN_NT] = v[q];
    }
  }
  for (int i = tid; i < LH * LH; i += GN_NT) Wol[i] = a.Wo[i];
  for (int i = tid; i < 2 * 2 * PK * PKP; i += GN_NT) (&hb[0][0][0])[i] = 0.f;
  if (tid < GN_LWIDE) zbuf[tid] = 0.f;
  if (tid < 128) xbuf[tid] = (a.S > 0 && tid < LH) ? a.x_seed[((size_t)n * a.S) * LH + tid] : 0.f;
  if (tid < a.C) wbuf[tid] = a.w[(size_t)n * a.C + tid];
  __syncthreads();

  // recurrent kernel slice of this lane's unit (gate pairs), per-sequence bias W.K_w + b of its gate s
  const float* U = enc ? a.U_enc : a.U_dec;
  const int zj = u_raw - LH;                                 // encoder role: surplus group index
  const bool is_z = !ZW && enc && zj >= 0 && 2 * zj < L;     // group carries latents 2zj, 2zj+1
  const int lat = 2 * zj + (s & 1);
  const bool lat_ok = is_z && lat < L;
  auto zcol = [&](int g) { const int l = 2 * zj + (g & 1); return l < L ? (g >> 1) * L + l : -1; };
  f2 Ur[PKK][2];
#pragma unroll
  for (int kk = 0; kk < PKK; ++kk)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int cix = zcol(g);
      const float* src = is_z ? a.Wz + (size_t)(PKK * s + kk) * 2 * L + max(cix, 0)
                              : U + (size_t)(PKK * s + kk) * LG + g * LH + u;
      const float v = *src;
      Ur[kk][g >> 1][g & 1] = (is_z && cix < 0) ? 0.f : v;
    }
  float rb;
  {
    const float* Kw = enc ? a.Kw_enc : a.Kw_dec;
    float acc = (enc ? a.b_enc : a.b_dec)[s * LH + u];
    for (int c0 = 0; c0 < a.C; c0 += 8) {
      float kv[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) kv[q] = Kw[(size_t)min(c0 + q, a.C - 1) * LG + s * LH + u];
#pragma unroll
      for (int q = 0; q < 8; ++q) acc = fmaf(c0 + q < a.C ? wbuf[c0 + q] : 0.f, kv[q], acc);
    }
    rb = acc;
  }
  // encoder role also owns the output head: unit u = note u, 4 k-slices
  float bor = 0.f, bzr = 0.f;
  // role registers (one allocation for both roles: a wave uses only its own view): decoder: z rows of its input
  // kernel, lane s takes latents s, s+4, ... -> RR[4q+g]; encoder, ZW: this lane group's head column -> RR[kk]
  constexpr int GN_RR = ZW ? 32 : 16;
  float RR[GN_RR];
  if (enc) {
#pragma unroll
    for (int kk = 0; kk < GN_RR; ++kk) {
      const bool own = ZW && kk < PKK && u_raw < 2 * L;
      const float v = a.Wz[(size_t)(PKK * s + min(kk, PKK - 1)) * 2 * L + min(u_raw, 2 * L - 1)];
      RR[kk] = own ? v : 0.f;
    }
  } else {
#pragma unroll
    for (int q = 0; q < GN_LQ; ++q)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int l = s + PK * q;
        const float v = a.Kz[(size_t)min(l, L - 1) * LG + g * LH + u];
        RR[4 * q + g] = l < L ? v : 0.f;
      }
  }
  if (enc) {
    bor = a.bo[u];
    const float bzv = a.bz[max(zcol(s), 0)];
    bzr = lat_ok ? bzv : 0.f;
  }
  const int hslot = PKP * (u / PKK) + (u % PKK);
  const int zpos = lat_ok ? (lat % PK) * GN_LQ + lat / PK : 0;
  float c = 0.f;                       // cell state of this lane's unit (its role's LSTM)
  const bool writer = s == 0 && u_raw < LH;        // one lane per unit publishes

  // The noise of a frame does not depend on the data: it is drawn one phase (uniform) / one frame (normal) ahead, in
  // slots where its lanes would otherwise wait at a barrier.
  const bool zdraw = ZW ? (tid < L) : (lat_ok && s < 2);            // lanes that own a latent's eps
  const uint64_t zidx = (uint64_t)n * L + (ZW ? tid : lat);
  float e_cur = zdraw ? philox_normal_at(zidx, a.k0, a.k1, 0u, 0u) : 0.f;
  for (int t = 0; t < T; ++t) {
    const int cur = t & 1;
    float x0, x1;
    unsigned long long m0, m1;
    frame_masks(xbuf, lane, x0, x1, m0, m1);       // xbuf = input frame of step t (seed frame or last sample)
    float u_cur = 0.f;
    float seed_next = 0.f;                         // teacher forcing: next seed frame, requested a whole frame early
    if (enc && writer && t + 1 < a.S) seed_next = a.x_seed[((size_t)n * a.S + t + 1) * LH + u];
    // ---- phase 1: encoder cell (enc waves) | decoder input-kernel rows from L2 (dec waves) ------------------------
    float xd = 0.f;
    if (enc) {
      float xv = rb;
      while (m0) {
        const int k = __builtin_ctzll(m0);
        m0 &= m0 - 1;
        const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x0), k));
        xv = fmaf(v, Kxl[k * LG + s * LH + u], xv);
      }
      while (m1) {
        const int k = __builtin_ctzll(m1);
        m1 &= m1 - 1;
        const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x1), k));
        xv = fmaf(v, Kxl[(k + 64) * LG + s * LH + u], xv);
      }
      if (!is_z) {
        f2 acc2[2];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc2[g >> 1][g & 1] = (s == g) ? xv : 0.f;
        slice_matvec_lr(&hb[0][cur][PKP * s], Ur, acc2);
        float z[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) z[g] = reduce_slices<PK>(acc2[g >> 1][g & 1]);
        float h, gg;
        lstm_cell<GATE>(z, c, h, gg);
        hb[0][cur ^ 1][hslot] = h;
      }
    } else if (a.has_xp) {
      // rows of the decoder's input kernel for the notes that are on, 4 loads in flight per round: they travel from
      // L2 while the encoder cell and the latent head run
      const unsigned lane_off = (unsigned)(s * LH + u);
      while (m0 | m1) {
        unsigned ko[4];              // 32-bit offsets from the (uniform) kernel base: one register per pending load
        float vv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool lo = m0 != 0, any = (m0 | m1) != 0;
          const int bit = lo ? __builtin_ctzll(m0) : (m1 ? __builtin_ctzll(m1) : 0);
          const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lo ? x0 : x1), bit));
          if (lo) m0 &= m0 - 1; else if (m1) m1 &= m1 - 1;
          ko[q] = (unsigned)(any ? bit + (lo ? 0 : 64) : 0) * LG + lane_off;
          vv[q] = any ? v : 0.f;
        }
        float kv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) kv[q] = a.Kx_dec[ko[q]];
#pragma unroll
        for (int q = 0; q < 4; ++q) xd = fmaf(vv[q], kv[q], xd);
      }
    }
    step_barrier();
    // ---- phase 2: latent head + z ------------------------------------------------------------------------------------
    if (!ZW) {          // surplus lane groups of the encoder's last wave
      if (enc && wave == GN_NW - 1) {
        f2 acc2[2];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc2[g >> 1][g & 1] = (s == g) ? bzr : 0.f;
        slice_matvec_lr(&hb[0][cur ^ 1][PKP * s], Ur, acc2);
        float z[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) z[g] = reduce_slices<PK>(acc2[g >> 1][g & 1]);
        if (lat_ok && s < 2) {
          const float m = a.z_prior ? 0.f : ((s & 1) ? z[1] : z[0]);
          const float lv = a.z_prior ? 0.f : ((s & 1) ? z[3] : z[2]);
          zbuf[zpos] = fmaf(expf(0.5f * lv), e_cur, m);
        }
      }
      step_barrier();
    } else {            // one head column per encoder lane group, then L lanes draw z
      if (enc) {
        const float4* hp = reinterpret_cast<const float4*>(&hb[0][cur ^ 1][PKP * s]);
        float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
        for (int q = 0; q < PKP / 4; ++q) {
          const float4 v = hp[q];
          if (4 * q < PKK) acc0 = fmaf(v.x, RR[min(4 * q, GN_RR - 1)], acc0);
          if (4 * q + 1 < PKK) acc1 = fmaf(v.y, RR[min(4 * q + 1, GN_RR - 1)], acc1);
          if (4 * q + 2 < PKK) acc0 = fmaf(v.z, RR[min(4 * q + 2, GN_RR - 1)], acc0);
          if (4 * q + 3 < PKK) acc1 = fmaf(v.w, RR[min(4 * q + 3, GN_RR - 1)], acc1);
        }
        const float za = reduce_slices<PK>(acc0 + acc1);
        if (s == 0 && u_raw < 2 * L) zargs_l[u_raw] = za + a.bz[u_raw];
      }
      step_barrier();
      if (tid < L) {
        const float m = a.z_prior ? 0.f : zargs_l[tid], lv = a.z_prior ? 0.f : zargs_l[L + tid];
        zbuf[(tid % PK) * GN_LQ + tid / PK] = fmaf(expf(0.5f * lv), e_cur, m);
      }
      step_barrier();
    }
    // ---- phase 3: decoder cell ---------------------------------------------------------------------------------------
    if (!enc) {
      f2 acc2[2];
      const float xv = xd + rb;
#pragma unroll
      for (int g = 0; g < 4; ++g) acc2[g >> 1][g & 1] = (s == g) ? xv : 0.f;
      {
#pragma unroll
        for (int q4 = 0; q4 < GN_LQ / 4; ++q4) {      // 4 latents of this lane per LDS word
          const float4 zq = *reinterpret_cast<const float4*>(&zbuf[GN_LQ * s + 4 * q4]);
          const float zl[4] = {zq.x, zq.y, zq.z, zq.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int q = 4 * q4 + j;
            const f2 zz = {zl[j], zl[j]};
            const f2 k01 = {RR[4 * q], RR[4 * q + 1]}, k23 = {RR[4 * q + 2], RR[4 * q + 3]};
            acc2[0] = __builtin_elementwise_fma(zz, k01, acc2[0]);
            acc2[1] = __builtin_elementwise_fma(zz, k23, acc2[1]);
          }
        }
      }
      slice_matvec_lr(&hb[1][cur][PKP * s], Ur, acc2);
      float z[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) z[g] = reduce_slices<PK>(acc2[g >> 1][g & 1]);
      float h, gg;
      lstm_cell<GATE>(z, c, h, gg);
      hb[1][cur ^ 1][hslot] = h;
    } else {
      // encoder waves are idle here: draw this frame's Bernoulli uniforms and the next frame's latent noise
      if (writer) u_cur = philox_uniform_at((uint64_t)n * LH + u, a.k0, a.k1, 1u, (uint32_t)t);
      if (zdraw) e_cur = philox_normal_at(zidx, a.k0, a.k1, 0u, (uint32_t)(t + 1));
    }
    step_barrier();
    // ---- phase 4: output head, Bernoulli sample, next input frame (enc waves) ------------------------------------
    if (enc) {
      // logit_u = sum_k h_dec[k] Wo[k][u]: this lane's k-slice, Wo rows from LDS (consecutive u: conflict-free)
      const float4* hp = reinterpret_cast<const float4*>(&hb[1][cur ^ 1][PKP * s]);
      const float* wo = Wol + (PKK * s) * LH + u;
      float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
      for (int q = 0; q < PKP / 4; ++q) {
        const float4 v = hp[q];
        if (4 * q < PKK) acc0 = fmaf(v.x, wo[(4 * q) * LH], acc0);
        if (4 * q + 1 < PKK) acc1 = fmaf(v.y, wo[(4 * q + 1) * LH], acc1);
        if (4 * q + 2 < PKK) acc0 = fmaf(v.z, wo[(4 * q + 2) * LH], acc0);
        if (4 * q + 3 < PKK) acc1 = fmaf(v.w, wo[(4 * q + 3) * LH], acc1);
      }
      float acc = acc0 + acc1;
      acc = reduce_slices<PK>(acc);
      if (writer) {
        const float p = sigmoidf_(acc + bor);
        const float xs = u_cur <= p ? 1.f : 0.f;
        if (a.xhat) a.xhat[((size_t)n * T + t) * LH + u] = p;
        if (t >= a.S) a.Xs[((size_t)n * a.nsteps + (t - a.S)) * LH + u] = xs;
        // teacher forcing: the next input is the next seed frame while there is one
        xbuf[u] = (t + 1 < a.S) ? seed_next : xs;
      }
    }
    step_barrier();
  }
}

}  // namespace clv

extern "C" int clv_vrnn_generate_supported(int D, int H, int L, int C) {
  return D == clv::LH && H == clv::LH && L >= 1 && L <= clv::GN_LWIDE && C >= 1 && C <= clv::GN_CMAX;
}

extern "C" int clv_vrnn_generate(int N, int S, int nsteps, int D, int H, int L, int C, int gate_act, int z_prior,
                                 uint64_t seed, const float* x_seed, const float* w,
                                 const float* Kx_enc, const float* Kw_enc, const float* b_enc, const float* U_enc,
                                 const float* Wz, const float* bz,
                                 const float* Kx_dec, const float* Kz, const float* Kw_dec, const float* b_dec,
                                 const float* U_dec, const float* Wo, const float* bo,
                                 float* Xs, float* xhat, void* stream) {
  using namespace clv;
  if (!clv_vrnn_generate_supported(D, H, L, C) || N <= 0 || S < 0 || nsteps < 0 || S + nsteps <= 0) return CLV_EINVAL;
  if (gate_act != CLV_GATE_HARD_SIGMOID && gate_act != CLV_GATE_SIGMOID) return CLV_EINVAL;
  if ((S > 0 && !x_seed) || !w || !Kx_enc || !Kw_enc || !b_enc || !U_enc || !Wz || !bz || !Kz || !Kw_dec || !b_dec ||
      !U_dec || !Wo || !bo || (nsteps > 0 && !Xs))
    return CLV_EINVAL;
  if (((uintptr_t)Kx_enc) % 16 != 0) return CLV_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  GenArgs a{N, S, nsteps, L, C, z_prior, Kx_dec != nullptr, (uint32_t)seed, (uint32_t)(seed >> 32), x_seed, w,
            Kx_enc, Kw_enc, b_enc, U_enc, Wz, bz, Kx_dec, Kz, Kw_dec, b_dec, U_dec, Wo, bo, Xs, xhat};
  const size_t lds = (size_t)(LH * LG + LH * LH) * sizeof(float);
  const bool hard = gate_act == CLV_GATE_HARD_SIGMOID, wide = L > GN_LMAX;
  void (*kern)(GenArgs) = hard ? (wide ? vrnn_generate_kernel<CLV_GATE_HARD_SIGMOID, true> : vrnn_generate_kernel<CLV_GATE_HARD_SIGMOID, false>)
                               : (wide ? vrnn_generate_kernel<CLV_GATE_SIGMOID, true> : vrnn_generate_kernel<CLV_GATE_SIGMOID, false>);
  if (int e = allow_dynamic_lds(reinterpret_cast<const void*>(kern), 156 * 1024)) return e;
  ProfScope p("vrnn_generate", s);
  hipLaunchKernelGGL(kern, dim3(N), dim3(GN_NT), lds, s, a);
  return launch_status();
}
