// vae_generate.hip -- cl_vae autoregressive generation as ONE persistent kernel per batch of sequences (gfx950).
//
// cl_vae/model.py:9-42 (generate_sample) produces a sequence frame by frame:
//     z_mean, z_log_var = z_encoder([x_prev, w])          h = relu([x_prev | w] . K_h + b_h); zargs = h . K_z + b_z
//     z ~ N(z_mean, exp(z_log_var))  (or N(0, 1) under use_z_prior)
//     x_hat = decoder([w, z, x_prev_t])                   h_d = relu([w | x_prev_t | z] . K_d + b_d); sigmoid(h_d . K_o + b_o)
//     x_t ~ Bernoulli(x_hat);  x_prev_t := x_prev;  x_prev := x_t          (the decoder's history lags one frame)
// Every frame depends on the previous sample, so the per-frame launches of the layer chain (a hipGraph replay of ~12
// kernels, ~90 us) are pure latency.  Here a workgroup owns a sequence for its whole length: the frame rows of the two
// hidden kernels sit in LDS (a frame is a handful of notes: its product is a gather of kernel rows), the head and output
// kernels in registers as 4-lane k-slices like the LSTM kernels', the label's contribution to both hidden layers is
// formed once, the noise comes from Philox in place (the values clv_philox_normal / _uniform give for (seed, frame,
// stream, index)), and a frame is four LDS barriers:
//     hidden layer of the z-encoder | latent head + z | hidden layer of the decoder | output layer + Bernoulli sample.
#include "lstm_common.h"
#include "philox.h"

namespace clv {

constexpr int VG_NT = 384;               // 96 output slots x 4 k-slices
constexpr int VG_LMAX = 32;              // latent dims
constexpr int VG_CMAX = 32;

struct VaeGenArgs {
  int N, nsteps, L, C, z_prior, has_xp;
  uint32_t k0, k1;              // Philox key (seed)
  const float* x_seed;          // [N,88]
  const float* w;               // [N,C]
  const float* Kh;              // h/kernel [88 + C, 88]: frame rows, then label rows
  const float* bh;              // [88]
  const float* Kz;              // zargs/kernel [88, 2L] = [z_mean | z_log_var]
  const float* bz;              // [2L]
  const float* Kd;              // decoder_h/kernel [C + (88) + L, 88]: label rows, history rows (has_xp), latent rows
  const float* bd;              // [88]
  const float* Ko;              // x_decoded_mean/kernel [88, 88]
  const float* bo;              // [88]
  float* Xs;                    // [N,nsteps,88]
  float* xhat;                  // [N,nsteps,88] or null
};

// sum over the notes that are on (two scalar masks: inputs 0..63 / 64..87) of row n of an LDS-resident [88][88] kernel,
// column j: every lane walks the same notes (scalar loop), all loads are issued before the first add
__device__ __forceinline__ float gather_rows(const float* Kl, int j, unsigned long long m0, unsigned long long m1) {
  float acc0 = 0.f, acc1 = 0.f;
  while (m0) {
    const int n0 = __builtin_ctzll(m0);
    m0 &= m0 - 1;
    float v1 = 0.f;
    if (m0) { const int n1 = __builtin_ctzll(m0); m0 &= m0 - 1; v1 = Kl[n1 * LH + j]; }
    acc0 += Kl[n0 * LH + j];
    acc1 += v1;
  }
  while (m1) {
    const int n0 = 64 + __builtin_ctzll(m1);
    m1 &= m1 - 1;
    acc0 += Kl[n0 * LH + j];
  }
  return acc0 + acc1;
}

__global__ __launch_bounds__(VG_NT) void vae_generate_kernel(VaeGenArgs a) {
  extern __shared__ __attribute__((aligned(16))) float vg_lds[];
  float* Khl = vg_lds;                        // [88][88] frame rows of the z-encoder's hidden kernel
  float* Kdl = Khl + LH * LH;                 // [88][88] history rows of the decoder's hidden kernel (has_xp)
  float* Kdz = Kdl + LH * LH;                 // [VG_LMAX][88] latent rows of the decoder's hidden kernel
  __shared__ __attribute__((aligned(16))) float hbuf[2][PK * PKP];      // sliced hidden vectors: z-encoder's, decoder's
  __shared__ float zbuf[VG_LMAX];
  __shared__ float xbuf[2][128];              // [parity]: the frame sampled last (0/1 per note)
  __shared__ float wbuf[VG_CMAX];
  const int tid = threadIdx.x, lane = tid & 63;
  const int s = tid & 3, o_raw = tid >> 2, o = min(o_raw, LH - 1);       // output slot (unit / note) and k-slice
  const int L = a.L, n = blockIdx.x;
  const bool writer = s == 0 && o_raw < LH;
  const int hslot = PKP * (o / PKK) + (o % PKK);

  // ---- one-time staging ---------------------------------------------------------------------------------------------
  for (int i = tid; i < LH * LH; i += VG_NT) {
    Khl[i] = a.Kh[i];
    Kdl[i] = a.has_xp ? a.Kd[(size_t)a.C * LH + i] : 0.f;
  }
  for (int i = tid; i < VG_LMAX * LH; i += VG_NT)
    Kdz[i] = i < L * LH ? a.Kd[(size_t)(a.C + (a.has_xp ? LH : 0)) * LH + i] : 0.f;
  for (int i = tid; i < 2 * PK * PKP; i += VG_NT) (&hbuf[0][0])[i] = 0.f;
  if (tid < VG_LMAX) zbuf[tid] = 0.f;
  if (tid < 128) { xbuf[0][tid] = tid < LH ? a.x_seed[(size_t)n * LH + tid] : 0.f; xbuf[1][tid] = xbuf[0][tid]; }
  if (tid < VG_CMAX) wbuf[tid] = tid < a.C ? a.w[(size_t)n * a.C + tid] : 0.f;
  // head kernel: slot c < 2L owns column c; the pairs (mean_l, log_var_l) sit in neighbouring slots 2l, 2l+1 so that the
  // log-variance reaches the mean's lanes by one DPP row shift (the kernel's own column order is [means | log-variances])
  const int zc = (o_raw & 1) * L + (o_raw >> 1);          // head column of this slot
  const bool zslot = o_raw < 2 * L;
  float Kzr[PKK], Kor[PKK];
#pragma unroll
  for (int kk = 0; kk < PKK; ++kk) {
    Kzr[kk] = zslot ? a.Kz[(size_t)(PKK * s + kk) * 2 * L + zc] : 0.f;
    Kor[kk] = a.Ko[(size_t)(PKK * s + kk) * LH + o];
  }
  const float bzr = zslot ? a.bz[zc] : 0.f, bor = a.bo[o];
  __syncthreads();
  // the label's share of both hidden layers (+ bias): constant over the sequence
  float ch = a.bh[o], cd = a.bd[o];
  for (int c = 0; c < a.C; ++c) {
    ch = fmaf(wbuf[c], a.Kh[(size_t)(LH + c) * LH + o], ch);
    cd = fmaf(wbuf[c], a.Kd[(size_t)c * LH + o], cd);
  }
  // notes of the current input frame and of the one before it (the decoder's history lags: cl_vae/model.py:38-40)
  unsigned long long cur0, cur1, his0, his1;
  {
    const float x0 = xbuf[0][lane], x1 = lane + 64 < LH ? xbuf[0][lane + 64] : 0.f;
    cur0 = __ballot(x0 != 0.f); cur1 = __ballot(x1 != 0.f);
    his0 = cur0; his1 = cur1;
  }

  for (int t = 0; t < a.nsteps; ++t) {
    // this frame's noise, drawn before anything depends on it
    const float u_cur = writer ? philox_uniform_at((uint64_t)n * LH + o, a.k0, a.k1, 1u, (uint32_t)t) : 2.f;
    const bool zdraw = s == 0 && zslot && !(o_raw & 1);                 // the mean slot of latent l = o_raw / 2
    const float eps = zdraw ? philox_normal_at((uint64_t)n * L + (o_raw >> 1), a.k0, a.k1, 0u, (uint32_t)t) : 0.f;
    // 1. z-encoder hidden layer: relu(x_prev . K_h[frame rows] + (w . K_h[label rows] + b_h))
    {
      const float h = fmaxf(ch + gather_rows(Khl, o, cur0, cur1), 0.f);
      if (writer) hbuf[0][hslot] = h;
    }
    step_barrier();
    // 2. latent head + sample
    {
      float hv[PKP];
      load_hslice(&hbuf[0][PKP * s], hv);
      float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
      for (int kk = 0; kk < PKK; kk += 2) { acc0 = fmaf(hv[kk], Kzr[kk], acc0); acc1 = fmaf(hv[kk + 1], Kzr[kk + 1], acc1); }
      const float za = reduce_slices<PK>(acc0 + acc1) + bzr;          // mean (even slots) / log-variance (odd slots)
      const float lv = dpp_mov<0x104>(za);                             // row_shl:4: the next slot's value
      if (zdraw) {
        const float mean = a.z_prior ? 0.f : za, lvv = a.z_prior ? 0.f : lv;
        zbuf[o_raw >> 1] = fmaf(__expf(0.5f * lvv), eps, mean);
      }
    }
    step_barrier();
    // 3. decoder hidden layer: relu(w . K_d[label rows] + b_d + x_prev_t . K_d[history rows] + z . K_d[latent rows])
    {
      float acc = cd + (a.has_xp ? gather_rows(Kdl, o, his0, his1) : 0.f);
      float zacc = 0.f;
      for (int l = s; l < L; l += PK) zacc = fmaf(zbuf[l], Kdz[l * LH + o], zacc);      // the 4 lanes of a slot share the latents
      acc += reduce_slices<PK>(zacc);
      if (writer) hbuf[1][hslot] = fmaxf(acc, 0.f);
    }
    step_barrier();
    // 4. output layer, Bernoulli sample
    {
      float hv[PKP];
      load_hslice(&hbuf[1][PKP * s], hv);
      float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
      for (int kk = 0; kk < PKK; kk += 2) { acc0 = fmaf(hv[kk], Kor[kk], acc0); acc1 = fmaf(hv[kk + 1], Kor[kk + 1], acc1); }
      const float lg = reduce_slices<PK>(acc0 + acc1) + bor;
      if (writer) {
        const float p = sigmoidf_(lg);
        const float xs = u_cur <= p ? 1.f : 0.f;
        if (a.xhat) a.xhat[((size_t)n * a.nsteps + t) * LH + o] = p;
        a.Xs[((size_t)n * a.nsteps + t) * LH + o] = xs;
        xbuf[(t + 1) & 1][o] = xs;
      }
    }
    step_barrier();
    {
      const float x0 = xbuf[(t + 1) & 1][lane], x1 = lane + 64 < LH ? xbuf[(t + 1) & 1][lane + 64] : 0.f;
      his0 = cur0; his1 = cur1;
      cur0 = __ballot(x0 != 0.f); cur1 = __ballot(x1 != 0.f);
    }
  }
}

}  // namespace clv

extern "C" int clv_vae_generate_supported(int D, int H, int L, int C) {
  return D == clv::LH && H == clv::LH && L >= 1 && L <= clv::VG_LMAX && C >= 1 && C <= clv::VG_CMAX;
}

extern "C" int clv_vae_generate(int N, int nsteps, int D, int H, int L, int C, int use_x_prev, int z_prior, uint64_t seed,
                                const float* x_seed, const float* w, const float* Kh, const float* bh, const float* Kz,
                                const float* bz, const float* Kd, const float* bd, const float* Ko, const float* bo,
                                float* Xs, float* xhat, void* stream) {
  using namespace clv;
  if (!clv_vae_generate_supported(D, H, L, C) || N <= 0 || nsteps <= 0) return CLV_EINVAL;
  if (!x_seed || !w || !Kh || !bh || !Kz || !bz || !Kd || !bd || !Ko || !bo || !Xs) return CLV_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  VaeGenArgs a{N, nsteps, L, C, z_prior, use_x_prev != 0, (uint32_t)seed, (uint32_t)(seed >> 32), x_seed, w, Kh, bh, Kz, bz, Kd, bd,
               Ko, bo, Xs, xhat};
  const size_t lds = (size_t)(2 * LH * LH + VG_LMAX * LH) * sizeof(float);
  if (int e = allow_dynamic_lds(reinterpret_cast<const void*>(vae_generate_kernel), 96 * 1024)) return e;
  ProfScope p("vae_generate", s);
  hipLaunchKernelGGL(vae_generate_kernel, dim3(N), dim3(VG_NT), lds, s, a);
  return launch_status();
}
