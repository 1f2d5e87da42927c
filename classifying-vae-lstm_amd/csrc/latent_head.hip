// latent_head.hip -- the latent head of cl_vrnn where the pair kernels do not carry it (latent_dim 9..32, or the
// large-batch sequence kernels): forward and backward, one launch each (gfx950).
//
// Reference: Zargs = TimeDistributed(Dense(2*latent_dim)) on the encoder states and the reparametrised sample
// (cl_vrnn/model.py:200-216), KL_z (cl_vrnn/model.py:243) and their gradients under K.gradients:
//   zargs = hs.Wz + bz  [R, 2L] = (mean | log_var),   z = mean + exp(log_var / 2) * eps,
//   rowkl = -0.5 sum_l (1 + log_var - mean^2 - exp(log_var))
//   dzargs = (dZ + kl * mean | dZ * eps * sd / 2 - kl * (1 - sd^2) / 2),   dhs = dzargs.Wz^T,
//   dWz = hs^T.dzargs,  dbz = sum_r dzargs
// As separate launches this was a K = 88 GEMM, an elementwise kernel, another elementwise kernel, an NT GEMM and a grouped TN
// GEMM: 260 us per step at configuration 5 (R = 262144 rows, L = 32), the GEMMs on the fp32 vector pipe at a third of its
// peak, zargs / dzargs written and read back twice.  Here (the structure of out_head.hip) a workgroup keeps Wz in LDS,
// takes 128 rows of hs (8 waves x one 16-row tile of v_mfma_f32_16x16x4_f32) and the elementwise math runs on the
// accumulators' C/D layout; the backward pass never stores dzargs unless asked to, hands it to the two products through
// LDS and leaves dWz / dbz as one [89, 2L] slab per workgroup for the deferred split-K reduce (fixed order: reproducible).
//
// Head columns are PADDED per half to a multiple of 16 (LP16 tiles each): column c of the mean sits at c, of log_var at
// 16*LP16 + c, so that a lane of the C/D layout holds both values of a latent.
#include "common.h"
#include "philox.h"
#include "reduce_job.h"

namespace clv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int ZH = 88;             // hidden units of the encoder LSTM
constexpr int ZH_T = 6;            // 16-wide tiles covering 88 (96)
constexpr int ZH_KS = ZH / 4;      // k-steps of a K = 88 product
constexpr int ZH_LD = 116;         // LDS row stride of the hs tiles and (backward) of Wz: r*116 + q hits 64 different banks
constexpr int ZD_LD = 84;          // ... of the dzargs tiles (<= 64 padded columns): r*84 + q likewise (84 % 64 = 20)
constexpr int ZH_NW = 8;           // waves per workgroup, one 16-row tile each
constexpr int ZH_RB = 16 * ZH_NW;  // rows per block
constexpr int ZH_TILE = 16 * ZH_LD;
constexpr int ZD_TILE = 16 * ZD_LD;
constexpr int ZH_SLAB_ROWS = ZH + 1;      // dWz rows + the dbz row

struct LatentFwdArgs {
  int R, L, ldz;
  const float* hs;        // [R,88]
  const float* Wz;        // [88,2L]
  const float* bz;        // [2L]
  float* eps;             // [R,L]: read, or drawn here and written (noise.on; the backward pass reads it)
  float* zargs;           // [R,2L]
  float* Z;               // [R] rows of stride ldz
  float* rowkl;           // [R] or null
  struct { int on; uint32_t k0, k1, stream, step; uint64_t first; const int32_t* step_dev; } noise;
};

// eps of a wave's 16-row tile drawn in the accumulators' C/D layout (lane (q, r), register reg: row 4 q + reg, column
// 16 jm + r): element (row, c) is the clv_philox_normal value at index first + row * L + c, bit for bit.
// A Philox counter yields the four normals of indices 4 k .. 4 k + 3: when L and `first` are multiples of 4 those are four
// neighbouring columns of ONE row, i.e. what the four lanes of a DPP quad hold in one register -- so lane i of a quad draws
// the counter of row 4 q + i (one Philox call instead of four per lane and column tile) and a 4 x 4 transpose inside the
// quad (two DPP exchange rounds) hands every lane its four rows.  Any other L: one call per element.
template <int CTRL>
__device__ __forceinline__ float zh_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int LP16>
__device__ __forceinline__ void zh_draw_eps(const LatentFwdArgs& a, uint32_t stp, int row0, int q, int r, float (&ev)[LP16][4]) {
  const int L = a.L;
  if ((L & 3) == 0 && (a.noise.first & 3) == 0) {
    const int i = r & 3;
    const uint64_t rowi = (uint64_t)min(row0 + 4 * q + i, a.R - 1);
#pragma unroll
    for (int jm = 0; jm < LP16; ++jm) {
      const int cg = min(16 * jm + (r & ~3), L - 4);
      const uint64_t ctr = (a.noise.first + rowi * (uint64_t)L + (uint64_t)cg) >> 2;
      uint32_t w[4];
      philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), a.noise.stream, stp, a.noise.k0, a.noise.k1, w);
      const float rad0 = sqrtf(-2.f * logf(u01(w[0]))), th0 = 6.283185307179586f * u01(w[1]);
      const float rad1 = sqrtf(-2.f * logf(u01(w[2]))), th1 = 6.283185307179586f * u01(w[3]);
      float n0 = rad0 * cosf(th0), n1 = rad0 * sinf(th0), n2 = rad1 * cosf(th1), n3 = rad1 * sinf(th1);
      // transpose: lane i element k  <->  lane k element i
      const bool o1 = i & 1, o2 = i & 2;
      const float s01 = zh_dpp<0xB1>(o1 ? n0 : n1), s23 = zh_dpp<0xB1>(o1 ? n2 : n3);      // quad_perm [1,0,3,2]
      if (o1) { n0 = s01; n2 = s23; } else { n1 = s01; n3 = s23; }
      const float s02 = zh_dpp<0x4E>(o2 ? n0 : n2), s13 = zh_dpp<0x4E>(o2 ? n1 : n3);      // quad_perm [2,3,0,1]
      if (o2) { n0 = s02; n1 = s13; } else { n2 = s02; n3 = s13; }
      ev[jm][0] = n0; ev[jm][1] = n1; ev[jm][2] = n2; ev[jm][3] = n3;
    }
    return;
  }
#pragma unroll
  for (int jm = 0; jm < LP16; ++jm)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const uint64_t e = (uint64_t)min(row0 + 4 * q + reg, a.R - 1) * (uint64_t)L + (uint64_t)min(16 * jm + r, L - 1);
      ev[jm][reg] = philox_normal_at(a.noise.first + e, a.noise.k0, a.noise.k1, a.noise.stream, stp);
    }
}

// this wave's 16 rows of hs: one contiguous 5.6 KB piece of HBM, 6 float4 per lane -> its LDS tile [16][ZH_LD]
__device__ __forceinline__ void zh_load_hs(const float* hs, int row0, int R, int lane, float4 (&hv)[6]) {
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int e = min(lane + 64 * i, 16 * (ZH / 4) - 1);
    const int rr = e / (ZH / 4), c4 = e - rr * (ZH / 4);
    hv[i] = *reinterpret_cast<const float4*>(hs + (size_t)min(row0 + rr, R - 1) * ZH + 4 * c4);
  }
}
__device__ __forceinline__ void zh_store_hs(float* myhs, int row0, int R, int lane, const float4 (&hv)[6], bool ones) {
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int e = lane + 64 * i;
    const int rr = e / (ZH / 4), c4 = e - rr * (ZH / 4);
    const float mk = row0 + rr < R ? 1.f : 0.f;
    if (e < 16 * (ZH / 4))
      *reinterpret_cast<float4*>(myhs + rr * ZH_LD + 4 * c4) = make_float4(hv[i].x * mk, hv[i].y * mk, hv[i].z * mk, hv[i].w * mk);
  }
  if (ones)       // columns 88..95 of the tile: the ones column (bias gradient) and zeros
    for (int e = lane; e < 16 * 8; e += 64) {
      const int rr = e >> 3, c = e & 7;
      myhs[rr * ZH_LD + ZH + c] = (c == 0 && row0 + rr < R) ? 1.f : 0.f;
    }
}

typedef __amdgpu_buffer_rsrc_t zh_rsrc_t;
constexpr unsigned ZH_OOB = 0x80000000u;
__device__ __forceinline__ zh_rsrc_t zh_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void zh_bstore(float v, zh_rsrc_t r, unsigned voff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, 0, 0);
}

template <int LP16>
__global__ __launch_bounds__(64 * ZH_NW) void latent_head_fwd_kernel(LatentFwdArgs a) {
  constexpr int NTZ = 2 * LP16, NP = 16 * NTZ;
  constexpr int WLD = NP + 16;                         // 48 / 80: the rows 4s + q of a k-step start 16 banks apart
  extern __shared__ __attribute__((aligned(16))) float zh_lds[];
  float* WzL = zh_lds;                                 // [88][WLD], padded head columns
  float* hsT = zh_lds + ZH * WLD;                      // [8][16][ZH_LD]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4, L = a.L;
  for (int e = tid; e < ZH * NP; e += 64 * ZH_NW) {
    const int k = e / NP, cp = e - k * NP;
    const bool half = cp >= 16 * LP16;
    const int c = cp - (half ? 16 * LP16 : 0);
    const float v = a.Wz[(size_t)k * 2 * L + (half ? L : 0) + min(c, L - 1)];
    WzL[k * WLD + cp] = c < L ? v : 0.f;
  }
  float bm[LP16], bl[LP16];
#pragma unroll
  for (int jm = 0; jm < LP16; ++jm) {
    const int c = min(16 * jm + r, L - 1);
    bm[jm] = a.bz[c]; bl[jm] = a.bz[L + c];
  }
  __syncthreads();
  float* myhs = hsT + wave * ZH_TILE;

  // A block's global inputs are requested one block ahead (a workgroup owns up to 8 blocks at configuration 5; with the
  // loads at the top of each block the kernel was a chain of exposed HBM round trips: 72 us for 3 GFLOP)
  float4 hv[6];
  float ev[LP16][4];                                   // eps of this lane's outputs (C/D layout: rows 4q + reg, column 16 jm + r)
  const bool draw = a.noise.on != 0;                   // (uniform)
  const uint32_t stp = a.noise.step + ((draw && a.noise.step_dev) ? (uint32_t)*a.noise.step_dev : 0u);
  auto fetch = [&](int blk) {                          // (rows beyond R: clamped addresses, masked where they are used)
    const int row0 = blk * ZH_RB + wave * 16;
    zh_load_hs(a.hs, row0, a.R, lane, hv);
    if (draw) return;
#pragma unroll
    for (int jm = 0; jm < LP16; ++jm)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
        ev[jm][reg] = a.eps[(size_t)min(row0 + 4 * q + reg, a.R - 1) * L + min(16 * jm + r, L - 1)];
  };
  fetch(blockIdx.x);
  for (int blk = blockIdx.x; blk * ZH_RB < a.R; blk += gridDim.x) {
    const int row0 = blk * ZH_RB + wave * 16;
    zh_store_hs(myhs, row0, a.R, lane, hv, false);
    float ec[LP16][4];
    if (draw) {        // the next block's loads go out first: the draw's VALU work (two Philox calls per lane) runs under them
      fetch(blk + gridDim.x);
      zh_draw_eps<LP16>(a, stp, row0, q, r, ec);
    } else {
#pragma unroll
      for (int jm = 0; jm < LP16; ++jm)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) ec[jm][reg] = ev[jm][reg];
      fetch(blk + gridDim.x);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the tile is wave-private: no barrier
    // this block's valid rows of every output as buffer resources (an absent output: zero records, its stores are dropped)
    const size_t rb0 = (size_t)blk * ZH_RB;
    const unsigned nrow = (unsigned)min(ZH_RB, a.R - blk * ZH_RB);
    const zh_rsrc_t r_za = zh_rsrc(a.zargs + rb0 * 2 * L, nrow * 2u * L * 4u);
    const zh_rsrc_t r_z = zh_rsrc(a.Z + rb0 * a.ldz, nrow * (unsigned)a.ldz * 4u);      // (columns beyond L of a row are other data: masked per lane)
    const zh_rsrc_t r_e = zh_rsrc(a.eps + rb0 * L, draw ? nrow * L * 4u : 0u);
    const zh_rsrc_t r_kl = zh_rsrc(a.rowkl ? a.rowkl + rb0 : a.zargs, a.rowkl ? nrow * 4u : 0u);

    f32x4 acc[NTZ];
#pragma unroll
    for (int j = 0; j < NTZ; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {   // operands of k-step s+1 are read while the MFMAs of step s issue (see out_head.hip)
      float av = myhs[r * ZH_LD + q], bv[NTZ];
#pragma unroll
      for (int j = 0; j < NTZ; ++j) bv[j] = WzL[q * WLD + 16 * j + r];
#pragma unroll
      for (int s = 0; s < ZH_KS; ++s) {
        float an = 0.f, bn[NTZ];
        if (s + 1 < ZH_KS) {
          an = myhs[r * ZH_LD + 4 * (s + 1) + q];
#pragma unroll
          for (int j = 0; j < NTZ; ++j) bn[j] = WzL[(4 * (s + 1) + q) * WLD + 16 * j + r];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NTZ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[j], acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < ZH_KS) {
          av = an;
#pragma unroll
          for (int j = 0; j < NTZ; ++j) bv[j] = bn[j];
        }
      }
    }
    // reparametrised sample and KL term (the arithmetic of gauss_fwd_kernel, csrc/pointwise.hip)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int row = row0 + 4 * q + reg;
      const bool rok = row < a.R;
      float ssum = 0.f;
#pragma unroll
      for (int jm = 0; jm < LP16; ++jm) {
        const int c = 16 * jm + r;
        const bool ok = rok && c < L;
        const float m = acc[jm][reg] + bm[jm], lv = acc[LP16 + jm][reg] + bl[jm];
        const float sd = expf(0.5f * lv);
        const float z = m + sd * ec[jm][reg];
        ssum += ok ? 1.f + lv - m * m - sd * sd : 0.f;
        // Buffer stores over this block's valid rows; a lane without a column carries an out-of-range offset.  No branch
        // around a store: the next block's hs rows were requested BEFORE these stores, and the wait in front of their use
        // must be able to count the stores behind them -- behind `if`s it was vmcnt(0), i.e. every block ended by waiting
        // for its own stores to be acknowledged (round 6, PERFLOG R6.14).
        const unsigned rb = (unsigned)(wave * 16 + 4 * q + reg);
        const unsigned oc = c < L ? 0u : ZH_OOB;
        zh_bstore(m, r_za, (rb * 2u * L + c) * 4u | oc);
        zh_bstore(lv, r_za, (rb * 2u * L + L + c) * 4u | oc);
        zh_bstore(z, r_z, (rb * (unsigned)a.ldz + c) * 4u | oc);
        zh_bstore(ec[jm][reg], r_e, (rb * L + c) * 4u | oc);
      }
      ssum += __shfl_xor(ssum, 8, 64);
      ssum += __shfl_xor(ssum, 4, 64);
      ssum += __shfl_xor(ssum, 2, 64);
      ssum += __shfl_xor(ssum, 1, 64);
      zh_bstore(-0.5f * ssum, r_kl, (unsigned)(wave * 16 + 4 * q + reg) * 4u | (r == 0 ? 0u : ZH_OOB));
    }
  }
}

struct LatentBwdArgs {
  int R, L, lddz;
  float kl_scale;
  const float* hs;        // [R,88]
  const float* Wz;        // [88,2L]
  const float* zargs;     // [R,2L]
  const float* eps;       // [R,L]
  const float* dZ;        // [R] rows of stride lddz
  float* dzargs;          // [R,2L] or null (not stored)
  float* dhs;             // [R,88]
  float* partial;         // [gridDim.x][89][2L]
};

template <int LP16>
__global__ __launch_bounds__(64 * ZH_NW) void latent_head_bwd_kernel(LatentBwdArgs a) {
  constexpr int NTZ = 2 * LP16, NP = 16 * NTZ;
  constexpr int NT3 = ZH_T * NTZ;                      // weight-gradient tiles (12 or 24)
  extern __shared__ __attribute__((aligned(16))) float zh_lds[];
  float* WzL = zh_lds;                                 // [88][ZH_LD], padded head columns (read transposed: B[k = column][n = unit])
  float* hsT = zh_lds + ZH * ZH_LD;                    // [8][16][ZH_LD]; column 88 = 1 for live rows (the dbz row of hs^T)
  float* dzT = hsT + ZH_NW * ZH_TILE;                  // [8][16][ZD_LD]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4, L = a.L;
  for (int e = tid; e < ZH * NP; e += 64 * ZH_NW) {
    const int k = e / NP, cp = e - k * NP;
    const bool half = cp >= 16 * LP16;
    const int c = cp - (half ? 16 * LP16 : 0);
    const float v = a.Wz[(size_t)k * 2 * L + (half ? L : 0) + min(c, L - 1)];
    WzL[k * ZH_LD + cp] = c < L ? v : 0.f;
  }
  // The weight gradient is WAVE-PRIVATE: every wave accumulates all ZH_T x NTZ tiles of [dWz ; dbz] over its own 16 rows
  // of every block (4 k-steps per block) and the eight partial results meet once, at the end.  With the tiles dealt to the
  // waves instead (each over all 128 rows of a block, as out_head.hip does) a block needs two workgroup barriers, the eight
  // waves move in lockstep and nothing hides a wave's LDS / HBM latencies: 162 us per launch at configuration 5 against
  // 42 us of MFMA time.
  f32x4 acc3[ZH_T][NTZ];
#pragma unroll
  for (int i = 0; i < ZH_T; ++i)
#pragma unroll
    for (int j = 0; j < NTZ; ++j) acc3[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float* myhs = hsT + wave * ZH_TILE;
  float* mydz = dzT + wave * ZD_TILE;
  __syncthreads();

  // the block's hs rows are requested one block ahead (see the forward kernel); the per-latent values at its top: with 96
  // accumulator registers there is no room to hold both across the products, and the waves are not in lockstep here
  float4 hv[6];
  zh_load_hs(a.hs, blockIdx.x * ZH_RB + wave * 16, a.R, lane, hv);
  for (int blk = blockIdx.x; blk * ZH_RB < a.R; blk += gridDim.x) {
    const int row0 = blk * ZH_RB + wave * 16;
    float mv[LP16][4], lvv[LP16][4], ev[LP16][4], dv[LP16][4];
#pragma unroll
    for (int jm = 0; jm < LP16; ++jm)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const size_t row = (size_t)min(row0 + 4 * q + reg, a.R - 1);
        const int c = min(16 * jm + r, L - 1);
        mv[jm][reg] = a.zargs[row * 2 * L + c];
        lvv[jm][reg] = a.zargs[row * 2 * L + L + c];
        ev[jm][reg] = a.eps[row * L + c];
        dv[jm][reg] = a.dZ[row * a.lddz + c];
      }
    // dzargs of this lane's (row, latent) pairs (the arithmetic of gauss_bwd_kernel, csrc/pointwise.hip) -> LDS tile
    zh_store_hs(myhs, row0, a.R, lane, hv, true);
#pragma unroll
    for (int jm = 0; jm < LP16; ++jm)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int row = row0 + 4 * q + reg, c = 16 * jm + r;
        const bool ok = row < a.R && c < L;
        const float m = mv[jm][reg], sd = expf(0.5f * lvv[jm][reg]), d = dv[jm][reg];
        const float dm = ok ? d + a.kl_scale * m : 0.f;
        const float dl = ok ? d * ev[jm][reg] * 0.5f * sd - 0.5f * a.kl_scale * (1.f - sd * sd) : 0.f;
        mydz[(4 * q + reg) * ZD_LD + c] = dm;
        mydz[(4 * q + reg) * ZD_LD + 16 * LP16 + c] = dl;
        if (ok && a.dzargs) {
          a.dzargs[(size_t)row * 2 * L + c] = dm;
          a.dzargs[(size_t)row * 2 * L + L + c] = dl;
        }
      }
    zh_load_hs(a.hs, (blk + gridDim.x) * ZH_RB + wave * 16, a.R, lane, hv);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // both tiles are wave-private: no barrier

    // ---- dhs = dzargs.Wz^T   (k = padded head column, n = hidden unit: B[k][n] = Wz[n][k])
    {
      f32x4 acc[ZH_T];
#pragma unroll
      for (int j = 0; j < ZH_T; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      int wrow[ZH_T];
#pragma unroll
      for (int j = 0; j < ZH_T; ++j) wrow[j] = min(16 * j + r, ZH - 1) * ZH_LD + q;      // units 88..95: repeat row 87 (never stored)
      float av = mydz[r * ZD_LD + q], bv[ZH_T];
#pragma unroll
      for (int j = 0; j < ZH_T; ++j) bv[j] = WzL[wrow[j]];
#pragma unroll
      for (int s = 0; s < NP / 4; ++s) {
        float an = 0.f, bn[ZH_T];
        if (s + 1 < NP / 4) {
          an = mydz[r * ZD_LD + 4 * (s + 1) + q];
#pragma unroll
          for (int j = 0; j < ZH_T; ++j) bn[j] = WzL[wrow[j] + 4 * (s + 1)];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < ZH_T; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[j], acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < NP / 4) {
          av = an;
#pragma unroll
          for (int j = 0; j < ZH_T; ++j) bv[j] = bn[j];
        }
      }
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int row = row0 + 4 * q + reg;
#pragma unroll
        for (int j = 0; j < ZH_T; ++j) {
          const int col = 16 * j + r;
          if (row < a.R && col < ZH) a.dhs[(size_t)row * ZH + col] = acc[j][reg];
        }
      }
    }
    // ---- [dWz ; dbz] += [hs | 1]^T . dzargs over this wave's 16 rows
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float av[ZH_T], bv[NTZ];
#pragma unroll
      for (int i = 0; i < ZH_T; ++i) av[i] = myhs[(4 * ks + q) * ZH_LD + 16 * i + r];
#pragma unroll
      for (int j = 0; j < NTZ; ++j) bv[j] = mydz[(4 * ks + q) * ZD_LD + 16 * j + r];
#pragma unroll
      for (int i = 0; i < ZH_T; ++i)
#pragma unroll
        for (int j = 0; j < NTZ; ++j) acc3[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc3[i][j], 0, 0, 0);
    }
  }
  // the eight waves' tiles meet in LDS (12 tiles per round: 8 x 12 KB), wave w sums tiles w, w + 8 of a round
  float* slab = a.partial + (size_t)blockIdx.x * ZH_SLAB_ROWS * 2 * L;
  float* red = zh_lds;
  __syncthreads();
#pragma unroll
  for (int round = 0; round < NT3 / 12; ++round) {
#pragma unroll
    for (int ti = 0; ti < 12; ++ti) {
      const int t = round * 12 + ti;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) red[((wave * 12 + ti) * 4 + reg) * 64 + lane] = acc3[t / NTZ][t % NTZ][reg];
    }
    __syncthreads();
    for (int ti = wave; ti < 12; ti += ZH_NW) {
      const int t = round * 12 + ti, tm = t / NTZ, tn = t - tm * NTZ;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < ZH_NW; ++w) v += red[((w * 12 + ti) * 4 + reg) * 64 + lane];
        const int h = 16 * tm + 4 * q + reg, cp = 16 * tn + r;
        const bool half = cp >= 16 * LP16;
        const int c = cp - (half ? 16 * LP16 : 0);
        if (h < ZH_SLAB_ROWS && c < L) slab[(size_t)h * 2 * L + (half ? L : 0) + c] = v;
      }
    }
    __syncthreads();
  }
}

static int latent_head_wgs(int R) {
  const int blocks = (R + ZH_RB - 1) / ZH_RB;
  return blocks < 256 ? blocks : 256;
}

}  // namespace clv

extern "C" int clv_latent_head_supported(int H, int L) { return H == clv::ZH && L >= 1 && L <= 32; }

extern "C" size_t clv_latent_head_bwd_workspace_bytes(int R, int L) {
  return (R > 0 && L > 0) ? (size_t)clv::latent_head_wgs(R) * clv::ZH_SLAB_ROWS * 2 * L * sizeof(float) : 0;
}

extern "C" int clv_latent_head_fwd(int R, int H, int L, const float* hs, const float* Wz, const float* bz, float* eps,
                                   float* zargs, float* Z, int ldz, float* rowkl, const clv_noise_draw* noise, void* stream) {
  using namespace clv;
  if (!clv_latent_head_supported(H, L) || R <= 0 || ldz < L) return CLV_EINVAL;
  if (!hs || !Wz || !bz || !eps || !zargs || !Z || ((uintptr_t)hs) % 16 != 0) return CLV_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  LatentFwdArgs a{R, L, ldz, hs, Wz, bz, eps, zargs, Z, rowkl, {}};
  if (noise) {
    a.noise.on = 1; a.noise.k0 = (uint32_t)noise->seed; a.noise.k1 = (uint32_t)(noise->seed >> 32);
    a.noise.stream = noise->stream; a.noise.step = noise->step; a.noise.first = noise->first;
    a.noise.step_dev = noise->step_dev;
  }
  const int wgs = latent_head_wgs(R);
  ProfScope p("latent_head_fwd", s);
  if (L <= 16) {
    const size_t lds = (size_t)(ZH * (32 + 16) + ZH_NW * ZH_TILE) * sizeof(float);
    if (int e = clv::allow_dynamic_lds(reinterpret_cast<const void*>(latent_head_fwd_kernel<1>), (int)lds)) return e;
    hipLaunchKernelGGL(latent_head_fwd_kernel<1>, dim3(wgs), dim3(64 * ZH_NW), lds, s, a);
  } else {
    const size_t lds = (size_t)(ZH * (64 + 16) + ZH_NW * ZH_TILE) * sizeof(float);
    if (int e = clv::allow_dynamic_lds(reinterpret_cast<const void*>(latent_head_fwd_kernel<2>), (int)lds)) return e;
    hipLaunchKernelGGL(latent_head_fwd_kernel<2>, dim3(wgs), dim3(64 * ZH_NW), lds, s, a);
  }
  return launch_status();
}

extern "C" int clv_latent_head_bwd(int R, int H, int L, const float* hs, const float* Wz, const float* zargs,
                                   const float* eps, const float* dZ, int lddz, float kl_scale, float* dzargs, float* dhs,
                                   float* dWz, float* dbz, void* ws, size_t ws_bytes, clv_reduce_job* job, void* stream) {
  using namespace clv;
  if (!clv_latent_head_supported(H, L) || R <= 0 || lddz < L) return CLV_EINVAL;
  if (!hs || !Wz || !zargs || !eps || !dZ || !dhs || !dWz || !dbz || ((uintptr_t)hs) % 16 != 0) return CLV_EINVAL;
  if (!ws || ws_bytes < clv_latent_head_bwd_workspace_bytes(R, L)) return CLV_EWORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = (size_t)(ZH * ZH_LD + ZH_NW * ZH_TILE + ZH_NW * ZD_TILE) * sizeof(float);
  const int wgs = latent_head_wgs(R);
  LatentBwdArgs a{R, L, lddz, kl_scale, hs, Wz, zargs, eps, dZ, dzargs, dhs, (float*)ws};
  {
    ProfScope p("latent_head_bwd", s);
    if (L <= 16) {
      if (int e = clv::allow_dynamic_lds(reinterpret_cast<const void*>(latent_head_bwd_kernel<1>), (int)lds)) return e;
      hipLaunchKernelGGL(latent_head_bwd_kernel<1>, dim3(wgs), dim3(64 * ZH_NW), lds, s, a);
    } else {
      if (int e = clv::allow_dynamic_lds(reinterpret_cast<const void*>(latent_head_bwd_kernel<2>), (int)lds)) return e;
      hipLaunchKernelGGL(latent_head_bwd_kernel<2>, dim3(wgs), dim3(64 * ZH_NW), lds, s, a);
    }
  }
  int st = launch_status();
  if (st) return st;
  ReduceJob j;
  memset(&j, 0, sizeof(j));
  j.partial = (const float*)ws; j.M = ZH_SLAB_ROWS; j.N = 2 * L; j.splits = wgs; j.nprob = 2;
  j.alpha = 1.f; j.beta = 0.f; j.act = CLV_ACT_NONE;
  j.prob[0] = ReduceProb{dWz, 2 * L, 0};
  j.prob[1] = ReduceProb{dbz, 2 * L, ZH};
  if (job && wgs > 1) {
    memcpy(job, &j, sizeof(j));   // the caller reduces later (clv_splitk_reduce_multi)
    return CLV_OK;
  }
  if (job) memset(job, 0, sizeof(*job));      // a single slab is finished here (the multi-reduce skips empty jobs)
  return launch_reduce(j, s);
}
