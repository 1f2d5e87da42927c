// out_head.hip -- the output head of cl_vrnn in training: forward, loss, and all three backward products in one pass
// over the decoder states (gfx950).
//
// Reference: X_decoded_mean = TimeDistributed(Dense(88, sigmoid)) on the decoder LSTM states (cl_vrnn/model.py:229-234),
// vae_loss = 88 * mean BCE(x, x_hat) (cl_vrnn/model.py:241-242) and their gradients under K.gradients:
//   logits = hs.Wo + bo                  [R,88], R = B*T
//   nll_r  = sum_j BCE(x_rj, sigmoid(logits_rj))   (Keras' 1e-7 clip),  dl = scale * (sigmoid(logits) - x)
//   dhs    = dl.Wo^T                     (upstream gradient of the decoder BPTT)
//   dWo    = hs^T.dl,  dbo = sum_r dl    (weight gradients)
// As three GEMM launches these are K = 88 products over [R,88] operands: each launch is prologue/epilogue-bound
// (5.5 k-tiles) and streams the same 11.5 MB arrays again; dl is written and read twice.  Here a workgroup keeps Wo in
// LDS, takes 128 rows of hs (8 waves x one 16-row MFMA tile), and runs the three products back to back on
// v_mfma_f32_16x16x4_f32 with dl handed from the C/D layout to the A and B layouts through LDS; dl and hs are read
// from HBM once and dl never has to exist there.  dWo/dbo accumulate in registers across a workgroup's row blocks and
// leave as one [89,88] slab per workgroup, summed by the deferred split-K reduce (fixed order: bit-reproducible).
#include "out_head_args.h"
#include "reduce_job.h"

namespace clv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int OH_T = 6;            // 16-wide tiles covering 88 (96)
constexpr int OH_KS = OH / 4;      // k-steps of a K = 88 product
constexpr int OH_LD = 116;         // LDS row stride: 116 % 64 = 52 -> the [row][k] operand reads (r*52 + q) hit 64 different banks
constexpr int OH_NW = 8;           // waves per workgroup, one 16-row tile each
static_assert(OH_RB == 16 * OH_NW, "rows per block");
constexpr int OH_TILE = 16 * OH_LD;
constexpr int OH_P3 = 5;           // weight-gradient tiles per wave (36 tiles over 8 waves: 4 x 5 + 4 x 4)

// -DOH_STAMPS: wave 0 of workgroup 0 records the shader clock at the phase boundaries of its first block
// (tools/out_head_stamps.py).  Configuration 3, one block per workgroup (us): Wo -> LDS + first loads + barrier 4.1, hs tile
// 0.7, first product 3.5, NLL + dl tile + logits stores 4.25, second product + dhs stores 3.6, barrier 1.2, weight gradient
// 5.1, slab store 2.15 = 24.6 of a 28.9 us launch.  Tried on that evidence: the 48 per-lane stores of a block as buffer
// instructions with immediate offsets and out-of-range lanes instead of predicated global stores -- SLOWER (NLL phase 4.8,
// second product 4.5, launch 30.4 -> 33.0 us), removed.
#ifdef OH_STAMPS
__device__ unsigned long long g_oh_stamps[16];
#define OHS(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_oh_stamps[k] = __builtin_readcyclecounter(); } while (0)
#else
#define OHS(k) do { } while (0)
#endif


__global__ __launch_bounds__(64 * OH_NW) void out_head_train_kernel(OutHeadArgs a) {
  extern __shared__ __attribute__((aligned(16))) float oh_lds[];
  float* WoL = oh_lds;                                   // [88][OH_LD]
  float* hsT = oh_lds + OH * OH_LD;                      // [8][16][OH_LD]; column 88 = 1 for live rows (the dbo row of hs^T)
  float* dlT = hsT + OH_NW * OH_TILE;                    // [8][16][OH_LD]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  OHS(0);

  float bias[OH_T];
#pragma unroll
  for (int j = 0; j < OH_T; ++j) bias[j] = a.bo[min(16 * j + r, OH - 1)];

  // weight-gradient tiles of this wave: tile t = 6*jm + jn covers rows 16*jm.. of [dWo ; dbo], columns 16*jn..
  const int t0 = wave < 4 ? OH_P3 * wave : 4 * OH_P3 + (OH_P3 - 1) * (wave - 4);
  const int nt = wave < 4 ? OH_P3 : OH_P3 - 1;
  int tm[OH_P3], tn[OH_P3];
  f32x4 acc3[OH_P3];
#pragma unroll
  for (int i = 0; i < OH_P3; ++i) {
    const int t = min(t0 + i, OH_T * OH_T - 1);
    tm[i] = t / OH_T; tn[i] = t - tm[i] * OH_T;
    acc3[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  float* myhs = hsT + wave * OH_TILE;
  float* mydl = dlT + wave * OH_TILE;

  // this wave's 16 rows of hs are one contiguous 5.6 KB piece of HBM: 6 float4 per lane
  float4 hv[6];
  auto load_hs = [&](int row0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int e = min(lane + 64 * i, 16 * (OH / 4) - 1);
      const int rr = e / (OH / 4), c4 = e - rr * (OH / 4);
      hv[i] = *reinterpret_cast<const float4*>(a.hs + (size_t)min(row0 + rr, a.R - 1) * OH + 4 * c4);
    }
  };
  // targets of this lane's outputs (C/D layout: column 16j + r, rows 4q + reg)
  float y[OH_T][4];
  auto load_y = [&](int row0) {
#pragma unroll
    for (int j = 0; j < OH_T; ++j)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
        y[j][reg] = static_cast<const float*>(a.Y)[(size_t)min(row0 + 4 * q + reg, a.R - 1) * a.ldy + min(16 * j + r, OH - 1)];
  };
  load_hs(blockIdx.x * OH_RB + wave * 16);               // both in flight while Wo is staged
  load_y(blockIdx.x * OH_RB + wave * 16);
  {
    constexpr int NV = OH * (OH / 4);                     // Wo -> LDS (float4 along the output index), loads issued together
    float4 wv[(NV + 64 * OH_NW - 1) / (64 * OH_NW)];
#pragma unroll
    for (int i = 0; i < (NV + 64 * OH_NW - 1) / (64 * OH_NW); ++i) {
      const int e = min(tid + 64 * OH_NW * i, NV - 1);
      const int h = e / (OH / 4), c4 = e - h * (OH / 4);
      wv[i] = *reinterpret_cast<const float4*>(a.Wo + h * OH + 4 * c4);
    }
#pragma unroll
    for (int i = 0; i < (NV + 64 * OH_NW - 1) / (64 * OH_NW); ++i) {
      const int e = tid + 64 * OH_NW * i;
      const int h = e / (OH / 4), c4 = e - h * (OH / 4);
      if (e < NV) *reinterpret_cast<float4*>(WoL + h * OH_LD + 4 * c4) = wv[i];
    }
    for (int e = tid; e < OH * 8; e += 64 * OH_NW) WoL[(e >> 3) * OH_LD + OH + (e & 7)] = 0.f;      // columns 88..95
  }
  __syncthreads();
  OHS(1);

  for (int blk = blockIdx.x; blk * OH_RB < a.R; blk += gridDim.x) {
    const int row0 = blk * OH_RB + wave * 16;
    if (blk != (int)blockIdx.x) load_y(row0);                  // (the block's hs rows were requested one block ago)
    {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int e = lane + 64 * i;
        const int rr = e / (OH / 4), c4 = e - rr * (OH / 4);
        const float mk = row0 + rr < a.R ? 1.f : 0.f;
        if (e < 16 * (OH / 4))
          *reinterpret_cast<float4*>(myhs + rr * OH_LD + 4 * c4) = make_float4(hv[i].x * mk, hv[i].y * mk, hv[i].z * mk, hv[i].w * mk);
      }
      // columns 88..95 of the tile: the ones column (bias gradient) and zeros
      for (int e = lane; e < 16 * 8; e += 64) {
        const int rr = e >> 3, c = e & 7;
        myhs[rr * OH_LD + OH + c] = (c == 0 && row0 + rr < a.R) ? 1.f : 0.f;
      }
    }
    if ((blk + (int)gridDim.x) * OH_RB < a.R) load_hs(row0 + (int)gridDim.x * OH_RB);      // the next block's rows, under this block's products
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the tile is wave-private: no barrier
    OHS(2);

    // ---- logits = hs.Wo
    f32x4 acc[OH_T];
#pragma unroll
    for (int j = 0; j < OH_T; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {   // operands of k-step s+1 are read while the MFMAs of step s issue (explicit register double buffer: left to
        // itself the compiler reads two operands, waits, issues two MFMAs)
      float av = myhs[r * OH_LD + q], bv[OH_T];
#pragma unroll
      for (int j = 0; j < OH_T; ++j) bv[j] = WoL[q * OH_LD + 16 * j + r];
#pragma unroll
      for (int s = 0; s < OH_KS; ++s) {
        float an = 0.f, bn[OH_T];
        if (s + 1 < OH_KS) {
          an = myhs[r * OH_LD + 4 * (s + 1) + q];
#pragma unroll
          for (int j = 0; j < OH_T; ++j) bn[j] = WoL[(4 * (s + 1) + q) * OH_LD + 16 * j + r];
        }
        __builtin_amdgcn_sched_barrier(0);      // keep the reads above the MFMAs (the scheduler sinks them to their uses)
#pragma unroll
        for (int j = 0; j < OH_T; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[j], acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < OH_KS) {
          av = an;
#pragma unroll
          for (int j = 0; j < OH_T; ++j) bv[j] = bn[j];
        }
      }
    }
    OHS(3);
    // ---- Bernoulli NLL with Keras' epsilon clip (same arithmetic as the gemm_bce epilogue); dl -> LDS tile
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int row = row0 + 4 * q + reg;
      const bool rok = row < a.R;
      float ssum = 0.f;
#pragma unroll
      for (int j = 0; j < OH_T; ++j) {
        const int col = 16 * j + r;
        const bool ok = rok && col < OH;
        const float lg = acc[j][reg] + bias[j];
        const float t = y[j][reg];
        const float l = fminf(fmaxf(lg, BCE_CLIP_LO), BCE_CLIP_HI);
        const float e = __expf(-fabsf(l));
        const float nl = fmaxf(l, 0.f) + __logf(1.f + e) - l * t;
        const float r1 = fast_rcp(1.f + e);
        const float sg = l >= 0.f ? r1 : e * r1;
        const bool inside = (lg >= BCE_CLIP_LO) && (lg <= BCE_CLIP_HI);
        const float dl = (ok && inside) ? a.scale * (sg - t) : 0.f;
        ssum += ok ? nl : 0.f;
        mydl[(4 * q + reg) * OH_LD + col] = dl;
        if (ok) {
          const size_t o = (size_t)row * OH + col;
          if (a.logits) a.logits[o] = lg;
          if (a.dlogits) a.dlogits[o] = dl;
        }
      }
      ssum += __shfl_xor(ssum, 8, 64);
      ssum += __shfl_xor(ssum, 4, 64);
      ssum += __shfl_xor(ssum, 2, 64);
      ssum += __shfl_xor(ssum, 1, 64);
      if (r == 0 && rok) a.rownll[row] = ssum;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    OHS(4);

    // ---- dhs = dl.Wo^T   (k = output note, n = hidden unit: B[k][n] = Wo[n][k])
#pragma unroll
    for (int j = 0; j < OH_T; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int wrow[OH_T];
#pragma unroll
    for (int j = 0; j < OH_T; ++j) wrow[j] = min(16 * j + r, OH - 1) * OH_LD + q;      // units 88..95: repeat row 87 (never stored)
    {
      float av = mydl[r * OH_LD + q], bv[OH_T];
#pragma unroll
      for (int j = 0; j < OH_T; ++j) bv[j] = WoL[wrow[j]];
#pragma unroll
      for (int s = 0; s < OH_KS; ++s) {
        float an = 0.f, bn[OH_T];
        if (s + 1 < OH_KS) {
          an = mydl[r * OH_LD + 4 * (s + 1) + q];
#pragma unroll
          for (int j = 0; j < OH_T; ++j) bn[j] = WoL[wrow[j] + 4 * (s + 1)];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < OH_T; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[j], acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < OH_KS) {
          av = an;
#pragma unroll
          for (int j = 0; j < OH_T; ++j) bv[j] = bn[j];
        }
      }
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int row = row0 + 4 * q + reg;
#pragma unroll
      for (int j = 0; j < OH_T; ++j) {
        const int col = 16 * j + r;
        if (row < a.R && col < OH) a.dhs[(size_t)row * OH + col] = acc[j][reg];
      }
    }
    OHS(5);
    __syncthreads();         // every wave's hs and dl tiles are in LDS
    OHS(6);

    // ---- [dWo ; dbo] += [hs | 1]^T . dl over the block's 128 rows: the 6 x 6 output tiles are dealt 5,5,5,5,4,4,4,4
    {
      auto p3off = [&](int ks) { return (ks >> 2) * OH_TILE + ((ks & 3) * 4 + q) * OH_LD + r; };     // row 4*ks + q of the block
      float av[OH_P3], bv[OH_P3];
#pragma unroll
      for (int i = 0; i < OH_P3; ++i) {
        av[i] = hsT[p3off(0) + 16 * tm[i]];
        bv[i] = dlT[p3off(0) + 16 * tn[i]];
      }
#pragma unroll 4
      for (int ks = 0; ks < OH_RB / 4; ++ks) {
        const int off = p3off(min(ks + 1, OH_RB / 4 - 1));        // the last step re-reads its own operands (unused)
        float an[OH_P3], bn[OH_P3];
#pragma unroll
        for (int i = 0; i < OH_P3; ++i) {
          an[i] = hsT[off + 16 * tm[i]];
          bn[i] = dlT[off + 16 * tn[i]];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < OH_P3; ++i)
          if (i < nt) acc3[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[i], acc3[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < OH_P3; ++i) { av[i] = an[i]; bv[i] = bn[i]; }
      }
    }
    OHS(7);
    __syncthreads();         // before the next block overwrites the tiles
  }
  float* slab = a.partial + (size_t)blockIdx.x * OH_SLAB_ROWS * OH;
#pragma unroll
  for (int i = 0; i < OH_P3; ++i)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int h = 16 * tm[i] + 4 * q + reg, o = 16 * tn[i] + r;
      if (i < nt && h < OH_SLAB_ROWS && o < OH) slab[h * OH + o] = acc3[i][reg];
    }
  OHS(8);
}

#ifdef OH_STAMPS
}
extern "C" int clv_debug_out_head_stamps(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(clv::g_oh_stamps), sizeof(unsigned long long) * 16);
}
namespace clv {
#endif
static int out_head_wgs(int R) {
  const int blocks = (R + OH_RB - 1) / OH_RB;
  return blocks < 256 ? blocks : 256;
}

}  // namespace clv

extern "C" int clv_out_head_train_supported(int H, int D) { return H == clv::OH && D == clv::OH; }

extern "C" size_t clv_out_head_train_workspace_bytes(int R) {
  return R > 0 ? (size_t)clv::out_head_wgs(R) * clv::OH_SLAB_ROWS * clv::OH * sizeof(float) : 0;
}

extern "C" int clv_out_head_train(int R, int H, int D, const float* hs, const float* Wo, const float* bo,
                                  const void* Y, int y_u8, int ldy, float scale, float* logits, float* rownll, float* dlogits,
                                  float* dhs, float* dWo, float* dbo, void* ws, size_t ws_bytes, clv_reduce_job* job,
                                  void* stream) {
  using namespace clv;
  if (!clv_out_head_train_supported(H, D) || R <= 0 || ldy < D) return CLV_EINVAL;
  if (!hs || !Wo || !bo || !Y || !rownll || !dhs || !dWo || !dbo) return CLV_EINVAL;
  if (((uintptr_t)hs) % 16 != 0 || ((uintptr_t)Wo) % 16 != 0) return CLV_EINVAL;
  if (!ws || ws_bytes < clv_out_head_train_workspace_bytes(R)) return CLV_EWORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = (size_t)(OH * OH_LD + 2 * OH_NW * OH_TILE) * sizeof(float);
  if (int e = clv::allow_dynamic_lds(reinterpret_cast<const void*>(out_head_train_kernel), (int)lds)) return e;
  const int wgs = out_head_wgs(R);
  OutHeadArgs a{R, ldy, scale, hs, Wo, bo, Y, y_u8 != 0, logits, rownll, dlogits, dhs, (float*)ws};
  // CLV_OUT_HEAD_F32=1 keeps the f32-MFMA kernel of this file (A/B runs; it also serves rows that are not 16-byte aligned)
  static const bool f32_only = [] { const char* e = getenv("CLV_OUT_HEAD_F32"); return e && e[0] == '1'; }();
  if (!(f32_only && !y_u8) && out_head_bf16_ok(a)) {
    if (int e = launch_out_head_bf16(a, wgs, s)) return e;
  } else {
    if (y_u8) return CLV_EINVAL;            // byte targets: the bf16-MFMA kernel only (rows 4-byte aligned, ldy a multiple of 4)
    ProfScope p("out_head_train", s);
    hipLaunchKernelGGL(out_head_train_kernel, dim3(wgs), dim3(64 * OH_NW), lds, s, a);
  }
  int st = launch_status();
  if (st) return st;
  ReduceJob j;
  memset(&j, 0, sizeof(j));
  j.partial = (const float*)ws; j.M = OH_SLAB_ROWS; j.N = OH; j.splits = wgs; j.nprob = 2;
  j.alpha = 1.f; j.beta = 0.f; j.act = CLV_ACT_NONE;
  j.prob[0] = ReduceProb{dWo, OH, 0};
  j.prob[1] = ReduceProb{dbo, OH, OH};
  if (job && wgs > 1) {
    memcpy(job, &j, sizeof(j));   // the caller reduces later (clv_splitk_reduce_multi)
    return CLV_OK;
  }
  if (job) memset(job, 0, sizeof(*job));      // a single slab is finished here (the multi-reduce skips empty jobs)
  return launch_reduce(j, s);
}
