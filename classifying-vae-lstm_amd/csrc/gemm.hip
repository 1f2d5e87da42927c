// gemm.hip -- fp32 GEMM on v_mfma_f32_16x16x4_f32 (gfx950), LDS-tiled, register-prefetched.
//
// C[M,N] = act(alpha * op(A) . op(B) + bias + beta * C); optional split-K with
// deterministic partial slabs.  The MFMA is a bit-exact k-ordered f32 fma chain
// (MI355X_MICROARCH.md, Matrix cores), so results are deterministic and within
// fp32 rounding of the fp64 oracle.
//
// Tile: BM = 16*WM*WAVES_M, BN = 16*WN*WAVES_N, BK = 16, 256 threads (4 waves).
// LDS holds both operands k-major ([k][m] / [k][n]) with a row stride == 16
// (mod 32) floats so the two k-rows a half-wave reads land on disjoint banks.
#include <stdlib.h>

#include "common.h"
#include "reduce_job.h"

namespace clv {

// Build with -DCLV_GEMM_STAMPS (make EXTRA=-DCLV_GEMM_STAMPS) to record, for the first wave of every k-group of one
// workgroup of the in-workgroup split-K kernel, the shader clock at five points of each of its first 16 k-tiles (loop
// top, loads issued, MFMAs issued, tile stored, after the barrier); tools/gemm_stamps.py prints them.
#ifdef CLV_GEMM_STAMPS
__device__ unsigned long long g_stamps[4][16][5];
#define CLV_STAMP(kt, i) do { if (stamp_on && (kt) < 16) g_stamps[kg][kt][i] = __builtin_readcyclecounter(); } while (0)
#else
#define CLV_STAMP(kt, i) do { } while (0)
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));

// one problem of a grouped launch: C_p[M_p,N] = op(A_p)[M_p,K] . B[K,N]; all problems share B, N, K.
struct GemmProb {
  const float* A; int lda; int M;
  float* C; int ldc;
  int a_shift;        // TN only: row k of op(A)^T is taken from row k - a_shift of A ...
  int a_zero_period;  // ... and is zero when k % a_zero_period == 0 (h_{t-1} of the first step of a sequence)
  int ones;           // A is an implicit row of ones (M == 1): column sums of B
  int tile0;          // first blockIdx.x of this problem
  int row0;           // first row of this problem in the split-K partial slabs
};

struct GemmArgs {
  int nprob;          // 0: single problem described by the fields below
  GemmProb prob[MAX_PROB];
  int M, N, K;
  float alpha, beta;
  const float* A; int lda;
  const float* B; int ldb;
  float* C; int ldc;
  const float* bias;
  int act;
  const float* aux;
  int k_chunk;       // K handled per blockIdx.z slice (multiple of 16)
  float* partial;    // != nullptr: raw partial sums [z][M][N]
  int vecA, vecB;    // 16-byte vector loads legal for A / B
  int xcd_remap;     // split-K launches: workgroups that share a K chunk run on the same XCD (see gemm_f32_kernel)
  // fused Bernoulli-NLL epilogue (output head): C = logits (may be null), plus d(loss)/d(logits) and the row NLL
  const float* bce_y; int bce_ldy; float bce_scale; float* bce_dl; float* bce_rownll;
};

template <int BMN>
struct LdsStride {  // == 16 (mod 32), multiple of 4
  static constexpr int value = (BMN % 32 == 16) ? BMN : BMN + 16;
};

// ---- global -> register staging -------------------------------------------
// MC: the tile dimension (m or n) is contiguous in memory: elem(mn,k) = p[k*ld + mn]
// KC: k is contiguous:                                  elem(mn,k) = p[mn*ld + k]
template <int BMN, bool KC, int BKT = 16>
struct TileLoader {
  static constexpr int BK = BKT;
  static constexpr int NV = BMN * BK / 4;              // float4 per tile
  static constexpr int PER = (NV + 255) / 256;         // float4 per thread
  static constexpr int LD = LdsStride<BMN>::value;

  // Fast path (vec: 16-byte aligned base, ld % 4 == 0): every thread issues its float4 loads unconditionally
  // from clamped, always-valid addresses and masks afterwards.  Loads under per-lane branches (the general path
  // below) make the compiler wait for each of them where it was issued, which serialises the prefetch of the
  // next k-tile with the MFMAs of the current one.
  // mk[i]: validity bits of r[i]'s 4 components; the masking itself happens in store(), i.e. AFTER the MFMAs of
  // the current tile: touching a loaded value right here would make the wave wait for the load before computing.
  __device__ static void load(float4 (&r)[PER], int (&mk)[PER], const float* __restrict__ p, int ld, int dim_mn,
                              int k_end, int mn0, int k0, int vec, int tid, int shift = 0, int zperiod = 0,
                              int ones = 0) {
#pragma unroll
    for (int i = 0; i < PER; ++i) mk[i] = 15;
    if (vec && ones != 1) {
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int idx = min(tid + i * 256, NV - 1);         // surplus threads repeat the last vector (never stored)
        float4 v;
        bool ok0, ok1, ok2, ok3;
        if (KC) {
          const int mn = idx / (BK / 4), k4 = (idx % (BK / 4)) * 4;
          const int gm = mn0 + mn, gk = k0 + k4;
          // (the clamp stays inside the operand's OWN k range, rounded up to a float4: an operand may be a column block of a
          // wider matrix -- pointer offset, ld = the parent's -- whose last row ends the allocation: round 5, a fault at
          // M = 4, K = 88, lda = 352 on the fourth gate block)
          const float* src = p + (size_t)min(gm, dim_mn - 1) * ld + min(gk, min(ld - 4, (k_end - 1) & ~3));
          v = *reinterpret_cast<const float4*>(src);
          const bool row = gm < dim_mn;
          ok0 = row && gk + 0 < k_end; ok1 = row && gk + 1 < k_end; ok2 = row && gk + 2 < k_end; ok3 = row && gk + 3 < k_end;
        } else {
          constexpr int RV = BMN / 4;
          const int k = idx / RV, c4 = (idx % RV) * 4;
          const int gk = k0 + k, gm = mn0 + c4;
          const bool row = gk < k_end && !(zperiod > 0 && gk % zperiod == 0);
          const int rk = max(min(gk, k_end - 1) - shift, 0);
          const float* src = p + (size_t)rk * ld + min(gm, min(ld - 4, (dim_mn - 1) & ~3));
          v = *reinterpret_cast<const float4*>(src);
          const int nreal = ones == 2 ? dim_mn - 1 : dim_mn;     // ones == 2: the last row of op(A)^T is implicit ones
          ok0 = row && gm + 0 < nreal; ok1 = row && gm + 1 < nreal; ok2 = row && gm + 2 < nreal; ok3 = row && gm + 3 < nreal;
          mk[i] = (ok0 ? 1 : 0) | (ok1 ? 2 : 0) | (ok2 ? 4 : 0) | (ok3 ? 8 : 0);
          if (ones == 2 && gk < k_end) {                          // bits 4..7: component is the ones row
            const int oc = dim_mn - 1 - gm;
            if (oc >= 0 && oc < 4) mk[i] |= 16 << oc;
          }
          r[i] = v;
          continue;
        }
        r[i] = v;
        mk[i] = (ok0 ? 1 : 0) | (ok1 ? 2 : 0) | (ok2 ? 4 : 0) | (ok3 ? 8 : 0);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      int idx = tid + i * 256;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < NV) {
        if (KC) {
          int mn = idx / (BK / 4), k4 = (idx % (BK / 4)) * 4;
          int gm = mn0 + mn, gk = k0 + k4;
          if (gm < dim_mn) {
            const float* src = p + (size_t)gm * ld + gk;
            if (vec && gk + 3 < k_end) {
              v = *reinterpret_cast<const float4*>(src);
            } else {
              if (gk + 0 < k_end) v.x = src[0];
              if (gk + 1 < k_end) v.y = src[1];
              if (gk + 2 < k_end) v.z = src[2];
              if (gk + 3 < k_end) v.w = src[3];
            }
          }
        } else {
          constexpr int RV = BMN / 4;
          int k = idx / RV, c4 = (idx % RV) * 4;
          int gk = k0 + k, gm = mn0 + c4;
          if (ones) {
            if (gk < k_end && gm == 0) v.x = 1.f;
          } else if (gk < k_end && !(zperiod > 0 && gk % zperiod == 0)) {
            const float* src = p + (size_t)(gk - shift) * ld + gm;
            if (vec && gm + 3 < dim_mn) {
              v = *reinterpret_cast<const float4*>(src);
            } else {
              if (gm + 0 < dim_mn) v.x = src[0];
              if (gm + 1 < dim_mn) v.y = src[1];
              if (gm + 2 < dim_mn) v.z = src[2];
              if (gm + 3 < dim_mn) v.w = src[3];
            }
          }
        }
      }
      r[i] = v;
    }
  }

  // ---- interior k-tiles: loop-invariant addressing ------------------------------------------------------------------
  // The index arithmetic above (divisions, clamps, 64-bit address products, validity bits: ~40 VALU instructions per
  // vector, 35 more for the zero-period modulo) used to run for every k-tile and made the kernel VALU-issue-bound
  // (ablation: the tile loop without MFMAs took 46 us of 64).  For a k-tile that lies fully inside [0, K) the address
  // of a thread's vector just advances by a constant, and nothing has to be masked along m/n at all: an out-of-range
  // row or column of a tile only ever feeds output rows/columns that the epilogue does not store.  Only the k
  // direction needs zeros (K tail, zero-period rows); the tail tile takes the general path.
  struct Iter {
    const float* p[PER];     // this thread's vector in the current k-tile (clamped to valid memory along m/n)
    int lofs[PER];           // LDS offset of the vector
    int zr[PER];             // (global k of the vector's row) % zperiod
    int oc[PER];             // component (0..3) of the vector that is the implicit ones row, or -1
  };
  __device__ static void iter_init(Iter& it, const float* __restrict__ base, int ld, int dim_mn, int mn0, int k0, int tid,
                                   int shift, int zperiod, int ones = 0) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int idx = min(tid + i * 256, NV - 1);
      if (KC) {
        const int mn = idx / (BK / 4), k4 = (idx % (BK / 4)) * 4;
        it.p[i] = base + (size_t)min(mn0 + mn, dim_mn - 1) * ld + (k0 + k4);
        it.lofs[i] = k4 * LD + mn;
        it.zr[i] = 1;
        it.oc[i] = -1;
      } else {
        constexpr int RV = BMN / 4;
        const int k = idx / RV, c4 = (idx % RV) * 4;
        it.p[i] = base + (size_t)(k0 + k - shift) * ld + min(mn0 + c4, min(ld - 4, (dim_mn - 1) & ~3));
        it.lofs[i] = k * LD + c4;
        it.zr[i] = zperiod > 0 ? (k0 + k) % zperiod : 1;
        const int oc = dim_mn - 1 - (mn0 + c4);
        it.oc[i] = (ones == 2 && oc >= 0 && oc < 4) ? oc : -1;
      }
    }
  }
  __device__ static void iter_load(Iter& it, float4 (&r)[PER], int ld, int zperiod) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      r[i] = *reinterpret_cast<const float4*>(it.p[i]);
      it.p[i] += KC ? BK : (size_t)BK * ld;
    }
  }
  // store the vectors loaded by iter_load (the row of tile t), then advance the zero-period phase to tile t+1
  __device__ static void iter_store(Iter& it, const float4 (&r)[PER], float* lds, int tid, int zperiod, int ones = 0) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      float4 v = r[i];
      if (!KC && ones == 2) {                        // uniform: implicit ones row of this problem
        v.x = it.oc[i] == 0 ? 1.f : v.x; v.y = it.oc[i] == 1 ? 1.f : v.y;
        v.z = it.oc[i] == 2 ? 1.f : v.z; v.w = it.oc[i] == 3 ? 1.f : v.w;
      }
      if (!KC && zperiod > 0) {                      // uniform
        const float f = it.zr[i] == 0 ? 0.f : 1.f;
        v.x *= f; v.y *= f; v.z *= f; v.w *= f;
        int z = it.zr[i] + BK;
        it.zr[i] = z >= zperiod ? z - zperiod : z;
      }
      if (tid + i * 256 < NV) {
        if (KC) {
          lds[it.lofs[i]] = v.x;
          lds[it.lofs[i] + LD] = v.y;
          lds[it.lofs[i] + 2 * LD] = v.z;
          lds[it.lofs[i] + 3 * LD] = v.w;
        } else {
          *reinterpret_cast<float4*>(&lds[it.lofs[i]]) = v;
        }
      }
    }
  }

  __device__ static void store(const float4 (&rr)[PER], const int (&mk)[PER], float* lds, int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      int idx = tid + i * 256;
      // arithmetic masks (operands are finite: a clamped address always points at real data); with selects the
      // compiler turns the masked load back into a load under a branch
      float4 r_i = make_float4(rr[i].x * ((mk[i] & 1) ? 1.f : 0.f), rr[i].y * ((mk[i] & 2) ? 1.f : 0.f),
                               rr[i].z * ((mk[i] & 4) ? 1.f : 0.f), rr[i].w * ((mk[i] & 8) ? 1.f : 0.f));
      if (mk[i] & 0xF0) {          // implicit ones row (GemmProb.ones == 2)
        if (mk[i] & 16) r_i.x = 1.f;
        if (mk[i] & 32) r_i.y = 1.f;
        if (mk[i] & 64) r_i.z = 1.f;
        if (mk[i] & 128) r_i.w = 1.f;
      }
      if (idx < NV) {
        if (KC) {
          int mn = idx / (BK / 4), k4 = (idx % (BK / 4)) * 4;
          lds[(k4 + 0) * LD + mn] = r_i.x;
          lds[(k4 + 1) * LD + mn] = r_i.y;
          lds[(k4 + 2) * LD + mn] = r_i.z;
          lds[(k4 + 3) * LD + mn] = r_i.w;
        } else {
          constexpr int RV = BMN / 4;
          int k = idx / RV, c4 = (idx % RV) * 4;
          *reinterpret_cast<float4*>(&lds[k * LD + c4]) = r_i;
        }
      }
    }
  }
};

__device__ __forceinline__ float apply_act(float v, int act, float aux) {
  if (act == CLV_ACT_RELU) return fmaxf(v, 0.f);
  if (act == CLV_ACT_SIGMOID) return sigmoidf_(v);
  if (act == CLV_ACT_MASKPOS) return aux > 0.f ? v : 0.f;
  return v;
}

// KG > 1: in-workgroup split-K.  The workgroup is KG groups of 4 waves; group kg runs the ordinary tile loop over its
// own K sub-chunk with its own LDS tiles, and the KG accumulators are summed through LDS (the lane -> (row, col) map
// is the same in every group) before group 0 writes ONE partial slab.  Same waves per CU as KG workgroups, 1/KG of
// the slab traffic.  Needs dynamic LDS (KG * tile bytes).
extern __shared__ __attribute__((aligned(16))) float gemm_dyn_smem[];
// PF: tiles in flight between HBM and LDS.  PF = 2 keeps two register sets and issues the loads of tile t+2 at the top
// of iteration t: at the weight-gradient shapes a load takes ~3600 cycles under load (in-kernel stamps),
// three times the MFMA phase of an iteration, so one tile of lookahead leaves every k-group waiting for its data.
template <int WM, int WN, int WAVES_M, int WAVES_N, bool TA, bool TB, int BK = 16, int KG = 1, int PF = 1>
__global__ __launch_bounds__(256 * KG) void gemm_f32_kernel(GemmArgs g) {
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  constexpr int BM = 16 * WM * WAVES_M, BN = 16 * WN * WAVES_N;
  using LA = TileLoader<BM, !TA, BK>;   // A row-major [M,K] => k contiguous
  using LB = TileLoader<BN, TB, BK>;    // B row-major [K,N] => n contiguous
  constexpr int LDA = LA::LD, LDB = LB::LD;
  constexpr int TILE_FLOATS = 2 * BK * (LDA + LDB);
  __shared__ __attribute__((aligned(16))) float smem_static[KG == 1 ? TILE_FLOATS : 4];
  const int kg = KG == 1 ? 0 : (int)(threadIdx.x >> 8);
  float* smem = KG == 1 ? smem_static : gemm_dyn_smem + kg * TILE_FLOATS;
  float* As = smem;
  float* Bs = smem + 2 * BK * LDA;

  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int wm = wave % WAVES_M, wn = wave / WAVES_M;
  // Workgroups are dealt to the 8 XCDs round-robin by linear id, and each XCD has its own L2.  In a split-K
  // launch the tiles of one K chunk read the same rows of A and B, so the linear id is re-decoded as
  // (xcd = chunk % 8, tile, chunk / 8): the chunk's rows are fetched from HBM into ONE L2 and reused there by
  // every tile, instead of once per XCD.
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (g.xcd_remap & 1) {
    const int ntx = gridDim.x, ntiles = gridDim.x * gridDim.y;
    const int lin = bx + ntx * (by + (int)gridDim.y * bz);
    const int xcd = lin & 7, rest = lin >> 3;
    const int tile = rest % ntiles;
    bz = (rest / ntiles) * 8 + xcd;
    bx = tile % ntx;
    by = tile / ntx;
  }
  // grouped launch: bx enumerates the m-tiles of every problem
  const float* __restrict__ Aptr = g.A;
  float* Cptr = g.C;
  int lda = g.lda, ldc = g.ldc, Mp = g.M, mtile = bx, prow0 = 0, a_shift = 0, a_zper = 0, a_ones = 0;
  if (g.nprob > 0) {
    int pi = 0;
#pragma unroll
    for (int i = 1; i < MAX_PROB; ++i)
      if (i < g.nprob && bx >= g.prob[i].tile0) pi = i;
    const GemmProb& pr = g.prob[pi];
    Aptr = pr.A; Cptr = pr.C; lda = pr.lda; ldc = pr.ldc; Mp = pr.M; mtile = bx - pr.tile0;
    prow0 = pr.row0; a_shift = pr.a_shift; a_zper = pr.a_zero_period; a_ones = pr.ones;
  }
  const int m0 = mtile * BM, n0 = by * BN;
  const int kbeg = min((bz * KG + kg) * g.k_chunk, g.K);
  const int kend = min(g.K, kbeg + g.k_chunk);
  // KG > 1: every group runs the same number of k-tiles (barriers are workgroup-wide); tiles past kend load zeros
  const int nk = KG == 1 ? (kend - kbeg + BK - 1) / BK : (g.k_chunk + BK - 1) / BK;

  f32x4 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float4 ra[LA::PER], rb[LB::PER];
  int ma[LA::PER], mb[LB::PER];
  if (nk > 0) {
    LA::load(ra, ma, Aptr, lda, Mp, kend, m0, kbeg, g.vecA, tid, a_shift, a_zper, a_ones);
    LB::load(rb, mb, g.B, g.ldb, g.N, kend, n0, kbeg, g.vecB, tid);
    LA::store(ra, ma, As, tid);
    LB::store(rb, mb, Bs, tid);
  }
  __syncthreads();
  // interior k-tiles (1 .. nfast): whole tile inside [kbeg, kend) -> incremental addressing, no masks along m/n
  const bool fastA = g.vecA && a_ones != 1 && !(a_ones == 2 && a_zper > 0) && (a_zper == 0 || a_zper >= BK) &&
                     !(g.xcd_remap & 2);
  const bool fastB = g.vecB != 0 && !(g.xcd_remap & 4);
  const int nfull = (kend - kbeg) / BK;              // k-tiles that are completely inside the chunk
  typename LA::Iter ia;
  typename LB::Iter ib;
  if (fastA) LA::iter_init(ia, Aptr, lda, Mp, m0, kbeg + BK, tid, a_shift, a_zper, a_ones);
  if (fastB) LB::iter_init(ib, g.B, g.ldb, g.N, n0, kbeg + BK, tid, 0, 0);

  auto mma = [&](int cur) {
    const float* as = As + cur * BK * LDA + wm * WM * 16 + r;
    const float* bs = Bs + cur * BK * LDB + wn * WN * 16 + r;
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      float a[WM], b[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) a[i] = as[(ks * 4 + q) * LDA + i * 16];
#pragma unroll
      for (int j = 0; j < WN; ++j) b[j] = bs[(ks * 4 + q) * LDB + j * 16];
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  };
  // tile t >= 1 takes the incremental path while it is a full interior tile (uniform), the general path otherwise
  auto load_tile = [&](int t, float4 (&xa)[LA::PER], int (&xma)[LA::PER], float4 (&xb)[LB::PER], int (&xmb)[LB::PER]) {
    const bool inner = t < nfull;
    if (fastA && inner) LA::iter_load(ia, xa, lda, a_zper);
    else LA::load(xa, xma, Aptr, lda, Mp, kend, m0, kbeg + t * BK, g.vecA, tid, a_shift, a_zper, a_ones);
    if (fastB && inner) LB::iter_load(ib, xb, g.ldb, 0);
    else LB::load(xb, xmb, g.B, g.ldb, g.N, kend, n0, kbeg + t * BK, g.vecB, tid);
  };
  auto store_tile = [&](int t, const float4 (&xa)[LA::PER], const int (&xma)[LA::PER], const float4 (&xb)[LB::PER],
                        const int (&xmb)[LB::PER]) {
    const bool inner = t < nfull;
    const int buf = t & 1;
    if (fastA && inner) LA::iter_store(ia, xa, As + buf * BK * LDA, tid, a_zper, a_ones);
    else LA::store(xa, xma, As + buf * BK * LDA, tid);
    if (fastB && inner) LB::iter_store(ib, xb, Bs + buf * BK * LDB, tid, 0);
    else LB::store(xb, xmb, Bs + buf * BK * LDB, tid);
  };

#ifdef CLV_GEMM_STAMPS
  const bool stamp_on = blockIdx.x == 1 && blockIdx.y == 1 && blockIdx.z == 3 && lane == 0 && wave == 0 && g.nprob > 0 && KG == 4;
#endif
  if (PF == 1) {
    for (int kt = 0; kt < nk; ++kt) {
      CLV_STAMP(kt, 0);
      if (kt + 1 < nk) load_tile(kt + 1, ra, ma, rb, mb);
      CLV_STAMP(kt, 1);
      mma(kt & 1);
      CLV_STAMP(kt, 2);
      if (kt + 1 < nk) store_tile(kt + 1, ra, ma, rb, mb);
      CLV_STAMP(kt, 3);
      __syncthreads();
      CLV_STAMP(kt, 4);
    }
  } else {
    // Two tiles in flight: (ra, rb) and (ra2, rb2) take turns; tile t+2 is requested at the top of iteration t, right
    // after its register set was emptied into LDS.  Host-checked: every tile of every chunk is a full interior tile on the
    // incremental path, so the steady-state body is straight-line code and the waits before the LDS stores stay
    // counted (the loads issued since remain in flight) instead of draining the queue.
    static_assert(PF == 1 || PF == 2, "1 or 2 tiles in flight");
    float4 ra2[LA::PER], rb2[LB::PER];
    auto ld = [&](float4 (&xa)[LA::PER], float4 (&xb)[LB::PER]) {
      LA::iter_load(ia, xa, lda, a_zper);
      LB::iter_load(ib, xb, g.ldb, 0);
    };
    auto st = [&](int buf, const float4 (&xa)[LA::PER], const float4 (&xb)[LB::PER]) {
      LA::iter_store(ia, xa, As + buf * BK * LDA, tid, a_zper, a_ones);
      LB::iter_store(ib, xb, Bs + buf * BK * LDB, tid, 0);
    };
    int kt = 0;
    if (nk > 1) ld(ra2, rb2);                      // tile 1
    for (; kt + 3 < nk; kt += 2) {
      CLV_STAMP(kt, 0);
      ld(ra, rb);                                  // tile kt+2
      CLV_STAMP(kt, 1);
      mma(0);
      CLV_STAMP(kt, 2);
      st(1, ra2, rb2);                             // tile kt+1
      CLV_STAMP(kt, 3);
      __syncthreads();
      CLV_STAMP(kt, 4);
      CLV_STAMP(kt + 1, 0);
      ld(ra2, rb2);                                // tile kt+3
      CLV_STAMP(kt + 1, 1);
      mma(1);
      CLV_STAMP(kt + 1, 2);
      st(0, ra, rb);                               // tile kt+2
      CLV_STAMP(kt + 1, 3);
      __syncthreads();
      CLV_STAMP(kt + 1, 4);
    }
    // tail: tile kt is in LDS buffer 0, tile kt+1 (if any) in (ra2, rb2), tile kt+2 (if any) not requested yet
    if (kt + 2 < nk) ld(ra, rb);
    mma(0);
    if (kt + 1 < nk) st(1, ra2, rb2);
    __syncthreads();
    if (kt + 1 < nk) {
      mma(1);
      if (kt + 2 < nk) st(0, ra, rb);
      __syncthreads();
      if (kt + 2 < nk) {
        mma(0);
        __syncthreads();
      }
    }
  }

  if (KG > 1) {
    // sum the groups' accumulators: groups 1.. park theirs in LDS (tile memory is free after the last barrier)
    static_assert(KG == 1 || (KG - 1) * WM * WN * 256 * 4 <= KG * TILE_FLOATS, "accumulators must fit the tile memory");
    f32x4* park = reinterpret_cast<f32x4*>(gemm_dyn_smem);
    if (kg > 0) {
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) park[((kg - 1) * WM * WN + i * WN + j) * 256 + tid] = acc[i][j];
    }
    __syncthreads();
    if (kg > 0) return;
#pragma unroll
    for (int gq = 0; gq < KG - 1; ++gq)
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          const f32x4 v = park[(gq * WM * WN + i * WN + j) * 256 + tid];
          acc[i][j][0] += v[0]; acc[i][j][1] += v[1]; acc[i][j][2] += v[2]; acc[i][j][3] += v[3];
        }
  }
  // epilogue: C/D map of 16x16x4: col = lane&15, row = (lane>>4)*4 + reg
  if (g.bce_y) {
    // logits -> Bernoulli NLL with Keras' epsilon clip (same arithmetic as bernoulli_nll_kernel).  One n-tile and
    // WAVES_N == 1 (host-checked): a row's columns sit in this wave, 16 lanes x WN tiles, so the row sum is
    // WN register adds and a 16-lane butterfly.
#pragma unroll
    for (int i = 0; i < WM; ++i) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int row = m0 + (wm * WM + i) * 16 + q * 4 + reg;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          const int col = n0 + (wn * WN + j) * 16 + r;
          if (row < Mp && col < g.N) {
            float a = acc[i][j][reg] * g.alpha;
            if (g.bias) a += g.bias[col];
            const float t = g.bce_y[(size_t)row * g.bce_ldy + col];
            const float l = fminf(fmaxf(a, BCE_CLIP_LO), BCE_CLIP_HI);
            const float e = __expf(-fabsf(l));
            s += fmaxf(l, 0.f) + __logf(1.f + e) - l * t;
            const float r1 = fast_rcp(1.f + e);
            const float sg = l >= 0.f ? r1 : e * r1;
            const bool inside = (a >= BCE_CLIP_LO) && (a <= BCE_CLIP_HI);
            const size_t o = (size_t)row * ldc + col;
            if (Cptr) Cptr[o] = a;
            if (g.bce_dl) g.bce_dl[o] = inside ? g.bce_scale * (sg - t) : 0.f;
          }
        }
        s += __shfl_xor(s, 8, 64);
        s += __shfl_xor(s, 4, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 1, 64);
        if (r == 0 && row < Mp && g.bce_rownll) g.bce_rownll[row] = s;
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < WM; ++i) {
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int col = n0 + (wn * WN + j) * 16 + r;
      if (col >= g.N) continue;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int row = m0 + (wm * WM + i) * 16 + q * 4 + reg;
        if (row >= Mp) continue;
        float v = acc[i][j][reg];
        if (g.partial) {
          g.partial[((size_t)bz * g.M + prow0 + row) * g.N + col] = v;
        } else {
          v *= g.alpha;
          if (g.bias) v += g.bias[col];
          const size_t o = (size_t)row * ldc + col;
          if (g.beta != 0.f) v += g.beta * Cptr[o];
          v = apply_act(v, g.act, g.act == CLV_ACT_MASKPOS ? g.aux[o] : 0.f);
          Cptr[o] = v;
        }
      }
    }
  }
}

static ReduceJob make_job(const GemmArgs& g, int splits) {
  ReduceJob j;
  memset(&j, 0, sizeof(j));
  j.partial = g.partial; j.M = g.M; j.N = g.N; j.splits = splits; j.nprob = g.nprob;
  j.alpha = g.alpha; j.beta = g.beta; j.bias = g.bias; j.aux = g.aux; j.act = g.act;
  if (g.nprob == 0) j.prob[0] = ReduceProb{g.C, g.ldc, 0};
  for (int i = 0; i < g.nprob; ++i) j.prob[i] = ReduceProb{g.prob[i].C, g.prob[i].ldc, g.prob[i].row0};
  return j;
}

// sum of `splits` partial slabs + epilogue.  64 outputs x 4 slab-lanes per block, 8 loads in flight
// per thread (a serial loop over the slabs is latency-bound: ~0.5 us per dependent HBM load).
__device__ __forceinline__ void reduce_block(const ReduceJob& g, unsigned blk, float (*red)[64]) {
  const int splits = g.splits;
  const size_t mn = (size_t)g.M * g.N;
  const int ex = threadIdx.x & 63, zy = threadIdx.x >> 6;
  const size_t idx = (size_t)blk * 64 + ex;
  float v = 0.f;
  if (idx < mn) {
    const float* p = g.partial + idx;
    int z = zy;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f;
    for (; z + 28 < splits; z += 32) {
      a0 += p[(size_t)(z + 0) * mn]; a1 += p[(size_t)(z + 4) * mn]; a2 += p[(size_t)(z + 8) * mn];
      a3 += p[(size_t)(z + 12) * mn]; a4 += p[(size_t)(z + 16) * mn]; a5 += p[(size_t)(z + 20) * mn];
      a6 += p[(size_t)(z + 24) * mn]; a7 += p[(size_t)(z + 28) * mn];
    }
    for (; z < splits; z += 4) a0 += p[(size_t)z * mn];
    v = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
  }
  red[zy][ex] = v;
  __syncthreads();
  if (zy == 0 && idx < mn) {
    v = (red[0][ex] + red[1][ex]) + (red[2][ex] + red[3][ex]);
    int row = (int)(idx / g.N);
    const int col = (int)(idx % g.N);
    int pi = 0;
#pragma unroll
    for (int i = 1; i < MAX_PROB; ++i)
      if (i < g.nprob && row >= g.prob[i].row0) pi = i;
    float* Cptr = g.prob[pi].C;
    const int ldc = g.prob[pi].ldc;
    row -= g.prob[pi].row0;
    v *= g.alpha;
    if (g.bias) v += g.bias[col];
    const size_t o = (size_t)row * ldc + col;
    if (g.beta != 0.f) v += g.beta * Cptr[o];
    v = apply_act(v, g.act, g.act == CLV_ACT_MASKPOS ? g.aux[o] : 0.f);
    Cptr[o] = v;
  }
}
// The same with 4 consecutive outputs per thread (16-byte loads of the slabs: a quarter of the load instructions; the
// slabs of the bf16 weight-gradient kernel are 32 MB per LSTM).  Needs N % 4 == 0 and 16-byte aligned rows everywhere
// (reduce_vec_ok).  A block still covers 64 outputs (the same number of blocks: ~4 per CU for an LSTM's gradients), now
// as 16 float4 lanes x 16 slab lanes, every thread with up to 8 loads in flight.
__device__ __forceinline__ void reduce_block_v4(const ReduceJob& g, unsigned blk, float4 (*red)[16]) {
  const int splits = g.splits;
  const size_t mn = (size_t)g.M * g.N;
  const int ex = threadIdx.x & 15, zy = threadIdx.x >> 4;
  const size_t idx = ((size_t)blk * 16 + ex) * 4;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  auto add = [](float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; };
  if (idx < mn) {
    const float* p = g.partial + idx;
    auto at = [&](int z) { return *reinterpret_cast<const float4*>(p + (size_t)z * mn); };
    float4 a0 = v, a1 = v, a2 = v, a3 = v, a4 = v, a5 = v, a6 = v, a7 = v;
    int z = zy;
    for (; z + 112 < splits; z += 128) {
      add(a0, at(z)); add(a1, at(z + 16)); add(a2, at(z + 32)); add(a3, at(z + 48));
      add(a4, at(z + 64)); add(a5, at(z + 80)); add(a6, at(z + 96)); add(a7, at(z + 112));
    }
    for (; z < splits; z += 16) add(a0, at(z));
    add(a0, a1); add(a2, a3); add(a4, a5); add(a6, a7);
    add(a0, a2); add(a4, a6);
    add(a0, a4);
    v = a0;
  }
  red[zy][ex] = v;
  __syncthreads();
  if (zy == 0 && idx < mn) {
    float4 t[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {                  // fixed order: ((0+1)+(2+3)) per group of four, then the groups
      t[q] = red[4 * q][ex];
      float4 u = red[4 * q + 2][ex];
      add(t[q], red[4 * q + 1][ex]); add(u, red[4 * q + 3][ex]);
      add(t[q], u);
    }
    add(t[0], t[1]); add(t[2], t[3]); add(t[0], t[2]);
    v = t[0];
    int row = (int)(idx / g.N);
    const int col = (int)(idx % g.N);
    int pi = 0;
#pragma unroll
    for (int i = 1; i < MAX_PROB; ++i)
      if (i < g.nprob && row >= g.prob[i].row0) pi = i;
    float* Cptr = g.prob[pi].C;
    row -= g.prob[pi].row0;
    const size_t o = (size_t)row * g.prob[pi].ldc + col;
    float r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float w = r[j] * g.alpha;
      if (g.bias) w += g.bias[col + j];
      if (g.beta != 0.f) w += g.beta * Cptr[o + j];
      r[j] = apply_act(w, g.act, g.act == CLV_ACT_MASKPOS ? g.aux[o + j] : 0.f);
    }
    *reinterpret_cast<float4*>(Cptr + o) = make_float4(r[0], r[1], r[2], r[3]);
  }
}
static bool reduce_vec_ok(const ReduceJob& j) {
  if (j.N % 4 || ((uintptr_t)j.partial) % 16) return false;
  const int n = j.nprob == 0 ? 1 : j.nprob;
  for (int i = 0; i < n; ++i)
    if (j.prob[i].ldc % 4 || ((uintptr_t)j.prob[i].C) % 16) return false;
  return true;
}
static unsigned reduce_blocks(const ReduceJob& j) {          // j.pad_ = 1: the 4-wide form
  const size_t mn = (size_t)j.M * j.N;
  return (unsigned)((mn + 63) / 64);
}
__global__ __launch_bounds__(256) void splitk_reduce_kernel(ReduceJob j) {
  __shared__ float4 red[16][16];
  if (j.pad_) reduce_block_v4(j, blockIdx.x, red);
  else reduce_block(j, blockIdx.x, reinterpret_cast<float (*)[64]>(red));
}
// several pending reductions in one launch (the weight gradients of a whole backward pass)
constexpr int MAX_JOBS = 16;
struct ReduceTable { int njobs; unsigned blk0[MAX_JOBS + 1]; ReduceJob job[MAX_JOBS]; };
// up to five strided means riding in the same launch (the loss terms of a step): one block each, after the jobs' blocks
struct MeanTerms { int n_terms; const float* x[5]; int n[5]; int stride[5]; float* out; };
__device__ __forceinline__ void mean_block(const MeanTerms& m, int k, float4 (*red)[16]) {
  const float* x = m.x[0]; int n = m.n[0], st = m.stride[0];
#pragma unroll
  for (int i = 1; i < 5; ++i)
    if (k == i) { x = m.x[i]; n = m.n[i]; st = m.stride[i]; }
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int i = threadIdx.x;
  if (st == 1 && ((uintptr_t)x) % 16 == 0) {        // contiguous term: float4 loads, 4 in flight per thread
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const int n4 = n / 4;
    int j = threadIdx.x;
    for (; j + 768 < n4; j += 1024) {
      const float4 a = x4[j], b = x4[j + 256], c = x4[j + 512], d = x4[j + 768];
      a0 += (a.x + a.y) + (a.z + a.w); a1 += (b.x + b.y) + (b.z + b.w);
      a2 += (c.x + c.y) + (c.z + c.w); a3 += (d.x + d.y) + (d.z + d.w);
    }
    for (; j < n4; j += 256) { const float4 a = x4[j]; a0 += (a.x + a.y) + (a.z + a.w); }
    i = 4 * n4 + threadIdx.x;
  }
  for (; i + 768 < n; i += 1024) {
    a0 += x[(size_t)i * st]; a1 += x[(size_t)(i + 256) * st];
    a2 += x[(size_t)(i + 512) * st]; a3 += x[(size_t)(i + 768) * st];
  }
  for (; i < n; i += 256) a0 += x[(size_t)i * st];
  const float acc = wave_sum((a0 + a1) + (a2 + a3));
  float* part = reinterpret_cast<float*>(red);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) m.out[k] = ((part[0] + part[1]) + (part[2] + part[3])) / (float)n;
}
// Few-row products over a short K riding in the same launch (clv_splitk_reduce_multi): C[r, :] = sum_k A[k, r] B[k, :] for
// r < R rows of A [K, lda] plus, with `ones`, the column sums of B as one more row -- the label rows and the bias of an
// LSTM input-kernel gradient over K = batch rows of sum_t dz (cl_vrnn/model.py:194,223: the RepeatVector(W) columns).
// A launch of their own was 8.6 us for 2 MFLOP.  A block owns 64 columns; 4 k-lanes stride through K with every row's
// accumulator in registers; the A chunk is staged in LDS (broadcast reads).
constexpr int SR_ROWS = 16, SR_KC = 32;
struct SkinnySet { const float* A; int lda, R, ones; const float* B; int ldb, N, K; float* C; int ldc; float* Cones; };
struct SkinnyRider { int nsets, blocks_per_set; SkinnySet set[2]; };
__device__ __forceinline__ void skinny_rider_block(const SkinnyRider& sr, int blk) {
  // one buffer: the A^T chunk [SR_KC][SR_ROWS] while the products run, the k-lanes' partial sums [3][SR_ROWS][64] at the end
  __shared__ __attribute__((aligned(16))) float sbuf[3 * SR_ROWS * 64];
  float (*At)[SR_ROWS] = reinterpret_cast<float (*)[SR_ROWS]>(sbuf);
  const bool second = blk >= sr.blocks_per_set;
  const SkinnySet g = second ? sr.set[1] : sr.set[0];
  const int tid = threadIdx.x, cx = tid & 63, kl = tid >> 6;
  const int col = (blk - (second ? sr.blocks_per_set : 0)) * 64 + cx;
  const bool live = col < g.N;
  const int R = g.R + (g.ones ? 1 : 0);
  float acc[SR_ROWS];
#pragma unroll
  for (int r = 0; r < SR_ROWS; ++r) acc[r] = 0.f;
  for (int kc = 0; kc < g.K; kc += SR_KC) {
    __syncthreads();
#pragma unroll
    for (int e0 = 0; e0 < SR_KC * SR_ROWS; e0 += 256) {      // A^T chunk (+ the ones row) -> LDS
      const int e = e0 + tid, kk = e / SR_ROWS, r = e % SR_ROWS, k = kc + kk;
      float v = 0.f;
      if (k < g.K) v = r < g.R ? g.A[(size_t)k * g.lda + r] : (r == g.R && g.ones ? 1.f : 0.f);
      At[kk][r] = v;
    }
    float breg[SR_KC / 4];
#pragma unroll
    for (int j = 0; j < SR_KC / 4; ++j) {                    // this thread's B values of the chunk, all in flight
      const int k = kc + kl + 4 * j;
      const float v = g.B[(size_t)min(k, g.K - 1) * g.ldb + min(col, g.N - 1)];
      breg[j] = v * ((live && k < g.K) ? 1.f : 0.f);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SR_KC / 4; ++j) {
      const float4* arow = reinterpret_cast<const float4*>(At[kl + 4 * j]);
#pragma unroll
      for (int r4 = 0; r4 < SR_ROWS / 4; ++r4) {
        const float4 a = arow[r4];
        acc[4 * r4] = fmaf(a.x, breg[j], acc[4 * r4]);
        acc[4 * r4 + 1] = fmaf(a.y, breg[j], acc[4 * r4 + 1]);
        acc[4 * r4 + 2] = fmaf(a.z, breg[j], acc[4 * r4 + 2]);
        acc[4 * r4 + 3] = fmaf(a.w, breg[j], acc[4 * r4 + 3]);
      }
    }
  }
  __syncthreads();
  float (*redk)[SR_ROWS][64] = reinterpret_cast<float (*)[SR_ROWS][64]>(sbuf);
  if (kl > 0) {
#pragma unroll
    for (int r = 0; r < SR_ROWS; ++r) redk[kl - 1][r][cx] = acc[r];
  }
  __syncthreads();
  if (kl == 0 && live) {
#pragma unroll
    for (int r = 0; r < SR_ROWS; ++r) {
      if (r < R) {
        const float v = ((acc[r] + redk[0][r][cx]) + (redk[1][r][cx] + redk[2][r][cx]));
        if (r < g.R) g.C[(size_t)r * g.ldc + col] = v;
        else g.Cones[col] = v;
      }
    }
  }
}

__global__ __launch_bounds__(256) void splitk_reduce_multi_kernel(ReduceTable t, MeanTerms m, SkinnyRider sr) {
  __shared__ float4 red[16][16];
  // the rider blocks come FIRST: they are short chains of dependent loads, and at the end of the grid they were the
  // launch's tail (the launch got as much longer as the products' own launch had taken)
  const unsigned nrider = (unsigned)(sr.nsets * sr.blocks_per_set);
  if (blockIdx.x < nrider) { skinny_rider_block(sr, blockIdx.x); return; }
  const unsigned bid = blockIdx.x - nrider;
  if (bid >= t.blk0[t.njobs]) { mean_block(m, bid - t.blk0[t.njobs], red); return; }
  int ji = 0;
#pragma unroll
  for (int i = 1; i < MAX_JOBS; ++i)
    if (i < t.njobs && bid >= t.blk0[i]) ji = i;
  if (t.job[ji].pad_) reduce_block_v4(t.job[ji], bid - t.blk0[ji], red);
  else reduce_block(t.job[ji], bid - t.blk0[ji], reinterpret_cast<float (*)[64]>(red));
}
int launch_reduce(const ReduceJob& j_in, hipStream_t s) {
  ProfScope p("gemm_splitk_reduce", s);
  ReduceJob j = j_in;
  j.pad_ = reduce_vec_ok(j) ? 1 : 0;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(reduce_blocks(j)), dim3(256), 0, s, j);
  return launch_status();
}

// Grouped TN product with a handful of output rows in total and a short K (the label rows and the bias of an
// LSTM kernel gradient: [C+1, 4H] over K = batch rows): the 96x96 MFMA tile would be >85 % padding and
// its split-K chain is a string of dependent round trips, so this runs on the VALU: a block owns 64 output
// columns, 16 k-lanes stride through K with every row's accumulator in registers, LDS-reduce at the end.
constexpr int SK_ROWS = 16, SK_KC = 256;
__device__ __forceinline__ void tn_skinny_body(const GemmArgs& g);
__global__ __launch_bounds__(1024) void tn_skinny_kernel(GemmArgs g) { tn_skinny_body(g); }
// two independent products of this kind (the two LSTMs of cl_vrnn) in one launch: blockIdx.y picks the set
__global__ __launch_bounds__(1024) void tn_skinny2_kernel(GemmArgs g0, GemmArgs g1) {
  if (blockIdx.y == 0) tn_skinny_body(g0); else tn_skinny_body(g1);
}
__device__ __forceinline__ void tn_skinny_body(const GemmArgs& g) {
  __shared__ __attribute__((aligned(16))) float At[SK_KC][SK_ROWS];     // A^T chunk, [k][row] (broadcast reads: no padding needed)
  __shared__ float red[8][16][64];
  const int tid = threadIdx.x, cx = tid & 63;
  const int kl = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = blockIdx.x * 64 + cx;
  const bool live = col < g.N;
  const int R = g.M;
  float acc[SK_ROWS];
#pragma unroll
  for (int r = 0; r < SK_ROWS; ++r) acc[r] = 0.f;
  for (int kc = 0; kc < g.K; kc += SK_KC) {
    // all of this thread's B values of the chunk in flight at once (a serial k loop would pay an HBM round
    // trip per iteration)
    float breg[SK_KC / 16];
#pragma unroll
    for (int j = 0; j < SK_KC / 16; ++j) {
      const int k = kc + kl + 16 * j;
      const float v = g.B[(size_t)min(k, g.K - 1) * g.ldb + min(col, g.N - 1)];     // unconditional (clamped) load
      breg[j] = v * ((live && k < g.K) ? 1.f : 0.f);     // arithmetic mask: a select would let the compiler sink the
                                                         // load into a branch and wait for each one separately
    }
    // A^T chunk -> LDS (rows of every problem side by side); the 4 loads of a thread are issued together
    float av[SK_KC * SK_ROWS / 1024];
#pragma unroll
    for (int it = 0; it < SK_KC * SK_ROWS / 1024; ++it) {
      const int e = tid + it * 1024;
      const int kk = e / SK_ROWS, r = e % SK_ROWS, k = kc + kk;
      // select the row's problem with static indices (a runtime index into the argument struct would send it
      // through scratch memory)
      const float* Ap = g.prob[0].A;
      int lda = g.prob[0].lda, row0 = 0, shift = g.prob[0].a_shift, zper = g.prob[0].a_zero_period, ones = g.prob[0].ones;
#pragma unroll
      for (int i = 1; i < MAX_PROB; ++i)
        if (i < g.nprob && r >= g.prob[i].row0) {
          Ap = g.prob[i].A; lda = g.prob[i].lda; row0 = g.prob[i].row0; shift = g.prob[i].a_shift;
          zper = g.prob[i].a_zero_period; ones = g.prob[i].ones;
        }
      const bool ok = r < R && k < g.K;
      const bool ld = ok && !ones && !(zper > 0 && k % zper == 0);
      const float* src = ld ? Ap + (size_t)(k - shift) * lda + (r - row0) : g.B;       // unconditional load
      const float v = *src;
      av[it] = fmaf(v, ld ? 1.f : 0.f, (ok && ones) ? 1.f : 0.f);
    }
#pragma unroll
    for (int it = 0; it < SK_KC * SK_ROWS / 1024; ++it) {
      const int e = tid + it * 1024;
      At[e / SK_ROWS][e % SK_ROWS] = av[it];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SK_KC / 16; ++j) {
      // wave-uniform address: 4 broadcast ds_read_b128 per k (the LDS pipe is shared by all 16 waves, so the
      // number of LDS instructions, not bytes, is what counts here)
      const float4* arow = reinterpret_cast<const float4*>(At[kl + 16 * j]);
#pragma unroll
      for (int r4 = 0; r4 < SK_ROWS / 4; ++r4) {
        const float4 a = arow[r4];
        acc[4 * r4] = fmaf(a.x, breg[j], acc[4 * r4]);
        acc[4 * r4 + 1] = fmaf(a.y, breg[j], acc[4 * r4 + 1]);
        acc[4 * r4 + 2] = fmaf(a.z, breg[j], acc[4 * r4 + 2]);
        acc[4 * r4 + 3] = fmaf(a.w, breg[j], acc[4 * r4 + 3]);
      }
    }
    __syncthreads();
  }
  // reduce the 16 k-lanes: 8 output rows per round through LDS, wave w sums row w of the round
#pragma unroll
  for (int r0 = 0; r0 < SK_ROWS; r0 += 8) {
    if (r0 < R) {
#pragma unroll
      for (int r = 0; r < 8; ++r) red[r][kl][cx] = acc[r0 + r];
      __syncthreads();
      const int r = r0 + kl;
      if (kl < 8 && r < R && live) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[kl][i][cx];
        float* Cp = g.prob[0].C;
        int ldc = g.prob[0].ldc, row0 = 0;
#pragma unroll
        for (int i = 1; i < MAX_PROB; ++i)
          if (i < g.nprob && r >= g.prob[i].row0) { Cp = g.prob[i].C; ldc = g.prob[i].ldc; row0 = g.prob[i].row0; }
        float* cp = Cp + (size_t)(r - row0) * ldc + col;
        *cp = (g.beta != 0.f ? g.beta * *cp : 0.f) + t;
      }
      __syncthreads();
    }
  }
}

template <int WM, int WN, int WAVES_M, int WAVES_N>
static void launch_cfg(const GemmArgs& g_in, int ta, int tb, int splits, hipStream_t s) {
  constexpr int BM = 16 * WM * WAVES_M, BN = 16 * WN * WAVES_N;
  GemmArgs g = g_in;
  {
    static const int remap = env_int("CLV_GEMM_XCD_REMAP", 1);
    g.xcd_remap = remap && splits > 1 && splits % 8 == 0;
    if (const char* e = getenv("CLV_GEMM_NOITER")) g.xcd_remap |= atoi(e);      // dev: 2 = no A iterator, 4 = no B iterator
  }
  dim3 grid(g.nprob > 0 ? g.prob[g.nprob - 1].tile0 + (g.prob[g.nprob - 1].M + BM - 1) / BM : (g.M + BM - 1) / BM,
            (g.N + BN - 1) / BN, splits);
  if (ta && !tb && g.nprob > 0 && BM >= 96 && BN == 96) {      // grouped weight gradients: optional deeper k-tile
    static const int bk32 = env_int("CLV_GEMM_BK32", 0);
    if (bk32) {
      hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, WAVES_M, WAVES_N, true, false, 32>), grid, dim3(256), 0, s, g);
      return;
    }
  }
  if (!ta && !tb) hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, WAVES_M, WAVES_N, false, false>), grid, dim3(256), 0, s, g);
  else if (!ta && tb) hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, WAVES_M, WAVES_N, false, true>), grid, dim3(256), 0, s, g);
  else if (ta && !tb) hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, WAVES_M, WAVES_N, true, false>), grid, dim3(256), 0, s, g);
  else hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, WAVES_M, WAVES_N, true, true>), grid, dim3(256), 0, s, g);
}

// tile choice by output shape (all shapes of the path: N in {1..352}, M in {2..262144})
enum TileCfg { T128x16, T64x32, T64x64, T96x96, T64x96, T64x176, T32x96 };
static TileCfg pick_tile(int M, int N, int K) {
  if (N <= 16) return T128x16;
  if (N <= 32) return T64x32;
  if (N <= 64) return T64x64;
  // tall output, short K (a Dense layer's weight gradient over one batch): enough 32-row tiles to fill the
  // chip without splitting K, so no partial slabs and no reduce pass
  if (N <= 96 && K <= 512 && M >= 32 * 256) return T32x96;
  if (N <= 96) return M <= 96 ? T96x96 : T64x96;
  if (N % 176 == 0 || N > 256) return M <= 96 ? T96x96 : T64x176;
  return T64x64;
}
static void tile_dims(TileCfg c, int& bm, int& bn) {
  switch (c) {
    case T128x16: bm = 128; bn = 16; break;
    case T64x32: bm = 64; bn = 32; break;
    case T64x64: bm = 64; bn = 64; break;
    case T96x96: bm = 96; bn = 96; break;
    case T64x96: bm = 64; bn = 96; break;
    case T32x96: bm = 32; bn = 96; break;
    default: bm = 64; bn = 176; break;
  }
}
// split K until ~4 workgroups per CU are in flight (each k-tile is a dependent HBM round trip, so a
// few long workgroups are latency-bound); chunks stay >= 32 deep.
static long split_target() {
  static long t = 0;
  if (!t) { const char* e = getenv("CLV_GEMM_WGS"); t = e ? atol(e) : 1024; if (t < 1) t = 1024; }
  return t;
}
static int auto_split(int M, int N, int K) {
  int bm, bn;
  tile_dims(pick_tile(M, N, K), bm, bn);
  const long tiles = (long)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
  if (tiles >= 256 || K < 128) return 1;       // each k-tile is a dependent ~1 us round trip; a reduce launch ~6 us
  long s = (split_target() + tiles - 1) / tiles;
  if (s > K / 64) s = K / 64;
  if (s > 512) s = 512;
  return s < 1 ? 1 : (int)s;
}

static void launch_gemm(const GemmArgs& g, int ta, int tb, int splits, hipStream_t s) {
  switch (pick_tile(g.M, g.N, g.K)) {
    case T128x16: launch_cfg<2, 1, 4, 1>(g, ta, tb, splits, s); break;
    case T64x32: launch_cfg<1, 2, 4, 1>(g, ta, tb, splits, s); break;
    case T64x64: launch_cfg<2, 2, 2, 2>(g, ta, tb, splits, s); break;
    case T96x96: launch_cfg<3, 3, 2, 2>(g, ta, tb, splits, s); break;
    case T64x96: launch_cfg<1, 6, 4, 1>(g, ta, tb, splits, s); break;
    case T32x96: launch_cfg<1, 3, 2, 2>(g, ta, tb, splits, s); break;
    default: launch_cfg<1, 11, 4, 1>(g, ta, tb, splits, s); break;
  }
}

}  // namespace clv

extern "C" int clv_gemm_auto_split(int M, int N, int K) { return clv::auto_split(M, N, K); }

extern "C" size_t clv_gemm_workspace_bytes(int M, int N, int split_k) {
  if (split_k <= 1) return 0;
  return (size_t)split_k * M * N * sizeof(float);
}

extern "C" int clv_gemm_f32(int transa, int transb, int M, int N, int K, float alpha,
                                     const float* A, int lda, const float* B, int ldb,
                                     float beta, float* C, int ldc,
                                     const float* bias, int act, const float* aux,
                                     int split_k, void* ws, size_t ws_bytes, clv_reduce_job* job, void* stream) {
  using namespace clv;
  if (job) memset(job, 0, sizeof(*job));
  if (M <= 0 || N <= 0 || K < 0 || !A || !B || !C) return CLV_EINVAL;
  if (act < CLV_ACT_NONE || act > CLV_ACT_MASKPOS) return CLV_EINVAL;
  if (act == CLV_ACT_MASKPOS && !aux) return CLV_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.nprob = 0;
  g.M = M; g.N = N; g.K = K; g.alpha = alpha; g.beta = beta;
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
  g.bias = bias; g.act = act; g.aux = aux;
  g.vecA = (lda % 4 == 0) && (((uintptr_t)A) % 16 == 0);
  g.vecB = (ldb % 4 == 0) && (((uintptr_t)B) % 16 == 0);
  int splits = split_k < 1 ? auto_split(M, N, K) : split_k;
  int kc = (K + splits - 1) / splits;
  kc = (kc + 15) / 16 * 16;
  if (kc == 0) kc = 16;
  splits = (K + kc - 1) / kc;
  if (splits < 1) splits = 1;
  g.k_chunk = kc;
  g.partial = nullptr;
  if (splits > 1) {
    size_t need = (size_t)splits * M * N * sizeof(float);
    if (!ws || ws_bytes < need) return CLV_EWORKSPACE;
    g.partial = (float*)ws;
  }
  {
    char label[48] = "gemm_f32";
    if (prof_on() && getenv("CLV_PROF_SHAPES"))
      snprintf(label, sizeof(label), "gemm %dx%dx%d %c%c s%d", M, N, K, transa ? 'T' : 'N', transb ? 'T' : 'N', splits);
    ProfScope p(label, s);
    launch_gemm(g, transa != 0, transb != 0, splits, s);
  }
  int st = launch_status();
  if (st) return st;
  if (splits > 1) {
    const ReduceJob j = make_job(g, splits);
    if (job) memcpy(job, &j, sizeof(j));       // the caller reduces later (clv_splitk_reduce_multi)
    else st = launch_reduce(j, s);
  }
  return st;
}

extern "C" int clv_splitk_reduce_multi(const clv_reduce_job* jobs, int njobs, const float* const* x, const int* n,
                                          const int* stride, int n_terms, float* means_out,
                                          const clv_skinny_product* riders, int n_riders, void* stream) {
  using namespace clv;
  SkinnyRider sr{};
  if (n_riders < 0 || n_riders > 2 || (n_riders > 0 && !riders)) return CLV_EINVAL;
  for (int i = 0; i < n_riders; ++i) {
    const clv_skinny_product& r = riders[i];
    if (!r.A || !r.B || !r.C || r.rows < 1 || r.rows + (r.bias_row ? 1 : 0) > SR_ROWS || r.N <= 0 || r.K <= 0 ||
        (i > 0 && r.N != riders[0].N))
      return CLV_EINVAL;
    sr.set[i] = SkinnySet{r.A, r.lda, r.rows, r.bias_row ? 1 : 0, r.B, r.ldb, r.N, r.K, r.C, r.ldc, r.bias_row};
  }
  sr.nsets = n_riders;
  sr.blocks_per_set = n_riders ? (riders[0].N + 63) / 64 : 0;
  if (njobs < 0 || (njobs > 0 && !jobs) || n_terms < 0 || n_terms > 5) return CLV_EINVAL;
  if (n_terms > 0 && (!x || !n || !stride || !means_out)) return CLV_EINVAL;
  MeanTerms m{};
  m.n_terms = n_terms; m.out = means_out;
  for (int i = 0; i < n_terms; ++i) {
    if (!x[i] || n[i] <= 0) return CLV_EINVAL;
    m.x[i] = x[i]; m.n[i] = n[i]; m.stride[i] = stride[i];
  }
  hipStream_t s = (hipStream_t)stream;
  ReduceTable t;
  t.njobs = 0;
  unsigned blk = 0;
  for (int i = 0; i < njobs; ++i) {
    ReduceJob j;
    memcpy(&j, &jobs[i], sizeof(j));
    if (j.splits <= 1 || !j.partial) continue;          // finished inside its GEMM
    if (t.njobs == MAX_JOBS) return CLV_EINVAL;
    j.pad_ = reduce_vec_ok(j) ? 1 : 0;
    t.blk0[t.njobs] = blk;
    t.job[t.njobs++] = j;
    blk += reduce_blocks(j);
  }
  if (t.njobs == 0 && n_terms == 0 && n_riders == 0) return CLV_OK;
  t.blk0[t.njobs] = blk;
  ProfScope p("gemm_splitk_reduce", s);
  hipLaunchKernelGGL(splitk_reduce_multi_kernel, dim3(blk + (unsigned)n_terms + (unsigned)(sr.nsets * sr.blocks_per_set)),
                     dim3(256), 0, s, t, m, sr);
  return launch_status();
}

// Grouped weight-gradient GEMM: C_p[M_p,N] = A_p^T . B for up to 4 problems that share B [K,N] (one pass over
// dz for all of an LSTM's kernel / recurrent-kernel gradients, or dW and db of a Dense layer).
static int grouped_tiles(const clv_gemm_prob* probs, int nprob, int bm) {   // total m-tiles of a grouped launch
  int t = 0;
  for (int i = 0; i < nprob; ++i) t += (probs[i].M + bm - 1) / bm;
  return t;
}

// tile of a grouped launch: narrow outputs (a latent head's 2L columns) get a narrow tile
static int grouped_tiles(const clv_gemm_prob* probs, int nprob, int bm);
static void grouped_tile(const clv_gemm_prob* probs, int nprob, int N, int& bm, int& bn) {
  if (N <= 16) { bm = 128; bn = 16; }
  else if (N <= 32) { bm = 64; bn = 32; }
  else {
    // 96- or 128-row tiles, whichever pads the problems' row counts less (e.g. [120 | 88] rows: 3 tiles of 96 = 288
    // padded rows, 2 tiles of 128 = 256)
    bn = 96;
    bm = grouped_tiles(probs, nprob, 128) * 128 < grouped_tiles(probs, nprob, 96) * 96 ? 128 : 96;
  }
}

extern "C" int clv_gemm_grouped_auto_split(const clv_gemm_prob* probs, int nprob, int N, int K) {
  if (!probs || nprob < 1 || nprob > clv::MAX_PROB) return 1;
  int rows = 0;
  for (int i = 0; i < nprob; ++i) rows += probs[i].M;
  if (rows <= clv::SK_ROWS && K <= 4096) return 1;      // skinny VALU kernel, never split
  int bm, bn;
  grouped_tile(probs, nprob, N, bm, bn);
  const long tiles = (long)grouped_tiles(probs, nprob, bm) * ((N + bn - 1) / bn);
  if (tiles >= 256 || K < 128) return 1;
  long s = (clv::split_target() + tiles - 1) / tiles;
  if (s > K / 64) s = K / 64;
  if (s > 512) s = 512;
  return s < 1 ? 1 : (int)s;
}

extern "C" size_t clv_gemm_grouped_workspace_bytes(const clv_gemm_prob* probs, int nprob, int N, int split_k) {
  if (!probs || split_k <= 1) return 0;
  size_t m = 0;
  for (int i = 0; i < nprob; ++i) m += probs[i].M;
  return (size_t)split_k * m * N * sizeof(float);
}

extern "C" int clv_gemm_grouped_tn(const clv_gemm_prob* probs, int nprob, int N, int K,
                                            const float* B, int ldb, float beta,
                                            int split_k, void* ws, size_t ws_bytes, clv_reduce_job* job, void* stream) {
  using namespace clv;
  if (job) memset(job, 0, sizeof(*job));
  if (!probs || nprob < 1 || nprob > MAX_PROB || N <= 0 || K <= 0 || !B) return CLV_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.nprob = nprob;
  int tile = 0, row = 0, vec = 1;
  bool has_ones2 = false;
  int bm, bn;
  grouped_tile(probs, nprob, N, bm, bn);
  for (int i = 0; i < nprob; ++i) {
    const clv_gemm_prob& p = probs[i];
    if (p.M <= 0 || !p.C || (p.ones != 1 && !p.A) || (p.ones == 1 && p.M != 1) || (p.ones == 2 && p.M < 2) ||
        p.ones < 0 || p.ones > 2)
      return CLV_EINVAL;
    if (p.a_shift < 0 || (p.a_shift > 0 && p.a_zero_period <= 0)) return CLV_EINVAL;
    g.prob[i] = GemmProb{p.A, p.lda, p.M, p.C, p.ldc, p.a_shift, p.a_zero_period, p.ones, tile, row};
    tile += (p.M + bm - 1) / bm;
    row += p.M;
    if (p.ones != 1) vec = vec && (p.lda % 4 == 0) && (((uintptr_t)p.A) % 16 == 0);
    if (p.ones == 2) { has_ones2 = true; if (p.a_zero_period > 0 || p.a_shift > 0) return CLV_EINVAL; }
  }
  g.M = row; g.N = N; g.K = K; g.alpha = 1.f; g.beta = beta;
  g.B = B; g.ldb = ldb; g.bias = nullptr; g.act = CLV_ACT_NONE; g.aux = nullptr;
  g.vecA = vec;
  g.vecB = (ldb % 4 == 0) && (((uintptr_t)B) % 16 == 0);
  char glabel[48] = "gemm_grouped_tn";
  if (prof_on() && getenv("CLV_PROF_SHAPES")) snprintf(glabel, sizeof(glabel), "ggemm %dx%dx%d", row, N, K);
  if (has_ones2 && !vec) return CLV_EINVAL;       // the appended ones row is only implemented on the vector-load path
  if (row <= SK_ROWS && K <= 4096 && split_k <= 1 && !has_ones2) {     // a few rows over a short K: VALU kernel, no split, no reduce
    ProfScope p(glabel, s);
    hipLaunchKernelGGL(tn_skinny_kernel, dim3((N + 63) / 64), dim3(1024), 0, s, g);
    return launch_status();
  }
  int splits = split_k < 1 ? clv_gemm_grouped_auto_split(probs, nprob, N, K) : split_k;
  int kc = (K + splits - 1) / splits;
  kc = (kc + 15) / 16 * 16;
  splits = (K + kc - 1) / kc;
  g.k_chunk = kc;
  // in-workgroup split-K: 4 K sub-chunks per workgroup, one slab per workgroup (96 x 96 tiles, enough splits)
  static const int kg_on = env_int("CLV_GEMM_KG", 1);
  const long wg_after = (long)tile * ((N + bn - 1) / bn) * (splits / 4);      // workgroups left if 4 chunks share one
  const bool kg4 = kg_on && bn == 96 && bm == 96 && splits >= 16 && splits % 4 == 0 && wg_after >= 256;
  // two tiles in flight per k-group where every tile of every chunk is a full interior tile on the incremental path
  static const int pf2_on = env_int("CLV_GEMM_PF2", 1);
  bool pf2 = kg4 && pf2_on && kc % 16 == 0 && K == kc * splits && vec && g.vecB;
  for (int i = 0; i < nprob && pf2; ++i)
    pf2 = probs[i].ones == 0 && (probs[i].a_zero_period == 0 || probs[i].a_zero_period >= 16);
  if (kg4) splits /= 4;                  // = number of slabs / workgroups along K
  g.partial = nullptr;
  if (splits > 1) {
    size_t need = (size_t)splits * row * N * sizeof(float);
    if (!ws || ws_bytes < need) return CLV_EWORKSPACE;
    g.partial = (float*)ws;
  }
  {
    ProfScope p(glabel, s);
    if (kg4) {
      constexpr int BMq = 96, BNq = 96;
      auto kern1 = gemm_f32_kernel<3, 3, 2, 2, true, false, 16, 4, 1>;
      auto kern2 = gemm_f32_kernel<3, 3, 2, 2, true, false, 16, 4, 2>;
      const size_t lds = 4 * (2 * 16 * (LdsStride<BMq>::value + LdsStride<BNq>::value)) * sizeof(float);
      if (int e = allow_dynamic_lds(reinterpret_cast<const void*>(kern1), (int)lds)) return e;
      if (int e = allow_dynamic_lds(reinterpret_cast<const void*>(kern2), (int)lds)) return e;
      static const int remap = env_int("CLV_GEMM_XCD_REMAP", 1);
      g.xcd_remap = remap && splits > 1 && splits % 8 == 0;
      dim3 grid(g.prob[g.nprob - 1].tile0 + (g.prob[g.nprob - 1].M + BMq - 1) / BMq, (g.N + BNq - 1) / BNq, splits);
      if (pf2) hipLaunchKernelGGL(kern2, grid, dim3(1024), lds, s, g);
      else hipLaunchKernelGGL(kern1, grid, dim3(1024), lds, s, g);
    }
    else if (bn == 16) launch_cfg<2, 1, 4, 1>(g, 1, 0, splits, s);     // 128 x 16
    else if (bn == 32) launch_cfg<1, 2, 4, 1>(g, 1, 0, splits, s);     // 64 x 32
    else if (bm == 128) launch_cfg<4, 3, 2, 2>(g, 1, 0, splits, s);    // 128 x 96
    else launch_cfg<3, 3, 2, 2>(g, 1, 0, splits, s);                   // 96 x 96
  }
  int st = launch_status();
  if (st) return st;
  if (splits > 1) {
    const ReduceJob j = make_job(g, splits);
    if (job) memcpy(job, &j, sizeof(j));
    else st = launch_reduce(j, s);
  }
  return st;
}

// Output head with the loss fused into the epilogue: logits = A.B + bias (C, optional), row NLL and
// d(NLL)/d(logits) * scale, see clvae.h.
extern "C" int clv_gemm_bce_f32(int M, int N, int K, const float* A, int lda, const float* B, int ldb, const float* bias,
                                const float* Y, int ldy, float scale, float* logits, float* dlogits, int ldc,
                                float* rownll, void* stream) {
  using namespace clv;
  if (M <= 0 || N <= 0 || N > 176 || K <= 0 || !A || !B || !Y || (!dlogits && !rownll && !logits)) return CLV_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.M = M; g.N = N; g.K = K; g.alpha = 1.f; g.beta = 0.f;
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = logits; g.ldc = ldc;
  g.bias = bias; g.act = CLV_ACT_NONE;
  g.vecA = (lda % 4 == 0) && (((uintptr_t)A) % 16 == 0);
  g.vecB = (ldb % 4 == 0) && (((uintptr_t)B) % 16 == 0);
  g.k_chunk = (K + 15) / 16 * 16;
  g.bce_y = Y; g.bce_ldy = ldy; g.bce_scale = scale; g.bce_dl = dlogits; g.bce_rownll = rownll;
  ProfScope p("gemm_bce", s);
  // tiles with one wave column (WAVES_N == 1) that cover N in one n-tile
  if (N <= 16) launch_cfg<2, 1, 4, 1>(g, 0, 0, 1, s);
  else if (N <= 32) launch_cfg<1, 2, 4, 1>(g, 0, 0, 1, s);
  else if (N <= 96) launch_cfg<1, 6, 4, 1>(g, 0, 0, 1, s);
  else launch_cfg<1, 11, 4, 1>(g, 0, 0, 1, s);
  return launch_status();
}

// Two few-row grouped products C_p = A_p^T . B (rows in total <= 16 each, K <= 4096) with different B operands in ONE
// launch -- the label rows and the bias of both LSTM input-kernel gradients of cl_vrnn (B = sum_t dz of each LSTM).
extern "C" int clv_gemm_grouped_tn_small2(const clv_gemm_prob* probs0, int nprob0, const float* B0,
                                          const clv_gemm_prob* probs1, int nprob1, const float* B1,
                                          int N, int K, int ldb, void* stream) {
  using namespace clv;
  if (!probs0 || !probs1 || !B0 || !B1 || N <= 0 || K <= 0 || K > 4096) return CLV_EINVAL;
  GemmArgs g[2];
  const clv_gemm_prob* pp[2] = {probs0, probs1};
  const int np[2] = {nprob0, nprob1};
  const float* Bs[2] = {B0, B1};
  for (int q = 0; q < 2; ++q) {
    if (np[q] < 1 || np[q] > MAX_PROB) return CLV_EINVAL;
    memset(&g[q], 0, sizeof(GemmArgs));
    int row = 0;
    for (int i = 0; i < np[q]; ++i) {
      const clv_gemm_prob& p = pp[q][i];
      if (p.M <= 0 || !p.C || (p.ones != 1 && !p.A) || (p.ones == 1 && p.M != 1) || p.ones < 0 || p.ones > 1) return CLV_EINVAL;
      g[q].prob[i] = GemmProb{p.A, p.lda, p.M, p.C, p.ldc, p.a_shift, p.a_zero_period, p.ones, 0, row};
      row += p.M;
    }
    if (row > SK_ROWS) return CLV_EINVAL;
    g[q].nprob = np[q]; g[q].M = row; g[q].N = N; g[q].K = K; g[q].alpha = 1.f; g[q].B = Bs[q]; g[q].ldb = ldb;
  }
  hipStream_t s = (hipStream_t)stream;
  ProfScope p("gemm_grouped_tn", s);
  hipLaunchKernelGGL(tn_skinny2_kernel, dim3((N + 63) / 64, 2), dim3(1024), 0, s, g[0], g[1]);
  return launch_status();
}

#ifdef CLV_GEMM_STAMPS
extern "C" int clv_dbg_gemm_stamps(void* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(clv::g_stamps), sizeof(clv::g_stamps));
}
#endif
