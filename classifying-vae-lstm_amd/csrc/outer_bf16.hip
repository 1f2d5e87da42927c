// outer_bf16.hip -- a Dense layer over a whole flattened window on the bf16 matrix cores when its input X is exactly
// representable in bf16 (piano-roll frames kept as bytes): the kernel gradient dK[j,:] = sum_b X[b,j] G[b,:], and the forward
// product X.K as split-K partial sums (dense_window_fwd_bf16_kernel, at the end of the file)   (gfx950)
//
// Reference: h_w = Dense(relu)(flat(X)) of cl_vrnn (cl_vrnn/model.py:174-176): 88 * seq_length inputs, 88 outputs; this is
// what K.gradients forms for its kernel.  sparse_outer_kernel (sparse_proj.hip) walks the ~4 % of notes that are on: each
// workgroup re-stages every block of G and pays an LDS round trip per note; at 1024 batch rows (configuration 5) that is 80 us
// for a product of 4 GFLOP.  Dense on the matrix cores it is: X is ONE bf16 piece (exact), G is three (x = p0 + p1 + p2
// exactly), the three piece products are exact and accumulate in fp32 -- 3 MFMAs per tile and 32 batch rows, the same
// products an fp32 FMA chain forms, in another order.  Then the kernel streams X once; that is its bound.
//
// A workgroup owns 96 consecutive inputs (6 row tiles, one per wave) and all N <= 96 output columns; it walks the batch in
// stages of 32 rows: every thread moves 2 float4 of X and 2 of G per stage (requested one stage ahead), X goes to a bf16
// [k][column] image as it is, G as three piece images, double buffered; fragments come out with ds_read_b64_tr_b16 (both
// operands are k-major in memory: bf16_images.h).  Every output row is written once: no split-K slabs, no reduction.
// One more workgroup forms the column sums of G (the bias gradient) and gdot (see clv_sparse_outer).
#include "bf16_images.h"

namespace clv {

typedef float od_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned od_u32x4 __attribute__((ext_vector_type(4)));

// waves per workgroup = row tiles of its inputs: 6 (96 inputs), or 3 (48) where 96 would leave half of the CUs without a
// workgroup (configuration 3: 11264 inputs = 118 or 235 workgroups; every workgroup converts all of G, so fewer, larger ones
// are the better deal once the grid is full)
constexpr int OD_KS = 32;                      // batch rows per stage = one MFMA k-step
constexpr int OD_P = 192;                      // image pitch (bytes): 96 columns
constexpr int OD_IMG = OD_KS * OD_P;           // one image of a stage
constexpr int OD_BUF = 4 * OD_IMG;             // X + three pieces of G
constexpr int OD_LDS = 2 * OD_BUF;

struct OuterBf16Args {
  int Bn, nx, N, ldx, ldg, ldo;
  const void* X;       // float, or uint8 (XU8 kernels: ldx counts bytes)
  const float* G;
  float* out;          // [nx, ldo]
  float* colsum;       // [N] or null
  const float* Hact; const float* hbias; float* gdot; int ldh;      // see SparseOuterArgs
};

typedef __amdgpu_buffer_rsrc_t od_rsrc_t;
constexpr unsigned OD_OOB = 0x80000000u;
__device__ __forceinline__ float od_u2f(unsigned u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ float4 od_load4(od_rsrc_t r, unsigned voff) {
  const od_u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0);
  return make_float4(od_u2f(x[0]), od_u2f(x[1]), od_u2f(x[2]), od_u2f(x[3]));
}
// Frames kept as BYTES (round 6: the training step of the large-batch path never widens its piano-roll frames to float): four
// consecutive inputs are one dword; it travels raw in the .x of the register slot a float4 would take and is widened where
// the slot is consumed (every byte value is exactly a bf16 number).
__device__ __forceinline__ float4 od_load_u8x4(od_rsrc_t r, unsigned voff) {
  return make_float4(od_u2f((unsigned)__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, 0, 0)), 0.f, 0.f, 0.f);
}
__device__ __forceinline__ float4 od_widen(const float4& raw) {
  const float r0 = raw.x;                     // (a scalar copy first: bit_cast of a vector ELEMENT reads element 0, tests/test_host_logic.py)
  const unsigned v = __builtin_bit_cast(unsigned, r0);
  return make_float4((float)(v & 0xffu), (float)((v >> 8) & 0xffu), (float)((v >> 16) & 0xffu), (float)(v >> 24));
}

template <int OD_NW, bool XU8>
__global__ __launch_bounds__(64 * OD_NW) void dense_outer_bf16_kernel(OuterBf16Args a) {
  constexpr int OD_NT = 64 * OD_NW, OD_JT = 16 * OD_NW;
  extern __shared__ __attribute__((aligned(16))) char od_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntile = (a.nx + OD_JT - 1) / OD_JT;
  if ((int)blockIdx.x >= ntile) {
    // ---- the extra workgroup: colsum[c] = sum_b G[b,c], gdot[c] = sum_b (Hact[b,c] - hbias[c]) G[b,c] -------------------
    float2* red = reinterpret_cast<float2*>(od_lds);     // [2][OD_NW][64]
    const int n2 = a.N / 2;
    float2 cs = make_float2(0.f, 0.f), gd = make_float2(0.f, 0.f);
    if (lane < n2) {
      const float2 hb = a.gdot ? make_float2(a.hbias[2 * lane], a.hbias[2 * lane + 1]) : make_float2(0.f, 0.f);
      for (int b0 = wave; b0 < a.Bn; b0 += 8 * OD_NW) {
        float2 hv[8], gv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {                    // all loads of a round in flight (clamped rows, masked below)
          const int bb = min(b0 + i * OD_NW, a.Bn - 1);
          gv[i] = *reinterpret_cast<const float2*>(a.G + (size_t)bb * a.ldg + 2 * lane);
          hv[i] = a.gdot ? *reinterpret_cast<const float2*>(a.Hact + (size_t)bb * a.ldh + 2 * lane) : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float mk = b0 + i * OD_NW < a.Bn ? 1.f : 0.f;
          cs.x += gv[i].x * mk; cs.y += gv[i].y * mk;
          gd.x = fmaf((hv[i].x - hb.x) * mk, gv[i].x, gd.x);
          gd.y = fmaf((hv[i].y - hb.y) * mk, gv[i].y, gd.y);
        }
      }
    }
    red[wave * 64 + lane] = cs;
    red[(OD_NW + wave) * 64 + lane] = gd;
    __syncthreads();
    if (wave == 0 && lane < n2) {
      float2 t = make_float2(0.f, 0.f), u = make_float2(0.f, 0.f);
#pragma unroll
      for (int w = 0; w < OD_NW; ++w) {
        t.x += red[w * 64 + lane].x; t.y += red[w * 64 + lane].y;
        u.x += red[(OD_NW + w) * 64 + lane].x; u.y += red[(OD_NW + w) * 64 + lane].y;
      }
      if (a.colsum) { a.colsum[2 * lane] = t.x; a.colsum[2 * lane + 1] = t.y; }
      if (a.gdot) { a.gdot[2 * lane] = u.x; a.gdot[2 * lane + 1] = u.y; }
    }
    return;
  }

  // the images' padding (columns N..95 of G, inputs beyond nx) is zeroed once and never written
  for (int i = tid; i < OD_LDS / 16; i += OD_NT) reinterpret_cast<float4*>(od_lds)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int j0 = blockIdx.x * OD_JT;
  const int n4 = a.N / 4;
  // which float4s of a stage this thread moves: X 32 rows x 24, G 32 rows x n4; slot e = tid + 384 i.  Rows beyond the
  // batch fall outside the descriptors (the loads return 0), inputs beyond nx and idle slots get an out-of-range offset.
  constexpr unsigned XE = XU8 ? 1u : 4u;                  // bytes per element of X
  const od_rsrc_t r_x = (od_rsrc_t)__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.X), 0, (int)(unsigned)((size_t)a.Bn * a.ldx * XE), 0x00020000);
  const od_rsrc_t r_g = (od_rsrc_t)__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.G), 0, (int)(unsigned)((size_t)a.Bn * a.ldg * 4), 0x00020000);
  constexpr int XC4 = OD_JT / 4;                          // float4 columns of the X tile
  constexpr int XS = OD_KS * XC4 / OD_NT, GS = (OD_KS * 24 + OD_NT - 1) / OD_NT;      // float4 slots per thread: 2, and 2 or 4
  static_assert(XS * OD_NT == OD_KS * XC4, "X slots");
  unsigned xg[XS], gg[GS];
  int xl[XS], gl[GS];
  bool gok[GS];
#pragma unroll
  for (int i = 0; i < XS; ++i) {
    const int e = tid + OD_NT * i;
    const int rx = e / XC4, cx = e - XC4 * rx;
    xg[i] = j0 + 4 * cx < a.nx ? XE * (unsigned)(rx * a.ldx + j0 + 4 * cx) : OD_OOB;
    xl[i] = rx * OD_P + 8 * cx;
  }
#pragma unroll
  for (int i = 0; i < GS; ++i) {
    const int e = tid + OD_NT * i;
    gok[i] = e < OD_KS * n4;
    const int eg = gok[i] ? e : 0, rg = eg / n4, cg = eg - n4 * rg;
    gg[i] = gok[i] ? 4u * (unsigned)(rg * a.ldg + 4 * cg) : OD_OOB;
    gl[i] = rg * OD_P + 8 * cg;
  }
  const int nst = (a.Bn + OD_KS - 1) / OD_KS;
  // FOUR stages of operands in flight per thread (register sets 0..3, set = stage % 4): a stage of 18 MFMAs per wave is a
  // fraction of a trip to HBM, with one stage of lookahead every stage waited for memory.  Every request is unconditional -- a stage beyond the batch lies outside the descriptors and costs
  // nothing -- so the compiler's wait counts stay exact: the body below is four stages, straight-line.
  constexpr int DEPTH = 4;
  float4 xr[DEPTH][XS], gr[DEPTH][GS];
  auto load_stage = [&](float4 (&xq)[XS], float4 (&gq)[GS], int s) {
    const unsigned kx = XE * (unsigned)(s * OD_KS * a.ldx), kg = 4u * (unsigned)(s * OD_KS * a.ldg);
#pragma unroll
    for (int i = 0; i < XS; ++i) {
      const unsigned xo = xg[i] == OD_OOB ? OD_OOB : xg[i] + kx;
      xq[i] = XU8 ? od_load_u8x4(r_x, xo) : od_load4(r_x, xo);
    }
#pragma unroll
    for (int i = 0; i < GS; ++i) gq[i] = od_load4(r_g, gg[i] == OD_OOB ? OD_OOB : gg[i] + kg);
  };
  auto store_stage = [&](const float4 (&xq)[XS], const float4 (&gq)[GS], int s) {
    char* buf = od_lds + (s & 1) * OD_BUF;
#pragma unroll
    for (int i = 0; i < XS; ++i) img_put4<1>(buf + xl[i], 0, XU8 ? od_widen(xq[i]) : xq[i]);
#pragma unroll
    for (int i = 0; i < GS; ++i)
      if (gok[i]) img_put4<3>(buf + OD_IMG + gl[i], OD_IMG, gq[i]);
  };
  __syncthreads();                 // the zeroes are in place
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) load_stage(xr[d], gr[d], d);
  store_stage(xr[0], gr[0], 0);
  load_stage(xr[0], gr[0], DEPTH);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  od_f32x4 acc[6];
#pragma unroll
  for (int n = 0; n < 6; ++n) acc[n] = od_f32x4{0.f, 0.f, 0.f, 0.f};
  const int fo = img_frag_lane_offset(OD_P, lane);
  // one stage: the products of stage s out of buffer s & 1; stage s + 1 (set K1) into the other buffer (last read in stage
  // s - 1: every wave has passed that stage's barrier); the request for stage s + 1 + DEPTH into the set that just emptied
  auto stage = [&](int s, float4 (&xq)[XS], float4 (&gq)[GS]) {
    const char* buf = od_lds + (s & 1) * OD_BUF;
    const img_bf16x8 ax = img_frag(buf, OD_P, 16 * wave, fo);
#pragma unroll
    for (int n = 0; n < 6; ++n) {
      img_bf16x8 b[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) b[p] = img_frag(buf + (1 + p) * OD_IMG, OD_P, 16 * n, fo);
#pragma unroll
      for (int p = 2; p >= 0; --p) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax, b[p], acc[n], 0, 0, 0);
    }
    store_stage(xq, gq, s + 1);
    load_stage(xq, gq, s + 1 + DEPTH);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // (not __syncthreads: the requests stay in flight)
  };
  for (int s = 0; s < nst; s += DEPTH) {         // (stages beyond nst: zero operands, the accumulators do not move)
    stage(s, xr[1], gr[1]);
    stage(s + 1, xr[2], gr[2]);
    stage(s + 2, xr[3], gr[3]);
    stage(s + 3, xr[0], gr[0]);
  }
  // C/D layout: column = lane & 15, row = 4 (lane >> 4) + register
#pragma unroll
  for (int n = 0; n < 6; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = j0 + 16 * wave + 4 * (lane >> 4) + r, c = 16 * n + (lane & 15);
      if (j < a.nx && c < a.N) a.out[(size_t)j * a.ldo + c] = acc[n][r];
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Forward: part[c][b][:] = sum over the inputs i of chunk c of X[b,i] K[i,:]   (h_w = relu(sum_c part + bias) is formed by the
// consumer: vrnn_label_fwd_x_kernel sums the chunks of its batch row, label_head.hip).
// The note-walking form gathers ~4 % of K's rows PER BATCH ROW from L2 (config 5: 1000 row gathers of 352 bytes per batch row,
// 360 MB through L2, 114 us): every row of K is fetched by ~45 batch rows.  Dense, K is one bf16-piece product per 32 inputs:
// a workgroup owns 16 NW batch rows (a wave's A operand is its 16 rows x 32 inputs straight from HBM: X is row-major, the k
// index is contiguous, one piece, exact) and a chunk of the inputs; K's 32 x N slab of a stage goes through LDS as three
// piece images shared by the waves.  Split-K because the output is tiny (B x 88): chunks = 256 / row blocks.
// ---------------------------------------------------------------------------------------------------------------------------
struct WindowFwdArgs {
  int Bn, nx, N, ldx, ldk, chunk;     // chunk: inputs per workgroup (a multiple of 32)
  const void* X;                      // float, or uint8 (XU8 kernel: ldx counts bytes)
  const float* K;
  float* part;                        // [chunks][Bn][N]
};
constexpr int WF_NW = 4, WF_NT = 64 * WF_NW;
constexpr int WF_BUF = 3 * OD_IMG, WF_LDS = 2 * WF_BUF;

template <bool XU8>
__global__ __launch_bounds__(WF_NT) void dense_window_fwd_bf16_kernel(WindowFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char od_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < WF_LDS / 16; i += WF_NT) reinterpret_cast<float4*>(od_lds)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int b0 = (blockIdx.x * WF_NW + wave) * 16;
  const int i0 = blockIdx.y * a.chunk, i1 = min(a.nx, i0 + a.chunk);
  const int nst = (i1 - i0 + OD_KS - 1) / OD_KS;
  const int n4 = a.N / 4;
  constexpr unsigned XE = XU8 ? 1u : 4u;
  const od_rsrc_t r_x = (od_rsrc_t)__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.X), 0, (int)(unsigned)((size_t)a.Bn * a.ldx * XE), 0x00020000);
  // K's descriptor ends with the chunk: the slab rows of the last stage that belong to the next chunk read as zeros
  const od_rsrc_t r_k = (od_rsrc_t)__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.K), 0, (int)(unsigned)((size_t)i1 * a.ldk * 4), 0x00020000);
  // A operand of this lane: row b0 + (lane & 15), inputs i0 + 32 s + 8 (lane >> 4) .. + 7.  An 8-group beyond the chunk is
  // multiplied by zero rows of K, one beyond nx must not be read (it would be the next row's data): nx % 8 == 0, so a group is
  // whole; rows beyond the batch fall outside the descriptor.
  const unsigned xa = XE * (unsigned)((b0 + (lane & 15)) * a.ldx + i0 + 8 * (lane >> 4));
  constexpr int KS_ = (OD_KS * 24 + WF_NT - 1) / WF_NT;       // float4 slots of K's slab per thread (3)
  unsigned kg[KS_];
  int kl[KS_];
  bool kok[KS_];
#pragma unroll
  for (int i = 0; i < KS_; ++i) {
    const int e = tid + WF_NT * i;
    kok[i] = e < OD_KS * n4;
    const int ek = kok[i] ? e : 0, rk = ek / n4, ck = ek - n4 * rk;
    kg[i] = kok[i] ? 4u * (unsigned)((i0 + rk) * a.ldk + 4 * ck) : OD_OOB;
    kl[i] = rk * OD_P + 8 * ck;
  }
  constexpr int DEPTH = 4;
  float4 xr[DEPTH][2], kr[DEPTH][KS_];
  auto load_stage = [&](float4 (&xq)[2], float4 (&kq)[KS_], int s) {
    const bool in = i0 + OD_KS * s + 8 * (lane >> 4) < a.nx;
    const unsigned xo = in ? xa + XE * (unsigned)(OD_KS * s) : OD_OOB;
    if (XU8) {       // eight inputs = two dwords, raw in the .x of the two slots
      xq[0] = od_load_u8x4(r_x, xo);
      xq[1] = od_load_u8x4(r_x, in ? xo + 4u : OD_OOB);
    } else {
      xq[0] = od_load4(r_x, xo);
      xq[1] = od_load4(r_x, in ? xo + 16u : OD_OOB);
    }
    const unsigned ko = 4u * (unsigned)(OD_KS * s * a.ldk);
#pragma unroll
    for (int i = 0; i < KS_; ++i) kq[i] = od_load4(r_k, kg[i] == OD_OOB ? OD_OOB : kg[i] + ko);
  };
  auto store_stage = [&](const float4 (&kq)[KS_], int s) {
    char* buf = od_lds + (s & 1) * WF_BUF;
#pragma unroll
    for (int i = 0; i < KS_; ++i)
      if (kok[i]) img_put4<3>(buf + kl[i], OD_IMG, kq[i]);
  };
  __syncthreads();
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) load_stage(xr[d], kr[d], d);
  store_stage(kr[0], 0);
  od_f32x4 acc[6];
#pragma unroll
  for (int n = 0; n < 6; ++n) acc[n] = od_f32x4{0.f, 0.f, 0.f, 0.f};
  const int fo = img_frag_lane_offset(OD_P, lane);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  // stage s: its A operand sits in set s % 4 (xs), its K images in buffer s & 1; K of stage s + 1 (set xn / kn) goes to the
  // other buffer; then the request for stage s + 4 into the set that held stage s
  auto stage = [&](int s, float4 (&xs)[2], float4 (&ks)[KS_], const float4 (&kn)[KS_]) {
    const char* buf = od_lds + (s & 1) * WF_BUF;
    const float4 x0 = XU8 ? od_widen(xs[0]) : xs[0], x1 = XU8 ? od_widen(xs[1]) : xs[1];
    const od_u32x4 au = {bf16_pack2(x0.x, x0.y), bf16_pack2(x0.z, x0.w), bf16_pack2(x1.x, x1.y), bf16_pack2(x1.z, x1.w)};
    const img_bf16x8 ax = __builtin_bit_cast(img_bf16x8, au);
#pragma unroll
    for (int n = 0; n < 6; ++n) {
      img_bf16x8 b[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) b[p] = img_frag(buf + p * OD_IMG, OD_P, 16 * n, fo);
#pragma unroll
      for (int p = 2; p >= 0; --p) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax, b[p], acc[n], 0, 0, 0);
    }
    store_stage(kn, s + 1);
    load_stage(xs, ks, s + DEPTH);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  for (int s = 0; s < nst; s += DEPTH) {         // (stages beyond the chunk: K reads as zeros, the accumulators do not move)
    stage(s, xr[0], kr[0], kr[1]);
    stage(s + 1, xr[1], kr[1], kr[2]);
    stage(s + 2, xr[2], kr[2], kr[3]);
    stage(s + 3, xr[3], kr[3], kr[0]);
  }
  float* out = a.part + (size_t)blockIdx.y * a.Bn * a.N;
#pragma unroll
  for (int n = 0; n < 6; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int b = b0 + 4 * (lane >> 4) + r, c = 16 * n + (lane & 15);
      if (b < a.Bn && c < a.N) out[(size_t)b * a.N + c] = acc[n][r];
    }
}

}  // namespace clv

extern "C" int clv_dense_outer_bf16_supported(int Bn, int nx, int N, int ldx, int ldg) {
  return Bn > 0 && nx > 0 && N >= 4 && N <= 96 && N % 4 == 0 && nx % 4 == 0 && ldx % 4 == 0 && ldg % 4 == 0 && ldx >= nx && ldg >= N &&
         (size_t)Bn * (size_t)ldx * 4 < 0x80000000ull && (size_t)Bn * (size_t)ldg * 4 < 0x80000000ull;
}

extern "C" int clv_dense_outer_bf16(int Bn, int nx, int N, const void* X, int x_u8, int ldx, const float* G, int ldg, float* out, int ldo,
                                    float* colsum, const float* Hact, int ldh, const float* hbias, float* gdot, void* stream) {
  using namespace clv;
  if (!clv_dense_outer_bf16_supported(Bn, nx, N, ldx, ldg) || !X || !G || !out || ldo < N) return CLV_EINVAL;
  if (((uintptr_t)X) % (x_u8 ? 4 : 16) != 0 || ((uintptr_t)G) % 16 != 0) return CLV_EINVAL;
  if (gdot && (!Hact || !hbias || ldh < N || ldh % 2 != 0 || ((uintptr_t)Hact) % 8 != 0)) return CLV_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  OuterBf16Args a{Bn, nx, N, ldx, ldg, ldo, X, G, out, colsum, Hact, hbias, gdot, ldh};
  const int extra = (colsum || gdot) ? 1 : 0;
  ProfScope p("dense_outer_bf16", s);
  auto go = [&](auto kern, int inputs, int threads) -> int {
    if (int e = allow_dynamic_lds(reinterpret_cast<const void*>(kern), OD_LDS)) return e;
    hipLaunchKernelGGL(kern, dim3((nx + inputs - 1) / inputs + extra), dim3(threads), OD_LDS, s, a);
    return launch_status();
  };
  // 96 inputs per workgroup (6 waves) from 100 workgroups on, else 48 (3).  Round 6: 100, not 200 -- with byte frames a row's piece
  // of a stage is then 96 bytes instead of 48 (configuration 3, 118 workgroups: 13.1 -> 11.1 us, profiles/r06_small_ab.txt).
  // ONE rule for float and byte frames: the two forms sum the column sums / gdot over the batch in different orders, and the
  // byte kernels promise the float kernels' results bit for bit.
  if ((nx + 95) / 96 >= 100) return x_u8 ? go(dense_outer_bf16_kernel<6, true>, 96, 384) : go(dense_outer_bf16_kernel<6, false>, 96, 384);
  return x_u8 ? go(dense_outer_bf16_kernel<3, true>, 48, 192) : go(dense_outer_bf16_kernel<3, false>, 48, 192);
}

// ---- forward ------------------------------------------------------------------------------------------------------------------
static int window_fwd_chunk(int Bn, int nx) {
  const int rb = (Bn + 16 * clv::WF_NW - 1) / (16 * clv::WF_NW);
  int chunks = (256 + rb - 1) / rb;                       // one workgroup per CU
  if (chunks > nx / 64) chunks = nx / 64;                 // ... of at least two stages
  if (chunks < 1) chunks = 1;
  const int chunk = ((nx + chunks - 1) / chunks + 31) / 32 * 32;
  return chunk;
}
extern "C" int clv_dense_window_fwd_bf16_supported(int Bn, int nx, int N, int ldx, int ldk) {
  return Bn > 0 && nx >= 8 && nx % 8 == 0 && N >= 4 && N <= 96 && N % 4 == 0 && ldx % 4 == 0 && ldk % 4 == 0 && ldx >= nx && ldk >= N &&
         (size_t)Bn * (size_t)ldx * 4 < 0x80000000ull && (size_t)nx * (size_t)ldk * 4 < 0x80000000ull;
}
extern "C" int clv_dense_window_fwd_bf16_splits(int Bn, int nx) {
  if (Bn <= 0 || nx <= 0) return 0;                       // (a size query never divides by zero: tests/test_host_logic.py)
  const int chunk = window_fwd_chunk(Bn, nx);
  return (nx + chunk - 1) / chunk;
}
extern "C" size_t clv_dense_window_fwd_bf16_workspace_bytes(int Bn, int nx, int N) {
  if (Bn <= 0 || nx <= 0 || N <= 0) return 0;
  return (size_t)clv_dense_window_fwd_bf16_splits(Bn, nx) * Bn * N * sizeof(float);
}
extern "C" int clv_dense_window_fwd_bf16(int Bn, int nx, int N, const void* X, int x_u8, int ldx, const float* K, int ldk, float* part,
                                         size_t part_bytes, void* stream) {
  using namespace clv;
  if (!clv_dense_window_fwd_bf16_supported(Bn, nx, N, ldx, ldk) || !X || !K) return CLV_EINVAL;
  if (((uintptr_t)X) % (x_u8 ? 4 : 16) != 0 || ((uintptr_t)K) % 16 != 0) return CLV_EINVAL;
  if (!part || part_bytes < clv_dense_window_fwd_bf16_workspace_bytes(Bn, nx, N)) return CLV_EWORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int chunk = window_fwd_chunk(Bn, nx);
  WindowFwdArgs a{Bn, nx, N, ldx, ldk, chunk, X, K, part};
  ProfScope p("dense_window_fwd_bf16", s);
  const dim3 grid((Bn + 16 * WF_NW - 1) / (16 * WF_NW), (nx + chunk - 1) / chunk);
  if (x_u8) {
    if (int e = allow_dynamic_lds(reinterpret_cast<const void*>(dense_window_fwd_bf16_kernel<true>), WF_LDS)) return e;
    hipLaunchKernelGGL(dense_window_fwd_bf16_kernel<true>, grid, dim3(WF_NT), WF_LDS, s, a);
  } else {
    if (int e = allow_dynamic_lds(reinterpret_cast<const void*>(dense_window_fwd_bf16_kernel<false>), WF_LDS)) return e;
    hipLaunchKernelGGL(dense_window_fwd_bf16_kernel<false>, grid, dim3(WF_NT), WF_LDS, s, a);
  }
  return launch_status();
}
