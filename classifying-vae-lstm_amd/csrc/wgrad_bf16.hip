// wgrad_bf16.hip -- every kernel gradient of one LSTM in one pass over dz, on the bf16 matrix cores from bf16 piece
// products: 6 of the 9 piece pairs of an fp32 x fp32 product (what is dropped is below 2^-25 of it, about the one rounding
// an fp32 multiply makes -- fp32-rounding accuracy, NOT exact), all pairs for byte-valued frames (gfx950).
//
//   dK_x [nx,4H] = X^T . dz      (input frames; binary piano-roll, or any values)
//   dU   [nh,4H] = H'^T . dz     (H'_k = h of the previous step: row k-1 of hs, zero at window starts)
//   dK_z [nz,4H] = Z^T . dz      (latent columns of the decoder input, may be absent)
// over K = B*T rows.  cl_vrnn/model.py:196-199,225-228 (the LSTM layers whose kernels these are); the products are
// what Keras' backward pass of `K.dot(x, kernel)` / `K.dot(h, recurrent_kernel)` computes.
//
// Why not the f32 MFMA (gemm.hip): v_mfma_f32_16x16x4_f32 runs at 1/16 of the bf16 rate.  An fp32 number is the exact
// sum of three bf16 numbers (8 + 8 + 8 mantissa bits, same exponent range), and a product of two bf16 numbers is exact
// in fp32, so  a.b = sum_{i,j} a_i.b_j  over the 3 x 3 piece pairs has EXACT partial products that accumulate in fp32 like
// any other summation order of the same products.  The three pairs with i + j >= 3 (a1.b2, a2.b1, a2.b2) together are below
// 2^-25 |a.b| -- less than the ONE rounding an fp32 multiply of a and b makes -- and are left out (round 4): 6 bf16 MFMAs
// (96 cycles per 16x16 tile and 32 k) replace 8 f32 MFMAs (256 cycles).  -DWB_PRODUCTS=9 keeps all nine (measured, us per
// launch: 36.0 -> 33.8 at K = 32768, 218 -> 196 at K = 262144 with six; profiles/r04_out_head_log.txt).  Frames that are
// exactly representable in bf16 (0/1 piano-roll, any uint8) need one piece: their 3 MFMAs are all kept, those products
// stay exact.  The kernel is then bound by its data-moving waves (profiles/r04_mx_log.txt section 13), not by the matrix pipe.
//
// Decomposition: a workgroup (4 waves, one per SIMD) owns ALL output rows x half of the 4H columns (176 = 11 column
// tiles, padded to 12) for K / splits rows; accumulators stay in registers (up to 168 per lane).  Per 32-row stage the
// operands go global -> registers -> (split into pieces) -> LDS as bf16 [k][column] images, double buffered; fragments
// come out of LDS with ds_read_b64_tr_b16 (both operands are k-major in memory, the MFMA wants k along the lane's
// registers).  Partial sums leave as one [rows,176] slab per workgroup into the deferred split-K reduction.
#include "reduce_job.h"

namespace clv {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WB_NT = 512;           // threads: 8 waves as 2 (row halves) x 4 (column quarters), two per SIMD
constexpr int WB_KS = 32;            // rows of dz per stage = one bf16 MFMA step
constexpr int WB_NC = 176;           // dz columns per workgroup
constexpr int WB_XT = 6;             // row tiles of the x block (nx <= 96)
// LDS row pitches in bytes.  A transposed read takes 4 rows x 32 bytes per 16-lane group and the two groups of a
// 32-lane half are 8 rows apart: with a pitch of 8 banks more than a multiple of 64 (dz: 416 B = 104 banks) or 48 banks
// (192 B) the 4 rows of a group fall on different banks; the second group lands on the first one's (2-way, unavoidable
// with a linear pitch: 8 rows x 8 banks would have to tile all 64 banks AND repeat after 8 rows).
// WB_SWZ (round 4): the 32-byte segments of image rows 8..15 and 24..31 swapped pairwise (address bit 5 XORed), so that the
// two 16-lane groups of a transposed read that are 8 rows apart fall on different banks too; needs an even number of
// 32-byte segments per row (pitches 448 / 320 instead of 416 / 288).  Measured (tools/sessions/r04_wbswz.sh): bank-conflict
// cycles per LDS-active cycle 0.37 -> 0.16, LDS-active cycles -24 % -- and the launch time does not move (32.5 against
// 32.6 us at K = 32768, 193 against 193 at 262144): the LDS was never what a stage waited for.  Kept on (-DWB_SWZ=0: linear).
#ifndef WB_SWZ
#define WB_SWZ 1
#endif
#define WB_SWZ_OF(r) (WB_SWZ ? ((((r) >> 3) & 1) << 5) : 0)
constexpr int WB_DZP = WB_SWZ ? 448 : 416;          // dz image: 192 columns + 16 pad
constexpr int WB_AP = 192;           // x image: 96 columns;  h image: 16 * HM columns, pitch below

template <int HM> struct WbGeo {
  static constexpr int HPB = HM == 6 ? 192 : (WB_SWZ ? 320 : 288);            // h image pitch (bytes): 96 columns, or 128 + 16 pad (72 banks)
  static constexpr int DZ_BYTES = 3 * WB_KS * WB_DZP;
  static constexpr int H_BYTES = 3 * WB_KS * HPB;
  template <int XP> static constexpr int x_bytes() { return XP * WB_KS * WB_AP; }
  template <int XP> static constexpr int buf_bytes() { return DZ_BYTES + H_BYTES + x_bytes<XP>(); }
  template <int XP> static constexpr int lds_bytes() { return 2 * buf_bytes<XP>(); }
};

struct WgradArgs {
  int K, N, kc;                      // rows, columns (352), rows per split
  const void* X; int ldx, nx, x_u8;      // frames: float rows, or (x_u8) the bytes themselves, ldx in bytes (one bf16 piece either way)
  const float* H; int ldh, nh, h_shift, h_zero_period;
  const float* Z; int ldz, nz;
  const float* dz; int lddz;
  float* partial;                    // [splits][nx + nh + nz][N]
};

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// x = p0 + p1 + p2 exactly, two values at a time: common.h (bf16_split_pair: 7 vector instructions per pair)
__device__ __forceinline__ unsigned pack2(float lo, float hi) { return bf16_pack2(lo, hi); }
__device__ __forceinline__ void split_pair(float a, float b, unsigned (&piece)[3]) { bf16_split_pair(a, b, piece); }
// 4 consecutive columns of one image row: one 8-byte LDS store per piece image
#ifndef WB_PRODUCER_PRIO
#define WB_PRODUCER_PRIO 1
#endif
#ifndef WB_PRODUCTS
#define WB_PRODUCTS 6    // piece pairs of an h . dz product: 6 = those with pa + pb <= 2 (default), 9 = all (-DWB_PRODUCTS=9)
#endif
// (round 5 tried the MFMA waves converting the h and x rows themselves: spills or no gain; the patch is
// tools/experiments/r05_wgrad_symmetric_waves.patch, the numbers profiles/r05_wgrad_symmetric_ab.txt.  The shipped schedule is
// round 4's: waves 0-3 issue MFMAs, waves 4-7 move and convert every operand.)
#ifndef WB_ABLATE
#define WB_ABLATE 0      // measurement builds (tools/build_variant.sh): 1 = no piece splitting, 2 = one MFMA term of nine,
#endif                   // 3 = no global loads after the first stages.  Results are wrong by design.
template <int NP>
__device__ __forceinline__ void put4(char* at, int piece_bytes, const float4& v) {
  unsigned lo[3], hi[3];
  if (NP == 1 || WB_ABLATE == 1) { lo[0] = lo[1] = lo[2] = pack2(v.x, v.y); hi[0] = hi[1] = hi[2] = pack2(v.z, v.w); }
  else { split_pair(v.x, v.y, lo); split_pair(v.z, v.w, hi); }
#pragma unroll
  for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x2*>(at + p * piece_bytes) = u32x2{lo[p], hi[p]};
}

// the 8 k-values x 16 columns fragment of a bf16 [k][column] image (k rows 0..31 of the stage, columns col0..col0+15):
// lane l = 16 g + i gets column col0 + i, k = 8 g + j in element j -- the layout of both operands of
// v_mfma_f32_16x16x32_bf16 when the image holds the operand k-major.  `lane_off` = frag_lane_offset(pitch, lane).
__device__ __forceinline__ int frag_lane_offset(int pitch, int lane) {
  const int g = lane >> 4, i = lane & 15;
  return (8 * g + (i >> 2)) * pitch + 8 * (i & 3);
}
__device__ __forceinline__ bf16x8 frag(const char* img, int pitch, int col0, int lane_off, int swz = 0) {
  const char* p = img + ((lane_off + 2 * col0) ^ swz);
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * pitch));
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// -DWB_STAMPS: every wave of workgroup (0, 0) records the shader clock at the top of each stage and at its arrival at the
// stage barrier (tools/wgrad_stamps.py)
#ifdef WB_STAMPS
__device__ unsigned long long g_wb_stamps[64][8][2];
// ... and every workgroup's wave 0 (a consumer) the 100 MHz clock at: entry, LDS cleared, first stage ready, stages done, slab stored
__device__ unsigned long long g_wb_wg[1024][5];
#define WBWG(k)                                                                                                      \
  do {                                                                                                               \
    if (threadIdx.x == 0) {                                                                                          \
      unsigned long long t__;                                                                                        \
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__));                                          \
      const unsigned w__ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;                            \
      if (w__ < 1024) g_wb_wg[w__][k] = t__;                                                                          \
    }                                                                                                                \
  } while (0)
#define WBSTAMP(slot, s_)                                                                              \
  do {                                                                                                  \
    unsigned long long t__;                                                                             \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__));                                   \
    if (blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x & 63) == 0 && (s_) < 64) g_wb_stamps[s_][threadIdx.x >> 6][slot] = t__; \
  } while (0)
#else
#define WBSTAMP(slot, s_)
#define WBWG(k)
#endif

// Stage barrier: this wave's LDS traffic is done, its global loads are NOT waited for (__syncthreads() would add
// s_waitcnt vmcnt(0) and drain the producers' prefetch at every stage).
__device__ __forceinline__ void stage_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float4 wb_widen(const float4& raw) {      // four byte frames that travelled as one dword in .x
  const float r0 = raw.x;                     // (a scalar copy first: bit_cast of a vector ELEMENT reads element 0, tests/test_host_logic.py)
  const unsigned v = __builtin_bit_cast(unsigned, r0);
  return make_float4((float)(v & 0xffu), (float)((v >> 8) & 0xffu), (float)((v >> 16) & 0xffu), (float)(v >> 24));
}

template <int HM, int XP, bool XU8 = false>
__device__ __forceinline__ void wgrad_body(const WgradArgs& a) {
  static_assert(!XU8 || XP == 1, "byte frames are one bf16 piece");
  using G = WbGeo<HM>;
  constexpr int HPB = G::HPB, BUF = G::template buf_bytes<XP>(), LDS_ALL = G::template lds_bytes<XP>();
  constexpr int HT = HM / 2;                     // h row tiles per consumer wave
  constexpr int XT = WB_XT / 2;                  // x row tiles per consumer wave
  constexpr int NTW = 6;                         // column tiles per consumer wave (2 halves x 6 = 12: 11 + 1 padding)
  constexpr int NP_T = WB_NT / 2;                // producer threads
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Waves w and w + 4 share a SIMD.  Waves 0-3 only issue MFMAs (and the LDS reads that feed them): 2 row halves x 2
  // column halves, all accumulators in registers.  Waves 4-7 only move data: global loads of stage s + 2, conversion
  // of stage s + 1 into bf16 pieces, LDS stores -- their VALU work runs in the shadow of the partner's MFMAs.
  const bool producer = wave >= 4;
  // grid = (splits, 2): the two column halves of a row range are 128 workgroup ids apart, i.e. on the same XCD under
  // round-robin placement, so the second read of X / H hits that XCD's L2 (speed only)
  const int n0 = blockIdx.y * WB_NC;             // first dz column of this workgroup
  const int split = blockIdx.x;
  const int k_begin = split * a.kc, k_end = min(a.K, k_begin + a.kc);
  const int nstage = (k_end - k_begin + WB_KS - 1) / WB_KS;

  WBWG(0);
  // The images' padding (x columns nx..95, h columns nh+nz.., dz columns 176..207) only reaches output rows / columns
  // that are never stored; everything is zeroed once so that every MFMA input is a finite number.
  for (int i = tid; i < LDS_ALL / 16; i += WB_NT) reinterpret_cast<float4*>(lds)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  WBWG(1);

  if (producer) {
    // the data movers go first when both waves of a SIMD have an instruction ready: the MFMA wave fills the gaps, and it
    // has slack (34.1 -> 32.8 us per launch at K = 32768, 216 -> 209 us at K = 262144; -DWB_PRODUCER_PRIO=0 to compare)
    __builtin_amdgcn_s_setprio(WB_PRODUCER_PRIO);
    // ---- which elements of a stage this thread moves (the same for every stage) ------------------------------------
    // dz: 32 rows x 44 float4 = 1408 slots;  h, x: 32 x (n/4);  z: 32 x nz scalars; 256 producer threads take slot
    // pt + 256 i.  Per slot: the byte offset from the stage's first row in global memory and the byte offset in the
    // LDS image.  Only the last slot of each kind can lie beyond the tile (its load is clamped, its store skipped).
    const int pt = tid - NP_T;
    constexpr int DZ_L = 6, A_L = 3, Z_L = HM == 8 ? 4 : 1;
    static_assert(A_L * NP_T >= WB_KS * (96 / 4), "h / x slots: 32 rows x 96 columns (clv_lstm_wgrad_supported: nh, nx <= 96)");
    static_assert(Z_L * NP_T >= WB_KS * (HM == 8 ? 32 : 8), "z slots: 32 rows x 8 (32) latent columns (wgrad_wide)");
    const int nh4 = a.nh / 4, nx4 = a.nx / 4;
    unsigned dz_g[DZ_L], h_g[A_L], x_g[A_L], z_g[Z_L];
    int dz_l[DZ_L], h_l[A_L], x_l[A_L], z_l[Z_L];          // LDS offsets; bit 30: row 0 of the stage (h), bit 31: idle slot
    constexpr int ROW0 = 1 << 30, ROW16 = 1 << 29, IDLE = 1 << 31, OFFM = ROW16 - 1;
#pragma unroll
    for (int i = 0; i < DZ_L; ++i) {
      const int e = pt + i * NP_T, ok = e < WB_KS * 44, ec = ok ? e : 0, r = ec / 44, c4 = ec % 44;
      dz_g[i] = 4u * (unsigned)(r * a.lddz + n0 + 4 * c4);
      dz_l[i] = ((r * WB_DZP + 8 * c4) ^ WB_SWZ_OF(r)) | (ok ? 0 : IDLE);
    }
#pragma unroll
    for (int i = 0; i < A_L; ++i) {
      const int e = pt + i * NP_T;
      const int okh = e < WB_KS * nh4, eh = okh ? e : 0, rh = eh / nh4, ch = eh % nh4;
      const int okx = e < WB_KS * nx4, ex = okx ? e : 0, rx = ex / nx4, cx = ex % nx4;
      h_g[i] = 4u * (unsigned)(rh * a.ldh + 4 * ch);
      h_l[i] = ((rh * HPB + 8 * ch) ^ WB_SWZ_OF(rh)) | (rh == 0 ? ROW0 : 0) | (rh == 16 ? ROW16 : 0) | (okh ? 0 : IDLE);
      x_g[i] = (XU8 ? 1u : 4u) * (unsigned)(rx * a.ldx + 4 * cx);
      x_l[i] = ((rx * WB_AP + 8 * cx) ^ WB_SWZ_OF(rx)) | (okx ? 0 : IDLE);
    }
#pragma unroll
    for (int i = 0; i < Z_L; ++i) {
      const int e = pt + i * NP_T, nz1 = max(a.nz, 1), ok = a.nz > 0 && e < WB_KS * a.nz, ez = ok ? e : 0;
      z_g[i] = 4u * (unsigned)((ez / nz1) * a.ldz + ez % nz1);
      z_l[i] = (((ez / nz1) * HPB + 2 * (a.nh + ez % nz1)) ^ WB_SWZ_OF(ez / nz1)) | (ok ? 0 : IDLE);
    }
    // A stage is "interior" when all of its 32 rows exist, the shifted H rows exist, and a window start can only be its
    // first row or its 17th (h_zero_period a multiple of 16 -- the reference's default seq_length is 16; k_begin is a
    // multiple of 32 by construction).  Interior stages take the fast path: no clamps, no masks, one select for the H
    // rows of a window start.
    const bool aligned = a.h_zero_period == 0 || a.h_zero_period % (WB_KS / 2) == 0;
    auto interior = [&](int s) { const int k0 = k_begin + s * WB_KS; return aligned && k0 + WB_KS <= k_end && k0 >= a.h_shift; };

    // Two stages of operands in flight per thread (register sets A and B): one stage of MFMAs (~1.5 us) is less than a
    // load takes under load, with one set the producers waited for memory every stage.
    struct Regs { float4 dz[DZ_L], h[A_L], x[A_L]; float z[Z_L]; };
    Regs ra, rb;
    auto load_stage = [&](Regs& q, int s) {        // nothing uses the values until store_stage()
      if (WB_ABLATE == 3 && s > 2) return;
      const int k0 = k_begin + s * WB_KS;
      if (interior(s)) {                           // uniform base + per-lane 32-bit offset
        const char* dzb = reinterpret_cast<const char*>(a.dz + (size_t)k0 * a.lddz);
        const char* hb = reinterpret_cast<const char*>(a.H + (size_t)(k0 - a.h_shift) * a.ldh);
        const char* xb = static_cast<const char*>(a.X) + (size_t)k0 * a.ldx * (XU8 ? 1 : 4);
#pragma unroll
        for (int i = 0; i < DZ_L; ++i) q.dz[i] = *reinterpret_cast<const float4*>(dzb + dz_g[i]);
#pragma unroll
        for (int i = 0; i < A_L; ++i) {
          q.h[i] = *reinterpret_cast<const float4*>(hb + h_g[i]);
          if (XU8) q.x[i].x = *reinterpret_cast<const float*>(xb + x_g[i]);       // (raw dword = four frames' bytes: widened in store_stage)
          else q.x[i] = *reinterpret_cast<const float4*>(xb + x_g[i]);
        }
        if (a.nz > 0) {
          const char* zb = reinterpret_cast<const char*>(a.Z + (size_t)k0 * a.ldz);
#pragma unroll
          for (int i = 0; i < Z_L; ++i) q.z[i] = *reinterpret_cast<const float*>(zb + z_g[i]);
        }
        return;
      }
      // first / last stage of the matrix: row indices clamped into it (the values of rows outside are masked later)
      auto row = [&](unsigned g, int ld, int shift, unsigned esz = 4u) { const int r = (int)(g / esz) / ld; return min(max(k0 + r - shift, 0), a.K - 1) - r; };
#pragma unroll
      for (int i = 0; i < DZ_L; ++i)
        q.dz[i] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(a.dz + (size_t)row(dz_g[i] - 4u * n0, a.lddz, 0) * a.lddz) + dz_g[i]);
#pragma unroll
      for (int i = 0; i < A_L; ++i) {
        q.h[i] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(a.H + (size_t)row(h_g[i], a.ldh, a.h_shift) * a.ldh) + h_g[i]);
        if (XU8) q.x[i].x = *reinterpret_cast<const float*>(static_cast<const char*>(a.X) + (size_t)row(x_g[i], a.ldx, 0, 1u) * a.ldx + x_g[i]);
        else q.x[i] = *reinterpret_cast<const float4*>(static_cast<const char*>(a.X) + (size_t)row(x_g[i], a.ldx, 0) * a.ldx * 4 + x_g[i]);
      }
      if (a.nz > 0) {
#pragma unroll
        for (int i = 0; i < Z_L; ++i)
          q.z[i] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.Z + (size_t)row(z_g[i], a.ldz, 0) * a.ldz) + z_g[i]);
      }
    };
    auto keep = [](const float4& v, bool live) { return live ? v : make_float4(0.f, 0.f, 0.f, 0.f); };
    auto store_stage = [&](const Regs& q, int s) {    // fp32 -> bf16 pieces -> LDS images of buffer s & 1
      const int k0 = k_begin + s * WB_KS;
      char* dzi = lds + (s & 1) * BUF;
      char* hi = dzi + G::DZ_BYTES;
      char* xi = hi + G::H_BYTES;
      const bool fast = interior(s);
      const bool wstart = a.h_zero_period != 0 && k0 % a.h_zero_period == 0;     // row 0 of the stage starts a window
      const bool wstart16 = a.h_zero_period != 0 && (k0 + WB_KS / 2) % a.h_zero_period == 0;      // ... row 16 does
#pragma unroll
      for (int i = 0; i < DZ_L; ++i) {
        const int r = (dz_l[i] & OFFM) / WB_DZP;
        const float4 v = fast ? q.dz[i] : keep(q.dz[i], k0 + r < k_end);
        if (i + 1 < DZ_L || !(dz_l[i] & IDLE)) put4<3>(dzi + (dz_l[i] & OFFM), WB_KS * WB_DZP, v);
      }
#pragma unroll
      for (int i = 0; i < A_L; ++i) {
        const int rh = (h_l[i] & OFFM) / HPB, rx = (x_l[i] & OFFM) / WB_AP, k = k0 + rh;
        // H'_k = h of the previous step; zero at the start of a window
        const bool live = fast ? !((wstart && (h_l[i] & ROW0)) || (wstart16 && (h_l[i] & ROW16)))
                               : (k < k_end && k >= a.h_shift && (a.h_zero_period == 0 || k % a.h_zero_period != 0));
        if (i + 1 < A_L || !(h_l[i] & IDLE)) put4<3>(hi + (h_l[i] & OFFM), WB_KS * HPB, keep(q.h[i], live));
        const float4 xv = XU8 ? wb_widen(q.x[i]) : q.x[i];
        if (i + 1 < A_L || !(x_l[i] & IDLE)) put4<XP>(xi + (x_l[i] & OFFM), WB_KS * WB_AP, fast ? xv : keep(xv, k0 + rx < k_end));
      }
      if (a.nz > 0) {
#pragma unroll
        for (int i = 0; i < Z_L; ++i) {
          unsigned p[3];
          const int r = (z_l[i] & OFFM) / HPB;
          split_pair((fast || k0 + r < k_end) ? q.z[i] : 0.f, 0.f, p);
          if (!(z_l[i] & IDLE)) {
#pragma unroll
            for (int w = 0; w < 3; ++w) *reinterpret_cast<unsigned short*>(hi + w * WB_KS * HPB + (z_l[i] & OFFM)) = (unsigned short)p[w];
          }
        }
      }
    };
    // stage s lives in set A for even s, set B for odd s
    if (nstage > 0) {
      load_stage(ra, 0);
      if (nstage > 1) load_stage(rb, 1);
      store_stage(ra, 0);
      if (nstage > 2) load_stage(ra, 2);
    }
    stage_barrier();
    // during stage s (consumers on buffer s & 1): convert stage s + 1 into the other buffer (last read in stage s - 1,
    // i.e. before the barrier every wave has passed), then request stage s + 3 into the registers that just emptied
    for (int s = 0; s < nstage; s += 2) {
      WBSTAMP(0, s);
      if (s + 1 < nstage) store_stage(rb, s + 1);
      if (s + 3 < nstage) load_stage(rb, s + 3);
      WBSTAMP(1, s);
      stage_barrier();
      if (s + 1 < nstage) {
        WBSTAMP(0, s + 1);
        if (s + 2 < nstage) store_stage(ra, s + 2);
        if (s + 4 < nstage) load_stage(ra, s + 4);
        WBSTAMP(1, s + 1);
        stage_barrier();
      }
    }
    return;
  }

  // ---- consumers ----------------------------------------------------------------------------------------------------
  const int mh = wave & 1, nh2 = wave >> 1;
  f32x4 accx[XT][NTW], acch[HT][NTW];
#pragma unroll
  for (int n = 0; n < NTW; ++n) {
#pragma unroll
    for (int m = 0; m < XT; ++m) accx[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < HT; ++m) acch[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int fo_dz = frag_lane_offset(WB_DZP, lane), fo_h = frag_lane_offset(HPB, lane), fo_x = frag_lane_offset(WB_AP, lane);
  const int swz = WB_SWZ_OF(lane >> 1);        // lane group g = lane >> 4 reads rows 8g ..: (g & 1) << 5
  stage_barrier();
  WBWG(2);
  for (int s = 0; s < nstage; ++s) {
    WBSTAMP(0, s);
    const char* dzi = lds + (s & 1) * BUF;
    const char* hi = dzi + G::DZ_BYTES;
    const char* xi = hi + G::H_BYTES;
    bf16x8 ax[XT][XP], ah[HT][3];
#pragma unroll
    for (int m = 0; m < XT; ++m)
#pragma unroll
      for (int p = 0; p < XP; ++p) ax[m][p] = frag(xi + p * WB_KS * WB_AP, WB_AP, 16 * (mh * XT + m), fo_x, swz);
#pragma unroll
    for (int m = 0; m < HT; ++m)
#pragma unroll
      for (int p = 0; p < 3; ++p) ah[m][p] = frag(hi + p * WB_KS * HPB, HPB, 16 * (mh * HT + m), fo_h, swz);
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
      bf16x8 b[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) b[p] = frag(dzi + p * WB_KS * WB_DZP, WB_DZP, 16 * (nh2 * NTW + n), fo_dz, swz);
#pragma unroll
      for (int m = 0; m < XT; ++m)
#pragma unroll
        for (int pa = 0; pa < XP; ++pa)
#pragma unroll
          for (int pb = 0; pb < 3; ++pb)
            accx[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[m][pa], b[pb], accx[m][n], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < HT; ++m)
#pragma unroll
        for (int pa = 0; pa < (WB_ABLATE == 2 ? 1 : 3); ++pa)
#pragma unroll
          for (int pb = 0; pb < (WB_ABLATE == 2 ? 1 : 3); ++pb)
            if (WB_PRODUCTS == 9 || pa + pb <= 2)
              acch[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[m][pa], b[pb], acch[m][n], 0, 0, 0);
    }
#ifdef WB_STAMPS
    asm volatile("" :: "v"(acch[HT - 1][NTW - 1][0]));       // the stage's last MFMA has delivered
#endif
    WBSTAMP(1, s);
    stage_barrier();
  }

  WBWG(3);
  // ---- slabs: accumulators -> LDS (row-major) -> 16-byte row pieces to the partial buffer ---------------------------
  // C/D layout of the MFMA: column = lane & 15, row = 4 (lane >> 4) + register.  (The producers are gone; the four
  // consumer waves stage in disjoint regions, no barrier needed: LDS operations of one wave execute in order.)
  constexpr int SP = 16 * NTW + 4;                // floats per staged row (96 columns + 4: the 4 row groups 2-way on banks)
  float* stg = reinterpret_cast<float*>(lds) + wave * (16 * (HT > XT ? HT : XT) * SP);
  const int rows_tot = a.nx + a.nh + a.nz;
  float* slab = a.partial + (size_t)split * rows_tot * a.N;
  const int c_loc = nh2 * 16 * NTW;               // first column of this wave inside the workgroup's 192
  const int colw = min(16 * NTW, WB_NC - c_loc);  // valid ones
  auto flush = [&](auto& acc, int tiles, int row_first, int row_limit, int slab_row0) {
#pragma unroll
    for (int m = 0; m < tiles; ++m)
#pragma unroll
      for (int n = 0; n < NTW; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) stg[(16 * m + 4 * (lane >> 4) + r) * SP + 16 * n + (lane & 15)] = acc[m][n][r];
    for (int e = lane; e < 16 * tiles * (4 * NTW); e += 64) {
      const int r = e / (4 * NTW), c4 = e % (4 * NTW), row = row_first + r;
      if (row < row_limit && 4 * c4 < colw) {
        const float4 v = *reinterpret_cast<const float4*>(&stg[r * SP + 4 * c4]);
        *reinterpret_cast<float4*>(slab + (size_t)(slab_row0 + row) * a.N + n0 + c_loc + 4 * c4) = v;
      }
    }
  };
  flush(accx, XT, mh * XT * 16, a.nx, 0);
  flush(acch, HT, mh * HT * 16, a.nh + a.nz, a.nx);
  WBWG(4);
}

template <int HM, int XP, bool XU8 = false>
__global__ __launch_bounds__(WB_NT) void lstm_wgrad_bf16_kernel(WgradArgs a) { wgrad_body<HM, XP, XU8>(a); }

// Two LSTMs in one launch (grid z): the encoder's and the decoder's gradients of a cl_vrnn step, whose dz both exist
// once the backward pass is through.  Launched one after the other at K = 32768 (configuration 3), each grid is one
// workgroup per CU with 8 stages -- 17 us of stages inside a 34 us launch (LDS clear and first loads in front, a 124 KB
// slab store behind, launch and drain around).  Together, with row ranges twice as long, the same 256 workgroups run 16
// stages each: the fixed part is paid once, and the split-K reduction reads half as many slabs.
template <int HM, int XP, bool XU8 = false>
__global__ __launch_bounds__(WB_NT) void lstm_wgrad_bf16_pair_kernel(WgradArgs a0, WgradArgs a1) {
  const WgradArgs a = blockIdx.z ? a1 : a0;        // (scalar selects: one copy of the body)
  wgrad_body<HM, XP, XU8>(a);
}

}  // namespace clv

// The producers' slot tables cover 32 rows x 96 columns of h and of x (A_L = 3 float4 slots per thread) and 32 x 8 z
// scalars in the 6-row-tile kernel, 32 x 32 in the 8-row-tile one: the wide kernel is taken when [h | z] needs more
// than 96 image columns OR more than 8 latent columns.
#ifdef WB_STAMPS
extern "C" int clv_debug_wb_stamps(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(clv::g_wb_stamps), sizeof(unsigned long long) * 64 * 8 * 2);
}
extern "C" int clv_debug_wb_wg(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(clv::g_wb_wg), sizeof(unsigned long long) * 1024 * 5);
}
#endif

static bool wgrad_wide(int nh, int nz) { return nh + nz > 96 || nz > 8; }

extern "C" int clv_lstm_wgrad_supported(int N, int nx, int nh, int nz, int x_exact_bf16) {
  if (!(N == 352 && nx > 0 && nx <= 96 && nx % 4 == 0 && nh > 0 && nh <= 96 && nh % 4 == 0 && nz >= 0 && nz <= 32 &&
        nh + nz <= 128))
    return 0;
  return !wgrad_wide(nh, nz) || x_exact_bf16;        // 8 h row tiles and three x pieces do not fit the LDS together
}

// rows per split: one workgroup per CU (2 column halves x 128 row ranges of whole 32-row stages); split_scale s: s times
// as many, shorter ranges (a grid of exactly one workgroup per CU needs a whole second round as soon as something else
// -- the gradient all-reduce's kernel -- holds a few CUs; 2 x 256 half-length workgroups lose only the share that is taken)
static int wgrad_kc(int K, int split_scale, int base_ranges = 128) {
  const int ranges = base_ranges * (split_scale < 1 ? 1 : split_scale);
  int kc = (K + ranges - 1) / ranges;
  return (kc + 31) / 32 * 32;
}
static int wgrad_splits(int K, int split_scale, int base_ranges = 128) {
  if (K <= 0) return 0;                                   // (size queries with no rows: 0 bytes, never a division by zero)
  const int kc = wgrad_kc(K, split_scale, base_ranges);
  return (K + kc - 1) / kc;
}

extern "C" size_t clv_lstm_wgrad_workspace_bytes(int K, int N, int nx, int nh, int nz, int split_scale) {
  return (size_t)wgrad_splits(K, split_scale) * (nx + nh + nz) * N * sizeof(float);
}
extern "C" int clv_lstm_wgrad(int K, int N, const void* X, int ldx, int nx, int x_exact_bf16,
                                 const float* H, int ldh, int nh, int h_shift, int h_zero_period,
                                 const float* Z, int ldz, int nz, const float* dz, int lddz,
                                 float* dKx, int ld_kx, float* dU, int ld_u, float* dKz, int ld_kz, float beta,
                                 int split_scale, void* ws, size_t ws_bytes, clv_reduce_job* job, void* stream) {
  using namespace clv;
  if (split_scale < 1 || split_scale > 8) return CLV_EINVAL;
  if (job) memset(job, 0, sizeof(*job));
  if (!clv_lstm_wgrad_supported(N, nx, nh, nz, x_exact_bf16) || K <= 0 || !X || !H || !dz || !dKx || !dU || (nz > 0 && (!Z || !dKz)))
    return CLV_EINVAL;
  const bool xu8 = x_exact_bf16 == CLV_FRAMES_U8;           // the frames as bytes (ldx in bytes): 4-byte aligned rows
  if (ldx % 4 || ldh % 4 || lddz % 4 || ((uintptr_t)H | (uintptr_t)dz) % 16 || ((uintptr_t)X) % (xu8 ? 4 : 16)) return CLV_EINVAL;
  if (clv_lstm_wgrad_workspace_bytes(K, N, nx, nh, nz, split_scale) > ws_bytes || !ws || ((uintptr_t)ws) % 16) return CLV_EWORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int splits = wgrad_splits(K, split_scale);
  const int kc = wgrad_kc(K, split_scale);
  WgradArgs a{K, N, kc, X, ldx, nx, xu8, H, ldh, nh, h_shift, h_zero_period, Z, ldz, nz, dz, lddz, (float*)ws};
  const bool wide = wgrad_wide(nh, nz);
  {
    ProfScope p("lstm_wgrad_bf16", s);
    dim3 grid(splits, N / WB_NC);
#define WB_LAUNCH(HM, XP, ...)                                                                              \
  do {                                                                                                      \
    auto kern = lstm_wgrad_bf16_kernel<HM, XP, ##__VA_ARGS__>;                                              \
    const int lds = WbGeo<HM>::template lds_bytes<XP>();                                                    \
    if (int e = allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return e;                      \
    hipLaunchKernelGGL(kern, grid, dim3(WB_NT), lds, s, a);                                                 \
  } while (0)
    if (wide) { if (xu8) WB_LAUNCH(8, 1, true); else if (x_exact_bf16) WB_LAUNCH(8, 1); else WB_LAUNCH(8, 3); }
    else { if (xu8) WB_LAUNCH(6, 1, true); else if (x_exact_bf16) WB_LAUNCH(6, 1); else WB_LAUNCH(6, 3); }
#undef WB_LAUNCH
  }
  int st = launch_status();
  if (st) return st;
  ReduceJob j;
  memset(&j, 0, sizeof(j));
  j.partial = (const float*)ws;
  j.M = nx + nh + nz; j.N = N; j.splits = splits; j.nprob = nz > 0 ? 3 : 2;
  j.alpha = 1.f; j.beta = beta; j.act = CLV_ACT_NONE;
  j.prob[0] = ReduceProb{dKx, ld_kx, 0};
  j.prob[1] = ReduceProb{dU, ld_u, nx};
  if (nz > 0) j.prob[2] = ReduceProb{dKz, ld_kz, nx + nh};
  // a single slab still goes through the reduction (it scatters the rows to the three tensors), but at once: the
  // deferred queue skips jobs without a split
  if (job && splits > 1) memcpy(job, &j, sizeof(j));
  else st = launch_reduce(j, s);
  return st;
}

// ---- both LSTMs of a step in one launch -------------------------------------------------------------------------------
static bool wgrad_problem_ok(const clv_wgrad_problem& p) {
  if (!clv_lstm_wgrad_supported(p.N, p.nx, p.nh, p.nz, p.x_exact_bf16) || p.K <= 0 || !p.X || !p.H || !p.dz || !p.dKx || !p.dU ||
      (p.nz > 0 && (!p.Z || !p.dKz)))
    return false;
  return !(p.ldx % 4 || p.ldh % 4 || p.lddz % 4 || ((uintptr_t)p.H | (uintptr_t)p.dz) % 16 ||
           ((uintptr_t)p.X) % (p.x_exact_bf16 == CLV_FRAMES_U8 ? 4 : 16));
}

extern "C" int clv_lstm_wgrad_pair_supported(const clv_wgrad_problem* p, const clv_wgrad_problem* q) {
  if (!p || !q || !wgrad_problem_ok(*p) || !wgrad_problem_ok(*q)) return 0;
  // one kernel instance, one grid: the same row count, the same tile geometry, the same number of frame pieces
  return p->K == q->K && p->N == q->N && wgrad_wide(p->nh, p->nz) == wgrad_wide(q->nh, q->nz) &&
         p->x_exact_bf16 == q->x_exact_bf16;
}

extern "C" size_t clv_lstm_wgrad_pair_workspace_bytes(int K, int N, int nx, int nh, int nz, int split_scale) {
  return (size_t)wgrad_splits(K, split_scale, 64) * (nx + nh + nz) * N * sizeof(float);
}

extern "C" int clv_lstm_wgrad_pair(const clv_wgrad_problem* p, const clv_wgrad_problem* q, int split_scale,
                                   clv_reduce_job* job_p, clv_reduce_job* job_q, void* stream) {
  using namespace clv;
  if (split_scale < 1 || split_scale > 8) return CLV_EINVAL;
  if (job_p) memset(job_p, 0, sizeof(*job_p));
  if (job_q) memset(job_q, 0, sizeof(*job_q));
  if (!clv_lstm_wgrad_pair_supported(p, q)) return CLV_EINVAL;
  const clv_wgrad_problem* pr[2] = {p, q};
  for (int i = 0; i < 2; ++i)
    if (clv_lstm_wgrad_pair_workspace_bytes(pr[i]->K, pr[i]->N, pr[i]->nx, pr[i]->nh, pr[i]->nz, split_scale) > pr[i]->ws_bytes ||
        !pr[i]->ws || ((uintptr_t)pr[i]->ws) % 16)
      return CLV_EWORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int K = p->K, N = p->N;
  const int splits = wgrad_splits(K, split_scale, 64), kc = wgrad_kc(K, split_scale, 64);
  WgradArgs a[2];
  for (int i = 0; i < 2; ++i)
    a[i] = WgradArgs{K, N, kc, pr[i]->X, pr[i]->ldx, pr[i]->nx, pr[i]->x_exact_bf16 == CLV_FRAMES_U8, pr[i]->H, pr[i]->ldh, pr[i]->nh, pr[i]->h_shift, pr[i]->h_zero_period,
                     pr[i]->Z, pr[i]->ldz, pr[i]->nz, pr[i]->dz, pr[i]->lddz, (float*)pr[i]->ws};
  {
    ProfScope ps("lstm_wgrad_bf16_pair", s);
    dim3 grid(splits, N / WB_NC, 2);
#define WB_LAUNCH2(HM, XP, ...)                                                                             \
  do {                                                                                                      \
    auto kern = lstm_wgrad_bf16_pair_kernel<HM, XP, ##__VA_ARGS__>;                                         \
    const int lds = WbGeo<HM>::template lds_bytes<XP>();                                                    \
    if (int e = allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return e;                      \
    hipLaunchKernelGGL(kern, grid, dim3(WB_NT), lds, s, a[0], a[1]);                                        \
  } while (0)
    const bool wide = wgrad_wide(p->nh, p->nz);
    const bool xu8 = p->x_exact_bf16 == CLV_FRAMES_U8;
    if (wide) { if (xu8) WB_LAUNCH2(8, 1, true); else if (p->x_exact_bf16) WB_LAUNCH2(8, 1); else WB_LAUNCH2(8, 3); }
    else { if (xu8) WB_LAUNCH2(6, 1, true); else if (p->x_exact_bf16) WB_LAUNCH2(6, 1); else WB_LAUNCH2(6, 3); }
#undef WB_LAUNCH2
  }
  int st = launch_status();
  if (st) return st;
  clv_reduce_job* jobs[2] = {job_p, job_q};
  for (int i = 0; i < 2 && !st; ++i) {
    const clv_wgrad_problem& w = *pr[i];
    ReduceJob j;
    memset(&j, 0, sizeof(j));
    j.partial = (const float*)w.ws;
    j.M = w.nx + w.nh + w.nz; j.N = N; j.splits = splits; j.nprob = w.nz > 0 ? 3 : 2;
    j.alpha = 1.f; j.beta = w.beta; j.act = CLV_ACT_NONE;
    j.prob[0] = ReduceProb{w.dKx, w.ld_kx, 0};
    j.prob[1] = ReduceProb{w.dU, w.ld_u, w.nx};
    if (w.nz > 0) j.prob[2] = ReduceProb{w.dKz, w.ld_kz, w.nx + w.nh};
    if (jobs[i] && splits > 1) memcpy(jobs[i], &j, sizeof(j));
    else st = launch_reduce(j, s);
  }
  return st;
}
