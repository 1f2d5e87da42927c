// out_head_bf16.hip -- the output head of cl_vrnn in training on the bf16 matrix cores, fp32 operands as bf16 pieces with 6 of
// the 9 piece pairs per product (relative error <= 2^-24 of a product: fp32-rounding accuracy, not exact) (gfx950):
// forward, loss and all three backward products of out_head.hip, every wave on its own 16 rows from start to end.
//
// Reference: X_decoded_mean = TimeDistributed(Dense(88, sigmoid)) on the decoder LSTM states (cl_vrnn/model.py:229-234),
// vae_loss = 88 * mean BCE(x, x_hat) (cl_vrnn/model.py:241-242) and their gradients under K.gradients:
//   logits = hs.Wo + bo,  nll_r = sum_j BCE(x_rj, sigmoid(logits_rj)) (Keras' 1e-7 clip),  dl = scale * (sigmoid(logits) - x)
//   dhs = dl.Wo^T,  dWo = hs^T.dl,  dbo = sum_r dl
//
// Why a second kernel.  out_head.hip runs the three K = 88 products on v_mfma_f32_16x16x4_f32: 13.6k matrix-pipe cycles
// per wave and 128-row block, two workgroup barriers per block (the weight gradient is dealt over the waves), dl and hs
// handed between layouts through fp32 LDS tiles that fill the LDS.  An fp32 number is the exact sum of three bf16
// numbers and a product of two bf16 numbers is exact in fp32 (wgrad_bf16.hip), so a.b = sum over piece pairs; the three
// pairs whose pieces are both beyond the first (a1.b2, a2.b1, a2.b2 <= 2^-24 |a.b|, the size of ONE fp32 rounding of the
// product) are left out: 6 bf16 MFMAs at 16 times the f32 rate.  5.2k cycles per wave and block.
//
// How the operands meet the MFMA layouts without LDS round trips (lane l: r = l & 15, q = l >> 4):
//   P1  logits^T [note][row] = Wo^T.hs^T   16x16x32, A = Wo pieces from an LDS image in FRAGMENT order (the 64 lanes'
//       16 bytes of one (piece, note tile, k-step) are 1 KB in lane order: conflict-free by construction), B = the wave's
//       hs rows straight from HBM: lane (row r, q) loads hs[r][32s + 8q .. +7].  Hidden index 88 is a ones column on the
//       hs side and the bias row on the Wo side.  Result: lane (row r, q) holds notes 16j + 4q + reg: Y / logits / dl move
//       as float4.
//   P2  dhs^T [hidden][row] = Wo.dl^T      the k-slots of an MFMA are ours to name: slot (q, e) of step s is note
//       16(2s + (e >> 2)) + 4q + (e & 3), which is what lane (r, q) already holds (tiles 2s, 2s + 1 of P1's result), so dl
//       is P2's B operand as it stands; a second image of Wo is laid out in that slot order.  Result: lane (row r, q) holds
//       hidden 16j + 4q + reg: float4 stores.
//   P3  [dWo ; dbo] += [hs | 1]^T.dl over the wave's 16 rows: 32x32x16 (K = 16 IS the tile), WAVE-PRIVATE accumulators
//       (9 tiles x 16 registers) across all the wave's blocks -- no barrier inside the row loop at all.  The sum runs
//       over rows, which sit on lanes in P1's layouts: dl goes through a 6 KB LDS tile per wave (written as it is held,
//       read as [note][8 rows]; both conflict-free, see ob_dlt_slot), hs^T comes from HBM/L2 a second time as 24 dword loads.
// The eight waves' partial gradients meet once, at the end, through LDS in a fixed order (bit-reproducible) and leave
// as one [89,88] slab per workgroup into the deferred split-K reduction, like out_head.hip's.
#include "out_head_args.h"

namespace clv {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int OB_NW = 8;                           // waves per workgroup, one 16-row tile each per block
constexpr int OB_FRAG = 1024;                      // bytes of one fragment: 64 lanes x 8 bf16
constexpr int OB_IMG = 3 * 6 * 3 * OB_FRAG;        // pieces x tiles x k-steps
constexpr int OB_DLT = 96 * 16;                    // floats of a wave's dl tile
constexpr int OB_LDS = 2 * OB_IMG + OB_NW * OB_DLT * 4;
static_assert(OB_LDS <= 160 * 1024, "LDS");
static_assert(OB_NW * 18 * 64 * 16 <= OB_LDS, "the final reduction's rounds fit");

// -DOB_STAMPS: wave 0 of workgroup 0 records the shader clock at the phase boundaries of its first block
#ifdef OB_STAMPS
__device__ unsigned long long g_ob_stamps[16];
__device__ unsigned long long g_ob_wg[256][4];     // per workgroup: start, loop entered, loop left (wave 0), end -- 100 MHz clock
#define OBW(k) do { if (threadIdx.x == 0) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)); g_ob_wg[blockIdx.x][k] = t_; } } while (0)
#define OBS(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_ob_stamps[k] = __builtin_readcyclecounter(); } while (0)
#else
#define OBW(k) do { } while (0)
#define OBS(k) do { } while (0)
#endif

// 8 floats -> their three bf16 pieces as MFMA fragments (element e of the fragment = v[e]); common.h: bf16_split_pair_dot2
__device__ __forceinline__ void ob_split8(const float (&v)[8], u32x4 (&f)[3]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned pc[3];
#ifdef OB_SHIFT_SPLIT
    bf16_split_pair(v[2 * i], v[2 * i + 1], pc);
#else
    bf16_split_pair_dot2(v[2 * i], v[2 * i + 1], pc);       // (-9 us per launch at 262144 rows against the shift form)
#endif
    f[0][i] = pc[0]; f[1][i] = pc[1]; f[2][i] = pc[2];
  }
}
__device__ __forceinline__ bf16x8 ob_frag(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }

// buffer instructions: out-of-range offsets read 0 / store nothing
typedef __amdgpu_buffer_rsrc_t ob_rsrc_t;
constexpr unsigned OB_OOB = 0x80000000u;
__device__ __forceinline__ ob_rsrc_t ob_rsrc(const void* p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(unsigned)bytes, 0x00020000);
}
// (by value: __builtin_bit_cast applied to a vector ELEMENT expression reads element 0 whatever the index -- clang 20, ROCm 7.2)
__device__ __forceinline__ float ob_u2f(unsigned u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ float ob_load1(ob_rsrc_t r, unsigned voff, int imm) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff + imm, 0, 0));
}
__device__ __forceinline__ void ob_load4(ob_rsrc_t r, unsigned voff, float (&v)[4]) {
  const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = ob_u2f(x[i]);
}
__device__ __forceinline__ void ob_load8(ob_rsrc_t r, unsigned voff, float (&v)[8]) {
  const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0);
  const u32x4 z = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff + 16, 0, 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[i] = ob_u2f(x[i]); v[4 + i] = ob_u2f(z[i]); }
}
__device__ __forceinline__ void ob_store4(const float (&v)[4], ob_rsrc_t r, unsigned voff) {
  const u32x4 x = {__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1]), __builtin_bit_cast(unsigned, v[2]),
                   __builtin_bit_cast(unsigned, v[3])};
  __builtin_amdgcn_raw_buffer_store_b128(x, r, (int)voff, 0, 0);
}
__device__ __forceinline__ void ob_store1(float v, ob_rsrc_t r, unsigned voff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, 0, 0);
}

// a.b over the piece pairs (i, j), i + j <= 2, smallest terms first
#define OB_PROD6(MFMA, acc, A, B)                                   \
  do {                                                              \
    acc = MFMA(ob_frag(A[2]), ob_frag(B[0]), acc, 0, 0, 0);         \
    acc = MFMA(ob_frag(A[1]), ob_frag(B[1]), acc, 0, 0, 0);         \
    acc = MFMA(ob_frag(A[0]), ob_frag(B[2]), acc, 0, 0, 0);         \
    acc = MFMA(ob_frag(A[1]), ob_frag(B[0]), acc, 0, 0, 0);         \
    acc = MFMA(ob_frag(A[0]), ob_frag(B[1]), acc, 0, 0, 0);         \
    acc = MFMA(ob_frag(A[0]), ob_frag(B[0]), acc, 0, 0, 0);         \
  } while (0)

// The wave's dl tile in LDS, 96 notes x 16 rows of floats.  A note's 16 rows are one 64-byte line; the lines are in the
// order the writer holds them (line = (4j + reg) * 4 + q for note 16j + 4q + reg: one ds_write_b32 of the 64 lanes fills
// 4 whole lines), and the four 16-byte chunks of a line are XORed with reg: the reader (lane = note, 8 rows = two
// ds_read_b128) then finds the 4 lines that share a bank group at 4 different chunks.
__device__ __forceinline__ int ob_dlt_line(int note) { return (((note >> 4) * 4 + (note & 3)) * 4 + ((note >> 2) & 3)) * 16; }

template <bool YU8>
__global__ __launch_bounds__(64 * OB_NW) void out_head_bf16_kernel(OutHeadArgs a) {
  extern __shared__ __attribute__((aligned(16))) char ob_lds[];
  char* A1 = ob_lds;                                     // P1: Wo^T pieces, [piece][note tile][k-step][lane] x 16 B
  char* A2 = ob_lds + OB_IMG;                            // P2: Wo pieces in P2's slot order, [piece][hidden tile][k-step][lane]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4, n32 = lane & 31, kh = lane >> 5;
  float* dlT = reinterpret_cast<float*>(ob_lds + 2 * OB_IMG) + wave * OB_DLT;
  OBS(0);
  OBW(0);

  // Every global access of the row loop is a buffer instruction: rows beyond R fall outside the descriptor (loads return 0,
  // stores are dropped), a lane that has nothing to move gets an out-of-range offset, an absent output a descriptor of 0
  // bytes -- no divergent branch in the loop, which the register allocator (144 accumulators live) cannot afford.
  const ob_rsrc_t r_hs = ob_rsrc(a.hs, (size_t)a.R * OH * 4), r_y = ob_rsrc(a.Y, (size_t)a.R * a.ldy * (YU8 ? 1 : 4));
  const ob_rsrc_t r_lg = ob_rsrc(a.logits, a.logits ? (size_t)a.R * OH * 4 : 0);
  const ob_rsrc_t r_dl = ob_rsrc(a.dlogits, a.dlogits ? (size_t)a.R * OH * 4 : 0);
  const ob_rsrc_t r_dh = ob_rsrc(a.dhs, (size_t)a.R * OH * 4), r_nl = ob_rsrc(a.rownll, (size_t)a.R * 4);
  const unsigned c_off = (16 * 0 + 4 * q) * 4u;           // byte offset of this lane's 4 outputs inside tile 0 of a row
  const bool c5 = q < 2;                                   // tile 5: notes / hidden units 88..95 do not exist

  // this lane's part of the wave's hs rows as P1's B operand: hidden 32s + 8q + e of row r (hidden 88..95: nothing to load;
  // element 0 of that group becomes the ones column where it is used).  The first block's rows are touched here, while the
  // images are built, so that the loop's own loads find them in L2 (keeping them in registers across the loop's back edge
  // would cost 24 registers through the whole body).
  float hv[3][8];
  auto load_hv = [&](int row0) {
#pragma unroll
    for (int s = 0; s < 3; ++s)
      ob_load8(r_hs, (s == 2 && q == 3) ? OB_OOB : (unsigned)(row0 + r) * (OH * 4u) + (32 * s + 8 * q) * 4u, hv[s]);
  };
  load_hv(blockIdx.x * OH_RB + wave * 16);

  // ---- the two images of Wo (and bo): 36 fragments of 64 lanes x 8 values, fragment F = 8 it + wave.  Which fragment is a
  // wave-uniform (scalar) matter, the lane supplies one offset per image, and values that do not exist come from offsets
  // outside the descriptor: ~60 vector instructions per fragment, nearly all of them the split.  (First version: per-thread
  // index arithmetic on 40 scalar loads, 2.5 us of instruction issue before the last load was even requested.)
  {
    const ob_rsrc_t r_wo = ob_rsrc(a.Wo, (size_t)OH * OH * 4), r_bo = ob_rsrc(a.bo, (size_t)OH * 4);
    const int m = r, qq = q;
    const unsigned l0 = (8 * qq * OH + m) * 4u;          // image 0: Wo[32s + 8qq + e][16j + m]
    const unsigned l1 = (m * OH + 4 * qq) * 4u;          // image 1: Wo[16j + m][32s + 16(e >> 2) + 4qq + (e & 3)]
    float wv[5][8];
#pragma unroll
    for (int it = 0; it < 5; ++it) {
      const int F = min(8 * it + wave, 35);
      const int img = F >= 18, fr = F - 18 * img, j = fr / 3, s = fr - 3 * j;
      if (!img) {
        // (the whole offset goes through the vector operand: the range check does not see a scalar offset)
        const unsigned vo = (16 * j + m < OH) ? l0 + (32 * s * OH + 16 * j) * 4u : OB_OOB;              // notes 88..95: zeros
#pragma unroll
        for (int e = 0; e < 8; ++e) wv[it][e] = ob_u2f(__builtin_amdgcn_raw_buffer_load_b32(r_wo, (int)vo + e * OH * 4, 0, 0));
        // hidden 88 (s = 2, qq = 3, e = 0) is the bias row; hidden 89..95 fall behind the array: zeros
        const float bias = ob_u2f(__builtin_amdgcn_raw_buffer_load_b32(r_bo, (s == 2 && qq == 3) ? (16 * j + m) * 4 : (int)OB_OOB, 0, 0));
        wv[it][0] = (s == 2 && qq == 3) ? bias : wv[it][0];
      } else {
        const unsigned vo = l1 + (16 * j * OH + 32 * s) * 4u;             // hidden 88..95 fall behind the array
        const u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(r_wo, (int)vo, 0, 0);
        const u32x4 hi = __builtin_amdgcn_raw_buffer_load_b128(r_wo, (s == 2 && qq >= 2) ? (int)OB_OOB : (int)vo + 64, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) { wv[it][e] = ob_u2f(lo[e]); wv[it][4 + e] = ob_u2f(hi[e]); }
      }
    }
#pragma unroll
    for (int it = 0; it < 5; ++it) {
      const int F = 8 * it + wave;
      if (F < 36) {
        u32x4 f[3];
        ob_split8(wv[it], f);
        char* base = (F >= 18 ? A2 + (F - 18) * OB_FRAG : A1 + F * OB_FRAG) + lane * 16;
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(base + p * 18 * OB_FRAG) = f[p];
      }
    }
  }

  f32x16 acc3[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc3[i][j][e] = 0.f;
#pragma unroll
  for (int s = 0; s < 3; ++s) asm volatile("" :: "v"(hv[s][0]), "v"(hv[s][4]));
  OBS(10);
  __syncthreads();
  OBS(1);
  OBW(1);

  for (int blk = blockIdx.x; blk * OH_RB < a.R; blk += gridDim.x) {
    const int row0 = blk * OH_RB + wave * 16;
    if (row0 >= a.R) continue;                           // (wave-uniform; no barrier inside this loop)
    const unsigned row = row0 + r;
    const bool rok = (int)row < a.R;
    const unsigned o_row = row * (OH * 4u) + c_off;        // + 64 j: this lane's float4 of tile j in an [R,88] array
    const unsigned o5 = c5 ? o_row + 5 * 64 : OB_OOB;

    load_hv(row0);                                       // (the first block's rows: out of L2, see above)
    OBS(2);

    // ---- P1: logits^T = Wo^T.hs^T (+ bias through the ones column)
    float y[6][4];             // targets of this lane's outputs: notes 16j + 4q .. + 3 of row r
    f32x4 acc[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      if (s == 2) hv[2][0] = q == 3 ? 1.f : hv[2][0];      // hidden 88: the ones column
      u32x4 B[3];
      ob_split8(hv[s], B);
      if (s == 2) {            // the targets are requested here, when two thirds of hv are dead
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          if (YU8) y[j][0] = ob_load1(r_y, row * (unsigned)a.ldy + 4 * q + 16 * j, 0);      // byte frames: this lane's four notes are one dword
          else ob_load4(r_y, row * (a.ldy * 4u) + c_off + 64 * j, y[j]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        u32x4 A[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) A[p] = *reinterpret_cast<const u32x4*>(A1 + ((p * 6 + j) * 3 + s) * OB_FRAG + lane * 16);
        OB_PROD6(__builtin_amdgcn_mfma_f32_16x16x32_bf16, acc[j], A, B);
      }
    }
    OBS(3);

    // ---- Bernoulli NLL with Keras' epsilon clip (the arithmetic of out_head.hip / the gemm_bce epilogue); dl stays in acc.
    // hs^T for P3 (lane = hidden 32jm + n32, rows 8kh .. + 7) is requested half way through, when half of y is dead
    float ht[3][8];
    float ssum = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      if (j == 3) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jm = 0; jm < 3; ++jm) {
          const int h = 32 * jm + n32;
          const unsigned o = h < OH ? (row0 + 8 * kh) * (OH * 4u) + h * 4u : OB_OOB;
#pragma unroll
          for (int e = 0; e < 8; ++e) ht[jm][e] = ob_load1(r_hs, o, e * OH * 4);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      const bool cok = j < 5 || c5;
      float lg4[4], dl4[4];
      const unsigned yraw = __builtin_bit_cast(unsigned, y[j][0]);
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const float lg = acc[j][reg];
        const float t = YU8 ? (float)((yraw >> (8 * reg)) & 0xffu) : y[j][reg];
        const float l = fminf(fmaxf(lg, BCE_CLIP_LO), BCE_CLIP_HI);
        const float e = __expf(-fabsf(l));
        const float nl = fmaxf(l, 0.f) + __logf(1.f + e) - l * t;
        const float r1 = fast_rcp(1.f + e);
        const float sg = l >= 0.f ? r1 : e * r1;
        const bool inside = (lg >= BCE_CLIP_LO) && (lg <= BCE_CLIP_HI);
        dl4[reg] = (rok && cok && inside) ? a.scale * (sg - t) : 0.f;
        lg4[reg] = lg;
        ssum += cok ? nl : 0.f;
        acc[j][reg] = dl4[reg];
      }
      const unsigned o = j < 5 ? o_row + 64 * j : o5;
      ob_store4(lg4, r_lg, o);
      ob_store4(dl4, r_dl, o);
    }
    // the next block's hs and target rows -> L2 (one dword per 128 bytes of a row; rows behind the arrays request nothing):
    // the row loop has no register to spare for a real prefetch, and its loads otherwise wait for HBM once per block
    const unsigned nrow = row + gridDim.x * OH_RB;
    const float pf0 = ob_load1(r_hs, q < 3 ? nrow * (OH * 4u) + q * 128u : OB_OOB, 0);
    const float pf1 = ob_load1(r_y, YU8 ? (q == 0 ? nrow * (unsigned)a.ldy : OB_OOB) : (q < 3 ? nrow * (a.ldy * 4u) + q * 128u : OB_OOB), 0);
    ssum += __shfl_xor(ssum, 16, 64);
    ssum += __shfl_xor(ssum, 32, 64);
    ob_store1(ssum, r_nl, q == 0 ? row * 4u : OB_OOB);
    // dl -> the wave's LDS tile (P3's B operand)
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
        dlT[((j * 4 + reg) * 4 + q) * 16 + (((r >> 2) ^ reg) << 2) + (r & 3)] = acc[j][reg];
    OBS(4);

    // ---- P3: [dWo ; dbo] += [hs | 1]^T.dl over this wave's 16 rows
    {
#pragma unroll
      for (int e = 0; e < 8; ++e) ht[2][e] = n32 == OH - 64 ? 1.f : ht[2][e];      // hidden 88: the dbo row
      u32x4 Hp[3][3];
#pragma unroll
      for (int jm = 0; jm < 3; ++jm) ob_split8(ht[jm], Hp[jm]);
#pragma unroll
      for (int jn = 0; jn < 3; ++jn) {
        const int note = 32 * jn + n32;
        const float* line = dlT + ob_dlt_line(note);
        const int key = note & 3;
        const float4 lo = *reinterpret_cast<const float4*>(line + (((2 * kh) ^ key) << 2));
        const float4 hi = *reinterpret_cast<const float4*>(line + (((2 * kh + 1) ^ key) << 2));
        const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        u32x4 Tp[3];
        ob_split8(v, Tp);
#pragma unroll
        for (int jm = 0; jm < 3; ++jm) OB_PROD6(__builtin_amdgcn_mfma_f32_32x32x16_bf16, acc3[jm][jn], Hp[jm], Tp);
      }
    }
    OBS(5);

    // ---- P2: dhs^T = Wo.dl^T, k-slot (q, e) of step s = note 16(2s + (e >> 2)) + 4q + (e & 3)
    {
      f32x4 acc2[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) acc2[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const float v[8] = {acc[2 * s][0], acc[2 * s][1], acc[2 * s][2], acc[2 * s][3],
                            acc[2 * s + 1][0], acc[2 * s + 1][1], acc[2 * s + 1][2], acc[2 * s + 1][3]};
        u32x4 B[3];
        ob_split8(v, B);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          u32x4 A[3];
#pragma unroll
          for (int p = 0; p < 3; ++p) A[p] = *reinterpret_cast<const u32x4*>(A2 + ((p * 6 + j) * 3 + s) * OB_FRAG + lane * 16);
          OB_PROD6(__builtin_amdgcn_mfma_f32_16x16x32_bf16, acc2[j], A, B);
        }
      }
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const float d4[4] = {acc2[j][0], acc2[j][1], acc2[j][2], acc2[j][3]};
        ob_store4(d4, r_dh, j < 5 ? o_row + 64 * j : o5);
      }
    }
    asm volatile("" :: "v"(pf0), "v"(pf1));
    OBS(6);
  }

  // ---- the eight waves' gradients meet in LDS, half of the 36 float4 per lane per round (8 x 18 KB), and leave summed in
  // wave order: float4 f = (3jm + jn) * 4 + i4 holds registers 4 i4 .. + 3 of tile (jm, jn) = 4 consecutive rows h of one note
  OBW(2);
  float* slab = a.partial + (size_t)blockIdx.x * OH_SLAB_ROWS * OH;
  float4* red = reinterpret_cast<float4*>(ob_lds);       // [wave][18][lane]
  __syncthreads();
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int ff = 0; ff < 18; ++ff) {
      const int f = 18 * half + ff, jm = f / 12, jn = (f >> 2) % 3, i4 = f & 3;
      red[(wave * 18 + ff) * 64 + lane] = make_float4(acc3[jm][jn][4 * i4], acc3[jm][jn][4 * i4 + 1], acc3[jm][jn][4 * i4 + 2],
                                                      acc3[jm][jn][4 * i4 + 3]);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int v = tid + 64 * OB_NW * k;                // (ff, lane) of four outputs
      if (v < 18 * 64) {
        const int ff = v >> 6, l = v & 63;
        float4 sum = red[ff * 64 + l];
#pragma unroll
        for (int w = 1; w < OB_NW; ++w) {
          const float4 x = red[(w * 18 + ff) * 64 + l];
          sum.x += x.x; sum.y += x.y; sum.z += x.z; sum.w += x.w;
        }
        const int f = 18 * half + ff, jm = f / 12, jn = (f >> 2) % 3, i4 = f & 3;
        const int h = 32 * jm + 8 * i4 + 4 * (l >> 5), note = 32 * jn + (l & 31);
        const float o4[4] = {sum.x, sum.y, sum.z, sum.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (h + i < OH_SLAB_ROWS && note < OH) slab[(h + i) * OH + note] = o4[i];
      }
    }
    if (half == 0) __syncthreads();
  }
  OBS(7);
  OBW(3);
}

#ifdef OB_STAMPS
}
extern "C" int clv_debug_out_head_bf16_stamps(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(clv::g_ob_stamps), sizeof(unsigned long long) * 16);
}
extern "C" int clv_debug_out_head_bf16_wg(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(clv::g_ob_wg), sizeof(unsigned long long) * 1024);
}
namespace clv {
#endif

bool out_head_bf16_ok(const OutHeadArgs& a) {
  auto al = [](const void* p) { return ((uintptr_t)p) % 16 == 0; };
  // (buffer descriptors: every array below 2 GiB, so that offset 0x80000000 is out of range for all of them)
  const size_t widest = (size_t)a.R * (size_t)(a.ldy > OH ? a.ldy : OH) * 4;
  const bool y_ok = a.y_u8 ? (((uintptr_t)a.Y) % 4 == 0) : al(a.Y);
  return widest < 0x80000000ull && al(a.hs) && y_ok && a.ldy % 4 == 0 && al(a.dhs) && al(a.logits) && al(a.dlogits);
}

int launch_out_head_bf16(const OutHeadArgs& a, int wgs, hipStream_t s) {
  ProfScope p("out_head_bf16", s);
  if (a.y_u8) {
    if (int e = allow_dynamic_lds(reinterpret_cast<const void*>(out_head_bf16_kernel<true>), OB_LDS)) return e;
    hipLaunchKernelGGL(out_head_bf16_kernel<true>, dim3(wgs), dim3(64 * OB_NW), OB_LDS, s, a);
  } else {
    if (int e = allow_dynamic_lds(reinterpret_cast<const void*>(out_head_bf16_kernel<false>), OB_LDS)) return e;
    hipLaunchKernelGGL(out_head_bf16_kernel<false>, dim3(wgs), dim3(64 * OB_NW), OB_LDS, s, a);
  }
  return CLV_OK;
}

}  // namespace clv
