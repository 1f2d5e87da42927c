// lstm_mx.hip -- the LSTM training pass for LARGE batches (>= 4 rows per CU: BASELINE configuration 5, 1024 rows per
// GPU) with the recurrent product on the bf16 matrix cores and EXACT fp32 products (gfx950).
//
//   z_t = x_t . K_x + z^lat_t . K_z + rowbias + h_{t-1} . U;   i,f,o = gate_act(z_i,z_f,z_o), g = tanh(z_c)
//   c_t = f c_{t-1} + i g;   h_t = o tanh(c_t)                          (cl_vrnn/model.py:196-199, 225-228; Keras LSTM)
//
// Why another pair of sequence kernels.  lstm.hip (VALU, U in registers) and lstm_mfma.hip (4x4x1 f32 MFMA) both take
// ~1.45 us per step for the four rows a CU owns at 1024 rows: the f32 pipes need ~1000 cycles per step for the product
// alone and the gate phase cannot overlap it.  Here:
//  * an fp32 number is the exact sum of three bf16 numbers and a bf16 x bf16 product is exact in fp32 (wgrad_bf16.hip),
//    so h . U = sum over the 3 x 3 piece pairs with exact partial products, accumulated in fp32;
//  * the three PIECES of a row's h sit in the MFMA's N dimension: column n = 4 r + p holds piece p of batch row r (four
//    rows x three pieces = 12 of 16 columns), the A operand walks the three pieces of U into the SAME accumulator, so the
//    nine piece products cost 3 v_mfma_f32_16x16x32_bf16 per tile and k-step instead of 9 -- 198 MFMAs per step and
//    workgroup (792 cycles per SIMD) against ~1060 cycles on the f32 matrix pipe, and the VALU does gate math only;
//  * the M dimension of a forward tile is (4 units x 4 gates), so a C/D lane holds the four gate sums of ONE unit for one
//    (row, piece); a 7-instruction butterfly over the piece lanes both sums the pieces and hands each lane a different
//    tile: after it every (row, unit) exists in exactly one lane -- no 4x replicated gate math;
//  * the input projection is gathered INSIDE the kernel: K_x (124 KB) stays in LDS as [k][unit][gate], one wave turns
//    the frames into note lists two steps ahead, and because the piece lanes are summed anyway, lane p adds the notes
//    p, p+4, ... of its row: no [B*T,352] projection buffer, no projection launch (2 x 110 us at configuration 5);
//  * the forward pass stores what the backward pass needs in the coefficient format of lstm_pair.hip:
//    coef [B*T,4H] = (ki, kf, kg, ko) = (g i', c_{t-1} f', i g', tanh(c) o'), aux [B*T,2,H] = (kcarry, kc) =
//    (f, o (1 - tanh(c)^2)), so a backward step is dc += dh kc; dz = (dc ki, dc kf, dc kg, dh ko); dc *= kcarry.
// Backward: dh_rec = dz_{t+1} . U^T with M = 16 units per tile, N = (row, piece) of dz, K = 352 gate columns (11 k-steps);
// the same butterfly leaves lane (unit quad, row, p) with unit 4 ul + p: 64 distinct (row, unit) cells per wave.  The
// decoder's dZ_t = dz_t . K_z^T is two more tiles whose "units" are latents.
#include <type_traits>

#include "lstm_common.h"

namespace clv {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

// measurement builds (tools/build_variant.sh, results are wrong by design), a bit mask: forward 1 = the producer idles
// after the prologue, 2 = no global stores, 4 = no input gather, 8 = one MFMA per tile, 16 = every step's stores land on
// the first frame; backward 32 = no per-step loads, 64 = no global stores, 128 = one MFMA, 256 = no dz image writes
#ifndef MX_ABL
#define MX_ABL 0
#endif
// (tried: s_setprio 2 for the first wave of every SIMD, so that the two waves sharing a SIMD leave their MFMA phases one
// after the other -- no change in either kernel, profiles/r04_mx_log.txt)

// -DMX_STAMPS: every wave of workgroup 0 records the shader clock at seven points of the forward steps 64..71
// (tools/mx_stamps.py); each stamp takes the value it follows as an operand, so it cannot move above its computation
#ifdef MX_STAMPS
__device__ unsigned long long g_mx_stamps[8][8][8];      // [step][wave][stamp]
#define MXSTAMP(k, dep) asm volatile("s_memtime %0" : "=s"(mst[k]) : "v"(dep))
#else
#define MXSTAMP(k, dep)
#endif

constexpr int MX_R = 4;              // batch rows per workgroup
constexpr int MX_KP = LH * 16 + 16;  // bytes per row of the K_x image [k][unit][gate]: 356 words = 36 mod 64, so the rows of the 16 (row,
                                     // piece) lane groups of a gather start on 16 different bank offsets (1408 B = 32 mod 64: two)
constexpr int MX_HP = 208;           // bytes per (row, piece) line of the h image: 96 bf16 + 16 pad = 52 banks: the 16 lines of
                                     // a ds_read_b128 lane group start at banks 52 n mod 64 = distinct multiples of 4
constexpr int MX_ZP = 80;            // z image: 32 bf16 + 16 pad = 20 banks (same property)
constexpr int MX_DP = 720;           // dz image (backward): 352 bf16 + 16 pad = 180 banks = 52 mod 64
constexpr int MX_CAP = 96;           // note-list capacity per row
constexpr int MX_FAST = 8;           // list entries handled without a loop (two rounds of the four piece lanes)
constexpr int MX_PAD = 16;           // entries the producer always pads
constexpr int MX_NXMAX = 96;

struct MxItem { int koff; float v; };       // byte offset of the K_x image row, input value

struct MxFwdArgs {
  int B, T;
  const float* X; int ldx, nx; const float* Kx;       // frames: B*T rows of stride ldx, nx columns; Kx [nx,352]
  const float* Z; int ldz, nz; const float* Kz;       // latent inputs (decoder): B*T rows of stride ldz; Kz [nz,352]
  const float* rowbias; const float* U;
  float* hs; float* coef; float* aux;
};

__device__ __forceinline__ unsigned short bf16_bits(__bf16 b) { return __builtin_bit_cast(unsigned short, b); }

// x = p0 + p1 + p2 exactly (round to nearest even at every step; the residuals are exact in fp32)
__device__ __forceinline__ void split3(float x, __bf16 (&p)[3]) {
  p[0] = (__bf16)x;
  const float r1 = x - (float)p[0];
  p[1] = (__bf16)r1;
  p[2] = (__bf16)(r1 - (float)p[1]);
}
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8 (&out)[3]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    __bf16 p[3];
    split3(v[e], p);
    out[0][e] = p[0]; out[1][e] = p[1]; out[2][e] = p[2];
  }
}

// Per-step global accesses of the backward pass are buffer instructions (lstm_pair.hip): a descriptor per array over the
// workgroup's rows, the lane's byte offset in one VGPR, the step's row offset in an SGPR; a lane (or a step) without an
// element gets an offset beyond num_records -- the load returns 0, the store is dropped -- so no access sits under a
// divergent branch and the compiler's vmcnt waits stay counted.
typedef __amdgpu_buffer_rsrc_t mx_rsrc_t;
constexpr unsigned MX_OOB = 0x80000000u;
__device__ __forceinline__ mx_rsrc_t mx_rsrc(const void* p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(unsigned)bytes, 0x00020000);
}
__device__ __forceinline__ float mx_load(mx_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void mx_store(float v, mx_rsrc_t r, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, (int)soff, 0);
}

// ---------------------------------------------------------------------------------------------------------------------
// forward.  8 waves; wave w < 7 owns three tiles = the units 12w .. 12w+11 (tile tl: units 12w + 3 j + tl, j = 0..3, so
// that after the butterfly the piece lanes p = 0, 1, 2 of a quad own three CONSECUTIVE units: 12-byte pieces per quad
// in every per-step store instead of isolated floats); wave 7 owns the units 84..87 (one tile) and is the PRODUCER:
// frames -> note lists (two steps ahead), z_t -> bf16 pieces in the B-operand image (one step ahead).
// Batch rows beyond B (the last workgroup of a batch that is no multiple of four) are clones of row B-1: same inputs,
// same values, same addresses -- every access stays unconditional.
// ---------------------------------------------------------------------------------------------------------------------
template <int GATE, bool HASZ, int NT, bool PROD>
__device__ __forceinline__ void mx_fwd_body(const MxFwdArgs& a, char* lds, int ubase) {
  const int lane = threadIdx.x & 63;
  const int ul = lane >> 4, n = lane & 15, r = n >> 2, p = n & 3;
  const int T = a.T;
  const int row0 = blockIdx.x * MX_R;
  char* Kimg = lds;
  const int zero_off = a.nx * MX_KP;
  char* hB = lds + (a.nx + 1) * MX_KP;
  char* zB = hB + 2 * 16 * MX_HP;
  MxItem* lists = reinterpret_cast<MxItem*>(zB + 2 * 16 * MX_ZP);
  int* maxcount = reinterpret_cast<int*>(lists + 2 * MX_R * MX_CAP);

  // ---- A operands: the three pieces of U (and K_z) for this wave's tiles, resident in registers ---------------------
  // lane l holds row m = l & 15 = (unit ubase + NT (m >> 2) + tl, gate m & 3) and k = 32 s + 8 (l >> 4) + e, e = 0..7
  bf16x8 Ar[NT][3][3];
  bf16x8 Az[NT][3];
  {
    const int m = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
      const int col = (m & 3) * LH + ubase + NT * (m >> 2) + tl;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = 32 * s + 8 * kg + e;
          const float w = a.U[(size_t)min(k, LH - 1) * LG + col];
          v[e] = k < LH ? w : 0.f;
        }
        split8(v, Ar[tl][s]);
      }
      if (HASZ) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = 8 * kg + e;
          const float w = a.Kz[(size_t)min(k, a.nz - 1) * LG + col];
          v[e] = k < a.nz ? w : 0.f;
        }
        split8(v, Az[tl]);
      }
    }
  }

  // ---- per-lane constants --------------------------------------------------------------------------------------------
  const size_t rowc = (size_t)min(row0 + r, a.B - 1);
  // accumulator layout (before the butterfly): tile tl, register i = gate i of unit ubase + NT ul + tl, for (row r, piece p)
  int ucol[NT];
#pragma unroll
  for (int tl = 0; tl < NT; ++tl) ucol[tl] = (ubase + NT * ul + tl) * 16;
  // after the butterfly this lane finishes tile tp of its wave: unit `unit` of row r (p == 3: a second copy of p == 2;
  // one tile: all four lanes hold the same cell)
  const int tp = NT == 3 ? min(p, 2) : 0;
  const int unit = ubase + NT * ul + tp;
  float rb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) rb[i] = a.rowbias ? a.rowbias[rowc * LG + i * LH + unit] : 0.f;
  const bool even = !(p & 1), lo = !(p & 2);
  float c = 0.f;
  float* hs_p = a.hs + rowc * T * LH + unit;
  float* coef_p = a.coef + rowc * T * LG + unit;
  float* aux_p = a.aux + rowc * T * 2 * LH + unit;
  // h pieces -> B-operand image: line 4 r + q, k = unit
  const int hw_off = (4 * r) * MX_HP + 2 * unit;

  // ---- producer state (wave 7) ---------------------------------------------------------------------------------------
  float fr[2][MX_R][2];      // two register sets of frames in flight (set = step parity)
  float zr[2][2];            // ... and of z pairs
  const int zrow = lane >> 4, zlat = 2 * (lane & 15);
  auto load_frames = [&](float (&f)[MX_R][2], int t) {
    const int tc = min(t, T - 1);
#pragma unroll
    for (int rr = 0; rr < MX_R; ++rr) {
      const float* fp = a.X + ((size_t)min(row0 + rr, a.B - 1) * T + tc) * a.ldx;
      f[rr][0] = fp[min(lane, a.nx - 1)];            // raw: nothing may touch a requested value before its consumer does
      f[rr][1] = fp[min(lane + 64, a.nx - 1)];       // (a select here is a wait for the load right behind its issue)
    }
  };
  auto load_z = [&](float (&z)[2], int t) {
    const int tc = min(t, T - 1);
    const float* zp = a.Z + ((size_t)min(row0 + zrow, a.B - 1) * T + tc) * a.ldz;
    z[0] = zp[min(zlat, a.nz - 1)];
    z[1] = zp[min(zlat + 1, a.nz - 1)];
  };
  auto compact = [&](const float (&fraw)[MX_R][2], int buf) {      // frames -> lists[buf], maxcount[buf]
    float f[MX_R][2];
#pragma unroll
    for (int rr = 0; rr < MX_R; ++rr) {
      f[rr][0] = lane < a.nx ? fraw[rr][0] : 0.f;
      f[rr][1] = lane + 64 < a.nx ? fraw[rr][1] : 0.f;
    }
    MxItem* L = lists + buf * MX_R * MX_CAP;
    L[(lane >> 4) * MX_CAP + (lane & 15)] = MxItem{zero_off, 0.f};       // padding first (LDS operations of a wave are in order)
    const unsigned long long lt = (1ull << lane) - 1ull;
    int mx = 0;
    int cnt[MX_R];
#pragma unroll
    for (int rr = 0; rr < MX_R; ++rr) {
      const unsigned long long m0 = __ballot(f[rr][0] != 0.f), m1 = __ballot(f[rr][1] != 0.f);
      const int n0 = __popcll(m0);
      if (f[rr][0] != 0.f) L[rr * MX_CAP + __popcll(m0 & lt)] = MxItem{lane * MX_KP, f[rr][0]};
      if (f[rr][1] != 0.f) L[rr * MX_CAP + n0 + __popcll(m1 & lt)] = MxItem{(lane + 64) * MX_KP, f[rr][1]};
      cnt[rr] = n0 + __popcll(m1);
      mx = max(mx, cnt[rr]);
    }
    if (mx > MX_PAD) {       // dense frames: pad every row up to the longest list (rounded up to a round of four)
      const int upto = (mx + 3) & ~3;
#pragma unroll
      for (int rr = 0; rr < MX_R; ++rr)
        for (int j = max(cnt[rr], MX_PAD) + lane; j < upto; j += 64) L[rr * MX_CAP + j] = MxItem{zero_off, 0.f};
    }
    if (lane == 0) maxcount[buf] = mx;
  };
  auto stage_z = [&](const float (&z)[2], int buf) {           // z pair -> three piece images, 4 bytes each
    __bf16 p0[3], p1[3];
    split3(zlat < a.nz ? z[0] : 0.f, p0);
    split3(zlat + 1 < a.nz ? z[1] : 0.f, p1);
    char* at = zB + buf * 16 * MX_ZP + (4 * zrow) * MX_ZP + 2 * zlat;
#pragma unroll
    for (int q = 0; q < 3; ++q)
      *reinterpret_cast<unsigned*>(at + q * MX_ZP) = (unsigned)bf16_bits(p0[q]) | ((unsigned)bf16_bits(p1[q]) << 16);
  };

  // next step's input contribution in the accumulator layout: lane p takes the notes p, p + 4, ... of its row.  The
  // reads (two dependent LDS round trips: list entry -> kernel row) are issued AHEAD of the step's MFMAs, the FMAs
  // behind them.
  float xinit[NT][4];
  struct Rows { float v[MX_FAST / 4]; float4 k[MX_FAST / 4][NT]; };
  auto gather_items = [&](int buf, MxItem (&items)[MX_FAST / 4]) {
    const MxItem* Lr = lists + (buf * MX_R + r) * MX_CAP;
#pragma unroll
    for (int rd = 0; rd < MX_FAST / 4; ++rd) items[rd] = Lr[p + 4 * rd];
  };
  auto gather_rows = [&](const MxItem (&items)[MX_FAST / 4], Rows& g) {
#pragma unroll
    for (int rd = 0; rd < MX_FAST / 4; ++rd) {
      g.v[rd] = items[rd].v;
#pragma unroll
      for (int tl = 0; tl < NT; ++tl) g.k[rd][tl] = *reinterpret_cast<const float4*>(Kimg + items[rd].koff + ucol[tl]);
    }
  };
  auto gather_finish = [&](int buf, const Rows& g) {
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
      xinit[tl][0] = g.v[0] * g.k[0][tl].x; xinit[tl][1] = g.v[0] * g.k[0][tl].y;
      xinit[tl][2] = g.v[0] * g.k[0][tl].z; xinit[tl][3] = g.v[0] * g.k[0][tl].w;
#pragma unroll
      for (int rd = 1; rd < MX_FAST / 4; ++rd) {
        xinit[tl][0] = fmaf(g.v[rd], g.k[rd][tl].x, xinit[tl][0]); xinit[tl][1] = fmaf(g.v[rd], g.k[rd][tl].y, xinit[tl][1]);
        xinit[tl][2] = fmaf(g.v[rd], g.k[rd][tl].z, xinit[tl][2]); xinit[tl][3] = fmaf(g.v[rd], g.k[rd][tl].w, xinit[tl][3]);
      }
    }
    const int mc = __builtin_amdgcn_readfirstlane(maxcount[buf]);
    const MxItem* Lr = lists + (buf * MX_R + r) * MX_CAP;
    for (int j = MX_FAST; j < mc; j += 4) {           // denser frames (rare for piano-rolls)
      const MxItem it = Lr[j + p];
#pragma unroll
      for (int tl = 0; tl < NT; ++tl) {
        const float4 kr = *reinterpret_cast<const float4*>(Kimg + it.koff + ucol[tl]);
        xinit[tl][0] = fmaf(it.v, kr.x, xinit[tl][0]); xinit[tl][1] = fmaf(it.v, kr.y, xinit[tl][1]);
        xinit[tl][2] = fmaf(it.v, kr.z, xinit[tl][2]); xinit[tl][3] = fmaf(it.v, kr.w, xinit[tl][3]);
      }
    }
  };

  // ---- prologue ------------------------------------------------------------------------------------------------------
  if (PROD) {
    if (a.nx > 0) {
      load_frames(fr[0], 0);
      load_frames(fr[1], 1);
      compact(fr[0], 0);
      compact(fr[1], 1);
      load_frames(fr[0], 2);
      load_frames(fr[1], 3);
    } else {
      for (int j = lane; j < 2 * MX_R * MX_CAP; j += 64) lists[j] = MxItem{zero_off, 0.f};
      if (lane < 2) maxcount[lane] = 0;                         // never rewritten: every list is padding
    }
    if (HASZ) {
      load_z(zr[0], 0);
      stage_z(zr[0], 0);
      load_z(zr[0], 1);       // set 0: z_{t+1} of step 0;  set 1: of step 1
      load_z(zr[1], 2);
    }
  }
  __syncthreads();
  {
    Rows g;
    MxItem items[MX_FAST / 4];
    gather_items(0, items);
    gather_rows(items, g);
    gather_finish(0, g);
  }
  // every load of the prologue has landed before the loop is entered (the wait-count bookkeeping merges the loop-entry
  // state with the back edge's: see lstm_pair.hip)
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();

  auto step = [&](int t, auto PAR) {
    constexpr int cur = decltype(PAR)::value;
    // LDS reads in the order they are needed: the list entries of the NEXT step's inputs (their kernel rows are a second,
    // dependent round trip), this step's B operands = pieces of h_{t-1} (and z_t), then the kernel rows, which come back
    // under the MFMAs
    Rows g;
    MxItem items[MX_FAST / 4];
#ifdef MX_STAMPS
    unsigned long long mst[8];
#endif
    MXSTAMP(0, c);
    if (!(MX_ABL & 4)) gather_items(cur ^ 1, items);
    const char* hb = hB + cur * 16 * MX_HP + n * MX_HP + (lane >> 4) * 16;
    bf16x8 bh[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) bh[s] = *reinterpret_cast<const bf16x8*>(hb + 64 * s);
    bf16x8 bz;
    if (HASZ) bz = *reinterpret_cast<const bf16x8*>(zB + cur * 16 * MX_ZP + n * MX_ZP + (lane >> 4) * 16);
    if (!(MX_ABL & 4)) gather_rows(items, g);
    MXSTAMP(1, bh[2][0]);                           // the B operands are here
    f32x4v acc[NT];
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) acc[tl] = f32x4v{xinit[tl][0], xinit[tl][1], xinit[tl][2], xinit[tl][3]};
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int tl = 0; tl < NT; ++tl)
          if (!(MX_ABL & 8) || (s == 0 && q == 0)) acc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ar[tl][s][q], bh[s], acc[tl], 0, 0, 0);
    if (HASZ) {
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int tl = 0; tl < NT; ++tl) acc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Az[tl][q], bz, acc[tl], 0, 0, 0);
    }
    MXSTAMP(2, acc[NT - 1][0]);                     // the last MFMA's result is here
    // work that does not depend on the recurrence: the producer's lists / z image, next step's input contribution
    if (PROD && !(MX_ABL & 1)) {
      if (a.nx > 0) compact(fr[cur], cur);     // frame t + 2 -> the list buffer step t - 1 finished with
      if (HASZ) stage_z(zr[cur], cur ^ 1);     // z_{t+1}
    }
    if (!(MX_ABL & 4)) gather_finish(cur ^ 1, g);
    MXSTAMP(3, xinit[0][0]);
    // butterfly over the piece lanes: sums the pieces (and the note shares) and deals the tiles to the lanes
    float z[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (NT == 3) {
        const float a0 = acc[0][i], a1 = acc[1][i], a2 = acc[2][i];
        const float keepA = even ? a0 : a1, sendA = even ? a1 : a0;
        const float wA = keepA + dpp_mov<0xB1>(sendA);
        const float wB = a2 + dpp_mov<0xB1>(a2);
        const float keep = lo ? wA : wB, send = lo ? wB : wA;
        z[i] = keep + dpp_mov<0x4E>(send) + rb[i];
      } else {
        float x = acc[0][i];
        x = dpp_add<0xB1>(x);
        x = dpp_add<0x4E>(x);
        z[i] = x + rb[i];
      }
    }
    MXSTAMP(4, z[3]);
    const float ig = gate_fn<GATE>(z[0]), fg = gate_fn<GATE>(z[1]), og = gate_fn<GATE>(z[3]);
    const float gg = fast_tanh(z[2]);
    const float kf = c * gate_grad<GATE>(z[1], fg);
    c = fg * c + ig * gg;
    const float tc = fast_tanh(c);
    const float h = og * tc;
    MXSTAMP(5, h);
    if (!(MX_ABL & 2)) {
      const size_t o = (MX_ABL & 16) ? 0 : (size_t)t;
      coef_p[o * LG] = gg * gate_grad<GATE>(z[0], ig);
      coef_p[o * LG + LH] = kf;
      coef_p[o * LG + 2 * LH] = ig * (1.f - gg * gg);
      coef_p[o * LG + 3 * LH] = tc * gate_grad<GATE>(z[3], og);
      aux_p[o * 2 * LH] = fg;
      aux_p[o * 2 * LH + LH] = og * (1.f - tc * tc);
      hs_p[o * LH] = h;
    }
    {
      __bf16 hp[3];
      split3(h, hp);
      char* at = hB + (cur ^ 1) * 16 * MX_HP + hw_off;
      if (NT == 3) {
#pragma unroll
        for (int q = 0; q < 3; ++q) *reinterpret_cast<unsigned short*>(at + q * MX_HP) = bf16_bits(hp[q]);
      } else {             // four lanes hold the same h: lane p writes piece min(p, 2)
        const __bf16 mine = p == 0 ? hp[0] : (p == 1 ? hp[1] : hp[2]);
        *reinterpret_cast<unsigned short*>(at + min(p, 2) * MX_HP) = bf16_bits(mine);
      }
    }
    // the producer's requests for two steps ahead, BEHIND the last use of the register set they land in (issued ahead of
    // it they need fresh registers, and the copies back at the loop's end wait for the loads just issued: every step
    // then costs a trip to HBM -- lstm_pair.hip)
    if (PROD && !(MX_ABL & 1)) {
      if (a.nx > 0) load_frames(fr[cur], t + 4);
      if (HASZ) load_z(zr[cur], t + 3);
    }
#ifdef MX_STAMPS
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(mst[6]));          // arrival at the barrier
    if (blockIdx.x == 0 && lane == 0 && t >= 64 && t < 72) {
      const int wv = threadIdx.x >> 6;
#pragma unroll
      for (int k = 0; k < 7; ++k) g_mx_stamps[t - 64][wv][k] = mst[k];
    }
#endif
    step_barrier();
  };

  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  int t = 0;
  for (; t + 1 < T; t += 2) {
    step(t, P0{});
    step(t + 1, P1{});
  }
  if (t < T) step(t, P0{});
}

template <int GATE, bool HASZ>
__global__ __launch_bounds__(512) void lstm_mx_fwd_kernel(MxFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char mx_lds[];
  const int tid = threadIdx.x;
  // K_x image [k][unit][gate] (one ds_read_b128 = the four gates of a unit), row pitch MX_KP, + a zero row for the lists'
  // padding
  {
    float* Kimg = reinterpret_cast<float*>(mx_lds);
    const int nv = a.nx * LG;
    for (int i = tid; i < nv; i += 512) {
      const int k = i / LG, rem = i - k * LG, g = rem / LH, u = rem - g * LH;
      Kimg[k * (MX_KP / 4) + u * 4 + g] = a.Kx[i];
    }
    float* rest = Kimg + a.nx * (MX_KP / 4);
    const int tail = (MX_KP + 2 * 16 * MX_HP + 2 * 16 * MX_ZP) / 4;      // zero row, h image, z image
    for (int i = tid; i < tail; i += 512) rest[i] = 0.f;
  }
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (wave < 7) mx_fwd_body<GATE, HASZ, 3, false>(a, mx_lds, 12 * wave);
  else mx_fwd_body<GATE, HASZ, 1, true>(a, mx_lds, 84);
}

// ---------------------------------------------------------------------------------------------------------------------
// backward.  One 16-unit tile per wave (6 waves; + ZT latent tiles for dZ = dz . K_z^T); coef is overwritten with dz.
// ---------------------------------------------------------------------------------------------------------------------
struct MxBwdArgs {
  int B, T;
  const float* U; const float* dhs; const float* aux;
  float* coef; float* dzsum;
  const float* Kz; int nz; float* dZ; int lddz;
};

// A operand of a backward tile: 16 rows of U (or K_z).  The reduction index is k = 4 u + gate (NOT the gate-major column
// order of dz in memory): a lane's four dz values of a step are then four consecutive bf16 of an image line, one 8-byte
// LDS store per piece instead of four 2-byte ones.  k = 32 s + 8 (l >> 4) + e  <->  column (k & 3) * 88 + (k >> 2).
__device__ __forceinline__ void mx_bwd_weights(const float* W, int first, int limit, bf16x8 (&Ar)[11][3]) {
  const int lane = threadIdx.x & 63, m = lane & 15, kg = lane >> 4;
  const int idx = first + m;
  const bool ok = idx < limit;
  const float* src = W + (size_t)min(idx, limit - 1) * LG;
#pragma unroll
  for (int s = 0; s < 11; ++s) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 32 * s + 8 * kg + e;
      const float w = src[(k & 3) * LH + (k >> 2)];
      v[e] = ok ? w : 0.f;
    }
    split8(v, Ar[s]);
  }
}
// four floats -> three piece images of four bf16 each (v_cvt_pk_bf16_f32 rounds and packs a pair)
__device__ __forceinline__ unsigned mx_pack2(float lo, float hi) {
  typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
  const bf16x2v v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void mx_split4(const float (&x)[4], uint2 (&piece)[3]) {
  float a0 = x[0], a1 = x[1], a2 = x[2], a3 = x[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const unsigned lo = mx_pack2(a0, a1), hi = mx_pack2(a2, a3);
    piece[q] = make_uint2(lo, hi);
    if (q < 2) {
      a0 -= __builtin_bit_cast(float, lo << 16); a1 -= __builtin_bit_cast(float, lo & 0xffff0000u);
      a2 -= __builtin_bit_cast(float, hi << 16); a3 -= __builtin_bit_cast(float, hi & 0xffff0000u);
    }
  }
}

// dz_{t+1} (image `buf`) . the tile's rows; the butterfly leaves lane (ul, r, p) with row 4 ul + p of the tile
__device__ __forceinline__ float mx_bwd_matvec(const char* dzB, int buf, const bf16x8 (&Ar)[11][3]) {
  const int lane = threadIdx.x & 63, n = lane & 15, p = n & 3;
  const bool even = !(p & 1), lo = !(p & 2);
  const char* bp = dzB + buf * 16 * MX_DP + n * MX_DP + (lane >> 4) * 16;
  bf16x8 b[11];
#pragma unroll
  for (int s = 0; s < 11; ++s) b[s] = *reinterpret_cast<const bf16x8*>(bp + 64 * s);
  f32x4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 11; ++s)
#pragma unroll
    for (int q = 0; q < 3; ++q)
      if (!(MX_ABL & 128) || (s == 0 && q == 0)) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ar[s][q], b[s], acc, 0, 0, 0);
  const float kA = even ? acc[0] : acc[1], sA = even ? acc[1] : acc[0];
  const float kB = even ? acc[2] : acc[3], sB = even ? acc[3] : acc[2];
  const float wA = kA + dpp_mov<0xB1>(sA);
  const float wB = kB + dpp_mov<0xB1>(sB);
  const float keep = lo ? wA : wB, send = lo ? wB : wA;
  return keep + dpp_mov<0x4E>(send);
}

// a unit tile: the BPTT of 16 units x 4 rows
__device__ __forceinline__ void mx_bwd_units(const MxBwdArgs& a, char* dzB, int wave) {
  const int lane = threadIdx.x & 63;
  const int ul = lane >> 4, n = lane & 15, r = n >> 2, p = n & 3;
  const int T = a.T;
  const int row0 = blockIdx.x * MX_R;
  const int nrows = min(MX_R, a.B - row0);
  bf16x8 Ar[11][3];
  mx_bwd_weights(a.U, 16 * wave, LH, Ar);
  const int idx = 16 * wave + 4 * ul + p;
  const bool valid = idx < LH;
  const int u = min(idx, LH - 1);
  const unsigned rloc = (unsigned)min(r, nrows - 1);              // rows beyond the batch: clones of its last row
  const mx_rsrc_t r_c = mx_rsrc(a.coef + (size_t)row0 * T * LG, (size_t)nrows * T * LG * 4);
  const mx_rsrc_t r_a = mx_rsrc(a.aux + (size_t)row0 * T * 2 * LH, (size_t)nrows * T * 2 * LH * 4);
  const mx_rsrc_t r_d = mx_rsrc(a.dhs + (size_t)row0 * T * LH, (size_t)nrows * T * LH * 4);
  const unsigned v_c = valid ? (rloc * T * LG + u) * 4u : MX_OOB;
  const unsigned v_a = valid ? (rloc * T * 2 * LH + u) * 4u : MX_OOB;
  const unsigned v_d = valid ? (rloc * T * LH + u) * 4u : MX_OOB;
  // dz pieces: line 4 r + q, k = 4 u + gate: 8 bytes; a lane without a unit writes into the line's padding
  const int dzl_off = (4 * r) * MX_DP + (valid ? 8 * u : 2 * LG);

  // Two register sets, the loop body is two steps: the coefficients of step t are requested at the end of step t + 2,
  // behind the last use of the set they land in (lstm_pair.hip).  An odd T runs one step more: step t = -1 lies beyond
  // num_records -- zero coefficients, dz = 0, stores dropped.
  struct Coef { float ki, kf, kg, ko, kcarry, kc, dh; };
  auto load_set = [&](Coef& k, int t) {
    const unsigned tc = (unsigned)t;
    k.ki = mx_load(r_c, v_c, tc * (LG * 4)); k.kf = mx_load(r_c, v_c + LH * 4, tc * (LG * 4));
    k.kg = mx_load(r_c, v_c + 2 * LH * 4, tc * (LG * 4)); k.ko = mx_load(r_c, v_c + 3 * LH * 4, tc * (LG * 4));
    k.kcarry = mx_load(r_a, v_a, tc * (2 * LH * 4)); k.kc = mx_load(r_a, v_a + LH * 4, tc * (2 * LH * 4));
    k.dh = mx_load(r_d, v_d, tc * (LH * 4));
  };
  Coef SA, SB;
  load_set(SA, T - 1);
  load_set(SB, T - 2);
  float dc = 0.f;
  float zs[4] = {0.f, 0.f, 0.f, 0.f};
  __builtin_amdgcn_s_waitcnt(0x0F70);          // the prologue's loads have landed (see the forward kernel)
  __syncthreads();

  auto step = [&](int i, auto PAR, Coef& k) {
    constexpr int par = decltype(PAR)::value;        // i & 1: this step writes image `par`, reads the other one
    const int t = T - 1 - i;
    const float x = mx_bwd_matvec(dzB, par ^ 1, Ar);
    const float dh = k.dh + x;
    dc = fmaf(dh, k.kc, dc);
    float dz[4];
    dz[0] = dc * k.ki; dz[1] = dc * k.kf; dz[2] = dc * k.kg; dz[3] = dh * k.ko;
    dc *= k.kcarry;
    char* at = dzB + par * 16 * MX_DP + dzl_off;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      zs[g] += dz[g];
      if (!(MX_ABL & 64)) mx_store(dz[g], r_c, v_c + g * LH * 4, (unsigned)t * (LG * 4));
    }
    uint2 pc[3];
    mx_split4(dz, pc);
#pragma unroll
    for (int q = 0; q < 3; ++q)
      if (!(MX_ABL & 256)) *reinterpret_cast<uint2*>(at + q * MX_DP) = pc[q];
    if (!(MX_ABL & 32)) load_set(k, t - 2);
    step_barrier();
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  for (int i = 0; i < T; i += 2) {
    step(i, P0{}, SA);
    step(i + 1, P1{}, SB);
  }
  if (valid) {
    const size_t rowc = (size_t)row0 + rloc;
#pragma unroll
    for (int g = 0; g < 4; ++g) a.dzsum[rowc * LG + g * LH + u] = zs[g];
  }
}

// a latent tile: dZ_t = dz_t . K_z^T for 16 latents x 4 rows
__device__ __forceinline__ void mx_bwd_latents(const MxBwdArgs& a, char* dzB, int tile) {
  const int lane = threadIdx.x & 63;
  const int ul = lane >> 4, n = lane & 15, r = n >> 2, p = n & 3;
  const int T = a.T;
  const int row0 = blockIdx.x * MX_R;
  const int nrows = min(MX_R, a.B - row0);
  bf16x8 Ar[11][3];
  mx_bwd_weights(a.Kz, 16 * tile, a.nz, Ar);
  const int lat = 16 * tile + 4 * ul + p;
  const unsigned rloc = (unsigned)min(r, nrows - 1);
  const mx_rsrc_t r_z = mx_rsrc(a.dZ + (size_t)row0 * T * a.lddz, (size_t)nrows * T * a.lddz * 4);
  const unsigned v_z = lat < a.nz ? (rloc * T * a.lddz + lat) * 4u : MX_OOB;
  __syncthreads();
  auto step = [&](int i, auto PAR) {
    constexpr int par = decltype(PAR)::value;
    const int t = T - 1 - i;
    const float x = mx_bwd_matvec(dzB, par ^ 1, Ar);                   // dZ_{t+1}; nothing to store at t == T-1
    mx_store(x, r_z, v_z, t + 1 < T ? (unsigned)(t + 1) * a.lddz * 4 : MX_OOB);
    step_barrier();
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  for (int i = 0; i < T; i += 2) {       // odd T: the padded step t = -1 already writes dZ_0
    step(i, P0{});
    step(i + 1, P1{});
  }
  mx_store(mx_bwd_matvec(dzB, (T - 1) & 1, Ar), r_z, v_z, 0);        // dz_0 is in the image of step i = T - 1
}

template <int ZT>
__global__ __launch_bounds__((6 + ZT) * 64) void lstm_mx_bwd_kernel(MxBwdArgs a) {
  __shared__ __attribute__((aligned(16))) char dzB[2 * 16 * MX_DP];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 2 * 16 * MX_DP / 4; i += (6 + ZT) * 64) reinterpret_cast<float*>(dzB)[i] = 0.f;
  if (ZT > 0 && wave >= 6) mx_bwd_latents(a, dzB, wave - 6);
  else mx_bwd_units(a, dzB, wave);
}

static bool mx_auto(int B) {
  static const int mode = env_int("CLV_LSTM_MX", -1);       // 0: never, 1: any batch, default: from 768 rows on (three rows per CU; at 512 rows half the CUs would idle)
  return mode == 1 || (mode < 0 && B >= 768);
}

}  // namespace clv

#ifdef MX_STAMPS
extern "C" int clv_debug_mx_stamps(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(clv::g_mx_stamps), sizeof(unsigned long long) * 8 * 8 * 8);
}
#endif

extern "C" int clv_lstm_mx_supported(int B, int H, int nx, int nz) {
  return H == clv::LH && B >= 1 && nx >= 0 && nx <= clv::MX_NXMAX && nz >= 0 && nz <= 32 && clv::mx_auto(B);
}

extern "C" int clv_lstm_mx_fwd(int B, int T, int H, int gate_act,
                               const float* X, int ldx, int nx, const float* Kx,
                               const float* Z, int ldz, int nz, const float* Kz,
                               const float* rowbias, const float* U,
                               float* hs, float* coef, float* aux, void* stream) {
  using namespace clv;
  if (H != LH || B <= 0 || T < 1 || nx < 0 || nx > MX_NXMAX || nz < 0 || nz > 32) return CLV_EINVAL;
  if ((nx > 0 && (!X || !Kx || ldx < nx)) || (nz > 0 && (!Z || !Kz || ldz < nz)) || !U || !hs || !coef || !aux) return CLV_EINVAL;
  if (gate_act != CLV_GATE_HARD_SIGMOID && gate_act != CLV_GATE_SIGMOID) return CLV_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  ProfScope p("lstm_mx_fwd", s);
  MxFwdArgs a{B, T, X, ldx, nx, Kx, Z, ldz, nz, Kz, rowbias, U, hs, coef, aux};
  const size_t lds = (size_t)(nx + 1) * MX_KP + 2 * 16 * MX_HP + 2 * 16 * MX_ZP + 2 * MX_R * MX_CAP * sizeof(MxItem) + 16;
  const dim3 grid((B + MX_R - 1) / MX_R), block(512);
  const bool hard = gate_act == CLV_GATE_HARD_SIGMOID;
#define MX_LAUNCH(G, Zf)                                                                      \
  do {                                                                                        \
    auto kern = lstm_mx_fwd_kernel<G, Zf>;                                                    \
    if (int e = allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds)) return e;   \
    hipLaunchKernelGGL(kern, grid, block, lds, s, a);                                         \
  } while (0)
  if (nz > 0) { if (hard) MX_LAUNCH(CLV_GATE_HARD_SIGMOID, true); else MX_LAUNCH(CLV_GATE_SIGMOID, true); }
  else { if (hard) MX_LAUNCH(CLV_GATE_HARD_SIGMOID, false); else MX_LAUNCH(CLV_GATE_SIGMOID, false); }
#undef MX_LAUNCH
  return launch_status();
}

extern "C" int clv_lstm_mx_bwd(int B, int T, int H, const float* U, const float* dhs, const float* aux,
                               float* coef_inout_dz, float* dzsum, const float* Kz, int nz, float* dZ, int lddz,
                               void* stream) {
  using namespace clv;
  if (H != LH || B <= 0 || T < 1 || !U || !dhs || !aux || !coef_inout_dz || !dzsum) return CLV_EINVAL;
  if (nz < 0 || nz > 32 || (nz > 0 && (!Kz || !dZ || lddz < nz))) return CLV_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  ProfScope p("lstm_mx_bwd", s);
  MxBwdArgs a{B, T, U, dhs, aux, coef_inout_dz, dzsum, Kz, nz, dZ, lddz};
  const dim3 grid((B + MX_R - 1) / MX_R);
  if (nz > 16) hipLaunchKernelGGL((lstm_mx_bwd_kernel<2>), grid, dim3(8 * 64), 0, s, a);
  else if (nz > 0) hipLaunchKernelGGL((lstm_mx_bwd_kernel<1>), grid, dim3(7 * 64), 0, s, a);
  else hipLaunchKernelGGL((lstm_mx_bwd_kernel<0>), grid, dim3(6 * 64), 0, s, a);
  return launch_status();
}
