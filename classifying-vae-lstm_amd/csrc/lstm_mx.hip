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

constexpr int MX_R = 4;              // batch rows per workgroup
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

// ---------------------------------------------------------------------------------------------------------------------
// forward.  8 waves; wave w < 7 owns the tiles 3w .. 3w+2 (units 12w .. 12w+11), wave 7 owns tile 21 (units 84..87) and
// is the PRODUCER: frames -> note lists (two steps ahead), z_t -> bf16 pieces in the B-operand image (one step ahead).
// ---------------------------------------------------------------------------------------------------------------------
template <int GATE, bool HASZ, int NT, bool PROD>
__device__ __forceinline__ void mx_fwd_body(const MxFwdArgs& a, char* lds, int tile0) {
  const int lane = threadIdx.x & 63;
  const int ul = lane >> 4, n = lane & 15, r = n >> 2, p = n & 3;
  const int T = a.T;
  const int row0 = blockIdx.x * MX_R;
  char* Kimg = lds;
  const int zero_off = a.nx * LH * 16;
  char* hB = lds + (a.nx + 1) * LH * 16;
  char* zB = hB + 2 * 16 * MX_HP;
  MxItem* lists = reinterpret_cast<MxItem*>(zB + 2 * 16 * MX_ZP);
  int* maxcount = reinterpret_cast<int*>(lists + 2 * MX_R * MX_CAP);

  // ---- A operands: the three pieces of U (and K_z) for this wave's tiles, resident in registers ---------------------
  // lane l holds row m = l & 15 = (unit 4 tile + (m >> 2), gate m & 3) and k = 32 s + 8 (l >> 4) + e, e = 0..7
  bf16x8 Ar[NT][3][3];
  bf16x8 Az[NT][3];
  {
    const int m = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
      const int col = (m & 3) * LH + 4 * (tile0 + tl) + (m >> 2);
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
  const int row = row0 + r;
  const size_t rowc = (size_t)min(row, a.B - 1);
  // accumulator layout (before the butterfly): tile tl, register i = gate i of unit 4 (tile0 + tl) + ul, for (row r, piece p);
  // the per-row bias enters through the p == 3 lanes (their B column is zero)
  float rbm[NT][4];
  int ucol[NT];
#pragma unroll
  for (int tl = 0; tl < NT; ++tl) {
    const int u = 4 * (tile0 + tl) + ul;
    ucol[tl] = u * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float b = a.rowbias ? a.rowbias[rowc * LG + i * LH + u] : 0.f;
      rbm[tl][i] = p == 3 ? b : 0.f;
    }
  }
  // after the butterfly this lane finishes tile tp of its wave: unit `unit` of row r
  const int tp = NT == 3 ? min(p, 2) : 0;
  const int unit = 4 * (tile0 + tp) + ul;
  const bool owner = (NT == 3 ? p < 3 : p == 0) && row < a.B;
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
      const float v0 = fp[min(lane, a.nx - 1)], v1 = fp[min(lane + 64, a.nx - 1)];
      f[rr][0] = lane < a.nx ? v0 : 0.f;
      f[rr][1] = lane + 64 < a.nx ? v1 : 0.f;
    }
  };
  auto load_z = [&](float (&z)[2], int t) {
    const int tc = min(t, T - 1);
    const float* zp = a.Z + ((size_t)min(row0 + zrow, a.B - 1) * T + tc) * a.ldz;
    const float v0 = zp[min(zlat, a.nz - 1)], v1 = zp[min(zlat + 1, a.nz - 1)];
    z[0] = zlat < a.nz ? v0 : 0.f;
    z[1] = zlat + 1 < a.nz ? v1 : 0.f;
  };
  auto compact = [&](const float (&f)[MX_R][2], int buf) {      // frames -> lists[buf], maxcount[buf]
    MxItem* L = lists + buf * MX_R * MX_CAP;
    L[(lane >> 4) * MX_CAP + (lane & 15)] = MxItem{zero_off, 0.f};       // padding first (LDS operations of a wave are in order)
    const unsigned long long lt = (1ull << lane) - 1ull;
    int mx = 0;
    int cnt[MX_R];
#pragma unroll
    for (int rr = 0; rr < MX_R; ++rr) {
      const unsigned long long m0 = __ballot(f[rr][0] != 0.f), m1 = __ballot(f[rr][1] != 0.f);
      const int n0 = __popcll(m0);
      if (f[rr][0] != 0.f) L[rr * MX_CAP + __popcll(m0 & lt)] = MxItem{lane * LH * 16, f[rr][0]};
      if (f[rr][1] != 0.f) L[rr * MX_CAP + n0 + __popcll(m1 & lt)] = MxItem{(lane + 64) * LH * 16, f[rr][1]};
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
    split3(z[0], p0);
    split3(z[1], p1);
    char* at = zB + buf * 16 * MX_ZP + (4 * zrow) * MX_ZP + 2 * zlat;
#pragma unroll
    for (int q = 0; q < 3; ++q)
      *reinterpret_cast<unsigned*>(at + q * MX_ZP) = (unsigned)bf16_bits(p0[q]) | ((unsigned)bf16_bits(p1[q]) << 16);
  };

  // next step's input contribution + bias, in the accumulator layout: lane p takes the notes p, p + 4, ... of its row
  float xinit[NT][4];
  auto gather = [&](int buf) {
#pragma unroll
    for (int tl = 0; tl < NT; ++tl)
#pragma unroll
      for (int i = 0; i < 4; ++i) xinit[tl][i] = rbm[tl][i];
    const MxItem* Lr = lists + (buf * MX_R + r) * MX_CAP;
    auto visit = [&](const MxItem it) {
#pragma unroll
      for (int tl = 0; tl < NT; ++tl) {
        const float4 kr = *reinterpret_cast<const float4*>(Kimg + it.koff + ucol[tl]);
        xinit[tl][0] = fmaf(it.v, kr.x, xinit[tl][0]);
        xinit[tl][1] = fmaf(it.v, kr.y, xinit[tl][1]);
        xinit[tl][2] = fmaf(it.v, kr.z, xinit[tl][2]);
        xinit[tl][3] = fmaf(it.v, kr.w, xinit[tl][3]);
      }
    };
#pragma unroll
    for (int rd = 0; rd < MX_FAST / 4; ++rd) visit(Lr[p + 4 * rd]);
    const int mc = __builtin_amdgcn_readfirstlane(maxcount[buf]);
    for (int j = MX_FAST; j < mc; j += 4) visit(Lr[j + p]);
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
  gather(0);
  __syncthreads();

  auto step = [&](int t, auto PAR) {
    constexpr int cur = decltype(PAR)::value;
    // B operands: pieces of h_{t-1} (and z_t)
    const char* hb = hB + cur * 16 * MX_HP + n * MX_HP + (lane >> 4) * 16;
    bf16x8 bh[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) bh[s] = *reinterpret_cast<const bf16x8*>(hb + 64 * s);
    bf16x8 bz;
    if (HASZ) bz = *reinterpret_cast<const bf16x8*>(zB + cur * 16 * MX_ZP + n * MX_ZP + (lane >> 4) * 16);
    f32x4v acc[NT];
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) acc[tl] = f32x4v{xinit[tl][0], xinit[tl][1], xinit[tl][2], xinit[tl][3]};
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int tl = 0; tl < NT; ++tl) acc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ar[tl][s][q], bh[s], acc[tl], 0, 0, 0);
    if (HASZ) {
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int tl = 0; tl < NT; ++tl) acc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Az[tl][q], bz, acc[tl], 0, 0, 0);
    }
    // work that does not depend on the recurrence: the producer's lists / z image, next step's input contribution
    if (PROD) {
      if (a.nx > 0) {
        compact(fr[cur], cur);                 // frame t + 2 -> the list buffer step t - 1 finished with
        load_frames(fr[cur], t + 4);
      }
      if (HASZ) {
        stage_z(zr[cur], cur ^ 1);             // z_{t+1}
        load_z(zr[cur], t + 3);
      }
    }
    gather(cur ^ 1);
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
        z[i] = keep + dpp_mov<0x4E>(send);
      } else {
        float x = acc[0][i];
        x = dpp_add<0xB1>(x);
        x = dpp_add<0x4E>(x);
        z[i] = x;
      }
    }
    const float ig = gate_fn<GATE>(z[0]), fg = gate_fn<GATE>(z[1]), og = gate_fn<GATE>(z[3]);
    const float gg = fast_tanh(z[2]);
    const float kf = c * gate_grad<GATE>(z[1], fg);
    c = fg * c + ig * gg;
    const float tc = fast_tanh(c);
    const float h = og * tc;
    if (owner) {
      const size_t o = (size_t)t;
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
        if (p < 3) {
#pragma unroll
          for (int q = 0; q < 3; ++q) *reinterpret_cast<unsigned short*>(at + q * MX_HP) = bf16_bits(hp[q]);
        }
      } else {             // four lanes hold the same h: lane p writes piece p
        const __bf16 mine = p == 0 ? hp[0] : (p == 1 ? hp[1] : hp[2]);
        if (p < 3) *reinterpret_cast<unsigned short*>(at + p * MX_HP) = bf16_bits(mine);
      }
    }
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
  // K_x image [k][unit][gate] (one ds_read_b128 = the four gates of a unit) + a zero row for the lists' padding
  {
    const int nv = a.nx * LG;
    float* Kimg = reinterpret_cast<float*>(mx_lds);
    for (int i = tid; i < nv; i += 512) {
      const int k = i / LG, rem = i - k * LG, g = rem / LH, u = rem - g * LH;
      Kimg[(k * LH + u) * 4 + g] = a.Kx[i];
    }
    const int tail = (2 * 16 * MX_HP + 2 * 16 * MX_ZP) / 4 + LH * 4;     // zero row, h image, z image
    for (int i = tid; i < tail; i += 512) Kimg[nv + i] = 0.f;
  }
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (wave < 7) mx_fwd_body<GATE, HASZ, 3, false>(a, mx_lds, 3 * wave);
  else mx_fwd_body<GATE, HASZ, 1, true>(a, mx_lds, 21);
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

template <int ZT>
__global__ __launch_bounds__((6 + ZT) * 64) void lstm_mx_bwd_kernel(MxBwdArgs a) {
  __shared__ __attribute__((aligned(16))) char dzB[2 * 16 * MX_DP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ul = lane >> 4, n = lane & 15, r = n >> 2, p = n & 3;
  const int T = a.T;
  const int row0 = blockIdx.x * MX_R;
  const bool is_lat = ZT > 0 && wave >= 6;
  for (int i = tid; i < 2 * 16 * MX_DP / 4; i += (6 + ZT) * 64) reinterpret_cast<float*>(dzB)[i] = 0.f;

  // A operand: rows of U (or K_z), k = gate column c = 32 s + 8 (l >> 4) + e
  bf16x8 Ar[11][3];
  {
    const int m = lane & 15, kg = lane >> 4;
    const int idx = is_lat ? 16 * (wave - 6) + m : 16 * wave + m;
    const bool ok = is_lat ? idx < a.nz : idx < LH;
    const float* src = is_lat ? a.Kz + (size_t)min(idx, max(a.nz, 1) - 1) * LG : a.U + (size_t)min(idx, LH - 1) * LG;
#pragma unroll
    for (int s = 0; s < 11; ++s) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float w = src[32 * s + 8 * kg + e];
        v[e] = ok ? w : 0.f;
      }
      split8(v, Ar[s]);
    }
  }
  // after the butterfly: lane (ul, r, p) owns row 4 ul + p of the tile, for batch row r
  const int idx = (is_lat ? 16 * (wave - 6) : 16 * wave) + 4 * ul + p;
  const int row = row0 + r;
  const size_t rowc = (size_t)min(row, a.B - 1);
  const bool valid = (is_lat ? idx < a.nz : idx < LH) && row < a.B;
  const int u = min(idx, LH - 1);
  const bool even = !(p & 1), lo = !(p & 2);
  const float* coef_r = a.coef + rowc * T * LG + u;
  const float* aux_r = a.aux + rowc * T * 2 * LH + u;
  const float* dh_r = a.dhs + rowc * T * LH + u;
  float* coef_w = a.coef + rowc * T * LG + u;
  float* dz_w = a.dZ + (is_lat ? rowc * T * a.lddz + min(idx, max(a.nz, 1) - 1) : 0);
  const int dzl_off = (4 * r) * MX_DP + 2 * u;          // dz pieces: line 4 r + q, k = gate * 88 + u

  struct Coef { float ki, kf, kg, ko, kcarry, kc, dh; };
  Coef S[2];
  auto load_set = [&](Coef& k, int t) {
    const size_t tc = (size_t)max(t, 0);
    k.ki = coef_r[tc * LG]; k.kf = coef_r[tc * LG + LH]; k.kg = coef_r[tc * LG + 2 * LH]; k.ko = coef_r[tc * LG + 3 * LH];
    k.kcarry = aux_r[tc * 2 * LH]; k.kc = aux_r[tc * 2 * LH + LH];
    k.dh = dh_r[tc * LH];
  };
  S[0] = S[1] = Coef{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (!is_lat && T > 0) {
    if (T & 1) { load_set(S[0], T - 1); load_set(S[1], T - 2); }        // compile-time indices: S stays in registers
    else { load_set(S[1], T - 1); load_set(S[0], T - 2); }
  }
  float dc = 0.f;
  float zs[4] = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  // dz_{t+1} (image `buf`) . this tile's rows -> the butterfly's result for this lane
  auto matvec = [&](int buf) {
    const char* bp = dzB + buf * 16 * MX_DP + n * MX_DP + (lane >> 4) * 16;
    bf16x8 b[11];
#pragma unroll
    for (int s = 0; s < 11; ++s) b[s] = *reinterpret_cast<const bf16x8*>(bp + 64 * s);
    f32x4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 11; ++s)
#pragma unroll
      for (int q = 0; q < 3; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ar[s][q], b[s], acc, 0, 0, 0);
    const float kA = even ? acc[0] : acc[1], sA = even ? acc[1] : acc[0];
    const float kB = even ? acc[2] : acc[3], sB = even ? acc[3] : acc[2];
    const float wA = kA + dpp_mov<0xB1>(sA);
    const float wB = kB + dpp_mov<0xB1>(sB);
    const float keep = lo ? wA : wB, send = lo ? wB : wA;
    return keep + dpp_mov<0x4E>(send);
  };

  auto step = [&](int t, auto PAR) {
    constexpr int par = decltype(PAR)::value;        // t & 1: this step writes image `par`, reads the other one
    const float x = matvec(par ^ 1);
    if (is_lat) {
      if (valid && t + 1 < T) dz_w[(size_t)(t + 1) * a.lddz] = x;         // dZ_{t+1}
    } else {
      const Coef k = S[par];
      load_set(S[par], t - 2);
      const float dh = k.dh + x;
      dc = fmaf(dh, k.kc, dc);
      float dz[4];
      dz[0] = dc * k.ki; dz[1] = dc * k.kf; dz[2] = dc * k.kg; dz[3] = dh * k.ko;
      dc *= k.kcarry;
      if (valid) {
        char* at = dzB + par * 16 * MX_DP + dzl_off;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          zs[g] += dz[g];
          coef_w[(size_t)t * LG + g * LH] = dz[g];
          __bf16 pc[3];
          split3(dz[g], pc);
#pragma unroll
          for (int q = 0; q < 3; ++q) *reinterpret_cast<unsigned short*>(at + q * MX_DP + 2 * g * LH) = bf16_bits(pc[q]);
        }
      }
    }
    step_barrier();
  };

  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  int t = T - 1;
  if (T & 1) { step(t, P0{}); --t; }          // T odd: step T-1 has even parity
  for (; t >= 1; t -= 2) {
    step(t, P1{});
    step(t - 1, P0{});
  }
  if (is_lat) {
    if (T > 0) {
      const float x = matvec(0);                 // dz_0 is in image 0
      if (valid) dz_w[0] = x;
    }
  } else if (valid) {
#pragma unroll
    for (int g = 0; g < 4; ++g) a.dzsum[rowc * LG + g * LH + u] = zs[g];
  }
}

static bool mx_auto(int B) {
  static const int mode = env_int("CLV_LSTM_MX", -1);       // 0: never, 1: any batch, default: from 512 rows on
  return mode == 1 || (mode < 0 && B >= 512);
}

}  // namespace clv

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
  const size_t lds = (size_t)(nx + 1) * LH * 16 + 2 * 16 * MX_HP + 2 * 16 * MX_ZP + 2 * MX_R * MX_CAP * sizeof(MxItem) + 16;
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
