// lstm_mx.hip -- the LSTM training pass for LARGE batches (>= 4 rows per CU: BASELINE configuration 5, 1024 rows per
// GPU) with the recurrent product on the bf16 matrix cores and EXACT fp32 products (gfx950).
//
//   z_t = x_t . K_x + z^lat_t . K_z + rowbias + h_{t-1} . U;   i,f,o = gate_act(z_i,z_f,z_o), g = tanh(z_c)
//   c_t = f c_{t-1} + i g;   h_t = o tanh(c_t)                          (cl_vrnn/model.py:196-199, 225-228; Keras LSTM)
//
// Why another pair of sequence kernels.  lstm.hip (VALU, U in registers) and round 2's 4x4x1 f32-MFMA forward (removed in round 6) both took
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
//  * the forward pass stores what the backward pass needs, the coefficients of lstm_pair.hip, as unit-major records:
//    coef [B*T,H,4] = (ki, kf, kg, ko) = (g i', c_{t-1} f', i g', tanh(c) o'), aux [B*T,H,2] = (kcarry, kc) =
//    (f, o (1 - tanh(c)^2)), so a backward step is dc += dh kc; dz = (dc ki, dc kf, dc kg, dh ko); dc *= kcarry, and a
//    lane moves its unit's record with one 16-byte and one 8-byte access (round 5: seven 4-byte stores / loads per lane
//    and step cost 22-33 us more per forward launch and 10-14 per backward launch at configuration 5).  The backward
//    pass writes dz over the record's row in the GATE-major order the kernel-gradient product reads: other lanes' records
//    are overwritten, which is safe because a step's dz goes out one step late -- behind the barrier every wave reaches
//    only after it has used its own record of that row.
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
__device__ unsigned long long g_mx_pstamps[8][4];        // [step][producer sub-stamp]
#define MXSTAMP(k, dep) asm volatile("s_memtime %0" : "=s"(mst[k]) : "v"(dep))
#define MXPSTAMP(k, dep) asm volatile("s_memtime %0" : "=s"(pst[k]) : "v"(dep))
#else
#define MXSTAMP(k, dep)
#define MXPSTAMP(k, dep)
#endif

constexpr int MX_R = 4;              // batch rows per workgroup
constexpr int MX_KP = LH * 16 + 16;  // bytes per row of the K_x image [k][unit][gate]: 356 words = 36 mod 64, so the rows of the 16 (row,
                                     // piece) lane groups of a gather start on 16 different bank offsets (1408 B = 32 mod 64: two)
// B-operand images (h, z pieces in the forward kernel, dz pieces in the backward kernel): a [16 columns] x [K] bf16 matrix
// stored by CHUNKS of 8 k values: chunk kc is one 256-byte row of the LDS (all 64 banks), column n's 16 bytes sit in slot
// mx_slot(n, kc).  A ds_read_b128 is served in four groups of 16 lanes that are NOT contiguous ({0-3,12-15,20-27}, ...:
// MI355X_MICROARCH.md, LDS): every group holds each column n exactly once, half of them with one k group and half with
// the next, so "a line of 208 bytes per column" (round-4's first layout) put 5 of a group's 16 lanes on a busy bank (SQ
// counters: 50-54 % of the LDS cycles were conflict cycles, and the shader-clock stamps showed every wave waiting 530-860
// cycles for its B operands at the top of a step).  With one bank row per chunk a group's 16 lanes read 16 different slots
// whatever their k group.  The slot order (piece-major, rows xor-swapped on odd chunks) is chosen for the WRITERS: the
// 16 lanes of a ds_write_b64 group of the backward kernel land on 16 different bank pairs.
constexpr int MX_CHUNK = 256;        // bytes per chunk row
constexpr int MX_HC = 12;            // chunks of the h image (K = 96)
constexpr int MX_ZC = 4;             // chunks of the z image (K = 32)
constexpr int MX_DC = 44;            // chunks of the dz image (K = 352)
__device__ __forceinline__ int mx_slot_q(int r, int q, int kc) { return (((q ^ (kc & 1)) << 2) | r); }      // column n = 4 r + q
constexpr int MX_CAP = 100;          // note-list stride per row (items): 800 bytes = 8 banks more than a multiple of 64, so the
                                     // four rows' lists start on different banks (96 put all four on the same ones)
constexpr int MX_FAST = 8;           // list entries handled without a loop (two rounds of the four piece lanes)
constexpr int MX_PAD = 16;           // entries the producer always pads
constexpr int MX_NXMAX = 96;

struct MxItem { int koff; float v; };       // byte offset of the K_x image row, input value

struct MxFwdArgs {
  int B, T;
  const void* X; int ldx, nx; const float* Kx;        // frames: B*T rows of stride ldx, nx columns (float, or bytes in the XU8 kernels); Kx [nx,352]
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
// Cache policy of the per-step stores (gfx950: 1 = sc0, 2 = nt, 16 = sc1).  The forward pass's records are a write-once
// stream of 550 MB per launch that nothing reads for the next ~1 ms: stored non-temporal they do not push the step's other
// arrays out of the L2 / MALL (config 5: 1.855-1.861 -> 1.818-1.827 ms per step; all stores nt: 1.823-1.832; h is read again
// by the next launches and keeps the default policy).  The backward pass's dz is read by the kernel-gradient product right
// after: nt there costs 4-10 us per launch (profiles/r05_mx_store_policy_ab.txt).
#ifndef MX_FWD_AUX
#define MX_FWD_AUX 2
#endif
#ifndef MX_FWD_HAUX
#define MX_FWD_HAUX 0
#endif
#ifndef MX_BWD_AUX
#define MX_BWD_AUX 0
#endif
typedef __amdgpu_buffer_rsrc_t mx_rsrc_t;
constexpr unsigned MX_OOB = 0x80000000u;
__device__ __forceinline__ mx_rsrc_t mx_rsrc(const void* p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(unsigned)bytes, 0x00020000);
}
__device__ __forceinline__ float mx_load(mx_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
template <int AUX = MX_BWD_AUX>
__device__ __forceinline__ void mx_store(float v, mx_rsrc_t r, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, (int)soff, AUX);
}
typedef unsigned mx_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned mx_u32x4 __attribute__((ext_vector_type(4)));
template <int AUX = MX_BWD_AUX>
__device__ __forceinline__ void mx_store2(float a, float b, mx_rsrc_t r, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b64((mx_u32x2){__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b)}, r, (int)voff, (int)soff, AUX);
}
template <int AUX = MX_BWD_AUX>
__device__ __forceinline__ void mx_store4(float a, float b, float c, float d, mx_rsrc_t r, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b128((mx_u32x4){__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b),
                                                    __builtin_bit_cast(unsigned, c), __builtin_bit_cast(unsigned, d)}, r, (int)voff, (int)soff, AUX);
}
// (the backward pass reads each record once: loaded nt, -8 to -10 us per config-5 step, same file)
#ifndef MX_BWD_LAUX
#define MX_BWD_LAUX 2
#endif
__device__ __forceinline__ void mx_load2(float& a, float& b, mx_rsrc_t r, unsigned voff, unsigned soff) {
  const mx_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, MX_BWD_LAUX);
  const unsigned x = v.x, y = v.y;      // (not __builtin_bit_cast(float, v.y): on a vector ELEMENT this clang reads element 0)
  a = __builtin_bit_cast(float, x); b = __builtin_bit_cast(float, y);
}
__device__ __forceinline__ void mx_load4(float& a, float& b, float& c, float& d, mx_rsrc_t r, unsigned voff, unsigned soff) {
  const mx_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, MX_BWD_LAUX);
  const unsigned x = v.x, y = v.y, z = v.z, w = v.w;
  a = __builtin_bit_cast(float, x); b = __builtin_bit_cast(float, y); c = __builtin_bit_cast(float, z); d = __builtin_bit_cast(float, w);
}

// ---------------------------------------------------------------------------------------------------------------------
// forward.  8 waves; wave w < 7 owns three tiles = the units 12w .. 12w+11 (tile tl: units 12w + 3 j + tl, j = 0..3, so
// that after the butterfly the piece lanes p = 0, 1, 2 of a quad own three CONSECUTIVE units: 12-byte pieces per quad
// in every per-step store instead of isolated floats); wave 7 owns the units 84..87 (one tile).  Wave 0 also turns the
// frames into note lists (two steps ahead), wave 1 the latent inputs z_t into bf16 pieces (one step ahead).
// Batch rows beyond B (the last workgroup of a batch that is no multiple of four) are clones of row B-1: same inputs,
// same values, same addresses -- every access stays unconditional.
//
// Order of a step (shader-clock stamps, tools/mx_stamps.py, decided it): the B operands are requested FIRST and alone --
// with the gather's kernel rows requested next to them every wave waited 530-860 cycles at the top of a step for the LDS
// to work through 8 waves x 9 conflicted ds_read_b128; the rows are requested behind the MFMAs instead and come back
// under the butterfly and the gate math, their FMAs are the last thing before the barrier.
// ---------------------------------------------------------------------------------------------------------------------
// ROLE: what a wave does besides its tiles -- bit 0: frames -> note lists (two steps ahead), bit 1: z_t -> pieces in the
// B-operand image (one step ahead).  The stamps showed a step's length to be the instruction count of its longest
// wave (~5 cycles per instruction of a wave, whatever the instruction): with both jobs in the one-tile wave 7 that wave
// ran ~400 instructions per step (2770 cycles) while waves 0-3 sat at the barrier after 1500; the jobs now ride in
// waves 0 and 1, which are the first to be served on their SIMDs (decoder: lists in wave 7, z pieces in wave 1).
template <int GATE, bool HASZ, int XMODE, int NT, int ROLE>
__device__ __forceinline__ void mx_fwd_body(const MxFwdArgs& a, char* lds, int ubase) {
  // HASX is a template parameter, not `a.nx > 0`: requests under a run-time branch, even a uniform one, count as "maybe
  // not issued" in the compiler's vmcnt bookkeeping, and the wait for the frames requested two steps ago then also waited
  // for the stores of the previous step -- 980 cycles per step in the producer wave (tools/mx_stamps.py)
  // XMODE: 0 no frames, 1 float frames, 2 frames as bytes (round 6: ldx in bytes; a requested byte travels raw, zero-extended, in
  // the register a float would take and is widened where the list is built)
  constexpr bool HASX = XMODE != 0, XU8 = XMODE == 2;
  constexpr bool PROD = (ROLE & 1) != 0, ZPROD = (ROLE & 2) != 0 && HASZ;
  const int lane = threadIdx.x & 63;
  const int ul = lane >> 4, n = lane & 15, r = n >> 2, p = n & 3;
  const int T = a.T;
  const int row0 = blockIdx.x * MX_R;
  const int nrows = min(MX_R, a.B - row0);
  char* Kimg = lds;
  const int zero_off = a.nx * MX_KP;
  char* hB = lds + (a.nx + 1) * MX_KP;                       // [2][MX_HC chunks][256]
  char* zB = hB + 2 * MX_HC * MX_CHUNK;                      // [2][MX_ZC chunks][256]
  MxItem* lists = reinterpret_cast<MxItem*>(zB + 2 * MX_ZC * MX_CHUNK);      // [2][4 rows][MX_CAP]
  MxItem* dump = lists + 2 * MX_R * MX_CAP;                  // [64]: where a producer lane without a note writes
  int* counts = reinterpret_cast<int*>(dump + 64);           // [2][4]

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
  const unsigned rloc = (unsigned)min(r, nrows - 1);
  const size_t rowc = (size_t)row0 + rloc;
  // accumulator layout (before the butterfly): tile tl, register i = gate i of unit ubase + NT ul + tl, for (row r, piece p)
  const int ucol0 = (ubase + NT * ul) * 16;                    // tile tl: + 16 tl
  // after the butterfly this lane finishes tile tp of its wave: unit `unit` of row r (p == 3: a second copy of p == 2;
  // one tile: all four lanes hold the same cell)
  const int tp = NT == 3 ? min(p, 2) : 0;
  const int unit = ubase + NT * ul + tp;
  float rb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) rb[i] = a.rowbias ? a.rowbias[rowc * LG + i * LH + unit] : 0.f;
  const bool even = !(p & 1), lo = !(p & 2);
  float c = 0.f;
  // per-step stores: buffer instructions, the step's row offset in an SGPR (no vector address arithmetic)
  const mx_rsrc_t r_c = mx_rsrc(a.coef + (size_t)row0 * T * LG, (size_t)nrows * T * LG * 4);
  const mx_rsrc_t r_a = mx_rsrc(a.aux + (size_t)row0 * T * 2 * LH, (size_t)nrows * T * 2 * LH * 4);
  const mx_rsrc_t r_h = mx_rsrc(a.hs + (size_t)row0 * T * LH, (size_t)nrows * T * LH * 4);
  const unsigned v_c = (rloc * T * LG + 4 * unit) * 4u, v_a = (rloc * T * 2 * LH + 2 * unit) * 4u, v_h = (rloc * T * LH + unit) * 4u;
  // h pieces -> B-operand image: chunk unit / 8, slot of column 4 r + q, element unit % 8
  int hw_off[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) hw_off[q] = (unit >> 3) * MX_CHUNK + mx_slot_q(r, q, unit >> 3) * 16 + 2 * (unit & 7);
  // B-operand reads: column n = 4 r + p of chunk 4 s + (lane >> 4)
  const int kg = lane >> 4;
  const int b_off = kg * MX_CHUNK + mx_slot_q(r, p, kg) * 16;     // + s * 4 * MX_CHUNK (chunk parity = kg & 1)

  // ---- producer (wave 0): lane = (row pr = lane >> 4, j = lane & 15) takes the columns 6 j .. 6 j + 5 of its row ------
  // Branch-free: per-lane note count, exclusive prefix over the row's 16 lanes by four DPP row shifts, six stores whose
  // address is the list slot or the lane's dump slot.  (Round 4's first version walked the rows one after the other with
  // ballots and stores under divergent branches: 1340 cycles of the producer's 2650-cycle step, with every other wave
  // parked at the barrier for 500-1200 cycles.)
  constexpr int PC = 6;
#ifdef MX_STAMPS
  unsigned long long pst[4] = {0, 0, 0, 0};
#endif
  const int pr = lane >> 4, pj = lane & 15;
  float fr[2][PC];           // two register sets of frame values in flight (set = step parity): requested at the end of step t
                             // for frame t + 4, compacted in step t + 2.  (One set, one step of lookahead: +70 us per launch --
                             // the wait then also covers the stores issued in between, and their acknowledgements are slow.)
  float zr[2][2];            // ... and of z pairs
  const int zlat = 2 * pj;
  auto load_frames = [&](float (&f)[PC], int t) {
    const size_t frame = ((size_t)min(row0 + pr, a.B - 1) * T + min(t, T - 1)) * a.ldx;
    if (XU8) {
      const unsigned char* bp = static_cast<const unsigned char*>(a.X) + frame;
#pragma unroll
      for (int i = 0; i < PC; ++i) f[i] = __builtin_bit_cast(float, (unsigned)bp[min(PC * pj + i, a.nx - 1)]);
    } else {
      const float* fp = static_cast<const float*>(a.X) + frame;
#pragma unroll
      for (int i = 0; i < PC; ++i) f[i] = fp[min(PC * pj + i, a.nx - 1)];    // raw: nothing touches a requested value before
    }                                                                         // its consumer does
  };
  auto load_z = [&](float (&z)[2], int t) {
    const float* zp = a.Z + ((size_t)min(row0 + pr, a.B - 1) * T + min(t, T - 1)) * a.ldz;
    z[0] = zp[min(zlat, a.nz - 1)];
    z[1] = zp[min(zlat + 1, a.nz - 1)];
  };
  auto compact = [&](const float (&f)[PC], int buf) {      // frame of row pr -> lists[buf][pr], counts[buf][pr]
    MxItem* L = lists + (buf * MX_R + pr) * MX_CAP;
    const MxItem padding = MxItem{zero_off, 0.f};
    L[pj] = padding;                                       // slots 0..15 (LDS operations of a wave are in order)
    bool on[PC];
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < PC; ++i) {
      on[i] = PC * pj + i < a.nx && (XU8 ? __builtin_bit_cast(unsigned, f[i]) != 0u : f[i] != 0.f);
      cnt += on[i] ? 1 : 0;
    }
    MXPSTAMP(0, cnt);                                      // the frame values are here
    int incl = cnt;                                        // inclusive prefix over the 16 lanes of the row
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, true);     // row_shr:1
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, true);     // row_shr:2
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, true);     // row_shr:4
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, true);     // row_shr:8
    const int total = __builtin_amdgcn_ds_bpermute((lane | 15) << 2, incl);  // the row's last lane holds its count
    MXPSTAMP(1, total);
    int pos = incl - cnt;
    MxItem* mine = dump + lane;
#pragma unroll
    for (int i = 0; i < PC; ++i) {
      MxItem* at = on[i] ? L + pos : mine;
      *at = MxItem{(PC * pj + i) * MX_KP, XU8 ? (float)__builtin_bit_cast(unsigned, f[i]) : f[i]};
      pos += on[i] ? 1 : 0;
    }
    // beyond 16 notes: pad up to the next round of four (a row's consumers walk its own count)
    MxItem* tail = (total >= MX_PAD && pj < ((total + 3) & ~3) - total) ? L + total + pj : mine;
    *tail = padding;
    if (pj == 0) counts[buf * MX_R + pr] = total;
    MXPSTAMP(2, pos);
  };
  auto stage_z = [&](const float (&z)[2], int buf) {           // z pair -> three piece images, 4 bytes each
    __bf16 p0[3], p1[3];
    split3(zlat < a.nz ? z[0] : 0.f, p0);
    split3(zlat + 1 < a.nz ? z[1] : 0.f, p1);
    const int kc = zlat >> 3;
    char* at = zB + buf * MX_ZC * MX_CHUNK + kc * MX_CHUNK + 2 * (zlat & 7);
#pragma unroll
    for (int q = 0; q < 3; ++q)
      *reinterpret_cast<unsigned*>(at + mx_slot_q(pr, q, kc) * 16) = (unsigned)bf16_bits(p0[q]) | ((unsigned)bf16_bits(p1[q]) << 16);
  };

  // next step's input contribution in the accumulator layout: lane p takes the notes p, p + 4, ... of its row
  float xinit[NT][4];
  struct Rows { float v[2]; float4 k[NT]; int koff1; int cnt; };      // one set of kernel rows: round 0, then reused by round 1
  auto gather_items = [&](int buf, MxItem (&items)[2], int& cnt) {
    const MxItem* Lr = lists + (buf * MX_R + r) * MX_CAP;
    items[0] = Lr[p];
    items[1] = Lr[p + 4];
    cnt = counts[buf * MX_R + r];
  };
  auto gather_rows = [&](const MxItem (&items)[2], int cnt, Rows& g) {          // round 0: the row's notes 0..3
    g.cnt = cnt;
    g.v[0] = items[0].v; g.v[1] = items[1].v;
    g.koff1 = items[1].koff;
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) g.k[tl] = *reinterpret_cast<const float4*>(Kimg + items[0].koff + ucol0 + 16 * tl);
  };
  auto gather_rows1 = [&](Rows& g) {                                            // round 1 (notes 4..7) into the same registers
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) g.k[tl] = *reinterpret_cast<const float4*>(Kimg + g.koff1 + ucol0 + 16 * tl);
  };
  auto gather_finish = [&](int buf, Rows& g) {           // the prologue's (unpipelined) form
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
      xinit[tl][0] = g.v[0] * g.k[tl].x; xinit[tl][1] = g.v[0] * g.k[tl].y;
      xinit[tl][2] = g.v[0] * g.k[tl].z; xinit[tl][3] = g.v[0] * g.k[tl].w;
    }
    gather_rows1(g);
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
      xinit[tl][0] = fmaf(g.v[1], g.k[tl].x, xinit[tl][0]); xinit[tl][1] = fmaf(g.v[1], g.k[tl].y, xinit[tl][1]);
      xinit[tl][2] = fmaf(g.v[1], g.k[tl].z, xinit[tl][2]); xinit[tl][3] = fmaf(g.v[1], g.k[tl].w, xinit[tl][3]);
    }
    const MxItem* Lr = lists + (buf * MX_R + r) * MX_CAP;
    for (int j = MX_FAST; j < g.cnt; j += 4) {        // denser frames (rare for piano-rolls); a row walks its own count
      const MxItem it = Lr[j + p];
#pragma unroll
      for (int tl = 0; tl < NT; ++tl) {
        const float4 kr = *reinterpret_cast<const float4*>(Kimg + it.koff + ucol0 + 16 * tl);
        xinit[tl][0] = fmaf(it.v, kr.x, xinit[tl][0]); xinit[tl][1] = fmaf(it.v, kr.y, xinit[tl][1]);
        xinit[tl][2] = fmaf(it.v, kr.z, xinit[tl][2]); xinit[tl][3] = fmaf(it.v, kr.w, xinit[tl][3]);
      }
    }
  };

  // ---- prologue ------------------------------------------------------------------------------------------------------
  if (PROD) {
    if (HASX) {
      load_frames(fr[0], 0);
      load_frames(fr[1], 1);
      compact(fr[0], 0);
      compact(fr[1], 1);
      load_frames(fr[0], 2);
      load_frames(fr[1], 3);
    } else {
      for (int j = lane; j < 2 * MX_R * MX_CAP; j += 64) lists[j] = MxItem{zero_off, 0.f};
      if (lane < 2 * MX_R) counts[lane] = 0;                    // never rewritten: every list is padding
    }
  }
  if (ZPROD) {
    load_z(zr[0], 0);
    stage_z(zr[0], 0);
    load_z(zr[0], 1);       // set 0: z_{t+1} of step 0;  set 1: of step 1
    load_z(zr[1], 2);
  }
  __syncthreads();
  {
    Rows g;
    MxItem items[2];
    int cnt;
    gather_items(0, items, cnt);
    gather_rows(items, cnt, g);
    gather_finish(0, g);
  }
  // every load of the prologue has landed before the loop is entered (the wait-count bookkeeping merges the loop-entry
  // state with the back edge's: see lstm_pair.hip)
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();

  // Software pipeline (MX_PIPE, default): what does not depend on a step's MFMAs is issued BETWEEN them, two vector
  // instructions per MFMA (an MFMA holds the issue port for 8 of its 16 cycles): the seven stores of the PREVIOUS step's
  // outputs (held in registers across the barrier) and the FMAs of the NEXT step's input gather (kernel rows requested
  // right behind the B operands).  A step's post-MFMA phase -- what every wave's SIMD partner waits out -- shrinks by
  // ~40 instructions per wave.
  float pend[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto store_pending = [&](int k, unsigned tt) {        // tt = the step the values belong to; (unsigned)-1: beyond num_records
    if (MX_ABL & 2) return;
    if (MX_ABL & 16) tt = tt == 0xffffffffu ? tt : 0u;
    switch (k) {        // one 16-byte, one 8-byte and one 4-byte store per lane and step (seven 4-byte ones: +9-27 us per launch)
      case 0: mx_store4<MX_FWD_AUX>(pend[0], pend[1], pend[2], pend[3], r_c, v_c, tt * (LG * 4)); break;
      case 1: mx_store2<MX_FWD_AUX>(pend[4], pend[5], r_a, v_a, tt * (2 * LH * 4)); break;
      case 2: mx_store<MX_FWD_HAUX>(pend[6], r_h, v_h, tt * (LH * 4)); break;
      default: break;
    }
  };
  float xin[2][NT][4];        // input contribution of the step of parity [.]: written during the previous step's MFMAs
#pragma unroll
  for (int tl = 0; tl < NT; ++tl)
#pragma unroll
    for (int i = 0; i < 4; ++i) { xin[0][tl][i] = xinit[tl][i]; xin[1][tl][i] = 0.f; }

  auto step = [&](int t, auto PAR) {
    constexpr int cur = decltype(PAR)::value;
#ifdef MX_STAMPS
    unsigned long long mst[8];
    mst[7] = 0;
#endif
    MXSTAMP(0, c);
    // list entries of the NEXT step's inputs (small), then this step's B operands = pieces of h_{t-1} (and z_t)
    MxItem items[2];
    int cnt = 0;
    if (!(MX_ABL & 4)) gather_items(cur ^ 1, items, cnt);
    const char* hb = hB + cur * MX_HC * MX_CHUNK + b_off;
    bf16x8 bh[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) bh[s] = *reinterpret_cast<const bf16x8*>(hb + s * 4 * MX_CHUNK);
    bf16x8 bz;
    if (HASZ) bz = *reinterpret_cast<const bf16x8*>(zB + cur * MX_ZC * MX_CHUNK + b_off);
    Rows g;
    if (!(MX_ABL & 4)) gather_rows(items, cnt, g);        // the kernel rows come back under the first MFMAs
    // the producer's lists / z image for later steps, AHEAD of its (few) MFMAs: nothing here depends on this step, and the
    // SIMD partner (a three-tile wave, served first) is in its MFMA phase now -- behind the MFMAs this work was the tail
    // every other wave waited for (wave 7: MFMA phase 1034 + compaction 634 cycles, stamps)
    if (PROD && HASX && !(MX_ABL & 1)) compact(fr[cur], cur);     // frame t + 2 -> the list buffer step t - 1 finished with
    if (ZPROD && !(MX_ABL & 1)) stage_z(zr[cur], cur ^ 1);        // z_{t+1}
    MXSTAMP(1, bh[2][0]);                           // the B operands are here
    f32x4v acc[NT];
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) acc[tl] = f32x4v{xin[cur][tl][0], xin[cur][tl][1], xin[cur][tl][2], xin[cur][tl][3]};
    // vector work of MFMA slot m -- slots 0..2: the previous step's three stores; A0 .. A0+2NT-1: round-0 FMAs (two per slot); slot A0+2NT: the round-1 rows are
    // requested into the same registers; B0 .. B0+2NT-1: round-1 FMAs
    // (the FMAs three slots earlier, right behind the stores: no change, profiles/r05_mx_wide_records_ab.txt)
    constexpr int NM = 9 * NT, A0 = NT == 3 ? 7 : 1, B0 = NT == 3 ? 20 : 6;
    auto slot = [&](int m) {
      if (m < 3) store_pending(m, (unsigned)(t - 1));
      if (!(MX_ABL & 4)) {
        if (m >= A0 && m < A0 + 2 * NT) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int e = 2 * (m - A0) + h, tl = e / 4, i = e % 4;
            const float kv = i == 0 ? g.k[tl].x : i == 1 ? g.k[tl].y : i == 2 ? g.k[tl].z : g.k[tl].w;
            xin[cur ^ 1][tl][i] = g.v[0] * kv;
          }
        }
        if (m == A0 + 2 * NT) gather_rows1(g);
        if (m >= B0 && m < B0 + 2 * NT) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int e = 2 * (m - B0) + h, tl = e / 4, i = e % 4;
            const float kv = i == 0 ? g.k[tl].x : i == 1 ? g.k[tl].y : i == 2 ? g.k[tl].z : g.k[tl].w;
            xin[cur ^ 1][tl][i] = fmaf(g.v[1], kv, xin[cur ^ 1][tl][i]);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int tl = 0; tl < NT; ++tl) {
          if (!(MX_ABL & 8) || (s == 0 && q == 0)) acc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ar[tl][s][q], bh[s], acc[tl], 0, 0, 0);
          slot((s * 3 + q) * NT + tl);
        }
    if (HASZ) {
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int tl = 0; tl < NT; ++tl) acc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Az[tl][q], bz, acc[tl], 0, 0, 0);
    }
    static_assert(B0 + 2 * NT <= NM && A0 + 2 * NT < B0 && NM >= 7, "every slot's work fits under the MFMAs");
    MXSTAMP(2, acc[NT - 1][0]);                     // the last MFMA's result is here
    MXSTAMP(3, acc[0][1]);
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
    {          // h pieces first: they are what the next step of EVERY wave waits for
      __bf16 hp[3];
      split3(h, hp);
      char* at = hB + (cur ^ 1) * MX_HC * MX_CHUNK;
      if (NT == 3) {
#pragma unroll
        for (int q = 0; q < 3; ++q) *reinterpret_cast<unsigned short*>(at + hw_off[q]) = bf16_bits(hp[q]);
      } else {             // four lanes hold the same h: lane p writes piece min(p, 2)
        const __bf16 mine = p == 0 ? hp[0] : (p == 1 ? hp[1] : hp[2]);
        const int off = p == 0 ? hw_off[0] : (p == 1 ? hw_off[1] : hw_off[2]);
        *reinterpret_cast<unsigned short*>(at + off) = bf16_bits(mine);
      }
    }
    // this step's outputs: stored under the next step's MFMAs
    pend[0] = gg * gate_grad<GATE>(z[0], ig);
    pend[1] = kf;
    pend[2] = ig * (1.f - gg * gg);
    pend[3] = tc * gate_grad<GATE>(z[3], og);
    pend[4] = fg;
    pend[5] = og * (1.f - tc * tc);
    pend[6] = h;
    if (!(MX_ABL & 4)) {           // denser frames (rare for piano-rolls): the notes beyond the eighth; a row walks its own count
      const MxItem* Lr = lists + ((cur ^ 1) * MX_R + r) * MX_CAP;
      for (int j = MX_FAST; j < g.cnt; j += 4) {
        const MxItem it = Lr[j + p];
#pragma unroll
        for (int tl = 0; tl < NT; ++tl) {
          const float4 kr = *reinterpret_cast<const float4*>(Kimg + it.koff + ucol0 + 16 * tl);
          xin[cur ^ 1][tl][0] = fmaf(it.v, kr.x, xin[cur ^ 1][tl][0]); xin[cur ^ 1][tl][1] = fmaf(it.v, kr.y, xin[cur ^ 1][tl][1]);
          xin[cur ^ 1][tl][2] = fmaf(it.v, kr.z, xin[cur ^ 1][tl][2]); xin[cur ^ 1][tl][3] = fmaf(it.v, kr.w, xin[cur ^ 1][tl][3]);
        }
      }
    }
    // the producer's requests for two steps ahead, BEHIND the last use of the register set they land in (issued ahead of
    // it they need fresh registers, and the copies back at the loop's end wait for the loads just issued: every step
    // then costs a trip to HBM -- lstm_pair.hip)
    if (PROD && HASX && !(MX_ABL & 1)) load_frames(fr[cur], t + 4);
    if (ZPROD && !(MX_ABL & 1)) load_z(zr[cur], t + 3);
#ifdef MX_STAMPS
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(mst[6]));          // arrival at the barrier
    if (blockIdx.x == 0 && lane == 0 && t >= 64 && t < 72) {
      const int wv = threadIdx.x >> 6;
#pragma unroll
      for (int k = 0; k < 7; ++k) g_mx_stamps[t - 64][wv][k] = mst[k];
      if (PROD) {
        g_mx_pstamps[t - 64][0] = pst[0] - mst[0]; g_mx_pstamps[t - 64][1] = pst[1] - pst[0];
        g_mx_pstamps[t - 64][2] = pst[2] - pst[1]; g_mx_pstamps[t - 64][3] = mst[1] - pst[2];
      }
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
#pragma unroll
  for (int k = 0; k < 3; ++k) store_pending(k, (unsigned)(T - 1));       // the last step's outputs
}

constexpr size_t mx_fwd_lds(int nx) {
  return (size_t)(nx + 1) * MX_KP + 2 * MX_HC * MX_CHUNK + 2 * MX_ZC * MX_CHUNK + (2 * MX_R * MX_CAP + 64) * sizeof(MxItem) + 2 * MX_R * 4;
}

template <int GATE, bool HASZ, int XMODE>
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
    const int tail = (MX_KP + 2 * MX_HC * MX_CHUNK + 2 * MX_ZC * MX_CHUNK) / 4;      // zero row, h images, z images
    for (int i = tid; i < tail; i += 512) rest[i] = 0.f;
  }
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // both producer jobs ride in the one-tile wave 7: a three-tile wave has no registers left for two sets of frame values in
  // flight (encoder wave 0: eight spilled registers; with ONE set in wave 3 or 0 the launch took 390-400 us against 320)
  if (wave < 7) mx_fwd_body<GATE, HASZ, XMODE, 3, 0>(a, mx_lds, 12 * wave);
  else mx_fwd_body<GATE, HASZ, XMODE, 1, 3>(a, mx_lds, 84);
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
// four floats -> three piece images of four bf16 each (common.h: bf16_split_pair, 7 instructions per pair)
__device__ __forceinline__ unsigned mx_pack2(float lo, float hi) { return bf16_pack2(lo, hi); }
__device__ __forceinline__ void mx_split4(const float (&x)[4], uint2 (&piece)[3]) {
  unsigned lo[3], hi[3];
  bf16_split_pair(x[0], x[1], lo);
  bf16_split_pair(x[2], x[3], hi);
#pragma unroll
  for (int q = 0; q < 3; ++q) piece[q] = make_uint2(lo[q], hi[q]);
}

// dz_{t+1} (image `buf`) . the tile's rows; the butterfly leaves lane (ul, r, p) with row 4 ul + p of the tile.
// slot(m) runs behind MFMA m (0..32): vector work that does not depend on this product (the unit waves store the previous
// step's dz there)
struct MxNoSlot { __device__ __forceinline__ void operator()(int) const {} };
template <class Slot = MxNoSlot>
__device__ __forceinline__ float mx_bwd_matvec(const char* dzB, int buf, const bf16x8 (&Ar)[11][3], Slot slot = Slot()) {
  const int lane = threadIdx.x & 63, n = lane & 15, p = n & 3, kg = lane >> 4;
  const bool even = !(p & 1), lo = !(p & 2);
  const char* bp = dzB + buf * MX_DC * MX_CHUNK + kg * MX_CHUNK + mx_slot_q(n >> 2, p, kg) * 16;
  bf16x8 b[11];
#pragma unroll
  for (int s = 0; s < 11; ++s) b[s] = *reinterpret_cast<const bf16x8*>(bp + s * 4 * MX_CHUNK);
  f32x4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 11; ++s)
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      if (!(MX_ABL & 128) || (s == 0 && q == 0)) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ar[s][q], b[s], acc, 0, 0, 0);
      slot(3 * s + q);
    }
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
  const unsigned v_c = valid ? (rloc * T * LG + u) * 4u : MX_OOB;              // dz, written in place: [gate][unit]
  const unsigned v_ci = valid ? (rloc * T * LG + 4 * u) * 4u : MX_OOB;         // the forward pass's record: [unit][4]
  const unsigned v_a = valid ? (rloc * T * 2 * LH + 2 * u) * 4u : MX_OOB;      // ... and [unit][2]
  const unsigned v_d = valid ? (rloc * T * LH + u) * 4u : MX_OOB;
  // dz pieces: k = 4 u + gate: chunk u / 2, the lane's four gates are 8 contiguous bytes; a lane without a unit writes to
  // its own dump slot behind the images
  int dzl_off[3];
#pragma unroll
  for (int q = 0; q < 3; ++q)
    dzl_off[q] = valid ? (u >> 1) * MX_CHUNK + mx_slot_q(r, q, u >> 1) * 16 + (u & 1) * 8 : 2 * MX_DC * MX_CHUNK + (int)(threadIdx.x & 63) * 8;

  // Two register sets, the loop body is two steps: the coefficients of step t are requested at the end of step t + 2,
  // behind the last use of the set they land in (lstm_pair.hip).  An odd T runs one step more: step t = -1 lies beyond
  // num_records -- zero coefficients, dz = 0, stores dropped.
  struct Coef { float ki, kf, kg, ko, kcarry, kc, dh; };
  auto load_set = [&](Coef& k, int t) {
    const unsigned tc = (unsigned)t;
    mx_load4(k.ki, k.kf, k.kg, k.ko, r_c, v_ci, tc * (LG * 4));
    mx_load2(k.kcarry, k.kc, r_a, v_a, tc * (2 * LH * 4));
    k.dh = mx_load(r_d, v_d, tc * (LH * 4));
  };
  Coef SA, SB;
  load_set(SA, T - 1);
  load_set(SB, T - 2);
  float dc = 0.f;
  float zs[4] = {0.f, 0.f, 0.f, 0.f};
  float pend[4] = {0.f, 0.f, 0.f, 0.f};         // the previous step's dz: stored (and summed) under this step's MFMAs
  __builtin_amdgcn_s_waitcnt(0x0F70);          // the prologue's loads have landed (see the forward kernel)
  __syncthreads();

  auto step = [&](int i, auto PAR, Coef& k) {
    constexpr int par = decltype(PAR)::value;        // i & 1: this step writes image `par`, reads the other one
    const int t = T - 1 - i;
#ifdef MX_STAMPS
    unsigned long long mst[8];
    mst[4] = mst[5] = mst[6] = mst[7] = 0;
#endif
    MXSTAMP(0, dc);
    // the previous step's (t + 1) dz goes out behind the first MFMAs; at i == 0 there is none (row T of a batch row would be
    // the next row's step 0: beyond num_records instead)
    const unsigned soff_prev = i > 0 ? (unsigned)(t + 1) * (LG * 4) : MX_OOB;
    const float x = mx_bwd_matvec(dzB, par ^ 1, Ar, [&](int m) {
      if (m < 4) {
        if (!(MX_ABL & 64)) mx_store(pend[m], r_c, v_c + m * LH * 4, soff_prev);
        zs[m] += pend[m];
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    MXSTAMP(1, x);                                   // MFMAs + butterfly done
    const float dh = k.dh + x;
    dc = fmaf(dh, k.kc, dc);
    float dz[4];
    dz[0] = dc * k.ki; dz[1] = dc * k.kf; dz[2] = dc * k.kg; dz[3] = dh * k.ko;
    dc *= k.kcarry;
    char* at = dzB + (valid ? par * MX_DC * MX_CHUNK : 0);
    uint2 pc[3];
    mx_split4(dz, pc);
#pragma unroll
    for (int q = 0; q < 3; ++q)
      if (!(MX_ABL & 256)) *reinterpret_cast<uint2*>(at + dzl_off[q]) = pc[q];
#pragma unroll
    for (int g = 0; g < 4; ++g) pend[g] = dz[g];
    MXSTAMP(2, zs[3]);                               // cell math, dz pieces written, stores issued
    if (!(MX_ABL & 32)) load_set(k, t - 2);
#ifdef MX_STAMPS
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(mst[3]));          // arrival at the barrier
    if (blockIdx.x == 0 && lane == 0 && i >= 64 && i < 72) {
#pragma unroll
      for (int q = 0; q < 4; ++q) g_mx_stamps[i - 64][wave][q] = mst[q];
    }
#endif
    step_barrier();
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  for (int i = 0; i < T; i += 2) {
    step(i, P0{}, SA);
    step(i + 1, P1{}, SB);
  }
  // the last executed step's dz (step 0, or the padded step -1 of an odd T whose dz is 0 and whose address lies beyond num_records)
  {
    const int tl = (T & 1) ? -1 : 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (!(MX_ABL & 64)) mx_store(pend[g], r_c, v_c + g * LH * 4, (unsigned)tl * (LG * 4));
      zs[g] += pend[g];
    }
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
  __shared__ __attribute__((aligned(16))) char dzB[2 * MX_DC * MX_CHUNK + 64 * 8];       // two images + the dump slots
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < (2 * MX_DC * MX_CHUNK + 64 * 8) / 4; i += (6 + ZT) * 64) reinterpret_cast<float*>(dzB)[i] = 0.f;
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
extern "C" int clv_debug_mx_pstamps(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(clv::g_mx_pstamps), sizeof(unsigned long long) * 8 * 4);
}
#endif

extern "C" int clv_lstm_mx_supported(int B, int H, int nx, int nz) {
  return H == clv::LH && B >= 1 && nx >= 0 && nx <= clv::MX_NXMAX && nz >= 0 && nz <= 32 && clv::mx_auto(B);
}

extern "C" int clv_lstm_mx_fwd(int B, int T, int H, int gate_act,
                               const void* X, int x_u8, int ldx, int nx, const float* Kx,
                               const float* Z, int ldz, int nz, const float* Kz,
                               const float* rowbias, const float* U,
                               float* hs, float* coef, float* aux, void* stream) {
  using namespace clv;
  if (H != LH || B <= 0 || T < 1 || nx < 0 || nx > MX_NXMAX || nz < 0 || nz > 32) return CLV_EINVAL;
  if ((nx > 0 && (!X || !Kx || ldx < nx)) || (nz > 0 && (!Z || !Kz || ldz < nz)) || !U || !hs || !coef || !aux) return CLV_EINVAL;
  if (gate_act != CLV_GATE_HARD_SIGMOID && gate_act != CLV_GATE_SIGMOID) return CLV_EINVAL;
  if (((uintptr_t)coef) % 16 != 0 || ((uintptr_t)aux) % 8 != 0) return CLV_EINVAL;      // the records move as 16- and 8-byte accesses
  hipStream_t s = (hipStream_t)stream;
  ProfScope p("lstm_mx_fwd", s);
  MxFwdArgs a{B, T, X, ldx, nx, Kx, Z, ldz, nz, Kz, rowbias, U, hs, coef, aux};
  const size_t lds = mx_fwd_lds(nx);
  const dim3 grid((B + MX_R - 1) / MX_R), block(512);
  const bool hard = gate_act == CLV_GATE_HARD_SIGMOID;
#define MX_LAUNCH(G, Zf, Xf)                                                                  \
  do {                                                                                        \
    auto kern = lstm_mx_fwd_kernel<G, Zf, Xf>;                                                \
    if (int e = allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds)) return e;   \
    hipLaunchKernelGGL(kern, grid, block, lds, s, a);                                         \
  } while (0)
#define MX_LAUNCH_G(G)                                                                        \
  do {                                                                                        \
    const int xm = nx > 0 ? (x_u8 ? 2 : 1) : 0;                                               \
    if (nz > 0) { if (xm == 2) MX_LAUNCH(G, true, 2); else if (xm) MX_LAUNCH(G, true, 1); else MX_LAUNCH(G, true, 0); }     \
    else { if (xm == 2) MX_LAUNCH(G, false, 2); else if (xm) MX_LAUNCH(G, false, 1); else MX_LAUNCH(G, false, 0); }        \
  } while (0)
  if (hard) MX_LAUNCH_G(CLV_GATE_HARD_SIGMOID); else MX_LAUNCH_G(CLV_GATE_SIGMOID);
#undef MX_LAUNCH_G
#undef MX_LAUNCH
  return launch_status();
}

extern "C" int clv_lstm_mx_bwd(int B, int T, int H, const float* U, const float* dhs, const float* aux,
                               float* coef_inout_dz, float* dzsum, const float* Kz, int nz, float* dZ, int lddz,
                               void* stream) {
  using namespace clv;
  if (H != LH || B <= 0 || T < 1 || !U || !dhs || !aux || !coef_inout_dz || !dzsum) return CLV_EINVAL;
  if (nz < 0 || nz > 32 || (nz > 0 && (!Kz || !dZ || lddz < nz))) return CLV_EINVAL;
  if (((uintptr_t)coef_inout_dz) % 16 != 0 || ((uintptr_t)aux) % 8 != 0) return CLV_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  ProfScope p("lstm_mx_bwd", s);
  MxBwdArgs a{B, T, U, dhs, aux, coef_inout_dz, dzsum, Kz, nz, dZ, lddz};
  const dim3 grid((B + MX_R - 1) / MX_R);
  if (nz > 16) hipLaunchKernelGGL((lstm_mx_bwd_kernel<2>), grid, dim3(8 * 64), 0, s, a);
  else if (nz > 0) hipLaunchKernelGGL((lstm_mx_bwd_kernel<1>), grid, dim3(7 * 64), 0, s, a);
  else hipLaunchKernelGGL((lstm_mx_bwd_kernel<0>), grid, dim3(6 * 64), 0, s, a);
  return launch_status();
}
