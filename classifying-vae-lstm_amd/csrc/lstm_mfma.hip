// lstm_mfma.hip -- the LSTM sequence forward for LARGE batches: the recurrent product on the matrix cores (gfx950).
//
// lstm.hip keeps the recurrence on the VALU because at the reference's batch sizes a GPU has one CU per batch row
// and a 16-row MFMA tile cannot be filled.  From ~4 rows per CU on (BASELINE configuration 5: 1024 rows per GPU) the
// multi-block f32 MFMA fits: v_mfma_f32_4x4x1_16B_f32 does 16 independent 4x4 outer products per instruction,
//     D_b[i][j] += A_b[i] * B_b[j],   block b = lane / 4,  A_b[i] in lane 4b+i,  B_b[j] in lane 4b+j,
//     D_b[i][j] in register i of lane 4b+j                      (layout and rate: tools/probes/mfma4x4_probe.hip;
//     8 cycles per instruction and SIMD, 40 cycles from one instruction to the next on the same accumulator),
// at the f32 vector rate but with no per-lane reduction tree, one instruction per 256 MACs, and the VALU left to
// the gate math.  A workgroup owns FOUR batch rows (the i index).  Block b = (gate cg = b / 4, k residue kb = b % 4):
//     A_b[i]  = h[row i][k = 4m + kb]                    (the same in all four gate blocks of a k residue)
//     B_b[j]  = U[4m + kb][gate cg, unit 4q + j]         (resident in registers: 22 k-steps per unit quad q)
// so 22 instructions accumulate, for the unit quad q, the partial sums over k = kb (mod 4) of all four gates of four
// units for four rows.  Two DPP row rotations (lanes 4 and 8 apart) add the four residues; lane (cg, kb, j) then
// finishes row kb: it activates its gate, the four gate lanes of a (row, unit) -- 16 lanes apart -- exchange the
// activated values through ds_bpermute, every one of them updates c and h (each stores a different output), and h
// goes back to LDS in the A-operand order [row][k residue][m].
// 8 waves, unit quads w, w+8, w+16 per wave (22 quads), 66 weights per lane.
// (The unit is the fastest lane index on purpose: with the GATE there the exchange is a DPP quad broadcast, but the
// per-step loads and stores of a quarter wave then touch 16 four-byte pieces instead of 4 sixteen-byte ones and the
// kernel ran 1.5x slower: 541 us against 354 us per launch at 1024 x 256.)
//
// Measured at configuration 5 (1024 rows, T = 256): 354 us per launch against 368 us for the VALU kernel at two rows
// per workgroup.  A step is the MFMA phase (66 instructions per wave, two waves per SIMD: ~1060 cycles) FOLLOWED by
// the gate phase (~140 vector instructions per wave: residue sums, activation, exchange, cell, stores): the next
// step's products need every h of this one, so the two phases of the four rows a CU owns cannot overlap.
// Round 3, tried: 16 waves (four per SIMD, one or two unit quads each, 122 registers) so that the gate phase's dependent
// chains have four waves to interleave: 1.29 ms for the two forward launches of configuration 5 against 0.767 ms -- with
// one or two quads a wave touches each accumulator every 16-32 cycles, inside the 40-cycle dependent latency of the 4x4x1
// MFMA, and the z variant spills.
#include <stdlib.h>

#include "lstm_common.h"

namespace clv {

typedef float f32x4m __attribute__((ext_vector_type(4)));

constexpr int MR = 4;             // batch rows per workgroup
constexpr int MNW = 8;            // waves
constexpr int MQ = 22;            // unit quads
constexpr int MKS = LH / 4;       // 22 k-steps of 4
constexpr int MHS = 28;           // LDS stride of one (row, k residue) slice of h: the 16 slices of a ds_read_b128 lane
                                  // group start at banks 28 c mod 64, four banks each, all distinct (see lstm_pair.hip)

__device__ float g_mfma_dump[128];

struct LstmMfmaFwdArgs {
  int B, T;
  const float* xproj;    // [B,T,352]
  const float* rowbias;  // [B,352] or null
  const float* U;        // [88,352]
  float* hs; float* cs; float* gates;      // [B,T,88] [B,T,88] [B,T,352] (z_i, z_f, tanh(z_c), z_o)
  float* hT; float* cT;                    // [B,88] or null
  // a second, per-step input that is multiplied inside the kernel: z_t [B*T rows of stride ldz, nz columns] . Kz [nz,352]
  // (the latent rows of the decoder's input kernel).  It joins the recurrent product as NZS more k-steps, so the
  // [B*T, 352] projection is neither a launch of its own nor 2 x 4 bytes per gate column of HBM traffic.
  const float* zin; const float* Kz; int ldz, nz;
};

constexpr int MZS = 12;           // LDS slice stride of z per (row, k residue): up to 8 k-steps (32 latent columns) used;
                                  // 12 floats: the 16 slices start at banks 12 c mod 64, all distinct (8: c and c+8 collide)

template <int GATE, int NQ, int NZS>
__device__ __forceinline__ void lstm_fwd_mfma_body(const LstmMfmaFwdArgs& a, float (*hA)[16 * MHS], float (*zA)[16 * MZS],
                                                   int wave) {
  const int lane = threadIdx.x & 63;
  const int j = lane & 3, kb = (lane >> 2) & 3, cg = lane >> 4;
  const int T = a.T;
  const int row = blockIdx.x * MR + kb;                 // the row this lane finishes
  const bool live = row < a.B;
  const size_t rowc = (size_t)min(row, a.B - 1);
  const Sel4 sel_kb(kb);

  float Ub[NQ][MKS];           // B operands: U[4m + kb][cg*88 + 4 quad + j]
  int unit[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    unit[q] = 4 * (wave + MNW * q) + j;
#pragma unroll
    for (int m = 0; m < MKS; ++m) Ub[q][m] = a.U[(size_t)(4 * m + kb) * LG + cg * LH + unit[q]];
  }
  float Kb[NQ][NZS > 0 ? NZS : 1];     // B operands of the z product: Kz[4m + kb][cg*88 + unit], zero rows beyond nz
  if (NZS > 0) {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int m = 0; m < NZS; ++m) {
        const int k = 4 * m + kb;
        const float v = a.Kz[(size_t)min(k, a.nz - 1) * LG + cg * LH + unit[q]];
        Kb[q][m] = k < a.nz ? v : 0.f;
      }
  }
  // z_t staging: thread e < 16 * 4 NZS moves element (row r, column k) of z_t from HBM (two steps ahead) into the
  // A-operand order of LDS (one step ahead); the columns beyond nz stay zero
  const int ze = threadIdx.x;
  const int zr = ze / (4 * NZS > 0 ? 4 * NZS : 1), zk = ze - zr * 4 * NZS;
  const bool zmine = NZS > 0 && ze < 4 * 4 * NZS && zk < a.nz && (int)(blockIdx.x * MR + zr) < a.B;
  const float* zp = a.zin + ((size_t)min((int)(blockIdx.x * MR + zr), a.B - 1) * T) * (NZS > 0 ? a.ldz : 0) + min(zk, max(a.nz - 1, 0));
  const int zdst = (zr * 4 + (zk & 3)) * MZS + (zk >> 2);
  float zn = 0.f;
  if (NZS > 0) {
    if (zmine) zA[0][zdst] = zp[0];
    zn = zmine ? zp[(size_t)min(1, T - 1) * a.ldz] : 0.f;
    __syncthreads();
  }
  float rb[NQ], c[NQ], xn[NQ], xn2[NQ];
  const float* xp[NQ];
  float* gptr[NQ];
  float* sptr[NQ];             // second store: h (gate lane 0), c (gate lane 1), nothing (gate lanes 2, 3)
  int sstr;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int col = cg * LH + unit[q];
    rb[q] = a.rowbias ? a.rowbias[rowc * LG + col] : 0.f;
    c[q] = 0.f;
    xp[q] = a.xproj + rowc * T * LG + col;
    xn[q] = xp[q][0];
    xn2[q] = xp[q][(size_t)min(1, T - 1) * LG];
    gptr[q] = live ? a.gates + rowc * T * LG + col : g_mfma_dump + lane;
    float* second = cg == 0 ? a.hs + rowc * T * LH + unit[q] : a.cs + rowc * T * LH + unit[q];
    sptr[q] = (live && cg < 2) ? second : g_mfma_dump + 64 + lane;
  }
  sstr = (live && cg < 2) ? LH : 0;
  const int gstr = live ? LG : 0;
  g_mfma_dump[lane] = 0.f;     // stores after the prologue's loads, like every iteration (counted vmcnt, see lstm.hip)
  g_mfma_dump[lane + 64] = 0.f;

  // A operand: h[row j][4m + kb], m = 0..21, contiguous in LDS
  const int aslice = (j * 4 + kb) * MHS;
  // h write position of unit u = 4 quad + j of row kb: k = u -> residue j, step quad
  int hpos[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) hpos[q] = (kb * 4 + j) * MHS + (wave + MNW * q);
  // source lanes (byte addresses for ds_bpermute) of the four activated gates of this lane's (row, unit)
  const int src0 = 4 * (lane & 15);

  float h_last[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) h_last[q] = 0.f;

  for (int t = 0; t < T; ++t) {
    const int cur = t & 1;
    float xv[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      xv[q] = xn[q] + rb[q];
      xn[q] = xn2[q];
      xn2[q] = xp[q][(size_t)min(t + 2, T - 1) * LG];
    }
    float zv[NZS > 0 ? NZS : 1];
    if (NZS > 0) {
      // z_{t+1} (loaded last step) goes to the other LDS buffer, z_{t+2} is requested
      if (zmine) zA[cur ^ 1][zdst] = zn;
      zn = zmine ? zp[(size_t)min(t + 2, T - 1) * a.ldz] : 0.f;
      const float4* zq = reinterpret_cast<const float4*>(&zA[cur][(j * 4 + kb) * MZS]);
#pragma unroll
      for (int i = 0; i < (NZS + 3) / 4; ++i) {
        const float4 v = zq[i];
        zv[4 * i] = v.x;
        if (4 * i + 1 < NZS) zv[4 * i + 1] = v.y;
        if (4 * i + 2 < NZS) zv[4 * i + 2] = v.z;
        if (4 * i + 3 < NZS) zv[4 * i + 3] = v.w;
      }
    }
    float av[MKS + 2];
    {
      const float4* ap = reinterpret_cast<const float4*>(&hA[cur][aslice]);
#pragma unroll
      for (int i = 0; i < (MKS + 2) / 4; ++i) {
        const float4 v = ap[i];
        av[4 * i] = v.x; av[4 * i + 1] = v.y; av[4 * i + 2] = v.z; av[4 * i + 3] = v.w;
      }
    }
    // Two accumulators per quad (even / odd k-steps) issued round-robin: an accumulator is touched every 2 NQ
    // instructions (48 cycles at three quads), past the 40-cycle dependent latency, so the matrix pipe never waits on
    // its own result.  The scheduling barriers pin the order: left alone the compiler issues each accumulator's 22
    // instructions back to back (2640 cycles per step instead of 528).
    f32x4m D[NQ], E[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) D[q] = E[q] = (f32x4m){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < MKS; m += 2) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) D[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[m], Ub[q][m], D[q], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NQ; ++q) E[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[m + 1], Ub[q][m + 1], E[q], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (NZS > 0) {
#pragma unroll
      for (int m = 0; m < NZS; m += 2) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) D[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(zv[m], Kb[q][m], D[q], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < NQ; ++q) E[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(zv[m + 1], Kb[q][m + 1], E[q], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      float zr[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x = D[q][r] + E[q][r];
        x = dpp_add<0x124>(x);           // row_ror:4, row_ror:8: the four k residues, total in every lane
        x = dpp_add<0x128>(x);
        zr[r] = x;
      }
      const float z = sel_kb(zr) + xv[q];
      const float sg = gate_fn<GATE>(z), th = fast_tanh(z);
      const int act = __builtin_bit_cast(int, cg == 2 ? th : sg);
      // the four gates of this (row, unit) sit 16 lanes apart: four reads through the LDS crossbar
      const float gi = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src0, act));
      const float gf = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src0 + 64, act));
      const float gg = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src0 + 128, act));
      const float go = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src0 + 192, act));
      c[q] = gf * c[q] + gi * gg;
      const float h = go * fast_tanh(c[q]);
      h_last[q] = h;
      *gptr[q] = cg == 2 ? th : z;
      *sptr[q] = cg == 0 ? h : c[q];
      gptr[q] += gstr;
      sptr[q] += sstr;
      if (cg == 0) hA[cur ^ 1][hpos[q]] = h;
    }
    step_barrier();
  }
  if (live && cg < 2) {
    float* dst = cg == 0 ? a.hT : a.cT;
    if (dst)
#pragma unroll
      for (int q = 0; q < NQ; ++q) dst[rowc * LH + unit[q]] = cg == 0 ? h_last[q] : c[q];
  }
}

template <int GATE, int NZS>
__global__ __launch_bounds__(MNW * 64) void lstm_fwd_mfma_kernel(LstmMfmaFwdArgs a) {
  __shared__ __attribute__((aligned(16))) float hA[2][16 * MHS];
  __shared__ __attribute__((aligned(16))) float zA[2][16 * MZS];
  for (int i = threadIdx.x; i < 2 * 16 * MHS; i += MNW * 64) (&hA[0][0])[i] = 0.f;
  for (int i = threadIdx.x; i < 2 * 16 * MZS; i += MNW * 64) (&zA[0][0])[i] = 0.f;
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (wave < MQ - 2 * MNW) lstm_fwd_mfma_body<GATE, 3, NZS>(a, hA, zA, wave);     // waves 0-5: quads w, w+8, w+16
  else lstm_fwd_mfma_body<GATE, 2, NZS>(a, hA, zA, wave);                         // waves 6, 7: quads w, w+8
}

template <int GATE>
static void launch_mfma_nzs(const LstmMfmaFwdArgs& a, hipStream_t s) {
  const dim3 grid((a.B + MR - 1) / MR), block(MNW * 64);
  const int nzs = a.zin ? (a.nz + 3) / 4 : 0;
  if (nzs == 0) hipLaunchKernelGGL((lstm_fwd_mfma_kernel<GATE, 0>), grid, block, 0, s, a);
  else if (nzs <= 2) hipLaunchKernelGGL((lstm_fwd_mfma_kernel<GATE, 2>), grid, block, 0, s, a);
  else if (nzs <= 4) hipLaunchKernelGGL((lstm_fwd_mfma_kernel<GATE, 4>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((lstm_fwd_mfma_kernel<GATE, 8>), grid, block, 0, s, a);
}

int launch_lstm_fwd_mfma(int B, int T, int gate_act, const float* xproj, const float* rowbias, const float* U,
                         float* hs, float* cs, float* gates, float* hT, float* cT,
                         const float* zin, int ldz, int nz, const float* Kz, hipStream_t s) {
  if (zin && (nz < 1 || nz > 32 || !Kz || ldz < nz)) return CLV_EINVAL;
  LstmMfmaFwdArgs a{B, T, xproj, rowbias, U, hs, cs, gates, hT, cT, zin, Kz, ldz, nz};
  if (gate_act == CLV_GATE_HARD_SIGMOID) launch_mfma_nzs<CLV_GATE_HARD_SIGMOID>(a, s);
  else launch_mfma_nzs<CLV_GATE_SIGMOID>(a, s);
  return launch_status();
}

}  // namespace clv
