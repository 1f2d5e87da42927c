// lstm.hip -- persistent LSTM sequence kernels for H = 88 (gfx950).
//
// Batch rows are independent, so a workgroup owns R rows for all T steps and no
// grid-level sync exists.  At the reference's batch sizes (256 rows per GPU) the
// chip has one CU per row: the recurrent product h.U is a matrix-vector product
// per CU, which the MFMA (16-row tiles) cannot fill, so it runs on the VALU with
// the whole recurrent kernel U [88,352] (124 KB) resident in REGISTERS.
//
// Thread layout (template KS = number of k-slices, 8 or 4):
//   lane = (unit_local = lane / KS, kslice = lane % KS); unit u = (64/KS)*wave + unit_local;
//   KS = 8: 11 waves, 44 U-floats per thread;  KS = 4: 6 waves, 88 U-floats per thread.
//   forward : thread (u,s) keeps U[KK*s .. KK*s+KK-1][{i,f,c,o} of u]   (KK = 88/KS)
//   backward: its own layout (4 units x 22 gate columns per thread), see lstm_bwd_kernel
// Per step a thread does 4*KK FMAs per row, the k-slices are summed with log2(KS) DPP
// adds, and the unit's gate math and cell state stay in that lane group's registers; only
// h_t (forward) / dz_t (backward) cross lanes through LDS, one barrier per step.
// Fewer slices = fewer redundant copies of the per-unit gate math and of every
// per-step overhead instruction (the kernels are VALU-issue-bound, not FMA-bound).
//
// Memory pipeline: all per-step loads and stores are unconditional and branch-free so that
// the compiler's s_waitcnt is a counted vmcnt(#younger ops); vmcnt retires in issue order, so
// a conditional or drained wait would expose the latency of the previous step's stores.
#include <stdlib.h>

#include "lstm_common.h"

namespace clv {

template <int KS>
struct Geo {
  static constexpr int UL = 64 / KS;                   // units per wave
  static constexpr int NW = (LH + UL - 1) / UL;        // waves
  static constexpr int NT = NW * 64;                   // threads
  static constexpr int KK = LH / KS;                   // k values per slice (forward)
  static constexpr int KP = (KK + 3) / 4 * 4;          // padded slice length in LDS
};

struct LstmFwdArgs {
  int B, T;
  const float* xproj;    // [B,T,352]
  const float* rowbias;  // [B,352] or null
  const float* U;        // [88,352]
  const float* h0;       // [B,88] or null
  const float* c0;
  float* hs;             // [B,T,88]
  float* cs;             // [B,T,88]
  float* gates;          // [B,T,352] (z_i, z_f, tanh(z_c), z_o) or null (inference)
  float* hT;             // [B,88] or null
  float* cT;
};

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
// ABL != 0 builds exist only in tools/lstm_ablate.hip (phase-removal timing, results are wrong by design)
template <int KS, int R, int GATE, bool SAVE, int ABL = 0>
__global__ __launch_bounds__(Geo<KS>::NT) void lstm_fwd_kernel(LstmFwdArgs a) {
  using G = Geo<KS>;
  constexpr int KK = G::KK, KP = G::KP;
  constexpr int NC = KS / R;                 // lane copies per (row, unit)
  constexpr int NX = (4 * R + KS - 1) / KS;  // xproj loads per lane per step
  __shared__ __attribute__((aligned(16))) float hbuf[2][R][KS * KP];   // slice s at KP*s (KK used)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s = lane % KS;
  const int u = min(wave * G::UL + lane / KS, LH - 1);   // surplus lane groups duplicate unit 87 (same values, same addresses)
  const int row0 = blockIdx.x * R;
  const int myrow = s % R, copy = s / R;     // which row this lane finishes, which outputs it stores
  const int T = a.T;

  // recurrent kernel slice -> registers
  f2 Ur[KK][2];               // gate pairs (i,f) and (c,o)
#pragma unroll
  for (int kk = 0; kk < KK; ++kk)
#pragma unroll
    for (int g = 0; g < 4; ++g) Ur[kk][g >> 1][g & 1] = a.U[(size_t)(KK * s + kk) * LG + g * LH + u];

  // per-lane share of xproj / rowbias: element e = s + KS*i -> (row e>>2, gate e&3)
  float rb[NX];
  size_t xoff[NX];
  bool xok[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int e = s + KS * i;
    xok[i] = e < 4 * R;
    const int rr = xok[i] ? (e >> 2) : 0, gg = e & 3;   // surplus lanes re-read a valid element
    xoff[i] = (size_t)(row0 + rr) * T * LG + gg * LH + u;
    rb[i] = (xok[i] && a.rowbias) ? a.rowbias[(size_t)(row0 + rr) * LG + gg * LH + u] : 0.f;
  }

  // initial state
  const int hslot = KP * (u / KK) + (u % KK);
  float c = a.c0 ? a.c0[(size_t)(row0 + myrow) * LH + u] : 0.f;
  if (copy == 0) hbuf[0][myrow][hslot] = a.h0 ? a.h0[(size_t)(row0 + myrow) * LH + u] : 0.f;
  float xn[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) xn[i] = (T > 0 && a.xproj) ? a.xproj[xoff[i]] : 0.f;

  __syncthreads();
  // output slots of this lane (loop-invariant): slot 0 h, 1 c, 2..5 gates (z_i, z_f, g, z_o)
  constexpr int NS = (6 + NC - 1) / NC;
  float* optr[NS];
  int ostr[NS], oslot[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    int slot = copy + j * NC;
    slot = slot < 6 ? slot : slot - 6;                    // surplus copies repeat slot 0/1 (same value, same address)
    oslot[j] = slot;
    const size_t bt0 = (size_t)(row0 + myrow) * T;
    if (SAVE) {
      optr[j] = slot == 0 ? a.hs + bt0 * LH + u : slot == 1 ? a.cs + bt0 * LH + u : a.gates + bt0 * LG + (slot - 2) * LH + u;
      ostr[j] = slot < 2 ? LH : LG;
    } else {
      optr[j] = a.hs + bt0 * LH + u;
      ostr[j] = LH;
    }
  }

  float hlast = 0.f;
  for (int t = 0; t < T; ++t) {
    const int cur = t & 1;
    float xv[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      xv[i] = xn[i] + rb[i];
      if (ABL != 5) xn[i] = a.xproj[xoff[i] + (size_t)min(t + 1, T - 1) * LG];   // prefetch, unconditional (clamped)
    }
    // acc[r][g] starts from the lane's xproj share, then KK FMAs per gate
    f2 acc2[R][2];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int e = r * 4 + g;              // compile-time
        acc2[r][g >> 1][g & 1] = (s == (e % KS) && xok[e / KS]) ? xv[e / KS] : 0.f;
      }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float4* hp = reinterpret_cast<const float4*>(&hbuf[cur][r][KP * s]);
      float hv[KP];
#pragma unroll
      for (int q = 0; q < KP / 4; ++q) {
        const float4 v = hp[q];
        hv[4 * q] = v.x; hv[4 * q + 1] = v.y; hv[4 * q + 2] = v.z; hv[4 * q + 3] = v.w;
      }
      if (ABL == 3) {
#pragma unroll
        for (int g = 0; g < 4; ++g) acc2[r][g >> 1][g & 1] += hv[g] * Ur[g][g >> 1][g & 1];
      } else {
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
          const f2 hh = {hv[kk], hv[kk]};
          acc2[r][0] = __builtin_elementwise_fma(hh, Ur[kk][0], acc2[r][0]);
          acc2[r][1] = __builtin_elementwise_fma(hh, Ur[kk][1], acc2[r][1]);
        }
      }
    }
    float acc[R][4];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[r][g] = acc2[r][g >> 1][g & 1];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[r][g] = ABL == 6 ? acc[r][g] : reduce_slices<KS>(acc[r][g]);
    // this lane finishes row `myrow`
    float z[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      z[g] = acc[0][g];
#pragma unroll
      for (int r = 1; r < R; ++r) z[g] = (myrow == r) ? acc[r][g] : z[g];
    }
    float ig, fg, og, gg, h;
    if (ABL == 2) {
      ig = z[0]; fg = z[1]; og = z[3]; gg = z[2];
      c = 0.5f * c + 0.01f * ig * gg;
      h = 0.1f * og + 0.1f * c + 0.01f * fg;
    } else {
      ig = gate_fn<GATE>(z[0]); fg = gate_fn<GATE>(z[1]); og = gate_fn<GATE>(z[3]);
      gg = fast_tanh(z[2]);
      c = fg * c + ig * gg;
      h = og * fast_tanh(c);
    }
    hlast = h;
    hbuf[cur ^ 1][myrow][hslot] = h;     // all copies write the same value
    // outputs: 6 values per (row, unit) shared among the NC lane copies; every lane issues the same
    // number of stores, unconditionally (see the header note on counted vmcnt)
    const float ov[6] = {h, c, z[0], z[1], gg, z[3]};
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      float val = ov[0];
#pragma unroll
      for (int q = 1; q < 6; ++q) val = (oslot[j] == q) ? ov[q] : val;
      if (ABL != 1) *optr[j] = SAVE ? val : h;
      optr[j] += ostr[j];
    }
    if (ABL == 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  if (ABL == 1) a.hs[(size_t)(row0 + myrow) * T * LH + u] = hlast + c;
  if (copy == 0) {
    if (a.hT) a.hT[(size_t)(row0 + myrow) * LH + u] = T > 0 ? hlast : (a.h0 ? a.h0[(size_t)(row0 + myrow) * LH + u] : 0.f);
    if (a.cT) a.cT[(size_t)(row0 + myrow) * LH + u] = c;
  }
}

// ---------------------------------------------------------------------------
// backward (BPTT); gates buffer is overwritten in place with dz
// ---------------------------------------------------------------------------
struct LstmBwdArgs {
  int B, T;
  const float* U;       // [88,352]
  const float* dhs;     // [B,T,88]
  const float* cs;      // [B,T,88]
  const float* c0;      // [B,88] or null
  float* gates;         // in: (z_i,z_f,g,z_o)  out: dz
  float* dzsum;         // [B,352]
  // ZW > 0 (the decoder): dZ_t = dz_t . Kz^T as well, by two more waves whose "units" are latents (the rows of Kz sit
  // where a unit group keeps its rows of U): the [B*T,352] x [352,nz] product is neither a launch nor a second read of dz
  const float* Kz;      // [nz,352]
  float* dZ;            // B*T rows of stride lddz
  int nz, lddz;
};

__device__ float g_bwd_dump[64];

// Thread layout of the backward kernel: lane = (unit group ug = lane / 16, column slice cs = lane % 16);
// a thread keeps U[4*ug + j][22*cs .. 22*cs+21] for its 4 units j, so one dz value read from LDS feeds
// 4 FMAs (two v_pk_fma_f32).  (One unit x 88 columns per thread -- the transpose of the forward layout --
// needs one LDS float per FMA and is bound by the LDS return path: 22 ds_read_b128 per wave per step.)
// The 16 slice partials of a unit are summed by a reduce-scatter: xor-1 and xor-2 exchanges that halve
// the number of units a lane carries, then two row rotations over the 4 quads.  Afterwards lane
// (ug, cs) owns unit 4*ug + (cs & 3), replicated over the 4 quads q = cs >> 2; as in the forward
// kernel the replicas split the rows (R) and the output slots between them.
constexpr int BW_NW = 6, BW_NT = BW_NW * 64;    // 24 unit groups (22 used)
// columns per slice, padded slice stride in LDS.  28 floats: the 16 slices of a ds_read_b128 lane group start at banks
// 28*cs mod 64, four banks each, all distinct; at a stride of 24 slices cs and cs+8 share banks and every read of the
// step was a 2-way conflict (SQ_LDS_BANK_CONFLICT 59 % of the LDS cycles at four rows per workgroup, where the kernel is
// LDS-bound: profiles/r02_e_sq_cfg5.json; lstm_pair.hip made the same change in round 2)
constexpr int BW_CW = 22, BW_CP = 28;
constexpr int BW_LDS = 16 * BW_CP;

template <int R, int GATE, int ZW = 0>
__global__ __launch_bounds__(BW_NT + 64 * ZW) void lstm_bwd_kernel(LstmBwdArgs a) {
  constexpr int NC = 4 / R;
  __shared__ __attribute__((aligned(16))) float dzbuf[2][R][BW_LDS];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cs = lane & 15, ug = wave * 4 + (lane >> 4);
  const int q = cs >> 2;
  const bool b0 = cs & 1, b1 = cs & 2;
  const int u = min(4 * ug + (cs & 3), LH - 1);       // surplus unit groups duplicate unit 87 (same values, same addresses)
  const int row0 = blockIdx.x * R;
  const int myrow = q % R, copy = q / R;
  const int T = a.T;
  // latent groups (ZW): the surplus groups 22, 23 and the two extra waves carry 4 latents each
  const int zg0 = 4 * (ug - 22);
  const bool zgroup = ZW > 0 && ug >= 22 && zg0 < a.nz;
  const int lat = zg0 + (cs & 3);
  const bool zlane = zgroup && lat < a.nz;

  f2 Ur[BW_CW][2];    // [column][unit pair]; a latent group holds rows of Kz instead
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float* rowp = zgroup ? a.Kz + (size_t)min(zg0 + j, a.nz - 1) * LG : a.U + (size_t)min(4 * ug + j, LH - 1) * LG;
    const float keep = (zgroup && zg0 + j >= a.nz) ? 0.f : 1.f;
    const float2* up = reinterpret_cast<const float2*>(rowp + BW_CW * cs);
#pragma unroll
    for (int c = 0; c < BW_CW / 2; ++c) {
      const float2 v = up[c];
      Ur[2 * c][j >> 1][j & 1] = v.x * keep;
      Ur[2 * c + 1][j >> 1][j & 1] = v.y * keep;
    }
  }
  for (int i = tid; i < 2 * R * BW_LDS; i += BW_NT + 64 * ZW) (&dzbuf[0][0][0])[i] = 0.f;

  const size_t rowbt = (size_t)(row0 + myrow) * T;
  float dc = 0.f;
  float zs[4] = {0.f, 0.f, 0.f, 0.f};   // running sum_t dz (this lane's row/unit)

  // Two-stage software pipeline.  Everything that does not depend on the recurrence (gate
  // activations, tanh(c_t), gate derivatives) is folded into 7 coefficients per step, computed one
  // iteration ahead from values loaded two iterations ahead, so a load has a full step to land and
  // the per-step critical path is: reduce -> 8 multiply/adds -> LDS write -> barrier.
  struct Raw { float zi, zf, g, zo, c, cp, dh; };
  struct Coef { float ko, kc, ki, kf, kg, kcarry, dhh; };
  const float* g_base = a.gates + rowbt * LG + u;
  const float* c_base = a.cs + rowbt * LH + u;
  const float* d_base = a.dhs + rowbt * LH + u;
  const float c0v = a.c0 ? a.c0[(size_t)(row0 + myrow) * LH + u] : 0.f;
  auto load_raw = [&](int t) {                     // t >= 0 (clamped by the caller)
    Raw r;
    const float* gp = g_base + (size_t)t * LG;
    r.zi = gp[0]; r.zf = gp[LH]; r.g = gp[2 * LH]; r.zo = gp[3 * LH];
    r.c = c_base[(size_t)t * LH];
    const float cprev = c_base[(size_t)max(t - 1, 0) * LH];
    r.cp = t > 0 ? cprev : c0v;
    r.dh = d_base[(size_t)t * LH];
    return r;
  };
  auto make_coef = [&](const Raw& r) {
    Coef k;
    const float ig = gate_fn<GATE>(r.zi), fg = gate_fn<GATE>(r.zf), og = gate_fn<GATE>(r.zo);
    const float tc = fast_tanh(r.c);
    k.ko = tc * gate_grad<GATE>(r.zo, og);          // dz_o = dh * ko
    k.kc = og * (1.f - tc * tc);                    // dc  += dh * kc
    k.ki = r.g * gate_grad<GATE>(r.zi, ig);         // dz_i = dc * ki
    k.kf = r.cp * gate_grad<GATE>(r.zf, fg);        // dz_f = dc * kf
    k.kg = ig * (1.f - r.g * r.g);                  // dz_g = dc * kg
    k.kcarry = fg;                                  // dc_{t-1} = dc * f
    k.dhh = r.dh;
    return k;
  };
  Raw raw_next = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  Coef coef_next = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (T > 0) {
    coef_next = make_coef(load_raw(T - 1));
    raw_next = load_raw(max(T - 2, 0));
  }
  // One store after the prologue's loads, exactly like every loop iteration issues after its loads: the
  // waitcnt pass merges the loop-entry and back-edge states, and with matching queues the wait for
  // `raw_next` stays vmcnt(#younger ops) instead of draining the previous dz store.
  a.dzsum[(size_t)(row0 + myrow) * LG + (copy & 3) * LH + u] = 0.f;
  constexpr int NSB = (4 + NC - 1) / NC;
  float* gptr[NSB];
  int lpos[NSB];      // LDS position of gate column slot*88 + u in the sliced layout
#pragma unroll
  for (int j = 0; j < NSB; ++j) {
    const int col = ((copy + j * NC) & 3) * LH + u;
    gptr[j] = a.gates + (rowbt + (T > 0 ? T - 1 : 0)) * LG + col;
    lpos[j] = BW_CP * (col / BW_CW) + col % BW_CW;
    if (zgroup) { gptr[j] = g_bwd_dump + lane; lpos[j] = BW_CP * cs + BW_CW; }    // a latent lane's dz is nobody's: the
  }                                                                               // slice's padding and a dump word
  const int gstr = zgroup ? 0 : LG;
  // dZ_{t+1} leaves at iteration t (the matvec of iteration t multiplies dz_{t+1}); iteration T-1 writes the zeros
  // of the empty buffer to row T-1, which iteration T-2 overwrites: the store stays unconditional
  float* zptr = g_bwd_dump + lane;
  int zstr = 0;
  if (ZW > 0 && zlane && T > 0) { zptr = a.dZ + (rowbt + (T - 1)) * a.lddz + lat; zstr = a.lddz; }
  __syncthreads();

  for (int t = T - 1; t >= 0; --t) {
    const int cur = (T - 1 - t) & 1;
    const Coef k = coef_next;
    const Raw rcur = raw_next;                     // values of step t-1 (loaded one iteration ago)
    raw_next = load_raw(max(t - 2, 0));            // lands during this whole iteration (redundant for t < 2)
    // partial dh_rec of this lane's 4 units over its 22 columns: sum_c dz_{t+1}[c] * U[u_j][c]
    float part[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float4* dp = reinterpret_cast<const float4*>(&dzbuf[cur][r][BW_CP * cs]);
      float dv[BW_CP];
#pragma unroll
      for (int j = 0; j < (BW_CW + 3) / 4; ++j) {
        const float4 v = dp[j];
        dv[4 * j] = v.x; dv[4 * j + 1] = v.y; dv[4 * j + 2] = v.z; dv[4 * j + 3] = v.w;
      }
      f2 acc01 = {0.f, 0.f}, acc23 = {0.f, 0.f};
#pragma unroll
      for (int c = 0; c < BW_CW; ++c) {
        const f2 dd = {dv[c], dv[c]};
        acc01 = __builtin_elementwise_fma(dd, Ur[c][0], acc01);
        acc23 = __builtin_elementwise_fma(dd, Ur[c][1], acc23);
      }
      // reduce-scatter over the 16 column slices
      const float keep_a = b0 ? acc01[1] : acc01[0], send_a = b0 ? acc01[0] : acc01[1];
      const float keep_b = b0 ? acc23[1] : acc23[0], send_b = b0 ? acc23[0] : acc23[1];
      const float wa = keep_a + dpp_mov<0xB1>(send_a);          // units {0,1}[b0] over a lane pair
      const float wb = keep_b + dpp_mov<0xB1>(send_b);          // units {2,3}[b0]
      const float keep = b1 ? wb : wa, send = b1 ? wa : wb;
      float x = keep + dpp_mov<0x4E>(send);                     // unit (cs & 3) over the quad
      x = dpp_add<0x124>(x);                                    // row_ror:4  -> two quads
      x = dpp_add<0x128>(x);                                    // row_ror:8  -> all four quads
      part[r] = x;
    }
    coef_next = make_coef(rcur);                   // off the critical path (unused after t == 0)
    float dhrec = part[0];
#pragma unroll
    for (int r = 1; r < R; ++r) dhrec = (myrow == r) ? part[r] : dhrec;
    if (ZW > 0) {                                  // latent lanes: dhrec is dZ_{t+1} of (row, latent)
      *zptr = dhrec;
      zptr -= (t < T - 1) ? zstr : 0;
    }

    const float dh = k.dhh + dhrec;
    dc = fmaf(dh, k.kc, dc);
    float dz[4];
    dz[0] = dc * k.ki;
    dz[1] = dc * k.kf;
    dz[2] = dc * k.kg;
    dz[3] = dh * k.ko;
    dc = dc * k.kcarry;
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) zs[gi] += dz[gi];
    // 4 values per (row, unit) shared among NC copies (unconditional stores)
#pragma unroll
    for (int j = 0; j < NSB; ++j) {
      const int slot = (copy + j * NC) & 3;
      float val = dz[0];
#pragma unroll
      for (int qq = 1; qq < 4; ++qq) val = (slot == qq) ? dz[qq] : val;
      dzbuf[cur ^ 1][myrow][lpos[j]] = val;
      *gptr[j] = val;
      gptr[j] -= gstr;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  if (ZW > 0 && wave >= 5 && T > 0) {              // dZ_0 = dz_0 . Kz^T: one more matvec by the waves that hold latents
    const int cur = T & 1;
    float dz0 = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float4* dp = reinterpret_cast<const float4*>(&dzbuf[cur][r][BW_CP * cs]);
      float dv[BW_CP];
#pragma unroll
      for (int j = 0; j < (BW_CW + 3) / 4; ++j) {
        const float4 v = dp[j];
        dv[4 * j] = v.x; dv[4 * j + 1] = v.y; dv[4 * j + 2] = v.z; dv[4 * j + 3] = v.w;
      }
      f2 acc01 = {0.f, 0.f}, acc23 = {0.f, 0.f};
#pragma unroll
      for (int c = 0; c < BW_CW; ++c) {
        const f2 dd = {dv[c], dv[c]};
        acc01 = __builtin_elementwise_fma(dd, Ur[c][0], acc01);
        acc23 = __builtin_elementwise_fma(dd, Ur[c][1], acc23);
      }
      const float keep_a = b0 ? acc01[1] : acc01[0], send_a = b0 ? acc01[0] : acc01[1];
      const float keep_b = b0 ? acc23[1] : acc23[0], send_b = b0 ? acc23[0] : acc23[1];
      const float wa = keep_a + dpp_mov<0xB1>(send_a);
      const float wb = keep_b + dpp_mov<0xB1>(send_b);
      const float keep = b1 ? wb : wa, send = b1 ? wa : wb;
      float x = keep + dpp_mov<0x4E>(send);
      x = dpp_add<0x124>(x);
      x = dpp_add<0x128>(x);
      dz0 = (myrow == r) ? x : dz0;
    }
    if (zlane) a.dZ[rowbt * a.lddz + lat] = dz0;
  }
  if (!zgroup) {
#pragma unroll
    for (int j = 0; j < NSB; ++j) {
      const int slot = (copy + j * NC) & 3;
      float val = zs[0];
#pragma unroll
      for (int qq = 1; qq < 4; ++qq) val = (slot == qq) ? zs[qq] : val;
      a.dzsum[(size_t)(row0 + myrow) * LG + slot * LH + u] = val;
    }
  }
}

// rows per workgroup: one while every CU can get its own row, then the smallest of 2 / 4 that brings the workgroups back
// to one per CU (a workgroup's rows share its weights and its barrier; two co-resident workgroups starve each other at
// the issue port).  Measured at T = 128 (tools/lstm_rows_bench.py, gpurun_out/s19; forward / backward, us):
//   B  384: R=1 125 / 171   R=2 122 / 131   R=4 181 / 196
//   B  512:     133 / 176       127 / 138       183 / 198
//   B  768:     186 / 270       200 / 255       186 / 206
//   B 1024:     290 / 343       210 / 280       195 / 222
//   B 2048:     553 / 693       498 / 539       397 / 455
static int rows_per_wg(int B) {
  static const int forced = env_int("CLV_LSTM_ROWS", 0);      // measurement knob: 1, 2 or 4 rows per workgroup
  if ((forced == 1 || forced == 2 || forced == 4) && B % forced == 0) return forced;
  if (B > 512 && B % 4 == 0) return 4;
  if (B > 256 && B % 2 == 0) return 2;
  return 1;
}
static int lstm_ks() {
  static const int ks = env_int("CLV_LSTM_KS", 4) == 8 ? 8 : 4;
  return ks;
}

template <int KS, int R, int GATE, bool SAVE>
static int launch_fwd_r(const LstmFwdArgs& a, hipStream_t s) {
  constexpr int NT = Geo<KS>::NT;
  hipLaunchKernelGGL((lstm_fwd_kernel<KS, R, GATE, SAVE, 0>), dim3(a.B / R), dim3(NT), 0, s, a);
  return launch_status();
}
template <int KS, int GATE, bool SAVE>
static int launch_fwd(const LstmFwdArgs& a, hipStream_t s) {
  const int R = rows_per_wg(a.B);
  if (R == 4) return launch_fwd_r<KS, 4, GATE, SAVE>(a, s);
  if (R == 2) return launch_fwd_r<KS, 2, GATE, SAVE>(a, s);
  return launch_fwd_r<KS, 1, GATE, SAVE>(a, s);
}
template <int GATE>
static int launch_bwd(const LstmBwdArgs& a, hipStream_t s) {
  const int B = a.B, R = rows_per_wg(B);
  if (a.Kz) {         // + dZ = dz . Kz^T: two more waves
    if (R == 4) hipLaunchKernelGGL((lstm_bwd_kernel<4, GATE, 2>), dim3(B / 4), dim3(BW_NT + 128), 0, s, a);
    else if (R == 2) hipLaunchKernelGGL((lstm_bwd_kernel<2, GATE, 2>), dim3(B / 2), dim3(BW_NT + 128), 0, s, a);
    else hipLaunchKernelGGL((lstm_bwd_kernel<1, GATE, 2>), dim3(B), dim3(BW_NT + 128), 0, s, a);
    return launch_status();
  }
  if (R == 4) hipLaunchKernelGGL((lstm_bwd_kernel<4, GATE>), dim3(B / 4), dim3(BW_NT), 0, s, a);
  else if (R == 2) hipLaunchKernelGGL((lstm_bwd_kernel<2, GATE>), dim3(B / 2), dim3(BW_NT), 0, s, a);
  else hipLaunchKernelGGL((lstm_bwd_kernel<1, GATE>), dim3(B), dim3(BW_NT), 0, s, a);
  return launch_status();
}

}  // namespace clv

static int lstm_fwd_dispatch(clv::LstmFwdArgs a, int gate_act, hipStream_t s) {
  using namespace clv;
  ProfScope p("lstm_seq_fwd", s);
  const bool save = a.gates != nullptr;
  const bool hard = gate_act == CLV_GATE_HARD_SIGMOID;
  if (lstm_ks() == 8) {
    if (hard) return save ? launch_fwd<8, CLV_GATE_HARD_SIGMOID, true>(a, s) : launch_fwd<8, CLV_GATE_HARD_SIGMOID, false>(a, s);
    return save ? launch_fwd<8, CLV_GATE_SIGMOID, true>(a, s) : launch_fwd<8, CLV_GATE_SIGMOID, false>(a, s);
  }
  if (hard) return save ? launch_fwd<4, CLV_GATE_HARD_SIGMOID, true>(a, s) : launch_fwd<4, CLV_GATE_HARD_SIGMOID, false>(a, s);
  return save ? launch_fwd<4, CLV_GATE_SIGMOID, true>(a, s) : launch_fwd<4, CLV_GATE_SIGMOID, false>(a, s);
}

extern "C" int clv_lstm_seq_fwd(int B, int T, int H, int gate_act,
                                const float* xproj, const float* rowbias, const float* U,
                                const float* h0, const float* c0,
                                float* hs, float* cs, float* gates, float* hT, float* cT,
                                void* stream) {
  using namespace clv;
  if (H <= 0 || B <= 0 || T < 0 || !xproj || !U || !hs || (gates && !cs)) return CLV_EINVAL;
  if (gate_act != CLV_GATE_HARD_SIGMOID && gate_act != CLV_GATE_SIGMOID) return CLV_EINVAL;
  // any other --intermediate_dim (cl_vrnn/train.py:90): csrc/lstm_any.hip (CLV_LSTM_ANY=1, read per call: also at 88 units -- tests)
  if (H != LH || env_int("CLV_LSTM_ANY", 0))
    return launch_lstm_any_fwd(B, T, H, gate_act, xproj, rowbias, U, h0, c0, hs, cs, gates, hT, cT, (hipStream_t)stream);
  LstmFwdArgs a{B, T, xproj, rowbias, U, h0, c0, hs, cs, gates, hT, cT};
  return lstm_fwd_dispatch(a, gate_act, (hipStream_t)stream);
}

extern "C" int clv_lstm_seq_bwd(int B, int T, int H, int gate_act,
                                const float* U, const float* dhs, const float* cs, const float* c0,
                                float* gates_inout_dz, float* dzsum, void* stream) {
  using namespace clv;
  if (H <= 0 || B <= 0 || T < 0 || !U || !dhs || !cs || !gates_inout_dz || !dzsum) return CLV_EINVAL;
  if (gate_act != CLV_GATE_HARD_SIGMOID && gate_act != CLV_GATE_SIGMOID) return CLV_EINVAL;
  if (H != LH || env_int("CLV_LSTM_ANY", 0)) return launch_lstm_any_bwd(B, T, H, gate_act, U, dhs, cs, c0, gates_inout_dz, dzsum, (hipStream_t)stream);
  LstmBwdArgs a{B, T, U, dhs, cs, c0, gates_inout_dz, dzsum, nullptr, nullptr, 0, 0};
  hipStream_t s = (hipStream_t)stream;
  ProfScope p("lstm_seq_bwd", s);
  const bool hard = gate_act == CLV_GATE_HARD_SIGMOID;
  return hard ? launch_bwd<CLV_GATE_HARD_SIGMOID>(a, s) : launch_bwd<CLV_GATE_SIGMOID>(a, s);
}

extern "C" int clv_lstm_seq_bwd_z(int B, int T, int H, int gate_act,
                                  const float* U, const float* dhs, const float* cs, const float* c0,
                                  float* gates_inout_dz, float* dzsum, const float* Kz, int nz, float* dZ, int lddz,
                                  void* stream) {
  using namespace clv;
  if (H != LH || B <= 0 || T < 0 || !U || !dhs || !cs || !gates_inout_dz || !dzsum) return CLV_EINVAL;
  if (!Kz || !dZ || nz < 1 || nz > 40 || lddz < nz) return CLV_EINVAL;      // 10 latent groups of 4
  if (gate_act != CLV_GATE_HARD_SIGMOID && gate_act != CLV_GATE_SIGMOID) return CLV_EINVAL;
  LstmBwdArgs a{B, T, U, dhs, cs, c0, gates_inout_dz, dzsum, Kz, dZ, nz, lddz};
  hipStream_t s = (hipStream_t)stream;
  ProfScope p("lstm_seq_bwd", s);
  const bool hard = gate_act == CLV_GATE_HARD_SIGMOID;
  return hard ? launch_bwd<CLV_GATE_HARD_SIGMOID>(a, s) : launch_bwd<CLV_GATE_SIGMOID>(a, s);
}
