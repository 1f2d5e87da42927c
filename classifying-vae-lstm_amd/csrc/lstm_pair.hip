// lstm_pair.hip -- the encoder and the decoder LSTM of cl_vrnn as ONE persistent kernel per pass (gfx950).
//
// At the reference's batch sizes a sequence kernel (lstm.hip) is one serial latency chain per step
// (LDS read -> 22 dependent FMAs -> lane reduce -> gate math -> LDS write -> barrier): neither the VALU
// nor the LDS is busy, so two chains that run on the SAME CU in different waves cost little more than
// one.  The decoder step t needs only z_t, i.e. the encoder's h_t, so the two recurrences are skewed
// by two steps and run side by side: a workgroup owns one batch row, waves 0-5 carry the encoder
// chain and waves 6-11 the decoder chain (3 waves per SIMD, <= 168 VGPRs each), one s_barrier per
// step for both.
//
// The latent head between them (cl_vrnn/model.py:200-216: Z_mean/Z_log_var Dense, z = mean +
// exp(log_var/2) eps, KL term) rides in the encoder's last wave: its 8 surplus lane groups (units
// 88..95 do not exist) hold columns of the fused head kernel instead of recurrent-kernel columns, so
// zargs_{t-1} = h_{t-1}.Wz falls out of the same FMA sequence that computes the gates of step t.
// Group j carries (mean_2j, mean_2j+1, log_var_2j, log_var_2j+1): latent_dim <= 16.
// The decoder adds z_t . K_z (the z rows of its input kernel) to its input projection itself, so
// the projection GEMM only covers the history frames x_{t-1}.
#include <stdlib.h>

#include "lstm_common.h"

namespace clv {

constexpr int PNW = 6;                  // waves per chain (16 units each)
constexpr int PNT = 2 * PNW * 64;       // 768 threads
constexpr int PLMAX = 16;               // latent dims the surplus groups can carry
constexpr int PLQ = PLMAX / PK;         // latents per decoder lane (z_t . K_z is split over the k-slice lanes)

// backward layout: gate columns per slice, padded LDS slice stride.  28 floats: the 16 slices of a ds_read_b128 lane
// group start at banks 28*cs mod 64 = {0,28,56,20,48,12,40,4,32,60,24,52,16,44,8,36}, four banks each, all distinct
// (a stride of 24 puts slices cs and cs+8 on the same banks: every read of the step was a 2-way conflict)
constexpr int BW_CW = 22, BW_CP = 28;
constexpr int BW_LDS = 16 * BW_CP;

// Tried and dropped (round 2, gpurun_out/s1): a two-barrier step with the chains half a step apart (one chain's FMAs
// over the other's LDS-write -> barrier -> LDS-read latency).  Every barrier makes all 12 waves wait for the slowest
// one, and two of them per step cost more than the overlap returns: +10 % per step (0.73 + 0.93 us against
// 0.67 + 0.83 us) for every split point of the FMA block tried (10, 12, 14 of 22).

__device__ float g_pair_dump[128];      // target of the stores of lanes that own no output (keeps every store unconditional)

// -DPAIR_STAMPS: the first decoder wave of workgroup 0 records the shader clock at six points of steps 32..39
// (tools/pair_stamps.py).  The stamps are issued without waiting (s_memtime returns under the step barrier's lgkmcnt(0))
// and each takes the value it follows as an operand, so it cannot move above that value's computation.
#ifdef PAIR_STAMPS
__device__ unsigned long long g_pair_stamps[8][8];
__device__ unsigned long long g_pair_arrive[8][12][2];     // [step][wave of the workgroup][top of step, arrival at the barrier]
#define PARRIVE(widx, t, dep)                                                                           \
  do {                                                                                                   \
    unsigned long long t1_;                                                                              \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1_) : "v"(dep));                         \
    if (blockIdx.x == 0 && lane == 0 && (t) >= 32 && (t) < 40) {                                         \
      g_pair_arrive[(t) - 32][widx][0] = ptop_; g_pair_arrive[(t) - 32][widx][1] = t1_;                  \
    }                                                                                                    \
  } while (0)
#define PTOP(dep) unsigned long long ptop_; asm volatile("s_memtime %0" : "=s"(ptop_) : "v"(dep))
#define PSTAMP(k, dep) asm volatile("s_memtime %0" : "=s"(pst[k]) : "v"(dep))
#else
#define PSTAMP(k, dep) do { } while (0)
#define PARRIVE(widx, t, dep) do { } while (0)
#define PTOP(dep) do { } while (0)
#endif

// ---------------------------------------------------------------------------
// Weights in lane order.  A workgroup needs every recurrent weight exactly once, one value per lane: read straight from
// the [88,352] kernels that is 88 dword loads per lane whose 64 lanes touch four 64-byte pieces of four different rows
// (1056 such wave loads per workgroup, every workgroup at the same time: ~13 us of each launch at config 3).  The pack
// kernel writes each lane's values as consecutive float4 (one wave load = 1 KB contiguous), once per step, for both
// passes: regions of [6 waves][n][64 lanes] float4.
//   PK_FE / PK_FD: forward encoder / decoder, n = 22 k values, components = gates (i,f,c,o); the encoder's latent lanes
//                  carry columns of the head kernel Wz instead (see pair_fwd_encoder)
//   PK_KZ:         forward decoder, n = 4: latent s + 4q of the decoder input kernel's z rows, components = gates
//   PK_BD / PK_BE: backward decoder / encoder, n = 22 gate columns of the lane's slice, components = the 4 units of the
//                  lane's group; the decoder's surplus groups carry rows of Kz (see pair_bwd_chain)
// ---------------------------------------------------------------------------
constexpr int PK_N = PNW * PKK * 64;               // float4 per 22-deep region
constexpr int PK_FE = 0, PK_FD = PK_N, PK_KZ = 2 * PK_N, PK_BD = PK_KZ + PNW * PLQ * 64, PK_BE = PK_BD + PK_N;
constexpr int PK_TOTAL = PK_BE + PK_N;             // float4

struct PairPackArgs { int L; const float* U_e; const float* U_d; const float* Kz; const float* Wz; float4* out; };

__global__ __launch_bounds__(256) void lstm_pair_pack_kernel(PairPackArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= PK_TOTAL) return;
  const int L = a.L;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (i < PK_KZ) {                                 // forward: lane = (unit, k-slice), n = kk
    const bool dec = i >= PK_FD;
    const int e = i - (dec ? PK_FD : PK_FE);
    const int wave = e / (PKK * 64), kk = (e / 64) % PKK, lane = e & 63;
    const int s = lane & 3, u_raw = wave * 16 + (lane >> 2), u = min(u_raw, LH - 1), zj = u_raw - LH;
    const bool is_z = !dec && zj >= 0 && 2 * zj < L;
    const float* U = dec ? a.U_d : a.U_e;
    // float4 kk of a lane = (k0, k1 | gate ga), (k0, k1 | gate gb) with k0 = 2 (kk / 2), (ga, gb) = (0, 1) for even kk
    // and (2, 3) for odd kk: the operand pairs of slice_fma_pairs
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) {
      const int g = 2 * (kk & 1) + (e2 >> 1), k = PKK * s + 2 * (kk >> 1) + (e2 & 1);
      if (is_z) {                                  // head column (mean_2j, mean_2j+1, log_var_2j, log_var_2j+1)
        const int l = 2 * zj + (g & 1);
        v[e2] = l < L ? a.Wz[(size_t)k * 2 * L + (g >> 1) * L + l] : 0.f;
      } else {
        v[e2] = U[(size_t)k * LG + g * LH + u];
      }
    }
  } else if (i < PK_BD) {                          // z rows of the decoder input kernel: latent s + 4q
    const int e = i - PK_KZ;
    const int wave = e / (PLQ * 64), q = (e / 64) % PLQ, lane = e & 63;
    const int s = lane & 3, u = min(wave * 16 + (lane >> 2), LH - 1), l = s + PK * q;
#pragma unroll
    for (int g = 0; g < 4; ++g) v[g] = l < L ? a.Kz[(size_t)l * LG + g * LH + u] : 0.f;
  } else {                                         // backward: lane = (unit group, column slice), n = column in the slice
    const bool dec = i < PK_BE;
    const int e = i - (dec ? PK_BD : PK_BE);
    const int wave = e / (BW_CW * 64), c = (e / 64) % BW_CW, lane = e & 63;
    const int cs = lane & 15, ug = wave * 4 + (lane >> 4), zg0 = 4 * (ug - 22);
    const bool zgroup = dec && ug >= 22 && zg0 < L;
    const float* U = dec ? a.U_d : a.U_e;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int lj = zg0 + j;
      if (zgroup) v[j] = lj < L ? a.Kz[(size_t)lj * LG + BW_CW * cs + c] : 0.f;
      else v[j] = U[(size_t)min(4 * ug + j, LH - 1) * LG + BW_CW * cs + c];
    }
  }
  a.out[i] = make_float4(v[0], v[1], v[2], v[3]);
}

struct PairFwdArgs {
  int B, T, L, ldz;
  const float* xproj_e;   // [B,T,352] x_t.K_x      (same buffer as gates_e)
  const float* rb_e;      // [B,352]   W.K_w + bias
  const float* xproj_d;   // [B,T,352] x_{t-1}.K_x  (same buffer as gates_d) or unused
  const float* rb_d;
  const float* pack;      // weights in lane order (clv_lstm_pair_pack)
  const float* bz;        // [2L]
  const float* eps;       // [B,T,L]
  float *hs_e, *cs_e, *gates_e, *hs_d, *cs_d, *gates_d;
  float* zargs;           // [B*T,2L]
  float* Z;               // [B*T] rows of stride ldz
  float* klterm;          // [B*T,L]  L * KL_l: the mean over all entries is the per-frame KL
};

// output slots of a regular lane: 6 values per unit (h, c, z_i, z_f, g, z_o) over the 4 slice lanes, 2 stores each
__device__ __forceinline__ void regular_slots(int s, int u, size_t bt0, float* hs, float* cs, float* gates,
                                              float* (&optr)[2], int (&ostr)[2], int (&oslot)[2]) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int slot = s + 4 * j;
    slot = slot < 6 ? slot : slot - 6;                   // lanes 2,3 repeat h and c (same value, same address)
    oslot[j] = slot;
    optr[j] = slot == 0 ? hs + bt0 * LH + u : slot == 1 ? cs + bt0 * LH + u : gates + bt0 * LG + (slot - 2) * LH + u;
    ostr[j] = slot < 2 ? LH : LG;
  }
}
// the two values lane s stores (slots s and s+4: see regular_slots): 3 selects each instead of a 6-way pick
__device__ __forceinline__ void pick_pair(int s, float h, float c, const float (&z)[4], float gg, float& v0, float& v1) {
  v0 = h;                       // flat selects (a nested ?: chain becomes branches around the stores)
  v0 = s == 1 ? c : v0;
  v0 = s == 2 ? z[0] : v0;
  v0 = s == 3 ? z[1] : v0;
  v1 = gg;
  v1 = s == 1 ? z[3] : v1;
  v1 = s == 2 ? h : v1;
  v1 = s == 3 ? c : v1;
}
__device__ __forceinline__ float pick_slot(int slot, float h, float c, const float (&z)[4], float gg) {
  float v = h;
  v = slot == 1 ? c : v;
  v = slot == 2 ? z[0] : v;
  v = slot == 3 ? z[1] : v;
  v = slot == 4 ? gg : v;
  v = slot == 5 ? z[3] : v;
  return v;
}

template <int GATE>
__device__ __forceinline__ void pair_fwd_encoder(const PairFwdArgs& a, int wave, int lane, float (*hb)[PK * PKP],
                                                 float (*zbuf)[PLMAX]) {
  const int s = lane & 3, b = blockIdx.x, T = a.T, L = a.L;
  const int u_raw = wave * 16 + (lane >> 2);
  const int u = min(u_raw, LH - 1);          // surplus groups that carry no latent duplicate unit 87
  const int zj = u_raw - LH;
  const bool is_z = zj >= 0 && 2 * zj < L;
  const int lat = 2 * zj + (s & 1);          // the latent this lane finishes
  const bool lat_ok = is_z && lat < L;
  const int zpos = (lat % PK) * PLQ + lat / PK;      // decoder lane s reads the latents s, s+4, .. as one 16-byte LDS word
  // head column held in accumulator g of a latent group: (mean_2j, mean_2j+1, log_var_2j, log_var_2j+1)
  auto zcol = [&](int g) { const int l = 2 * zj + (g & 1); return l < L ? (g >> 1) * L + l : -1; };
  const Sel4 sel_s(s);

  f2 Up[PKK / 2][4];     // [k pair][gate] = (U[2j][g], U[2j+1][g])
  {
    const float4* pw = reinterpret_cast<const float4*>(a.pack) + PK_FE + wave * PKK * 64 + lane;
#pragma unroll
    for (int kk = 0; kk < PKK; ++kk) {
      const float4 v = pw[kk * 64];
      Up[kk >> 1][2 * (kk & 1)] = (f2){v.x, v.y}; Up[kk >> 1][2 * (kk & 1) + 1] = (f2){v.z, v.w};
    }
  }
  const size_t bt0 = (size_t)b * T;
  const float* xp = a.xproj_e + bt0 * LG + s * LH + u;
  float rb, xmask;
  {
    const float* src = is_z ? a.bz + max(zcol(s), 0) : a.rb_e + (size_t)b * LG + s * LH + u;
    rb = *src;
    rb = (is_z && !lat_ok) ? 0.f : rb;
    xmask = is_z ? 0.f : 1.f;
  }
  const float* ep = a.eps + bt0 * L + (lat_ok ? lat : 0);

  float* optr[2];
  int ostr[2], oslot[2];
  regular_slots(s, u, bt0, a.hs_e, a.cs_e, a.gates_e, optr, ostr, oslot);
  if (is_z) {     // store 0: head pre-activation column; store 1: z (lanes 0,1) or the KL term (lanes 2,3)
    optr[0] = lat_ok ? a.zargs + bt0 * 2 * L + zcol(s) : g_pair_dump + lane;
    ostr[0] = lat_ok ? 2 * L : 0;
    optr[1] = !lat_ok ? g_pair_dump + lane : (s < 2 ? a.Z + bt0 * a.ldz + lat : a.klterm + bt0 * L + lat);
    ostr[1] = !lat_ok ? 0 : (s < 2 ? a.ldz : L);
  }
  const int hslot = PKP * (u / PKK) + (u % PKK);
  const float klscale = -0.5f * (float)L;

  float c = 0.f;
  // The per-step loads are requested two steps ahead: one step (0.75 us) is less than a load takes from HBM under
  // load, and with one step of lookahead the launch time moved 117..130 us from run to run with the latency.
  float xn = xp[0], xn2 = xp[(size_t)min(1, T - 1) * LG];      // projections of steps i and i+1
  float en = 0.f, en2 = ep[0];                                 // eps of steps i-1 and i
  // two stores after the prologue's loads, like every iteration issues after its loads: the loop-entry and
  // back-edge memory queues then match and the wait for `xn` stays a counted vmcnt (see lstm.hip)
  g_pair_dump[lane] = 0.f;
  g_pair_dump[lane + 64] = 0.f;
  auto latent = [&](const float (&acc)[4], float e, float& zv, float& klv) {
    const float m = (s & 1) ? acc[1] : acc[0], lv = (s & 1) ? acc[3] : acc[2];
    const float sd = __expf(0.5f * lv);
    zv = fmaf(sd, e, m);
    klv = klscale * (1.f + lv - m * m - sd * sd);
  };
  // the 4 gate sums of this lane's unit: h (LDS) . U slice, reduced over the k-slices
  auto gate_sums = [&](const float* hslice, float x0, float (&z)[4]) {
    f2 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = (f2){(s == g) ? x0 : 0.f, 0.f};
    float hv[PKP];
    load_hslice(hslice, hv);
    slice_fma_pairs<0, PKK / 2>(hv, Up, acc);
#pragma unroll
    for (int g = 0; g < 4; ++g) z[g] = reduce_slices<PK>(acc[g][0] + acc[g][1]);
  };

  for (int i = 0; i < T; ++i) {
    const int cur = i & 1;
    const float xv = fmaf(xn, xmask, rb);
    PTOP(xv);
    xn = xn2;
    // prefetch two steps ahead, unconditional (clamped); three: no gain.  The row offset is a 32-bit SCALAR product:
    // `(size_t)step * LG` per lane is a quarter-rate v_mad_i64_i32 on a SIMD that is issue-bound (tools/pair_stamps.py)
    xn2 = xp[(unsigned)min(i + 2, T - 1) * (unsigned)LG];
    const float ecur = en;                              // eps of step i-1
    en = en2;
    en2 = ep[(unsigned)min(i + 1, T - 1) * (unsigned)L];
    float z[4];
    gate_sums(&hb[cur][PKP * s], xv, z);
    float h, gg;
    lstm_cell<GATE>(z, c, h, gg);
    float v0, v1;
    pick_pair(s, h, c, z, gg, v0, v1);
    if (wave == PNW - 1) {          // wave-uniform: the latent head of step i-1 (garbage at i == 0, rewritten at i == 1)
      float zv, klv;
      latent(z, ecur, zv, klv);
      v0 = is_z ? sel_s(z) : v0;
      v1 = is_z ? (s < 2 ? zv : klv) : v1;
      if (lat_ok && s < 2) zbuf[(i + 1) & 1][zpos] = zv;
    }
    if (!is_z) hb[cur ^ 1][hslot] = h;
    *optr[0] = v0;
    *optr[1] = v1;
    const bool hold = is_z && i == 0;                   // the head lags one step: its row pointer starts moving at i == 1
    optr[0] += hold ? 0 : ostr[0];
    optr[1] += hold ? 0 : ostr[1];
    PARRIVE(wave, i + 2, v0 + v1);                      // the encoder runs two steps ahead of the decoder's step index
    step_barrier();
  }
  // iteration T: only the latent head of step T-1 is left
  if (wave == PNW - 1) {
    f2 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = (f2){(s == g) ? rb : 0.f, 0.f};
    float hv[PKP];
    load_hslice(&hb[T & 1][PKP * s], hv);
    slice_fma_pairs<0, PKK / 2>(hv, Up, acc);
    float z[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) z[g] = reduce_slices<PK>(acc[g][0] + acc[g][1]);
    float zv, klv;
    latent(z, en, zv, klv);
    if (lat_ok) {
      *optr[0] = sel_s(z);
      *optr[1] = s < 2 ? zv : klv;
      if (s < 2) zbuf[(T + 1) & 1][zpos] = zv;
    }
  }
  step_barrier();          // the decoder chain runs two steps behind
  step_barrier();
}

// ZQ: latents per decoder lane actually present (ceil(latent_dim / 4)): lane s multiplies the latents s, s+4, ..; the
// slots beyond latent_dim hold zero weights, so they are not issued at all
template <int GATE, bool HASXP, int ZQ>
__device__ __forceinline__ void pair_fwd_decoder(const PairFwdArgs& a, int wave, int lane, float (*hb)[PK * PKP],
                                                 float (*zbuf)[PLMAX]) {
  const int s = lane & 3, b = blockIdx.x, T = a.T;
  const int u = min(wave * 16 + (lane >> 2), LH - 1);
  f2 Up[PKK / 2][4];     // [k pair][gate] = (U[2j][g], U[2j+1][g])
  f2 Kzr[ZQ][2];         // lane s takes the latents s, s+4, ...
  {
    const float4* pw = reinterpret_cast<const float4*>(a.pack) + PK_FD + wave * PKK * 64 + lane;
#pragma unroll
    for (int kk = 0; kk < PKK; ++kk) {
      const float4 v = pw[kk * 64];
      Up[kk >> 1][2 * (kk & 1)] = (f2){v.x, v.y}; Up[kk >> 1][2 * (kk & 1) + 1] = (f2){v.z, v.w};
    }
    const float4* pz = reinterpret_cast<const float4*>(a.pack) + PK_KZ + wave * PLQ * 64 + lane;
#pragma unroll
    for (int q = 0; q < ZQ; ++q) {
      const float4 v = pz[q * 64];
      Kzr[q][0][0] = v.x; Kzr[q][0][1] = v.y; Kzr[q][1][0] = v.z; Kzr[q][1][1] = v.w;
    }
  }
  const size_t bt0 = (size_t)b * T;
  const float* xp = a.xproj_d + bt0 * LG + s * LH + u;
  const float rb = a.rb_d[(size_t)b * LG + s * LH + u];
  float* optr[2];
  int ostr[2], oslot[2];
  regular_slots(s, u, bt0, a.hs_d, a.cs_d, a.gates_d, optr, ostr, oslot);
  const int hslot = PKP * (u / PKK) + (u % PKK);
  float c = 0.f;
  float xn = HASXP ? xp[0] : 0.f, xn2 = HASXP ? xp[(size_t)min(1, T - 1) * LG] : 0.f;
  g_pair_dump[lane] = 0.f;     // see the encoder
  g_pair_dump[lane + 64] = 0.f;
  step_barrier();          // the encoder is two steps ahead
  step_barrier();
  for (int t = 0; t < T; ++t) {
    const int cur = t & 1;
#ifdef PAIR_STAMPS
    unsigned long long pst[8];
#endif
    const float xv = xn + rb;
    PSTAMP(0, xv);
    PTOP(xv);
    xn = xn2;
    if (HASXP) xn2 = xp[(unsigned)min(t + 2, T - 1) * (unsigned)LG];
    // scalar FMAs throughout: with v_pk_fma the allocator pairs a prefetch's destination register with an h value
    // inside a packed operand, and the wave waits for the load in the middle of the FMA block (+4 % step throughput)
    f2 acc[4];       // (even k, odd k) partial sums; the input projection and z_t . K_z start the odd halves
    {   // z_t . K_z: branch-free (rows of K_z beyond latent_dim are zero registers); does not depend on h
      float zk[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) zk[g] = 0.f;
      const float4 zq = *reinterpret_cast<const float4*>(&zbuf[cur][PLQ * s]);
      const float zl[PLQ] = {zq.x, zq.y, zq.z, zq.w};
#pragma unroll
      for (int q = 0; q < ZQ; ++q) {
        zk[0] = fmaf(zl[q], Kzr[q][0][0], zk[0]); zk[1] = fmaf(zl[q], Kzr[q][0][1], zk[1]);
        zk[2] = fmaf(zl[q], Kzr[q][1][0], zk[2]); zk[3] = fmaf(zl[q], Kzr[q][1][1], zk[3]);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = (f2){(s == g) ? xv : 0.f, zk[g]};
    }
    float hv[PKP];
    load_hslice(&hb[cur][PKP * s], hv);
    PSTAMP(1, hv[0]);                      // the first 16 bytes of h have arrived
    PSTAMP(2, hv[PKP - 4]);                // the last
    slice_fma_pairs<0, PKK / 2>(hv, Up, acc);
    PSTAMP(3, acc[3][0] + acc[0][1]);
    float z[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) z[g] = reduce_slices<PK>(acc[g][0] + acc[g][1]);
    PSTAMP(4, z[0] + z[3]);
    float h, gg;
    lstm_cell<GATE>(z, c, h, gg);
    PSTAMP(5, h);
    hb[cur ^ 1][hslot] = h;
    float v0, v1;
    pick_pair(s, h, c, z, gg, v0, v1);
    *optr[0] = v0;
    *optr[1] = v1;
    optr[0] += ostr[0];
    optr[1] += ostr[1];
    PSTAMP(6, v0 + v1);
    PARRIVE(PNW + wave, t, v0 + v1);
    step_barrier();
#ifdef PAIR_STAMPS
    if (blockIdx.x == 0 && wave == 0 && lane == 0 && t >= 32 && t < 40) {
      unsigned long long now;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now));
#pragma unroll
      for (int k = 0; k < 7; ++k) g_pair_stamps[t - 32][k] = pst[k];
      g_pair_stamps[t - 32][7] = now;
    }
#endif
  }
}

template <int GATE, bool HASXP, int ZQ>
__global__ __launch_bounds__(PNT) void lstm_pair_fwd_kernel(PairFwdArgs a) {
  __shared__ __attribute__((aligned(16))) float hbuf[2][2][PK * PKP];      // [chain][parity][sliced h]
  __shared__ __attribute__((aligned(16))) float zbuf[2][PLMAX];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 2 * 2 * PK * PKP; i += PNT) (&hbuf[0][0][0])[i] = 0.f;
  if (tid < 2 * PLMAX) (&zbuf[0][0])[tid] = 0.f;
  __syncthreads();
  if (wave < PNW) pair_fwd_encoder<GATE>(a, wave, lane, hbuf[0], zbuf);
  else pair_fwd_decoder<GATE, HASXP, ZQ>(a, wave - PNW, lane, hbuf[1], zbuf);
}

// ---------------------------------------------------------------------------
// backward: decoder BPTT, the latent head's backward and encoder BPTT in one launch
// ---------------------------------------------------------------------------
// Waves 0-5 run the decoder chain (layout of lstm_bwd_kernel: lane = (unit group, column slice), 4 units x
// 22 gate columns of U per thread), waves 6-11 the encoder chain two steps behind.  Between them:
//   dZ_t[l]   = sum_c dz_dec_t[c] * Kz[l][c]              -- rows of Kz ride in the decoder's surplus unit
//                                                            groups (units 88..95: latent_dim <= 8), so dZ_{t+1}
//                                                            falls out of the FMA sequence of step t
//   dzargs_t  = (dZ + kl*mean, dZ*eps*sd/2 - kl*(1-sd^2)/2)  -- same lanes; to HBM (head weight gradient) and LDS
//   dh_enc_t  = dzargs_t . Wz^T                            -- 2L FMAs per encoder lane (its row of Wz in registers)
// so neither dZ nor the encoder's upstream gradient ever exists in HBM.
constexpr int QL = 8;                   // latent dims the surplus unit groups of the decoder chain carry
constexpr int QZ = 2 * QL;              // head columns

struct PairBwdArgs {
  int B, T, L;
  float kl_scale;                 // kl_weight / (B*T)
  const float* pack;              // weights in lane order (clv_lstm_pair_pack)
  const float* Wz;                // [88,2L]
  const float* dhs_d;             // [B,T,88] dL/dh of the decoder (output head)
  const float* cs_d;
  const float* cs_e;
  float* gates_d;                 // in: (z_i,z_f,g,z_o) of the forward pass   out: dz
  float* gates_e;
  float* dzsum_d;                 // [B,352] sum_t dz
  float* dzsum_e;
  const float* zargs;             // [B*T,2L]
  const float* eps;               // [B*T,L]
  float* dzargs;                  // [B*T,2L]
};

template <int GATE, bool DEC, int ZP>
__device__ __forceinline__ void pair_bwd_chain(const PairBwdArgs& a, int wave, int lane, float (*dzb)[BW_LDS],
                                               float (*dza)[QZ]) {
  const int cs = lane & 15, ug = wave * 4 + (lane >> 4), q = cs >> 2;
  const bool b0 = cs & 1, b1 = cs & 2;
  const int b = blockIdx.x, T = a.T, L = a.L;
  const int u = min(4 * ug + (cs & 3), LH - 1);      // surplus groups without a latent duplicate unit 87
  const int zg0 = 4 * (ug - 22);                     // first latent of a surplus group
  const bool zgroup = DEC && ug >= 22 && zg0 < L;
  const int lat = zg0 + (cs & 3);                    // latent this lane finishes after the reduce-scatter
  const bool zlane = zgroup && lat < L;
  const Sel4 sel_q(q);
  float* gates = DEC ? a.gates_d : a.gates_e;
  const float* csp = DEC ? a.cs_d : a.cs_e;
  float* dzsum = DEC ? a.dzsum_d : a.dzsum_e;

  f2 Ur[BW_CW][2];    // [column][unit pair]; a latent group holds rows of Kz instead
  {
    const float4* pw = reinterpret_cast<const float4*>(a.pack) + (DEC ? PK_BD : PK_BE) + wave * BW_CW * 64 + lane;
#pragma unroll
    for (int c = 0; c < BW_CW; ++c) {
      const float4 v = pw[c * 64];
      Ur[c][0][0] = v.x; Ur[c][0][1] = v.y; Ur[c][1][0] = v.z; Ur[c][1][1] = v.w;
    }
  }
  // encoder: this unit's row of the head kernel (dh_enc = dzargs . Wz^T), zero beyond 2L
  float Wzr[ZP];
  if (!DEC) {
#pragma unroll
    for (int j = 0; j < ZP; ++j) {
      const float v = a.Wz[(size_t)u * 2 * L + min(j, 2 * L - 1)];
      Wzr[j] = j < 2 * L ? v : 0.f;
    }
  }

  const size_t rowbt = (size_t)b * T;
  float dc = 0.f;
  float zsum = 0.f;                  // sum_t dz of this lane's gate column (the 4 replicas of a unit share the 4 gates)

  struct Raw { float zi, zf, g, zo, c, cp, dh; };
  struct Coef { float ko, kc, ki, kf, kg, kcarry, dhh; };
  const float* g_base = gates + rowbt * LG + u;
  // per-lane streams behind the (c_t, c_{t-1}, dh_t) load slots: a latent lane reads (mean, log_var, eps) of its
  // latent there instead (one step later in time, see below), so every lane issues the same loads
  const float* pc = zlane ? a.zargs + rowbt * 2 * L + lat : csp + rowbt * LH + u;
  const float* pd = zlane ? a.eps + rowbt * L + lat : (DEC ? a.dhs_d + rowbt * LH + u : pc);
  const int stc = zlane ? 2 * L : LH, std_ = zlane ? L : (DEC ? LH : 0);
  const int ppo = zlane ? L : 0;                       // offset of the second stream (log_var column / same array)
  const float hk = 0.5f * a.kl_scale;
  // Offsets stay 32-bit and their per-lane products go through v_mul_u32_u24 (time index < 2^24, strides <= 88): the
  // 64-bit `(size_t)t * stride` forms compile to quarter-rate v_mul_lo_u32 / v_mad_u64_u32, ~30 issue slots of a step.
  auto load_raw = [&](int tr) {       // regular lanes: step max(tr, 0); latent lanes: step tr + 1 (clamped to the window)
    Raw r;
    const int t = max(tr, 0);
    const float* gp = g_base + (unsigned)t * (unsigned)LG;                   // uniform offset: scalar multiply
    r.zi = gp[0]; r.zf = gp[LH]; r.g = gp[2 * LH]; r.zo = gp[3 * LH];
    const int tz = zlane ? min(max(tr + 1, 0), T - 1) : t;
    r.c = pc[__umul24(tz, stc)];
    const float cprev = pc[__umul24(zlane ? tz : max(tz - 1, 0), stc) + (unsigned)ppo];
    r.cp = (t > 0 || zlane) ? cprev : 0.f;
    r.dh = DEC ? pd[__umul24(tz, std_)] : 0.f;
    return r;
  };
  auto make_coef = [&](const Raw& r) {
    Coef k;
    const float ig = gate_fn<GATE>(r.zi), fg = gate_fn<GATE>(r.zf), og = gate_fn<GATE>(r.zo);
    const float tc = fast_tanh(r.c);
    k.ko = tc * gate_grad<GATE>(r.zo, og);
    k.kc = og * (1.f - tc * tc);
    k.ki = r.g * gate_grad<GATE>(r.zi, ig);
    k.kf = r.cp * gate_grad<GATE>(r.zf, fg);
    k.kg = ig * (1.f - r.g * r.g);
    k.kcarry = fg;
    k.dhh = r.dh;
    if (DEC && wave == PNW - 1) {
      // latent lanes: (c, cp, dh) = (mean, log_var, eps); dzargs = dZ * ki + kf with
      //   mean column (replica 0): ki = 1, kf = kl*mean;  log_var column: ki = eps*sd/2, kf = -kl*(1 - sd^2)/2
      const float sd = __expf(0.5f * r.cp);
      const float zi_ = q == 0 ? 1.f : 0.5f * r.dh * sd;
      const float zf_ = q == 0 ? a.kl_scale * r.c : -hk * (1.f - sd * sd);
      k.ki = zgroup ? zi_ : k.ki;
      k.kf = zgroup ? zf_ : k.kf;
    }
    return k;
  };
  // With the +1 offset above the coefficients a latent lane holds during iteration (step t) are those of step
  // t+1, whose dZ its matvec has just produced from dz_dec_{t+1}.
  // Two steps of loads in flight: the values of step t are requested at iteration t+2 and turned into coefficients at
  // the top of iteration t (off the recurrence: the coefficients are first used after the matvec).  One step of
  // lookahead is less than a load takes from HBM under load.
  Raw raw0 = load_raw(T - 1);         // step t
  Raw raw1 = load_raw(T - 2);         // step t-1
  g_pair_dump[lane] = 0.f;           // one store after the prologue's loads (see lstm_bwd_kernel)

  // output slot: regular lanes: dz of gate q (4 replicas share the 4 gates); latent lanes: replica 0 the
  // mean column of dzargs, replica 1 the log_var column
  const int col = q * LH + u;
  float* gptr = gates + (rowbt + T - 1) * LG + col;
  int gstr = -LG;
  int lpos = BW_CP * (col / BW_CW) + col % BW_CW;
  if (zgroup) {
    const bool live = zlane && q < 2;
    gptr = live ? a.dzargs + (rowbt + T - 1) * 2 * L + q * L + lat : g_pair_dump + lane;
    gstr = live ? -2 * L : 0;
    lpos = q * L + lat;              // position in the dzargs LDS vector
  }

  auto matvec = [&](int cur) {        // reduce-scattered: this lane's unit (or latent) total
    const float4* dp = reinterpret_cast<const float4*>(&dzb[cur][BW_CP * cs]);
    float dv[BW_CP];
#pragma unroll
    for (int j = 0; j < BW_CP / 4; ++j) {
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
    return x;
  };

  if (!DEC) {                        // the encoder chain runs two iterations behind: dzargs_t leaves the decoder
    step_barrier();                  // chain at the end of the iteration after dz_dec_t
    step_barrier();
  }

  for (int i = 0; i < T; ++i) {
    const int t = T - 1 - i;
    const int cur = i & 1;
    const Raw raw2 = load_raw(t - 2);
    const Coef k = make_coef(raw0);
    float dhup;
    if (DEC) {
      dhup = k.dhh;
    } else {                          // dh_enc_t = dzargs_t . Wz[u,:]  (dzargs_t was written one iteration ago)
      const int par = (i + 1) & 1;    // written by decoder iteration i + 1
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int j4 = 0; j4 < ZP; j4 += 4) {           // columns beyond 2L: zero weights, zero LDS
        const float4 v = *reinterpret_cast<const float4*>(&dza[par][j4]);
        s0 = fmaf(v.x, Wzr[j4], s0); s1 = fmaf(v.y, Wzr[j4 + 1], s1);
        s0 = fmaf(v.z, Wzr[j4 + 2], s0); s1 = fmaf(v.w, Wzr[j4 + 3], s1);
      }
      dhup = s0 + s1;
    }
    const float dhrec = matvec(cur);
    const float dh = dhup + dhrec;
    dc = fmaf(dh, k.kc, dc);
    float dz[4];
    dz[0] = dc * k.ki;
    dz[1] = dc * k.kf;
    dz[2] = dc * k.kg;
    dz[3] = dh * k.ko;
    dc = dc * k.kcarry;
    float val = sel_q(dz);
    zsum += val;
    if (DEC && wave == PNW - 1) {     // wave-uniform: latent lanes turn dZ_{t+1} into dzargs_{t+1}
      const float zv = fmaf(dhrec, k.ki, k.kf);
      val = zgroup ? zv : val;
      if (zlane && q < 2) dza[i & 1][lpos] = zv;
    }
    if (!zgroup) dzb[cur ^ 1][lpos] = val;
    *gptr = val;
    gptr += (zgroup && i == 0) ? 0 : gstr;            // the head lags one step
    raw0 = raw1;
    raw1 = raw2;
    step_barrier();
  }
  if (DEC) {
    // iteration T: dZ_0 -> dzargs_0
    if (wave == PNW - 1) {
      const float dZ = matvec(T & 1);
      const Coef klast = make_coef(raw0);          // raw0 = load_raw(-1): a latent lane's values of step 0
      const float zv = fmaf(dZ, klast.ki, klast.kf);
      if (zlane && q < 2) {
        *gptr = zv;
        dza[T & 1][lpos] = zv;
      }
    }
    step_barrier();
    step_barrier();
  }
  if (!zgroup) dzsum[(size_t)b * LG + col] = zsum;
}

template <int GATE, int ZP>
__global__ __launch_bounds__(PNT) void lstm_pair_bwd_kernel(PairBwdArgs a) {
  __shared__ __attribute__((aligned(16))) float dzbuf[2][2][BW_LDS];       // [chain][parity][sliced dz]
  __shared__ __attribute__((aligned(16))) float dza[2][QZ];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 2 * 2 * BW_LDS; i += PNT) (&dzbuf[0][0][0])[i] = 0.f;
  if (tid < 2 * QZ) (&dza[0][0])[tid] = 0.f;
  __syncthreads();
  if (wave < PNW) pair_bwd_chain<GATE, true, ZP>(a, wave, lane, dzbuf[0], dza);
  else pair_bwd_chain<GATE, false, ZP>(a, wave - PNW, lane, dzbuf[1], dza);
}

}  // namespace clv

#ifdef PAIR_STAMPS
extern "C" int clv_debug_pair_stamps(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(clv::g_pair_stamps), sizeof(unsigned long long) * 64);
}
extern "C" int clv_debug_pair_arrive(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(clv::g_pair_arrive), sizeof(unsigned long long) * 8 * 12 * 2);
}
#endif

extern "C" int clv_lstm_pair_supported(int H, int L) { return H == clv::LH && L >= 1 && L <= clv::QL; }

extern "C" size_t clv_lstm_pair_pack_floats(void) { return (size_t)clv::PK_TOTAL * 4; }

extern "C" int clv_lstm_pair_pack(int H, int L, const float* U_enc, const float* U_dec, const float* Kz, const float* Wz,
                                  float* pack, void* stream) {
  using namespace clv;
  if (!clv_lstm_pair_supported(H, L) || !U_enc || !U_dec || !Kz || !Wz || !pack || ((uintptr_t)pack) % 16 != 0)
    return CLV_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  PairPackArgs a{L, U_enc, U_dec, Kz, Wz, reinterpret_cast<float4*>(pack)};
  ProfScope p("lstm_pair_pack", s);
  hipLaunchKernelGGL(lstm_pair_pack_kernel, dim3((PK_TOTAL + 255) / 256), dim3(256), 0, s, a);
  return launch_status();
}

extern "C" int clv_lstm_pair_fwd(int B, int T, int H, int L, int gate_act,
                                 float* gates_enc, const float* rowbias_enc,
                                 float* gates_dec, int dec_has_xproj, const float* rowbias_dec,
                                 const float* pack, const float* bz, const float* eps,
                                 float* hs_enc, float* cs_enc, float* hs_dec, float* cs_dec,
                                 float* zargs, float* Z, int ldz, float* klterm, void* stream) {
  using namespace clv;
  if (!clv_lstm_pair_supported(H, L) || B <= 0 || T <= 0 || ldz < L) return CLV_EINVAL;
  if (gate_act != CLV_GATE_HARD_SIGMOID && gate_act != CLV_GATE_SIGMOID) return CLV_EINVAL;
  if (!gates_enc || !rowbias_enc || !gates_dec || !rowbias_dec || !pack || !bz || !eps ||
      !hs_enc || !cs_enc || !hs_dec || !cs_dec || !zargs || !Z || !klterm)
    return CLV_EINVAL;
  PairFwdArgs a{B, T, L, ldz, gates_enc, rowbias_enc, gates_dec, rowbias_dec, pack, bz, eps,
                hs_enc, cs_enc, gates_enc, hs_dec, cs_dec, gates_dec, zargs, Z, klterm};
  hipStream_t s = (hipStream_t)stream;
  ProfScope p("lstm_pair_fwd", s);
  const bool hard = gate_act == CLV_GATE_HARD_SIGMOID;
#define PAIR_FWD_Z(G, X, Z) hipLaunchKernelGGL((lstm_pair_fwd_kernel<G, X, Z>), dim3(B), dim3(PNT), 0, s, a)
#define PAIR_FWD(G, X) do { if (L <= 4) PAIR_FWD_Z(G, X, 1); else if (L <= 8) PAIR_FWD_Z(G, X, 2); else PAIR_FWD_Z(G, X, 4); } while (0)
  if (hard) { if (dec_has_xproj) PAIR_FWD(CLV_GATE_HARD_SIGMOID, true); else PAIR_FWD(CLV_GATE_HARD_SIGMOID, false); }
  else { if (dec_has_xproj) PAIR_FWD(CLV_GATE_SIGMOID, true); else PAIR_FWD(CLV_GATE_SIGMOID, false); }
#undef PAIR_FWD_Z
#undef PAIR_FWD
  return launch_status();
}

extern "C" int clv_lstm_pair_bwd(int B, int T, int H, int L, int gate_act, float kl_scale,
                                 const float* pack, const float* Wz,
                                 const float* dhs_dec, const float* cs_dec, const float* cs_enc,
                                 float* gates_dec_inout_dz, float* gates_enc_inout_dz,
                                 float* dzsum_dec, float* dzsum_enc,
                                 const float* zargs, const float* eps, float* dzargs, void* stream) {
  using namespace clv;
  if (!clv_lstm_pair_supported(H, L) || B <= 0 || T <= 0) return CLV_EINVAL;
  if (gate_act != CLV_GATE_HARD_SIGMOID && gate_act != CLV_GATE_SIGMOID) return CLV_EINVAL;
  if (!pack || !Wz || !dhs_dec || !cs_dec || !cs_enc || !gates_dec_inout_dz || !gates_enc_inout_dz ||
      !dzsum_dec || !dzsum_enc || !zargs || !eps || !dzargs)
    return CLV_EINVAL;
  PairBwdArgs a{B, T, L, kl_scale, pack, Wz, dhs_dec, cs_dec, cs_enc, gates_dec_inout_dz,
                gates_enc_inout_dz, dzsum_dec, dzsum_enc, zargs, eps, dzargs};
  hipStream_t s = (hipStream_t)stream;
  ProfScope p("lstm_pair_bwd", s);
  const bool hard = gate_act == CLV_GATE_HARD_SIGMOID;
#define PAIR_BWD(G, Z) hipLaunchKernelGGL((lstm_pair_bwd_kernel<G, Z>), dim3(B), dim3(PNT), 0, s, a)
#define PAIR_BWD_Z(Z) do { if (hard) PAIR_BWD(CLV_GATE_HARD_SIGMOID, Z); else PAIR_BWD(CLV_GATE_SIGMOID, Z); } while (0)
  if (2 * L <= 4) PAIR_BWD_Z(4);
  else if (2 * L <= 8) PAIR_BWD_Z(8);
  else PAIR_BWD_Z(16);
#undef PAIR_BWD_Z
#undef PAIR_BWD
  return launch_status();
}
