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
#include "lstm_pair_pack.h"
#include "label_bwd_row.h"
#include "philox.h"
#include "reduce_job.h"

namespace clv {

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
// -DPAIR_PHASES: every wave of workgroup 0 sums, over all steps of the forward kernel, the shader-clock time it spends
//   [0] from leaving the barrier to having its prefetched input (the s_waitcnt vmcnt at the top of the step),
//   [1] from there to the reduced gate sum (LDS reads, 44 packed FMAs, slice reduce),
//   [2] from there to its arrival at the barrier (activation, cell, LDS write, stores issued),
//   [3] in the barrier (incl. the lgkmcnt(0) in front of it), until it runs again
// in SGPR accumulators (nothing is stored inside the loop); tools/pair_phases.py prints them per wave.
#ifdef PAIR_PHASES
__device__ unsigned g_pair_phase[12][4];
#define PH_DECL unsigned long long phA = 0, phB = 0, phD = 0, phC = 0; unsigned phPrev = 0, phAcc0 = 0, phAcc1 = 0, phAcc2 = 0, phAcc3 = 0, phN = 0
#define PH_A() asm volatile("s_memtime %0" : "=s"(phA))
#define PH_B(dep) asm volatile("s_memtime %0" : "=s"(phB) : "v"(dep))
#define PH_D(dep) asm volatile("s_memtime %0" : "=s"(phD) : "v"(dep))
#define PH_C(dep) asm volatile("s_memtime %0" : "=s"(phC) : "v"(dep))
#define PH_ACC()                                                                     \
  do {                                                                               \
    phAcc0 += (unsigned)phB - (unsigned)phA; phAcc1 += (unsigned)phD - (unsigned)phB;   \
    phAcc2 += (unsigned)phC - (unsigned)phD;                                          \
    phAcc3 += phN ? (unsigned)phA - phPrev : 0u;                                      \
    phPrev = (unsigned)phC; phN = 1;                                                  \
  } while (0)
#define PH_OUT(widx)                                                                 \
  do {                                                                               \
    if (blockIdx.x == 0 && lane == 0) {                                              \
      g_pair_phase[widx][0] = phAcc0; g_pair_phase[widx][1] = phAcc1;                \
      g_pair_phase[widx][2] = phAcc2; g_pair_phase[widx][3] = phAcc3;                \
    }                                                                                \
  } while (0)
#else
#define PH_DECL do { } while (0)
#define PH_A() do { } while (0)
#define PH_B(dep) do { } while (0)
#define PH_D(dep) do { } while (0)
#define PH_C(dep) do { } while (0)
#define PH_ACC() do { } while (0)
#define PH_OUT(widx) do { } while (0)
#endif

__global__ __launch_bounds__(256) void lstm_pair_pack_kernel(PairPackArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= PK_TOTAL) return;
  a.out[i] = pair_pack_element(i, a.L, a.U_e, a.U_d, a.Kz, a.Wz, [](const float* p) { return *p; });
}

struct PairFwdArgs {
  int B, T, L, ldz;
  const float* xproj_e;   // [B,T,352] x_t.K_x      (same buffer as gates_e)
  const float* rb_e;      // [B,352]   W.K_w + bias
  const float* xproj_d;   // [B,T,352] x_{t-1}.K_x  (same buffer as gates_d) or unused
  const float* rb_d;
  const float* pack;      // weights in lane order (clv_lstm_pair_pack)
  const float* bz;        // [2L]
  // note lists (clv_gather_rows_multi, notes_out) of the frames x_t / x_{t-1} and the kernels' frame rows [88,352]: when
  // given, the input projections are gathered in this kernel and the gate buffers are not read
  const unsigned char* notes_e; const float* Kx_e;
  const unsigned char* notes_d; const float* Kx_d;
  float* eps;             // [B,T,L]: read, or drawn in the kernel's prologue and written (noise.on)
  struct { int on; uint32_t k0, k1, stream, step; uint64_t first; const int32_t* step_dev; } noise;
  float *hs_e, *aux_e, *gates_e, *hs_d, *aux_d, *gates_d;      // aux: [B*T, 2, 88] = (kcarry, kc)
  float* zargs;           // [B*T,2L]
  float* Z;               // [B*T] rows of stride ldz
  float* klterm;          // [B*T,L]  L * KL_l: the mean over all entries is the per-frame KL
};

// lane masks of the k-slice index s (SGPR pairs; selects on them are single v_cndmask: see Sel4 in lstm_common.h)
struct SliceMasks {
  unsigned long long m1, m2, odd;
  __device__ __forceinline__ explicit SliceMasks(int s) {
    m1 = lane_mask(s == 1);
    m2 = lane_mask(s == 2);
    odd = lane_mask(s & 1);
  }
};

// quad broadcast of lane K of every quad (DPP operand of the consuming instruction where the ISA has the form)
template <int K>
__device__ __forceinline__ float quad_bcast(float v) { return dpp_mov<K * 0x55>(v); }

// One LSTM cell on the reduce-scattered layout.  z: this lane's gate pre-activation (lane s: gate s = i, f, g, o).
// Returns h (all four lanes); c is updated in all four lanes.  What lane s stores afterwards:
//   vA = (ki, kf, kg, ko)[s] -> gates[., s*88 + u];  lane 0: h -> hs;  lane 1: act = f = kcarry, lane 2: kc -> aux
template <int GATE>
__device__ __forceinline__ float pair_cell(const SliceMasks& sm, float z, float& c, float& vA, float& act, float& kc) {
  float a, d;                        // activation of this lane's gate and its derivative
  const float th = PAIR_ABL == 4 ? z * 0.25f : fast_tanh(z);
  if (GATE == CLV_GATE_HARD_SIGMOID) {
    const float y = fmaf(0.2f, z, 0.5f);
    const float hsv = __builtin_amdgcn_fmed3f(y, 0.f, 1.f);
    const float hd = (y == hsv) ? 0.2f : 0.f;          // TF's clip passes the gradient at ties
    a = Sel4::pick(sm.m2, th, hsv);
    d = Sel4::pick(sm.m2, fmaf(-th, th, 1.f), hd);
  } else {
    const float sg = sigmoidf_(z);
    a = Sel4::pick(sm.m2, th, sg);
    d = fmaf(-a, a, Sel4::pick(sm.m2, 1.f, a));       // 1 - g^2  |  a - a^2
  }
  const float gg = quad_bcast<2>(a);
  const float igg = quad_bcast<0>(a) * gg;
  const float cn = fmaf(quad_bcast<1>(a), c, igg);
  const float tc = PAIR_ABL == 4 ? cn * 0.25f : fast_tanh(cn);
  const float og = quad_bcast<3>(a);
  const float h = og * tc;
  // k of this lane: d * (gg, c_{t-1}, ig, tc)[s]: lanes 0 / 2 swap their activations, lanes 1 / 3 take c_{t-1} / tc
  const float sw = dpp_mov<0xC6>(a);                   // quad_perm [2,1,0,3]
  const float other = Sel4::pick(sm.m1, c, tc);
  vA = d * Sel4::pick(sm.odd, other, sw);
  kc = og * fmaf(-tc, tc, 1.f);
  act = a;
  c = cn;
  return h;
}

// sum of the four k-slice partials, scattered: lane s ends with the total of ITS accumulator 0 = gate s
__device__ __forceinline__ float reduce_scatter4(float a0, float a1, float a2, float a3) {
  const float r0 = a0 + dpp_mov<0xB1>(a1);             // partner s^1: its accumulator 1 is gate s
  const float r2 = a2 + dpp_mov<0xB1>(a3);             //              its accumulator 3 is gate s^2
  return r0 + dpp_mov<0x4E>(r2);                       // partner s^2: its r2 is gate s
}

template <int P> struct ParC { static constexpr int value = P; };

// PAIR_CONTIG=1 (measurement build): the forward's one load and three stores per step in the contiguous lane order, through
// ds_bpermute.  rocprofv3, same session (tools/sessions/r03_contig.sh, profiles/r03_pair_lane_order_ab.txt): 100.5 us
// against 95.5-96.5 us in the recurrence's own order -- with so few accesses per step the permutes' latency on the step's
// critical path costs more than the line lookups they save (with the 8 loads of the fused projections it is the other
// way round: that path always uses the contiguous order).
#ifndef PAIR_CONTIG
#define PAIR_CONTIG 0
#endif

// Per-step global accesses are buffer instructions: a resource descriptor per array (base = this batch row's first
// frame, num_records = its T frames: 4 SGPRs), the lane's byte offset in ONE VGPR that never changes, the step's row offset
// in an SGPR.  No vector address arithmetic at all, and a lane that has nothing to store (or load) gets an offset
// beyond num_records: the hardware drops the store (returns 0), so no store sits under a divergent branch -- which
// matters beyond the branch itself: with stores on conditional paths the compiler's vmcnt bookkeeping assumes they may
// not have been issued and waits for the NEXT newer load instead (the waits of the two-step lookahead collapse).
// cache-policy bits of the per-step stores (measurement builds: 2 = nt, 1 = sc0, 17 = sc0 sc1; the default write-back policy
// is the fastest, profiles/r03_pair_ablation.txt)
#ifndef PAIR_STORE_AUX
#define PAIR_STORE_AUX 0
#endif
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr unsigned BUF_OOB = 0x80000000u;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void buf_store(float v, rsrc_t r, unsigned voff, unsigned soff) {
  if (PAIR_ABL == 5) { asm volatile("" :: "v"(v)); return; }
  if (PAIR_ABL == 9) soff = 0;      // every step's stores land on the row's first frame: no write traffic beyond L2
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, (int)soff, PAIR_STORE_AUX);
}
__device__ __forceinline__ float lane_permute(int addr, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v)));
}

// Every load of the prologue (weights, the first two steps' values) has landed before the step loop is entered: the
// compiler's wait-count bookkeeping merges the loop-entry state with the back edge's, and with ~25 prologue loads still
// "in flight" at the entry it places s_waitcnt vmcnt(<small>) INSIDE the loop for the weights' first use -- which in the
// steady state waits for the loads the previous step has just issued.  vmcnt(0) expcnt(7) lgkmcnt(15): the builtin (not
// inline asm) so that the bookkeeping sees it.
__device__ __forceinline__ void prologue_loads_done() { __builtin_amdgcn_s_waitcnt(0x0F70); }

// h slice of a unit in LDS: 22 values + 2 spare floats.  The DECODER's spare floats carry z_t: slice s' holds the latents
// s' and s' + 4 behind its h values, so the decoder's sixth ds_read_b128 brings them along (no separate z buffer, no read)
__device__ __forceinline__ int pair_hslot(int u) { return PKP * (u / PKK) + (u % PKK); }

// The input projection x_t . K_x of a chain, two steps ahead of its use, in two register sets (even / odd steps).
//   XL == false: one value per lane and step, read from the gate buffer (written by clv_sparse_proj or a GEMM);
//   XL == true:  gathered HERE: the frame's note list (bytes, CLV_NOTE_NONE-terminated) comes in through two scalar
//                registers -- requested two steps before the loads that use them -- and the kernel rows of the (up to) 8
//                (Not a scalar load: s_load shares lgkmcnt with the LDS, and the step barrier's lgkmcnt(0) then waits
//                for a scalar-cache miss issued moments before -- 0.6 us per step.  The list's first 8 bytes come through
//                a broadcast buffer load, counted in vmcnt like everything else, and two v_readfirstlane.)
//                first notes are 8 buffer loads whose row offset is an SGPR (note * 1408 bytes): no vector address
//                arithmetic, and an absent note (88) points beyond num_records, i.e. returns 0 without touching
//                memory.  The kernel rows are L2-resident (124 KB per LSTM).  More than 8 notes in a frame (1.6 % of
//                the frames at piano-roll density): the rest is added one load at a time, in place.
//                Lane order of these loads: NOT the (unit, k-slice) order of the recurrence -- there the four lanes of a
//                quad read four different 64-byte pieces of a kernel row, and the texture path looks up a cache line per
//                (quad, piece): 64 lookups per wave load instead of 4; with 8 loads per wave and step that alone was
//                0.7 us per step.  Lane l = 16 g + j loads column g*88 + u0 + j (four contiguous runs of 64 bytes), sums its
//                notes there, and ONE ds_bpermute per step moves the sum to lane 4 j + g, which owns gate g of unit u0 + j.
//                Why here: the separate projection launch wrote 92 MB and this kernel read them back (26 us + ~35 us of
//                HBM time per step); the price is 7 more vector adds per lane and step.
template <bool XL> struct XSet { float v[XL ? 8 : 1]; bool more; };      // more (uniform): the frame has more than 8 notes
template <bool XL>
struct XProj {
  rsrc_t r_kx;            // XL: the kernel's frame rows; else: the gate buffer of this batch row
  unsigned vo;            // lane's byte offset (its gate column); BUF_OOB for lanes without a unit
  // XL: this batch row's note lists, through the CONSTANT address space: nothing in this kernel writes them, and only a
  // load the compiler knows to be invariant becomes a scalar load (the generic-pointer form was a vector load +
  // s_waitcnt vmcnt(0) + v_readfirstlane at the top of every step)
  typedef unsigned nn_t __attribute__((ext_vector_type(2)));      // a list's first 8 bytes
  typedef const nn_t __attribute__((address_space(4))) * notes_ptr;
  typedef const unsigned __attribute__((address_space(4))) * notes_ptr32;
  notes_ptr nrow;
  rsrc_t r_n;             // XL: the same lists as a buffer (the first 8 bytes of a list travel through vmcnt)
  int T;
  __device__ __forceinline__ static notes_ptr as_notes(const unsigned char* p) { return (notes_ptr)(uintptr_t)p; }
  // the first 8 bytes of frame f's list, in every lane (a broadcast load: one address)
  __device__ __forceinline__ uint2 notes(int f) const {
    const nn_t v = __builtin_amdgcn_raw_buffer_load_b64(r_n, 0, (int)((unsigned)min(f, T - 1) * (unsigned)CLV_NOTE_ROW), 0);
    return make_uint2(v.x, v.y);
  }
  __device__ __forceinline__ void load(XSet<XL>& x, uint2 nv, int f) const {
    f = min(f, T - 1);
    if constexpr (!XL) {
      x.v[0] = buf_load(r_kx, vo, (unsigned)f * (unsigned)(LG * 4));
    } else {
      uint2 nn;
      nn.x = __builtin_amdgcn_readfirstlane(nv.x);
      nn.y = __builtin_amdgcn_readfirstlane(nv.y);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        x.v[j] = buf_load(r_kx, vo, ((nn.x >> (8 * j)) & 255u) * (unsigned)(LG * 4));
        x.v[4 + j] = buf_load(r_kx, vo, ((nn.y >> (8 * j)) & 255u) * (unsigned)(LG * 4));
      }
      x.more = (nn.y >> 24) != CLV_NOTE_NONE;
    }
  }
  // the 9th.. note of frame f (uniform, rare: 1.6 % of the frames at piano-roll density): added where the set is consumed,
  // one waited load at a time -- nothing loop-carried depends on this path
  __device__ __forceinline__ float extra(int f, int paddr) const {
    const notes_ptr32 w = (notes_ptr32)nrow + (unsigned)min(f, T - 1) * (CLV_NOTE_ROW / 4) + 2;
    float e = 0.f;
    for (int q = 0; q < (CLV_NOTE_ROW - 8) / 4; ++q) {
      const unsigned ww = w[q];
#pragma unroll
      for (int j = 0; j < 4; ++j) e += buf_load(r_kx, vo, ((ww >> (8 * j)) & 255u) * (unsigned)(LG * 4));
      if ((ww >> 24) == CLV_NOTE_NONE) break;
    }
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(paddr, __builtin_bit_cast(int, e)));
  }
  // pins the use of a set behind the previous step's barrier (else the compiler computes the sum half a step early and
  // waits there for loads issued moments before)
  __device__ __forceinline__ static void pin(XSet<XL>& x) {
    if constexpr (XL) asm volatile("" : "+v"(x.v[0]), "+v"(x.v[1]), "+v"(x.v[2]), "+v"(x.v[3]), "+v"(x.v[4]), "+v"(x.v[5]), "+v"(x.v[6]), "+v"(x.v[7]));
    else asm volatile("" : "+v"(x.v[0]));
  }
  // byte offset of the column a lane LOADS (XL): wave's first unit u0, lane l = 16 g + j -> column g*88 + u0 + j
  // (unit slots beyond the last unit duplicate it, like the lanes they feed)
  __device__ __forceinline__ static unsigned load_offset(int u0, int lane, int nunits) {
    const int g = lane >> 4, j = lane & 15;
    return (unsigned)(g * LH + min(u0 + j, nunits - 1)) * 4u;
  }
  // ds_bpermute address that brings lane (4 j + g) the value of lane (16 g + j)
  __device__ __forceinline__ static int perm_addr(int lane) { return 4 * (16 * (lane & 3) + (lane >> 2)); }
  // ds_bpermute address for the way back (stores): lane (16 g + j) takes the value of lane (4 j + g)
  __device__ __forceinline__ static int perm_addr_inv(int lane) { return 4 * (4 * (lane & 15) + (lane >> 4)); }
  __device__ __forceinline__ static float sum(const XSet<XL>& x, float rb, int paddr, float mask = 1.f) {
    float t = x.v[0];
    if constexpr (XL) t = ((x.v[0] + x.v[1]) + (x.v[2] + x.v[3])) + ((x.v[4] + x.v[5]) + (x.v[6] + x.v[7]));
    if constexpr (XL || PAIR_CONTIG) t = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(paddr, __builtin_bit_cast(int, t)));
    return fmaf(t, mask, rb);
  }
};

// LATW: this is the encoder's last wave: its 8 surplus lane groups (units 88..95) carry the latent head
template <int GATE, bool LATW, bool XL>
__device__ __forceinline__ void pair_fwd_encoder(const PairFwdArgs& a, int wave, int lane, float (*hb)[PK * PKP],
                                                 float (*hbd)[PK * PKP]) {
  const int s = lane & 3, b = blockIdx.x, T = a.T, L = a.L;
  const int u_raw = wave * 16 + (lane >> 2);
  const int u = min(u_raw, LH - 1);          // surplus groups that carry no latent duplicate unit 87
  const int zj = u_raw - LH;
  const bool is_z = LATW && zj >= 0 && 2 * zj < L;
  const int lat = 2 * zj + (s & 1);          // the latent this lane finishes
  const bool lat_ok = is_z && lat < L;
  const int zslot = PKP * (lat % PK) + PKK + lat / PK;       // where decoder lane s' = lat % 4 finds it (see pair_hslot)
  // head column this lane ends with after the reduce-scatter: (mean_2j, mean_2j+1, log_var_2j, log_var_2j+1)[s]
  const int zcol = (s >> 1) * L + lat;
  const SliceMasks sm(s);
  const unsigned long long m_lo = lane_mask(s < 2);

  f2 Up[PKK / 2][4];     // [k pair][acc] = (U[2j][g], U[2j+1][g]), g = acc ^ s
  {
    const float4* pw = reinterpret_cast<const float4*>(a.pack) + PK_FE + wave * PKK * 64 + lane;
#pragma unroll
    for (int kk = 0; kk < PKK; ++kk) {
      const float4 v = pw[kk * 64];
      Up[kk >> 1][2 * (kk & 1)] = (f2){v.x, v.y}; Up[kk >> 1][2 * (kk & 1) + 1] = (f2){v.z, v.w};
    }
  }
  const size_t bt0 = (size_t)b * T;
  const unsigned loff = s * LH + u;                          // gate column of this lane
  const rsrc_t r_g = make_rsrc(a.gates_e + bt0 * LG, T * LG * 4);       // x_t.K_x in, k out (same buffer)
  const rsrc_t r_h = make_rsrc(a.hs_e + bt0 * LH, T * LH * 4);
  const rsrc_t r_a = make_rsrc(a.aux_e + bt0 * 2 * LH, T * 2 * LH * 4);
  const rsrc_t r_e = make_rsrc(a.eps + bt0 * L, T * L * 4);
  const rsrc_t r_za = make_rsrc(a.zargs + bt0 * 2 * L, T * 2 * L * 4);
  const rsrc_t r_z = make_rsrc(a.Z + bt0 * a.ldz, T * a.ldz * 4);
  const rsrc_t r_kl = make_rsrc(a.klterm + bt0 * L, T * L * 4);
  // The CONTIGUOUS lane order (lane 16 g + j <-> gate g of unit u0 + j: four runs of 64 bytes per wave) with a
  // ds_bpermute from / to the recurrence's (unit, k-slice) order: in the latter the four lanes of a quad touch four
  // different 64-byte pieces and the texture path looks up a line per (quad, piece) -- 64 lookups per wave access
  // instead of 4.  Pays for the 8 loads of the fused projections (profiles/r03_notes_fusion_ab.txt), not for the one load
  // and three stores of the default path: see PAIR_CONTIG
  const int cg = lane >> 4, cj = lane & 15, cu_raw = wave * 16 + cj, cu = min(cu_raw, LH - 1);
  const bool c_is_z = LATW && cu_raw >= LH && 2 * (cu_raw - LH) < L;          // the slot of a latent group
  const int paddr_inv = XProj<XL>::perm_addr_inv(lane);
  unsigned vo_g = PAIR_CONTIG ? (c_is_z ? BUF_OOB : (unsigned)(cg * LH + cu) * 4u) : (is_z ? BUF_OOB : loff * 4);
  XProj<XL> xp;
  xp.r_kx = XL ? make_rsrc(a.Kx_e, CLV_NOTE_NONE * LG * 4) : r_g;
  // contiguous lane order for the loads (latent lanes mask what arrives: xmask)
  xp.vo = (XL || PAIR_CONTIG) ? XProj<XL>::load_offset(wave * 16, lane, LH) : loff * 4;
  const int paddr = XProj<XL>::perm_addr(lane);
  xp.nrow = XProj<XL>::as_notes(a.notes_e + (XL ? bt0 * CLV_NOTE_ROW : 0));
  xp.r_n = XL ? make_rsrc(a.notes_e + bt0 * CLV_NOTE_ROW, T * CLV_NOTE_ROW) : r_g;
  xp.T = T;
  unsigned vo_h = PAIR_CONTIG ? ((!c_is_z && cg == 0) ? cu * 4 : BUF_OOB) : ((!is_z && s == 0) ? u * 4 : BUF_OOB);
  unsigned vo_a = PAIR_CONTIG ? ((!c_is_z && (cg == 1 || cg == 2)) ? ((cg - 1) * LH + cu) * 4 : BUF_OOB)
                                    : ((!is_z && (s == 1 || s == 2)) ? ((s - 1) * LH + u) * 4 : BUF_OOB);
  const unsigned vo_e = (lat_ok ? lat : 0) * 4;
  const unsigned vo_za = lat_ok ? zcol * 4 : BUF_OOB;
  if (PAIR_ABL == 6) {      // timing ablation: lane-contiguous addresses (wrong layout, same instruction count)
    vo_g = is_z ? BUF_OOB : (wave * 64 + lane) * 4;
    vo_h = (!is_z && lane < 16) ? (wave * 16 + lane) * 4 : BUF_OOB;
    vo_a = (!is_z && lane >= 16 && lane < 48) ? (wave * 32 + lane - 16) * 4 : BUF_OOB;
    xp.vo = (wave * 64 + lane) * 4;
  }
  const unsigned vo_z = (lat_ok && s < 2) ? lat * 4 : BUF_OOB;
  const unsigned vo_kl = (lat_ok && s >= 2) ? lat * 4 : BUF_OOB;
  float rb;
  {
    const float* src = lat_ok ? a.bz + zcol : a.rb_e + (size_t)b * LG + loff;
    rb = *src;
    rb = (is_z && !lat_ok) ? 0.f : rb;
  }
  const float xmask = is_z ? 0.f : 1.f;
  const int hslot = pair_hslot(u);
  const float klscale = -0.5f * (float)L;

  float c = 0.f;
  // The per-step loads are requested two steps ahead into TWO register sets (the loop body is two steps): a value is
  // consumed where it landed, nothing rotates
  XSet<XL> xA, xB;                                                       // projections of steps 0 and 1
  uint2 nA = make_uint2(0, 0), nB = nA;                                  // note lists of steps 2 and 3
  if constexpr (XL) { xp.load(xA, xp.notes(0), 0); xp.load(xB, xp.notes(1), 1); nA = xp.notes(2); nB = xp.notes(3); }
  else { xp.load(xA, nA, 0); xp.load(xB, nB, 1); }
  float eA = 0.f, eB = buf_load(r_e, vo_e, 0);                           // eps of steps -1 and 0
  prologue_loads_done();

  // latent lanes: zc = this lane's head column of step i-1; (mean, log_var) of its latent sit in lanes s and s^2
  auto latent = [&](float zc, float e, float& zv, float& klv) {
    const float other = dpp_mov<0x4E>(zc);
    const float m = Sel4::pick(m_lo, zc, other), lv = Sel4::pick(m_lo, other, zc);
    const float sd = __expf(0.5f * lv);
    zv = fmaf(sd, e, m);
    klv = klscale * (1.f + lv - m * m - sd * sd);
  };
  auto gate_sum = [&](const float* hslice, float x0) {
    f2 acc[4];
    acc[0] = (f2){x0, 0.f};
#pragma unroll
    for (int j = 1; j < 4; ++j) acc[j] = (f2){0.f, 0.f};
    float hv[PKP];
    load_hslice(hslice, hv);
    slice_fma_pairs<0, PKK / 2>(hv, Up, acc);
    return reduce_scatter4(acc[0][0] + acc[0][1], acc[1][0] + acc[1][1], acc[2][0] + acc[2][1], acc[3][0] + acc[3][1]);
  };
  auto store_latent = [&](int row, float zc, float zv, float klv) {      // lanes without a latent: dropped (offset out of range)
    buf_store(zc, r_za, vo_za, (unsigned)row * (unsigned)(2 * L * 4));
    buf_store(zv, r_z, vo_z, (unsigned)row * (unsigned)(a.ldz * 4));
    buf_store(klv, r_kl, vo_kl, (unsigned)row * (unsigned)(L * 4));
  };

  PH_DECL;
  auto step = [&](auto parc, int i, XSet<XL>& xa, uint2& na, float& ea) {
    constexpr int cur = decltype(parc)::value;
    PH_A();
    if (PAIR_ABL != 7) XProj<XL>::pin(xa);
    float xv = PAIR_ABL == 7 ? rb : XProj<XL>::sum(xa, rb, paddr, LATW ? xmask : 1.f);
    if constexpr (XL) { if (xa.more) xv = fmaf(xp.extra(i, paddr), LATW ? xmask : 1.f, xv); }
    PTOP(xv);
    PH_B(xv);
    const float z = gate_sum(&hb[cur][PKP * s], xv);
    PH_D(z);
    float vA, act, kc;
    const float h = pair_cell<GATE>(sm, z, c, vA, act, kc);
    if (!is_z) hb[cur ^ 1][hslot] = h;
    if (LATW) {                     // the latent head of step i-1 (garbage at i == 0, rewritten at i == 1)
      float zv, klv;
      asm volatile("" : "+v"(ea));
      latent(z, ea, zv, klv);       // ea = eps of step i-1
      if (lat_ok && s < 2) hbd[cur ^ 1][zslot] = zv;
      store_latent(max(i - 1, 0), z, zv, klv);
    }
    float pA = vA, pB = Sel4::pick(sm.m2, kc, Sel4::pick(sm.m1, act, h));             // (h, kcarry, kc, -)[s]
    if (PAIR_CONTIG) { pA = lane_permute(paddr_inv, pA); pB = lane_permute(paddr_inv, pB); }      // -> lane 16 g + j
    buf_store(pA, r_g, vo_g, (unsigned)i * (unsigned)(LG * 4));
    buf_store(pB, r_h, vo_h, (unsigned)i * (unsigned)(LH * 4));
    buf_store(pB, r_a, vo_a, (unsigned)i * (unsigned)(2 * LH * 4));
    // the loads of step i+2 into the registers this step has just consumed; issued BEHIND the stores, so that the wait at
    // the top of step i+2 lets everything step i+1 issues stay in flight
    if (PAIR_ABL == 7) XProj<XL>::pin(xa);      // (timing ablation: the prefetched value is "used" here, 0.8 steps later)
    xp.load(xa, na, i + 2);
    if constexpr (XL) na = xp.notes(i + 4);
    if (LATW) ea = buf_load(r_e, vo_e, (unsigned)min(i + 1, T - 1) * (unsigned)(L * 4));
    PARRIVE(wave, i + 2, vA + h);                       // the encoder runs two steps ahead of the decoder's step index
    PH_C(vA + h);
    step_barrier();
    PH_ACC();
  };
  // An odd T runs one step more (no separate tail: a second copy of the step after the loop costs register moves at the
  // loop header): step T reads clamped rows, its stores lie beyond num_records and are dropped, and its latent head is
  // the one of step T-1, which the epilogue below writes again.
  for (int i = 0; i < T; i += 2) {
    step(ParC<0>(), i, xA, nA, eA);
    step(ParC<1>(), i + 1, xB, nB, eB);
  }
  PH_OUT(wave);
  // iteration T: only the latent head of step T-1 is left
  if (LATW) {
    const float z = gate_sum(&hb[T & 1][PKP * s], rb);
    float zv, klv;
    latent(z, buf_load(r_e, vo_e, (unsigned)(T - 1) * (unsigned)(L * 4)), zv, klv);
    if (lat_ok && s < 2) hbd[(T + 1) & 1][zslot] = zv;
    store_latent(T - 1, z, zv, klv);
  }
  step_barrier();          // the decoder chain runs two steps behind
  step_barrier();
}

// ZQ: latents per decoder lane actually present (ceil(latent_dim / 4) <= 2): lane s multiplies the latents s, s+4
template <int GATE, bool HASXP, int ZQ, bool XL>
__device__ __forceinline__ void pair_fwd_decoder(const PairFwdArgs& a, int wave, int lane, float (*hb)[PK * PKP]) {
  const int s = lane & 3, b = blockIdx.x, T = a.T;
  const int u = min(wave * 16 + (lane >> 2), LH - 1);
  const SliceMasks sm(s);
  f2 Up[PKK / 2][4];     // [k pair][acc] = (U[2j][g], U[2j+1][g]), g = acc ^ s
  float Kzr[ZQ][4];      // lane s takes the latents s, s+4; [acc]
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
      Kzr[q][0] = v.x; Kzr[q][1] = v.y; Kzr[q][2] = v.z; Kzr[q][3] = v.w;
    }
  }
  const size_t bt0 = (size_t)b * T;
  const unsigned loff = s * LH + u;
  const rsrc_t r_g = make_rsrc(a.gates_d + bt0 * LG, T * LG * 4);       // x_{t-1}.K_x in, k out (same buffer)
  const rsrc_t r_h = make_rsrc(a.hs_d + bt0 * LH, T * LH * 4);
  const rsrc_t r_a = make_rsrc(a.aux_d + bt0 * 2 * LH, T * 2 * LH * 4);
  const int cg = lane >> 4, cu = min(wave * 16 + (lane & 15), LH - 1);        // contiguous order: see the encoder
  const int paddr_inv = XProj<XL>::perm_addr_inv(lane);
  unsigned vo_g = PAIR_CONTIG ? (unsigned)(cg * LH + cu) * 4u : loff * 4;
  unsigned vo_h = PAIR_CONTIG ? (cg == 0 ? cu * 4 : BUF_OOB) : (s == 0 ? u * 4 : BUF_OOB);
  unsigned vo_a = PAIR_CONTIG ? ((cg == 1 || cg == 2) ? ((cg - 1) * LH + cu) * 4 : BUF_OOB)
                                    : ((s == 1 || s == 2) ? ((s - 1) * LH + u) * 4 : BUF_OOB);
  XProj<XL> xp;
  xp.r_kx = XL ? make_rsrc(a.Kx_d, CLV_NOTE_NONE * LG * 4) : r_g;
  xp.vo = (XL || PAIR_CONTIG) ? XProj<XL>::load_offset(wave * 16, lane, LH) : loff * 4;
  const int paddr = XProj<XL>::perm_addr(lane);
  xp.nrow = XProj<XL>::as_notes(a.notes_d + (XL ? bt0 * CLV_NOTE_ROW : 0));
  xp.r_n = XL ? make_rsrc(a.notes_d + bt0 * CLV_NOTE_ROW, T * CLV_NOTE_ROW) : r_g;
  xp.T = T;
  const float rb = a.rb_d[(size_t)b * LG + loff];
  const int hslot = pair_hslot(u);
  if (PAIR_ABL == 6) {
    vo_g = min(wave * 64 + lane, LG - 1) * 4;
    vo_h = lane < 16 ? min(wave * 16 + lane, LH - 1) * 4 : BUF_OOB;
    vo_a = (lane >= 16 && lane < 48) ? min(wave * 32 + lane - 16, 2 * LH - 1) * 4 : BUF_OOB;
    xp.vo = min(wave * 64 + lane, LG - 1) * 4;
  }
  float c = 0.f;
  XSet<XL> xA, xB;
  uint2 nA = make_uint2(0, 0), nB = nA;
#pragma unroll
  for (int j = 0; j < (XL ? 8 : 1); ++j) xA.v[j] = xB.v[j] = 0.f;
  xA.more = xB.more = false;
  if constexpr (HASXP) {
    if constexpr (XL) { xp.load(xA, xp.notes(0), 0); xp.load(xB, xp.notes(1), 1); nA = xp.notes(2); nB = xp.notes(3); }
    else { xp.load(xA, nA, 0); xp.load(xB, nB, 1); }
  }
  prologue_loads_done();
  step_barrier();          // the encoder is two steps ahead
  step_barrier();
  PH_DECL;
  auto step = [&](auto parc, int t, XSet<XL>& xa, uint2& na) {
    constexpr int cur = decltype(parc)::value;
#ifdef PAIR_STAMPS
    unsigned long long pst[8];
#endif
    PH_A();
    if (PAIR_ABL != 7) XProj<XL>::pin(xa);
    float xv = PAIR_ABL == 7 ? rb : XProj<XL>::sum(xa, rb, paddr);
    if constexpr (XL) { if (xa.more) xv += xp.extra(t, paddr); }
    PSTAMP(0, xv);
    PTOP(xv);
    PH_B(xv);
    f2 acc[4];       // (even k, odd k) partial sums
    acc[0] = (f2){xv, 0.f};
#pragma unroll
    for (int j = 1; j < 4; ++j) acc[j] = (f2){0.f, 0.f};
    float hv[PKP];   // 22 h values, then z_t[s], z_t[s + 4]
    load_hslice(&hb[cur][PKP * s], hv);
    PSTAMP(1, hv[0]);                      // the first 16 bytes of h have arrived
    PSTAMP(2, hv[PKP - 4]);                // the last
    slice_fma_pairs<0, PKK / 2>(hv, Up, acc);
    PSTAMP(3, acc[3][0] + acc[0][1]);
    float as[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {          // + z_t . K_z (rows of K_z beyond latent_dim are zero registers)
      as[j] = acc[j][0] + acc[j][1];
#pragma unroll
      for (int q = 0; q < ZQ; ++q) as[j] = fmaf(hv[PKK + q], Kzr[q][j], as[j]);
    }
    const float z = reduce_scatter4(as[0], as[1], as[2], as[3]);
    PSTAMP(4, z);
    PH_D(z);
    float vA, act, kc;
    const float h = pair_cell<GATE>(sm, z, c, vA, act, kc);
    PSTAMP(5, h);
    hb[cur ^ 1][hslot] = h;
    float pA = vA, pB = Sel4::pick(sm.m2, kc, Sel4::pick(sm.m1, act, h));
    if (PAIR_CONTIG) { pA = lane_permute(paddr_inv, pA); pB = lane_permute(paddr_inv, pB); }
    buf_store(pA, r_g, vo_g, (unsigned)t * (unsigned)(LG * 4));
    buf_store(pB, r_h, vo_h, (unsigned)t * (unsigned)(LH * 4));
    buf_store(pB, r_a, vo_a, (unsigned)t * (unsigned)(2 * LH * 4));
    if constexpr (HASXP) {              // behind the stores: see the encoder
      if (PAIR_ABL == 7) XProj<XL>::pin(xa);
      xp.load(xa, na, t + 2);
      if constexpr (XL) na = xp.notes(t + 4);
    }
    PSTAMP(6, vA + h);
    PARRIVE(PNW + wave, t, vA + h);
    PH_C(vA + h);
    step_barrier();
    PH_ACC();
#ifdef PAIR_STAMPS
    if (blockIdx.x == 0 && wave == 0 && lane == 0 && t >= 32 && t < 40) {
      unsigned long long now;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now));
#pragma unroll
      for (int k = 0; k < 7; ++k) g_pair_stamps[t - 32][k] = pst[k];
      g_pair_stamps[t - 32][7] = now;
    }
#endif
  };
  for (int t = 0; t < T; t += 2) {     // an odd T runs one step more: see the encoder
    step(ParC<0>(), t, xA, nA);
    step(ParC<1>(), t + 1, xB, nB);
  }
  PH_OUT(PNW + wave);
}

template <int GATE, bool HASXP, int ZQ, bool XL>
__global__ __launch_bounds__(PNT) void lstm_pair_fwd_kernel(PairFwdArgs a) {
  __shared__ __attribute__((aligned(16))) float hbuf[2][2][PK * PKP];      // [chain][parity][sliced h (+ z in the decoder's)]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 2 * 2 * PK * PKP; i += PNT) (&hbuf[0][0][0])[i] = 0.f;
  if (a.noise.on) {
    // this row's eps_z, the values clv_philox_normal writes at the row's global indices (a launch of its own was
    // 5-6 us of a 450 us step); the latent lanes read them back from memory like given noise
    const int n = a.T * a.L;
    const uint32_t stp = a.noise.step + (a.noise.step_dev ? (uint32_t)*a.noise.step_dev : 0u);
    const uint64_t first = a.noise.first + (uint64_t)blockIdx.x * n;
    float* out = a.eps + (size_t)blockIdx.x * n;
    for (int e = tid; e < n; e += PNT) out[e] = philox_normal_at(first + e, a.noise.k0, a.noise.k1, a.noise.stream, stp);
  }
  __syncthreads();
  if (wave < PNW - 1) pair_fwd_encoder<GATE, false, XL>(a, wave, lane, hbuf[0], hbuf[1]);
  else if (wave == PNW - 1) pair_fwd_encoder<GATE, true, XL>(a, wave, lane, hbuf[0], hbuf[1]);
  else pair_fwd_decoder<GATE, HASXP, ZQ, XL>(a, wave - PNW, lane, hbuf[1]);
}

// ---------------------------------------------------------------------------
// backward: decoder BPTT, the latent head's backward and encoder BPTT in one launch
// ---------------------------------------------------------------------------
// Waves 0-5 run the decoder chain (lane = (unit group, column slice), 4 units x 22 gate columns of U per thread),
// waves 6-11 the encoder chain two steps behind.  Between them:
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
  const float* aux_d;             // [B*T,2,88] (kcarry, kc) of the forward pass
  const float* aux_e;
  float* gates_d;                 // in: (ki,kf,kg,ko) of the forward pass   out: dz
  float* gates_e;
  float* dzsum_d;                 // [B,352] sum_t dz
  float* dzsum_e;
  const float* zargs;             // [B*T,2L]
  const float* eps;               // [B*T,L]
  float* dzargs;                  // [B*T,2L]
  // WZG: the latent head's weight gradient [hs_enc | 1]^T . dzargs, accumulated per batch row while the chains run
  // (a GEMM of its own re-read hs_enc and cost a 10 us launch): hs_e [B*T,88] in, wz_slab [B][89][2L] out
  const float* hs_e;
  float* wz_slab;
  // the label path's backward of the same batch row as this workgroup's epilogue (label_bwd_row.h): lab_on != 0
  LabelBwdArgs lab;
  int lab_on;
};

// LATW: the decoder chain's last wave: its surplus unit groups hold rows of Kz and finish dzargs
// WZG: also accumulate this batch row's share of the latent head's kernel / bias gradient (PairBwdArgs::wz_slab)
template <int GATE, bool DEC, bool LATW, int ZP, bool WZG>
__device__ __forceinline__ void pair_bwd_chain(const PairBwdArgs& a, int wave, int lane, float (*dzb)[BW_LDS],
                                               float (*dza)[QZ]) {
  const int cs = lane & 15, ug = wave * 4 + (lane >> 4), q = cs >> 2;
  const int b = blockIdx.x, T = a.T, L = a.L;
  const int u = min(4 * ug + (cs & 3), LH - 1);      // surplus groups without a latent duplicate unit 87
  const int zg0 = 4 * (ug - 22);                     // first latent of a surplus group
  const bool zgroup = LATW && ug >= 22 && zg0 < L;
  const int lat = zg0 + (cs & 3);                    // latent this lane finishes after the reduce-scatter
  const bool zlane = zgroup && lat < L;
  const bool zlive = zlane && q < 2;                 // replica 0: the mean column of dzargs, replica 1: the log_var column
  const unsigned long long m_q3 = lane_mask(q == 3);
  const unsigned long long m_q0 = lane_mask(q == 0);
  float* gates = DEC ? a.gates_d : a.gates_e;
  const float* aux = DEC ? a.aux_d : a.aux_e;
  float* dzsum = DEC ? a.dzsum_d : a.dzsum_e;

  f2 Ur[BW_CW][2];    // [column][acc pair]: acc j = unit j ^ (cs & 3) of the group; a latent group holds rows of Kz instead
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

  // What a lane needs of step t: k of its gate, kc, kcarry and (decoder) the upstream dh: (uniform row pointer) +
  // (32-bit lane offset) each.  A latent lane needs (mean, log_var, eps) of its latent ONE STEP LATER in time: the
  // values it holds during iteration (step t) are those of step t+1, whose dZ its matvec has just produced from
  // dz_dec_{t+1}.
  struct Raw { float kq, kc, kcarry, dh, m, lv, e, hh; };
  const unsigned col = q * LH + u;
  const rsrc_t r_g = make_rsrc(gates + rowbt * LG, T * LG * 4);              // k in, dz out
  const rsrc_t r_a = make_rsrc(aux + rowbt * 2 * LH, T * 2 * LH * 4);
  const rsrc_t r_d = make_rsrc(a.dhs_d + rowbt * LH, T * LH * 4);
  const rsrc_t r_za = make_rsrc(a.zargs + rowbt * 2 * L, T * 2 * L * 4);
  const rsrc_t r_e = make_rsrc(a.eps + rowbt * L, T * L * 4);
  const rsrc_t r_dz = make_rsrc(a.dzargs + rowbt * 2 * L, T * 2 * L * 4);
  const rsrc_t r_hs = make_rsrc((WZG && !DEC) ? a.hs_e + rowbt * LH : a.zargs, (WZG && !DEC) ? T * LH * 4 : 4);
  constexpr int WZC = ZP / 4;              // head columns per replica lane: replica q takes columns q*WZC ..
  float wz[WZC];
#pragma unroll
  for (int j = 0; j < WZC; ++j) wz[j] = 0.f;
  float zb = 0.f;                          // latent lanes (WZG): sum_t dzargs of their column = the head's bias gradient
  const unsigned vo_g = zgroup ? BUF_OOB : col * 4;                            // dz store (latent groups: none)
  const unsigned vo_m = zlane ? lat * 4 : BUF_OOB, vo_lv = zlane ? (L + lat) * 4 : BUF_OOB;
  const float hk = 0.5f * a.kl_scale;
  auto load_raw = [&](int tr, Raw& r) {      // tr < 0 (the steps beyond the window's start): beyond num_records, i.e. zeros
    const unsigned t = (unsigned)tr;
    r.kq = buf_load(r_g, col * 4, t * (unsigned)(LG * 4));
    r.kcarry = buf_load(r_a, u * 4, t * (unsigned)(2 * LH * 4));
    r.kc = buf_load(r_a, (LH + u) * 4, t * (unsigned)(2 * LH * 4));
    if (DEC) r.dh = buf_load(r_d, u * 4, t * (unsigned)(LH * 4));
    if (WZG && !DEC) r.hh = buf_load(r_hs, u * 4, t * (unsigned)(LH * 4));
    if (LATW) {                    // lanes without a latent read beyond num_records: 0
      const unsigned tz = (unsigned)min(max(tr + 1, 0), T - 1);
      r.m = buf_load(r_za, vo_m, tz * (unsigned)(2 * L * 4));
      r.lv = buf_load(r_za, vo_lv, tz * (unsigned)(2 * L * 4));
      r.e = buf_load(r_e, vo_m, tz * (unsigned)(L * 4));
    }
  };
  // Two register sets, the loop body is two steps: the values of step t are requested at the end of iteration t+2
  // (behind the last use of the set) and consumed where they landed
  Raw rA, rB;
  rA.m = rA.lv = rA.e = rB.m = rB.lv = rB.e = 0.f;
  rA.dh = rB.dh = rA.hh = rB.hh = 0.f;
  load_raw(T - 1, rA);
  load_raw(T - 2, rB);
  prologue_loads_done();

  // LDS slot: regular lanes: dz of gate q (4 replicas share the 4 gates); latent lanes: their dzargs column
  const int lpos = zgroup ? q * L + lat : BW_CP * ((int)col / BW_CW) + (int)col % BW_CW;
  const unsigned vo_dz = zlive ? (unsigned)lpos * 4 : BUF_OOB;

  auto matvec = [&](const float* dzs) {        // reduce-scattered: this lane's unit (or latent) total
    const float4* dp = reinterpret_cast<const float4*>(dzs + BW_CP * cs);
    float dv[BW_CP];
#pragma unroll
    for (int j = 0; j < BW_CP / 4; ++j) {
      float4 v;
      if (PAIR_ABL == 2) { asm volatile("; no read" : "=v"(v.x), "=v"(v.y), "=v"(v.z), "=v"(v.w)); }
      else if (PAIR_ABL == 1 && j >= 3) { v = make_float4(dv[4 * (j - 3)], dv[4 * (j - 3) + 1], dv[4 * (j - 3) + 2], dv[4 * (j - 3) + 3]); }
      else v = dp[j];
      dv[4 * j] = v.x; dv[4 * j + 1] = v.y; dv[4 * j + 2] = v.z; dv[4 * j + 3] = v.w;
    }
    f2 acc01 = {0.f, 0.f}, acc23 = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < BW_CW; ++c) {
      const f2 dd = {dv[c], dv[c]};
      acc01 = __builtin_elementwise_fma(dd, Ur[c][0], acc01);
      acc23 = __builtin_elementwise_fma(dd, Ur[c][1], acc23);
    }
    const float r0 = acc01[0] + dpp_mov<0xB1>(acc01[1]);      // partner m^1: its accumulator 1 is unit m
    const float r2 = acc23[0] + dpp_mov<0xB1>(acc23[1]);      //              its accumulator 3 is unit m^2
    float x = r0 + dpp_mov<0x4E>(r2);                         // partner m^2: its r2 is unit m
    x = dpp_add<0x124>(x);                                    // the four quads of the row (16 column slices)
    x = dpp_add<0x128>(x);
    return x;
  };
  auto latent_dz = [&](const Raw& k, float dZ) {    // mean column (replica 0): dZ + kl*mean; log_var column: dZ*eps*sd/2 - kl*(1 - sd^2)/2
    const float sd = __expf(0.5f * k.lv);
    const float zi_ = Sel4::pick(m_q0, 1.f, 0.5f * k.e * sd);
    const float zf_ = Sel4::pick(m_q0, a.kl_scale * k.m, -hk * (1.f - sd * sd));
    return fmaf(dZ, zi_, zf_);
  };

  if (!DEC) {                        // the encoder chain runs two iterations behind: dzargs_t leaves the decoder
    step_barrier();                  // chain at the end of the iteration after dz_dec_t
    step_barrier();
  }

  auto step = [&](auto parc, int i, Raw& k) {
    constexpr int cur = decltype(parc)::value;
    const int t = T - 1 - i;
    float dhup;
    if (DEC) {
      dhup = k.dh;
    } else {                          // dh_enc_t = dzargs_t . Wz[u,:]  (dzargs_t was written one iteration ago)
      float s0 = 0.f, s1 = 0.f;       // by decoder iteration i + 1: parity cur ^ 1
#pragma unroll
      for (int j4 = 0; j4 < ZP; j4 += 4) {           // columns beyond 2L: zero weights, zero LDS
        const float4 v = *reinterpret_cast<const float4*>(&dza[cur ^ 1][j4]);
        s0 = fmaf(v.x, Wzr[j4], s0); s1 = fmaf(v.y, Wzr[j4 + 1], s1);
        s0 = fmaf(v.z, Wzr[j4 + 2], s0); s1 = fmaf(v.w, Wzr[j4 + 3], s1);
      }
      dhup = s0 + s1;
      if (WZG) {                      // dWz[u][q*WZC + j] += h_enc_t[u] * dzargs_t[q*WZC + j]  (h of a step beyond the window: 0)
        const float* dzc = &dza[cur ^ 1][q * WZC];
#pragma unroll
        for (int j = 0; j < WZC; ++j) wz[j] = fmaf(k.hh, dzc[j], wz[j]);
      }
    }
    const float dhrec = matvec(dzb[cur]);
    const float dh = dhup + dhrec;
    dc = fmaf(dh, k.kc, dc);
    const float val = Sel4::pick(m_q3, dh, dc) * k.kq;      // dz_o = dh ko; dz_{i,f,g} = dc k
    dc = dc * k.kcarry;
    zsum += val;
    if (!zgroup) dzb[cur ^ 1][lpos] = val;
    buf_store(val, r_g, vo_g, (unsigned)t * (unsigned)(LG * 4));
    if (PAIR_ABL == 10) {             // timing ablation: what writing dz as three bf16 pieces (for the kernel-gradient product) would add --
      const __bf16 p0 = (__bf16)val;  // the split and three 2-byte stores (into the aux rows of this step: already consumed)
      const float r1 = val - (float)p0;
      const __bf16 p1 = (__bf16)r1;
      const __bf16 p2 = (__bf16)(r1 - (float)p1);
      const unsigned vo_p = zgroup ? BUF_OOB : col * 2;
      __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, p0), r_a, (int)vo_p, (int)((unsigned)t * (unsigned)(2 * LH * 4)), 0);
      __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, p1), r_a, (int)vo_p, (int)((unsigned)t * (unsigned)(2 * LH * 4)), 0);
      __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, p2), r_a, (int)vo_p, (int)((unsigned)t * (unsigned)(2 * LH * 4)), 0);
    }
    if (LATW) {                       // latent lanes turn dZ_{t+1} into dzargs_{t+1}
      const float zv = latent_dz(k, dhrec);
      if (zlive) dza[cur][lpos] = zv;
      if (WZG) zb = fmaf(zv, (i >= 1 && i < T) ? 1.f : 0.f, zb);      // i == 0: no step T; i == T (odd T): the epilogue's step 0
      buf_store(zv, r_dz, vo_dz, (unsigned)min(t + 1, T - 1) * (unsigned)(2 * L * 4));      // i == 0: garbage into row T-1, rewritten at i == 1
    }
    load_raw(t - 2, k);
    step_barrier();
  };
  // An odd T runs one step more (no separate tail, see the forward kernel): step t = -1 has zero coefficients (its
  // loads lie beyond num_records), so dz = 0 and the column sums are untouched, its stores are dropped, and in the latent
  // lanes it IS the iteration-T work below (dZ_0 -> dzargs_0), which is then done twice with the same result.
  for (int i = 0; i < T; i += 2) {
    step(ParC<0>(), i, rA);
    step(ParC<1>(), i + 1, rB);
  }
  if (DEC) {
    // iteration T: dZ_0 -> dzargs_0
    if (LATW) {
      const float dZ = matvec(dzb[T & 1]);
      Raw k0;                                                  // a latent lane's values of step 0
      k0.m = buf_load(r_za, vo_m, 0); k0.lv = buf_load(r_za, vo_lv, 0); k0.e = buf_load(r_e, vo_m, 0);
      const float zv = latent_dz(k0, dZ);
      buf_store(zv, r_dz, vo_dz, 0);
      if (zlive) dza[T & 1][lpos] = zv;
      if (WZG && zlive) a.wz_slab[((size_t)b * (LH + 1) + LH) * 2 * L + lpos] = zb + zv;
    }
    step_barrier();
    step_barrier();
  }
  if (!zgroup) dzsum[(size_t)b * LG + col] = zsum;
  if (WZG && !DEC && 4 * ug + (cs & 3) < LH) {
#pragma unroll
    for (int j = 0; j < WZC; ++j)
      if (q * WZC + j < 2 * L) a.wz_slab[((size_t)b * (LH + 1) + u) * 2 * L + q * WZC + j] = wz[j];
  }
}

template <int GATE, int ZP, bool WZG>
__global__ __launch_bounds__(PNT) void lstm_pair_bwd_kernel(PairBwdArgs a) {
  __shared__ __attribute__((aligned(16))) float dzbuf[2][2][BW_LDS];       // [chain][parity][sliced dz]
  __shared__ __attribute__((aligned(16))) float dza[2][QZ];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 2 * 2 * BW_LDS; i += PNT) (&dzbuf[0][0][0])[i] = 0.f;
  if (tid < 2 * QZ) (&dza[0][0])[tid] = 0.f;
  __syncthreads();
  if (wave < PNW - 1) pair_bwd_chain<GATE, true, false, ZP, WZG>(a, wave, lane, dzbuf[0], dza);
  else if (wave == PNW - 1) pair_bwd_chain<GATE, true, true, ZP, WZG>(a, wave, lane, dzbuf[0], dza);
  else pair_bwd_chain<GATE, false, false, ZP, WZG>(a, wave - PNW, lane, dzbuf[1], dza);
  if (a.lab_on) {            // uniform: row b's sum_t dz of both chains is complete in HBM / L2 after the barrier
    __syncthreads();
    label_bwd_row<PNT>(a.lab, (int)blockIdx.x, tid);
  }
}

}  // namespace clv

#ifdef PAIR_PHASES
extern "C" int clv_debug_pair_phases(unsigned* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(clv::g_pair_phase), sizeof(unsigned) * 48);
}
#endif
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
                                 const float* pack, const float* bz, float* eps,
                                 float* hs_enc, float* aux_enc, float* hs_dec, float* aux_dec,
                                 float* zargs, float* Z, int ldz, float* klterm,
                                 const unsigned char* notes_enc, const float* Kx_enc,
                                 const unsigned char* notes_dec, const float* Kx_dec,
                                 const clv_noise_draw* noise, void* stream) {
  using namespace clv;
  if (!clv_lstm_pair_supported(H, L) || B <= 0 || T <= 0 || ldz < L) return CLV_EINVAL;
  const bool xl = notes_enc != nullptr;
  if (xl && (!Kx_enc || (dec_has_xproj && (!notes_dec || !Kx_dec)) || ((uintptr_t)notes_enc | (uintptr_t)notes_dec) % 8))
    return CLV_EINVAL;
  if (gate_act != CLV_GATE_HARD_SIGMOID && gate_act != CLV_GATE_SIGMOID) return CLV_EINVAL;
  if (!gates_enc || !rowbias_enc || !gates_dec || !rowbias_dec || !pack || !bz || !eps ||
      !hs_enc || !aux_enc || !hs_dec || !aux_dec || !zargs || !Z || !klterm)
    return CLV_EINVAL;
  PairFwdArgs a{B, T, L, ldz, gates_enc, rowbias_enc, gates_dec, rowbias_dec, pack, bz,
                notes_enc, Kx_enc, notes_dec, Kx_dec, eps, {0, 0, 0, 0, 0, 0, nullptr},
                hs_enc, aux_enc, gates_enc, hs_dec, aux_dec, gates_dec, zargs, Z, klterm};
  if (noise) {
    a.noise.on = 1; a.noise.k0 = (uint32_t)noise->seed; a.noise.k1 = (uint32_t)(noise->seed >> 32);
    a.noise.stream = noise->stream; a.noise.step = noise->step; a.noise.first = noise->first;
    a.noise.step_dev = noise->step_dev;
  }
  hipStream_t s = (hipStream_t)stream;
  ProfScope p("lstm_pair_fwd", s);
  const bool hard = gate_act == CLV_GATE_HARD_SIGMOID;
#define PAIR_FWD_ZX(G, X, Z, XL) hipLaunchKernelGGL((lstm_pair_fwd_kernel<G, X, Z, XL>), dim3(B), dim3(PNT), 0, s, a)
#define PAIR_FWD_Z(G, X, Z) do { if (xl) PAIR_FWD_ZX(G, X, Z, true); else PAIR_FWD_ZX(G, X, Z, false); } while (0)
#define PAIR_FWD(G, X) do { if (L <= 4) PAIR_FWD_Z(G, X, 1); else PAIR_FWD_Z(G, X, 2); } while (0)
  if (hard) { if (dec_has_xproj) PAIR_FWD(CLV_GATE_HARD_SIGMOID, true); else PAIR_FWD(CLV_GATE_HARD_SIGMOID, false); }
  else { if (dec_has_xproj) PAIR_FWD(CLV_GATE_SIGMOID, true); else PAIR_FWD(CLV_GATE_SIGMOID, false); }
#undef PAIR_FWD_ZX
#undef PAIR_FWD_Z
#undef PAIR_FWD
  return launch_status();
}

extern "C" int clv_lstm_pair_bwd(int B, int T, int H, int L, int gate_act, float kl_scale,
                                    const float* pack, const float* Wz,
                                    const float* dhs_dec, const float* aux_dec, const float* aux_enc,
                                    float* gates_dec_inout_dz, float* gates_enc_inout_dz,
                                    float* dzsum_dec, float* dzsum_enc,
                                    const float* zargs, const float* eps, float* dzargs,
                                    const float* hs_enc, float* dWz, float* dbz, void* ws, size_t ws_bytes, clv_reduce_job* job,
                                    const clv_label_bwd_rider* label, void* stream) {
  using namespace clv;
  if (job) memset(job, 0, sizeof(*job));
  if (label && label->job) memset(label->job, 0, sizeof(*label->job));
  if (label) {
    const clv_label_bwd_rider& r = *label;
    if (r.D <= 0 || r.D > 128 || r.C < 2 || r.C > LH_MAXC) return CLV_EINVAL;
    if (!r.Kenc_w || !r.Kdec_w || !r.wargs || !r.eps || !r.onehot || !r.W || !r.hW || !r.Ka || !r.dwargs || !r.dhW)
      return CLV_EINVAL;
    if (r.dKa && (!r.dba || !r.ws || r.ws_bytes < (size_t)B * (r.D + 1) * 2 * (r.C - 1) * sizeof(float))) return CLV_EWORKSPACE;
  }
  if (!clv_lstm_pair_supported(H, L) || B <= 0 || T <= 0) return CLV_EINVAL;
  const bool wzg = hs_enc != nullptr;
  if (wzg && (!dWz || !dbz || !ws || ws_bytes < clv_lstm_pair_bwd_workspace_bytes(B, H, L))) return CLV_EWORKSPACE;
  if (gate_act != CLV_GATE_HARD_SIGMOID && gate_act != CLV_GATE_SIGMOID) return CLV_EINVAL;
  if (!pack || !Wz || !dhs_dec || !aux_dec || !aux_enc || !gates_dec_inout_dz || !gates_enc_inout_dz ||
      !dzsum_dec || !dzsum_enc || !zargs || !eps || !dzargs)
    return CLV_EINVAL;
  PairBwdArgs a{B, T, L, kl_scale, pack, Wz, dhs_dec, aux_dec, aux_enc, gates_dec_inout_dz,
                gates_enc_inout_dz, dzsum_dec, dzsum_enc, zargs, eps, dzargs, hs_enc, (float*)ws, {}, 0};
  if (label) {
    const clv_label_bwd_rider& r = *label;
    a.lab = LabelBwdArgs{B, r.D, r.C, LG, dzsum_enc, dzsum_dec, r.Kenc_w, r.Kdec_w, r.wargs, r.eps, r.onehot, r.W, r.hW, r.Ka,
                         r.prior_logvar, r.class_weight, r.w_kl_weight, r.inv_b, r.dwargs, r.dhW,
                         r.dKa ? (float*)r.ws : nullptr};
    a.lab_on = 1;
  }
  hipStream_t s = (hipStream_t)stream;
  ProfScope p("lstm_pair_bwd", s);
  const bool hard = gate_act == CLV_GATE_HARD_SIGMOID;
#define PAIR_BWD_W(G, Z, W) hipLaunchKernelGGL((lstm_pair_bwd_kernel<G, Z, W>), dim3(B), dim3(PNT), 0, s, a)
#define PAIR_BWD(G, Z) do { if (wzg) PAIR_BWD_W(G, Z, true); else PAIR_BWD_W(G, Z, false); } while (0)
#define PAIR_BWD_Z(Z) do { if (hard) PAIR_BWD(CLV_GATE_HARD_SIGMOID, Z); else PAIR_BWD(CLV_GATE_SIGMOID, Z); } while (0)
  if (2 * L <= 4) PAIR_BWD_Z(4);
  else if (2 * L <= 8) PAIR_BWD_Z(8);
  else PAIR_BWD_Z(16);
#undef PAIR_BWD_Z
#undef PAIR_BWD
#undef PAIR_BWD_W
  int st = launch_status();
  if (st) return st;
  if (label && label->dKa) {       // the Wargs layer's per-row slabs, exactly as clv_vrnn_label_bwd leaves them
    ReduceJob jl;
    memset(&jl, 0, sizeof(jl));
    const int NA = 2 * (label->C - 1);
    jl.partial = (const float*)label->ws;
    jl.M = label->D + 1; jl.N = NA; jl.splits = B; jl.nprob = 2;
    jl.alpha = 1.f; jl.beta = 0.f; jl.act = CLV_ACT_NONE;
    jl.prob[0] = ReduceProb{label->dKa, NA, 0};
    jl.prob[1] = ReduceProb{label->dba, NA, label->D};
    if (label->job && B > 1) memcpy(label->job, &jl, sizeof(jl));
    else st = launch_reduce(jl, s);
    if (st) return st;
  }
  if (!wzg) return st;
  // the per-row slabs [B][89][2L] -> dWz [88,2L] and dbz [2L]: a pending reduction like a split-K product's
  ReduceJob j;
  memset(&j, 0, sizeof(j));
  j.partial = (const float*)ws;
  j.M = LH + 1; j.N = 2 * L; j.splits = B; j.nprob = 2;
  j.alpha = 1.f; j.beta = 0.f; j.act = CLV_ACT_NONE;
  j.prob[0] = ReduceProb{dWz, 2 * L, 0};
  j.prob[1] = ReduceProb{dbz, 2 * L, LH};
  if (job && B > 1) memcpy(job, &j, sizeof(j));
  else st = launch_reduce(j, s);
  return st;
}

extern "C" size_t clv_lstm_pair_bwd_workspace_bytes(int B, int H, int L) {
  return (size_t)B * (H + 1) * 2 * L * sizeof(float);
}
