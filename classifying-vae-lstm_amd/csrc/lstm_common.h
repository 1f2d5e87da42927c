// lstm_common.h -- helpers shared by the LSTM sequence kernels (lstm.hip, lstm_pair.hip)
#pragma once
#include "common.h"

namespace clv {

constexpr int LH = 88;          // hidden units
constexpr int LG = 4 * LH;      // gate columns

// lstm_any.hip: the same contract for any number of hidden units (1..1024); clv_lstm_seq_fwd / _bwd dispatch here for H != 88
int launch_lstm_any_fwd(int B, int T, int H, int gate_act, const float* xproj, const float* rowbias, const float* U,
                        const float* h0, const float* c0, float* hs, float* cs, float* gates, float* hT, float* cT,
                        hipStream_t s);
int launch_lstm_any_bwd(int B, int T, int H, int gate_act, const float* U, const float* dhs, const float* cs,
                        const float* c0, float* gates_inout_dz, float* dzsum, hipStream_t s);

typedef float f2 __attribute__((ext_vector_type(2)));   // register pair: v_pk_fma_f32 does two fp32 FMAs per issue slot

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true);
  return v + __builtin_bit_cast(float, t);
}
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over the KS consecutive lanes of a k-slice group; every lane gets the total
template <int KS>
__device__ __forceinline__ float reduce_slices(float v) {
  v = dpp_add<0xB1>(v);                  // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);                  // quad_perm [2,3,0,1]
  if (KS == 8) v = dpp_add<0x141>(v);    // row_half_mirror
  return v;
}

template <int GATE>
__device__ __forceinline__ float gate_fn(float z) {
  return GATE == CLV_GATE_HARD_SIGMOID ? hard_sigmoid(z) : sigmoidf_(z);
}
// y = gate_fn(z).  hard_sigmoid: the clip changes the value exactly when it is out of range, so "gradient passes"
// (inside the range, ties included, like TF's clip) == "clipped value equals the unclipped one": one compare against
// the fma the forward expression already holds instead of two range compares.
template <int GATE>
__device__ __forceinline__ float gate_grad(float z, float y) {
  return GATE == CLV_GATE_HARD_SIGMOID ? ((0.2f * z + 0.5f) == y ? 0.2f : 0.0f) : y * (1.f - y);
}

// ---- pieces shared by the kernels that keep a unit's 4 gate columns in 4 k-slice lanes (lstm_pair.hip, generate.hip)
constexpr int PK = 4;                   // k-slices per unit
constexpr int PKK = LH / PK;            // 22 k values per slice
constexpr int PKP = 24;                 // padded slice stride in LDS (16-byte aligned)

#ifndef PAIR_ABL
#define PAIR_ABL 0
#endif
// (PAIR_ABL == 3: timing ablation without the barrier itself)
__device__ __forceinline__ void step_barrier() {
  if (PAIR_ABL == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// A lane mask (ballot of a per-lane condition) kept in an SGPR pair, opaque to the optimiser: the empty asm holds no
// instruction, it only hides that the mask is `cond` per lane (see Sel4).
__device__ __forceinline__ unsigned long long lane_mask(bool cond) {
  unsigned long long m = __builtin_amdgcn_ballot_w64(cond);
  asm volatile("" : "+s"(m));
  return m;
}

// v[i] for a per-lane i in 0..3 as three v_cndmask_b32 on lane masks kept in SGPR pairs.  Written as a chain of selects
// on `i == const` the compiler recognises a dynamic index into a 4-element array and puts the array in LDS
// (ds_write_b128 + ds_read_b32 and a full lgkmcnt wait on the critical path of every step); selects on opaque masks cost
// no VGPR and no LDS.
// pick() is the compiler's own select on inverse_ballot(mask) -- since round 5; until then it was an inline-asm
// v_cndmask_b32_e64, and that was a BUG on gfx940/gfx950: a VALU instruction that reads the result of a transcendental
// one (v_rcp_f32, v_exp_f32) needs one wait state in between (LLVM's "trans forwarding hazard"); the compiler inserts
// the s_nop for instructions it emits but does not look inside an asm statement.  With SLP vectorisation on, the sigmoid
// instance of the pair forward kernel scheduled `v_rcp_f32 v97` directly in front of the asm select that reads v97 and
// the select saw the stale register (wrong states, logits off by 0.1: round 4's "-fno-slp-vectorize is a correctness
// requirement"); without SLP a v_fma happened to sit in between.  tools/probes/trans_hazard_asm.hip reproduces the
// pattern in 20 lines; tests/test_host_logic.py checks that no VALU inline asm is left in csrc/.
struct Sel4 {
  unsigned long long m1, m2, m3;
  __device__ __forceinline__ explicit Sel4(int i) : m1(lane_mask(i == 1)), m2(lane_mask(i == 2)), m3(lane_mask(i == 3)) {}
  __device__ __forceinline__ static float pick(unsigned long long m, float a, float b) {     // m ? a : b
    return __builtin_amdgcn_inverse_ballot_w64(m) ? a : b;
  }
  __device__ __forceinline__ float operator()(float v0, float v1, float v2, float v3) const {
    return pick(m3, v3, pick(m2, v2, pick(m1, v1, v0)));
  }
  __device__ __forceinline__ float operator()(const float (&v)[4]) const { return (*this)(v[0], v[1], v[2], v[3]); }
};

// h (LDS, sliced layout) . U slice -> the 4 gate sums of this lane's unit, reduced over the k-slices
__device__ __forceinline__ void slice_matvec(const float* hslice, const f2 (&Ur)[PKK][2], f2 (&acc2)[2]) {
  const float4* hp = reinterpret_cast<const float4*>(hslice);
  float hv[PKP];
#pragma unroll
  for (int q = 0; q < PKP / 4; ++q) {
    const float4 v = hp[q];
    hv[4 * q] = v.x; hv[4 * q + 1] = v.y; hv[4 * q + 2] = v.z; hv[4 * q + 3] = v.w;
  }
#pragma unroll
  for (int kk = 0; kk < PKK; ++kk) {
    const f2 hh = {hv[kk], hv[kk]};
    acc2[0] = __builtin_elementwise_fma(hh, Ur[kk][0], acc2[0]);
    acc2[1] = __builtin_elementwise_fma(hh, Ur[kk][1], acc2[1]);
  }
}

// the same product in pieces: the slice's h values, then the k range [K0, K1) with scalar FMAs
// -DPAIR_ABL=1 / 2 (timing ablations, wrong results): only the first half of a slice is read (the rest reuses it) / nothing is
// read at all -- what the 72 ds_read_b128 of a step cost on its critical path (profiles/r03_pair_ablation.txt)
#ifndef PAIR_ABL
#define PAIR_ABL 0
#endif
__device__ __forceinline__ void load_hslice(const float* hslice, float (&hv)[PKP]) {
  const float4* hp = reinterpret_cast<const float4*>(hslice);
#pragma unroll
  for (int q = 0; q < PKP / 4; ++q) {
    float4 v;
    if (PAIR_ABL == 2) { asm volatile("; no read" : "=v"(v.x), "=v"(v.y), "=v"(v.z), "=v"(v.w)); }
    else if (PAIR_ABL == 1 && q >= 3) { v = make_float4(hv[4 * (q - 3)], hv[4 * (q - 3) + 1], hv[4 * (q - 3) + 2], hv[4 * (q - 3) + 3]); }
    else v = hp[q];
    hv[4 * q] = v.x; hv[4 * q + 1] = v.y; hv[4 * q + 2] = v.z; hv[4 * q + 3] = v.w;
  }
}
template <int K0, int K1>
__device__ __forceinline__ void slice_fma_range(const float (&hv)[PKP], const f2 (&Ur)[PKK][2], float (&a)[4]) {
#pragma unroll
  for (int kk = K0; kk < K1; ++kk) {
    a[0] = fmaf(hv[kk], Ur[kk][0][0], a[0]);
    a[1] = fmaf(hv[kk], Ur[kk][0][1], a[1]);
    a[2] = fmaf(hv[kk], Ur[kk][1][0], a[2]);
    a[3] = fmaf(hv[kk], Ur[kk][1][1], a[3]);
  }
}

// Packed form without a broadcast: k values in pairs.  Up[j][g] = (U[2j][g], U[2j+1][g]) and a[g] collects (even k,
// odd k) partial sums, so every v_pk_fma_f32 operand is a natural register pair -- two adjacent h values of a
// ds_read_b128, two weights the pack kernel stored side by side -- and no operand borrows a foreign register as its
// unused half (the broadcast form does: a pending load's destination there makes the FMA block wait for the load).
template <int J0, int J1>
__device__ __forceinline__ void slice_fma_pairs(const float (&hv)[PKP], const f2 (&Up)[PKK / 2][4], f2 (&a)[4]) {
#pragma unroll
  for (int j = J0; j < J1; ++j) {
    const f2 hp = {hv[2 * j], hv[2 * j + 1]};
#pragma unroll
    for (int g = 0; g < 4; ++g) a[g] = __builtin_elementwise_fma(hp, Up[j][g], a[g]);
  }
}

// the same product with scalar FMAs: no register pairs, so no v_pk operand can alias the destination of a load in flight
__device__ __forceinline__ void slice_matvec_scalar(const float* hslice, const f2 (&Ur)[PKK][2], f2 (&acc2)[2]) {
  const float4* hp = reinterpret_cast<const float4*>(hslice);
  float hv[PKP];
#pragma unroll
  for (int q = 0; q < PKP / 4; ++q) {
    const float4 v = hp[q];
    hv[4 * q] = v.x; hv[4 * q + 1] = v.y; hv[4 * q + 2] = v.z; hv[4 * q + 3] = v.w;
  }
  float a0 = acc2[0][0], a1 = acc2[0][1], a2 = acc2[1][0], a3 = acc2[1][1];
#pragma unroll
  for (int kk = 0; kk < PKK; ++kk) {
    a0 = fmaf(hv[kk], Ur[kk][0][0], a0);
    a1 = fmaf(hv[kk], Ur[kk][0][1], a1);
    a2 = fmaf(hv[kk], Ur[kk][1][0], a2);
    a3 = fmaf(hv[kk], Ur[kk][1][1], a3);
  }
  acc2[0][0] = a0; acc2[0][1] = a1; acc2[1][0] = a2; acc2[1][1] = a3;
}

template <int GATE>
__device__ __forceinline__ void lstm_cell(const float (&z)[4], float& c, float& h, float& gg) {
  const float ig = gate_fn<GATE>(z[0]), fg = gate_fn<GATE>(z[1]), og = gate_fn<GATE>(z[3]);
  gg = fast_tanh(z[2]);
  c = fg * c + ig * gg;
  h = og * fast_tanh(c);
}

__device__ __forceinline__ float pick4(int i, const float (&v)[4]) {
  float r = v[0];
  r = i == 1 ? v[1] : r;
  r = i == 2 ? v[2] : r;
  r = i == 3 ? v[3] : r;
  return r;
}

}  // namespace clv
