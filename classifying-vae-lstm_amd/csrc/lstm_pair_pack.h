// lstm_pair_pack.h -- the lane-order weight pack of the pair LSTM kernels (lstm_pair.hip): layout constants and the
// element function, shared with the kernels that write the pack as a by-product (label_head.hip: the label forward kernel
// runs right before the pair forward and has idle threads; a launch of its own was 4.7 us of a 430 us step).
#pragma once
#include "lstm_common.h"

namespace clv {

constexpr int PNW = 6;                  // waves per chain (16 units each)
constexpr int PNT = 2 * PNW * 64;       // 768 threads
constexpr int PLMAX = 16;               // latent slots of the K_z pack region (the kernels carry latent_dim <= 8)
constexpr int PLQ = PLMAX / PK;         // latents per decoder lane (z_t . K_z is split over the k-slice lanes)

// backward layout: gate columns per slice, padded LDS slice stride.  28 floats: the 16 slices of a ds_read_b128 lane
// group start at banks 28*cs mod 64 = {0,28,56,20,48,12,40,4,32,60,24,52,16,44,8,36}, four banks each, all distinct
// (a stride of 24 puts slices cs and cs+8 on the same banks: every read of the step was a 2-way conflict)
constexpr int BW_CW = 22, BW_CP = 28;
constexpr int BW_LDS = 16 * BW_CP;

// ---------------------------------------------------------------------------
// Lane layout (round 3).  Forward: a unit's 4 lanes (s = lane & 3) each hold one k-slice of the recurrent kernel (22 k
// values x 4 gates) -- but lane s keeps the gates in the order (s, s^1, s^2, s^3): accumulator j collects gate j ^ s.
// The sum over the k-slices is then a REDUCE-SCATTER of three v_add_dpp (no selects: the partner's accumulator 1 / 3 is
// exactly the gate this lane keeps in 0 / 2), after which lane s holds the pre-activation of gate s only.  It applies
// ITS activation once (lane 2: tanh, the others: the gate function), and the cell update reads the other three gates as
// DPP quad-broadcast operands.  The backward pass needs per (unit, step) only six numbers that are products of those
// activations and their derivatives:
//     ki = g i'   kf = c_{t-1} f'   kg = i g'   ko = tanh(c) o'   kc = o (1 - tanh(c)^2)   kcarry = f
// (dz_i = dc ki, dz_f = dc kf, dz_g = dc kg, dz_o = dh ko, dc += dh kc, dc_{t-1} = dc kcarry), and lane s of the forward
// pass can form "its" k from its own derivative and one neighbour value: so the forward pass stores (ki, kf, kg, ko) in
// the gate buffer and (kcarry, kc) in the aux buffer, and the backward pass is four loads and six instructions per lane
// and step where it used to rebuild three gate functions, a tanh and their derivatives from seven loaded values.
// Backward: thread = 4 units x 22 gate columns as before, accumulator j = unit j ^ (cs & 3) of the lane's group: the
// same select-free reduce-scatter.
// ---------------------------------------------------------------------------
// Weights in lane order.  A workgroup needs every recurrent weight exactly once, one value per lane: read straight from
// the [88,352] kernels that is 88 dword loads per lane whose 64 lanes touch four 64-byte pieces of four different rows
// (1056 such wave loads per workgroup, every workgroup at the same time: ~13 us of each launch at config 3).  The pack
// kernel writes each lane's values as consecutive float4 (one wave load = 1 KB contiguous), once per step, for both
// passes: regions of [6 waves][n][64 lanes] float4.
//   PK_FE / PK_FD: forward encoder / decoder, n = 22: float4 kk = (k0, k1 | acc ja), (k0, k1 | acc jb), k0 = 22 s +
//                  2 (kk / 2), (ja, jb) = (0, 1) for even kk, (2, 3) for odd kk, acc j = gate j ^ s; the encoder's
//                  latent lanes carry columns of the head kernel Wz instead (see pair_fwd_encoder)
//   PK_KZ:         forward decoder, n = 4: latent s + 4q of the decoder input kernel's z rows, components = acc 0..3
//   PK_BD / PK_BE: backward decoder / encoder, n = 22 gate columns of the lane's slice, components = acc 0..3 = units
//                  j ^ (cs & 3) of the lane's group; the decoder's surplus groups carry rows of Kz (see pair_bwd_chain)
// ---------------------------------------------------------------------------
constexpr int PK_N = PNW * PKK * 64;               // float4 per 22-deep region
constexpr int PK_FE = 0, PK_FD = PK_N, PK_KZ = 2 * PK_N, PK_BD = PK_KZ + PNW * PLQ * 64, PK_BE = PK_BD + PK_N;
constexpr int PK_TOTAL = PK_BE + PK_N;             // float4

struct PairPackArgs { int L; const float* U_e; const float* U_d; const float* Kz; const float* Wz; float4* out; };

// element i of the pack from the weights as they are in the flat parameter buffer (clv_lstm_pair_pack's kernel, and the
// label forward kernel, which writes the pack as a by-product: label_head.hip)
template <class Load>
__device__ __forceinline__ float4 pair_pack_element(int i, int L, const float* U_e, const float* U_d, const float* Kz,
                                                    const float* Wz, Load ld) {
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (i < PK_KZ) {                                 // forward: lane = (unit, k-slice), n = kk
    const bool dec = i >= PK_FD;
    const int e = i - (dec ? PK_FD : PK_FE);
    const int wave = e / (PKK * 64), kk = (e / 64) % PKK, lane = e & 63;
    const int s = lane & 3, u_raw = wave * 16 + (lane >> 2), u = min(u_raw, LH - 1), zj = u_raw - LH;
    const bool is_z = !dec && zj >= 0 && 2 * zj < L;
    const float* U = dec ? U_d : U_e;
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) {
      const int j = 2 * (kk & 1) + (e2 >> 1), g = j ^ s, k = PKK * s + 2 * (kk >> 1) + (e2 & 1);
      if (is_z) {                                  // head column g of the group: (mean_2j, mean_2j+1, log_var_2j, log_var_2j+1)
        const int l = 2 * zj + (g & 1);
        v[e2] = l < L ? ld(Wz + (size_t)k * 2 * L + (g >> 1) * L + l) : 0.f;
      } else {
        v[e2] = ld(U + (size_t)k * LG + g * LH + u);
      }
    }
  } else if (i < PK_BD) {                          // z rows of the decoder input kernel: latent s + 4q
    const int e = i - PK_KZ;
    const int wave = e / (PLQ * 64), q = (e / 64) % PLQ, lane = e & 63;
    const int s = lane & 3, u = min(wave * 16 + (lane >> 2), LH - 1), l = s + PK * q;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = l < L ? ld(Kz + (size_t)l * LG + (j ^ s) * LH + u) : 0.f;
  } else {                                         // backward: lane = (unit group, column slice), n = column in the slice
    const bool dec = i < PK_BE;
    const int e = i - (dec ? PK_BD : PK_BE);
    const int wave = e / (BW_CW * 64), c = (e / 64) % BW_CW, lane = e & 63;
    const int cs = lane & 15, m = cs & 3, ug = wave * 4 + (lane >> 4), zg0 = 4 * (ug - 22);
    const bool zgroup = dec && ug >= 22 && zg0 < L;
    const float* U = dec ? U_d : U_e;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int lj = zg0 + (j ^ m);
      if (zgroup) v[j] = lj < L ? ld(Kz + (size_t)lj * LG + BW_CW * cs + c) : 0.f;
      else v[j] = ld(U + (size_t)min(4 * ug + (j ^ m), LH - 1) * LG + BW_CW * cs + c);
    }
  }
  return make_float4(v[0], v[1], v[2], v[3]);
}

}  // namespace clv
