// lstm_common.h -- helpers shared by the LSTM sequence kernels (lstm.hip, lstm_pair.hip)
#pragma once
#include "common.h"

namespace clv {

constexpr int LH = 88;          // hidden units
constexpr int LG = 4 * LH;      // gate columns

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
template <int GATE>
__device__ __forceinline__ float gate_grad(float z, float y) {
  return GATE == CLV_GATE_HARD_SIGMOID ? hard_sigmoid_grad(z) : y * (1.f - y);
}

}  // namespace clv
