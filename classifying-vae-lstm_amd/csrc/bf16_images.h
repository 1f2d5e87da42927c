// bf16_images.h -- LDS images of bf16 pieces in [k][column] order and their MFMA fragments (gfx950).
// Shared by the kernels that form X^T.G products on v_mfma_f32_16x16x32_bf16 from k-major operands (outer_bf16.hip;
// wgrad_bf16.hip keeps its own copies with its measurement hooks).
#pragma once
#include "common.h"

namespace clv {

typedef __bf16 img_bf16x8 __attribute__((ext_vector_type(8)));
typedef short img_s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int img_u32x2 __attribute__((ext_vector_type(2)));

// 4 consecutive columns of one image row: one 8-byte LDS store per piece image (NP = 1: the values are exact in bf16)
template <int NP>
__device__ __forceinline__ void img_put4(char* at, int piece_bytes, const float4& v) {
  unsigned lo[3], hi[3];
  if (NP == 1) { lo[0] = bf16_pack2(v.x, v.y); hi[0] = bf16_pack2(v.z, v.w); }
  else { bf16_split_pair(v.x, v.y, lo); bf16_split_pair(v.z, v.w, hi); }
#pragma unroll
  for (int p = 0; p < NP; ++p) *reinterpret_cast<img_u32x2*>(at + p * piece_bytes) = img_u32x2{lo[p], hi[p]};
}

// the 8 k-values x 16 columns fragment of a bf16 [k][column] image (k rows 0..31 of the stage, columns col0..col0+15):
// lane l = 16 g + i gets column col0 + i, k = 8 g + j in element j -- the layout of both operands of
// v_mfma_f32_16x16x32_bf16 when the image holds the operand k-major.  `lane_off` = img_frag_lane_offset(pitch, lane).
// A transposed read takes 4 rows x 32 bytes per 16-lane group; with a pitch of 48 banks (192 bytes = 96 columns) the 4 rows of
// a group fall on different banks.
__device__ __forceinline__ int img_frag_lane_offset(int pitch, int lane) {
  const int g = lane >> 4, i = lane & 15;
  return (8 * g + (i >> 2)) * pitch + 8 * (i & 3);
}
__device__ __forceinline__ img_bf16x8 img_frag(const char* img, int pitch, int col0, int lane_off) {
  const char* p = img + lane_off + 2 * col0;
  typedef __attribute__((address_space(3))) img_s16x4 lds_s16x4;
  const img_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  const img_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * pitch));
  return __builtin_bit_cast(img_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

}  // namespace clv
