// out_head_args.h -- launch arguments shared by the two output-head kernels (out_head.hip, out_head_bf16.hip)
#pragma once
#include "common.h"

namespace clv {

constexpr int OH = 88;                    // hidden units == output notes
constexpr int OH_SLAB_ROWS = OH + 1;      // dWo rows + the dbo row
constexpr int OH_RB = 128;                // rows per block (8 waves x one 16-row tile)

struct OutHeadArgs {
  int R, ldy;
  float scale;
  const float* hs;        // [R,88]
  const float* Wo;        // [88,88]
  const float* bo;        // [88]
  const void* Y;          // [R,ldy] targets: float, or uint8 frames (y_u8: ldy in bytes; out_head_bf16.hip only)
  int y_u8;
  float* logits;          // [R,88] or null
  float* rownll;          // [R]
  float* dlogits;         // [R,88] or null
  float* dhs;             // [R,88]
  float* partial;         // [gridDim.x][89][88]
};

// out_head_bf16.hip: the split-bf16 kernel; needs 16-byte aligned rows of Y / logits / dlogits / dhs (out_head_bf16_ok)
bool out_head_bf16_ok(const OutHeadArgs& a);
int launch_out_head_bf16(const OutHeadArgs& a, int wgs, hipStream_t s);

}  // namespace clv
