// philox.h -- Philox4x32-10 counter RNG, shared by the stand-alone draws (runtime.hip) and the kernels that
// draw in place (generate.hip).  Element i of a stream uses counter (i >> 2) and word / Box-Muller branch (i & 3):
// a value is a pure function of (seed, step, stream, index).
#pragma once
#include <stdint.h>

#include "common.h"

namespace clv {

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }


// the i-th normal / uniform of stream `stream_id` at step `st` (same values as philox_kernel writes)
__device__ __forceinline__ float philox_normal_at(uint64_t i, uint32_t k0, uint32_t k1, uint32_t stream_id, uint32_t st) {
  const uint64_t ctr = i >> 2;
  uint32_t r[4];
  philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), stream_id, st, k0, k1, r);
  const int j = (int)(i & 3);
  const float rad = sqrtf(-2.f * logf(u01(j < 2 ? r[0] : r[2])));
  const float th = 6.283185307179586f * u01(j < 2 ? r[1] : r[3]);
  return (j & 1) ? rad * sinf(th) : rad * cosf(th);
}
__device__ __forceinline__ float philox_uniform_at(uint64_t i, uint32_t k0, uint32_t k1, uint32_t stream_id, uint32_t st) {
  const uint64_t ctr = i >> 2;
  uint32_t r[4];
  philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), stream_id, st, k0, k1, r);
  const int j = (int)(i & 3);
  return u01(j == 0 ? r[0] : j == 1 ? r[1] : j == 2 ? r[2] : r[3]);
}

}  // namespace clv
