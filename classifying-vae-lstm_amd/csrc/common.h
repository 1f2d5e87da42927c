// common.h -- shared host/device helpers for libclvae_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/clvae.h"

#define CLV_WAVE 64

#define CLV_HIP_TRY(expr)                         \
  do {                                            \
    hipError_t e__ = (expr);                      \
    if (e__ != hipSuccess) return (int)e__;       \
  } while (0)

namespace clv {

// ---- opt-in profiler (clv_prof_*): events on the launch stream -------------
void prof_begin(const char* name, hipStream_t s);
void prof_end(hipStream_t s);
bool prof_on();

struct ProfScope {
  hipStream_t s;
  bool on;
  ProfScope(const char* name, hipStream_t st) : s(st), on(prof_on()) {
    if (on) prof_begin(name, s);
  }
  ~ProfScope() {
    if (on) prof_end(s);
  }
};

inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? CLV_OK : (int)e;
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// measurement knobs: `static const int x = env_int("NAME", dflt);` (a function-local static is initialised once, thread-safe)
inline int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

// hipFuncAttributeMaxDynamicSharedMemorySize for `kernel` on the CURRENT device, set once per (device, kernel): safe
// from several host threads and for a process that drives more than one GPU.  Returns a hipError_t / CLV_OK.
int allow_dynamic_lds(const void* kernel, int bytes);

// ---- Bernoulli NLL clip (cl_vae/model.py:190-191, cl_vrnn/model.py:241-242: keras.losses.binary_crossentropy) -------
// Keras clips p = sigmoid(a) to [eps, 1 - eps] and takes log(p / (1 - p)); on the logits that is a clip of a, gradient 0
// outside.  The comparand is the FLOAT32 Keras path: float32(1 - 1e-7) is 1 - 2^-23, so the upper end is
// log((1 - 2^-23) / 2^-23) = log(2^23 - 1) = 15.942385, while the lower end stays log(1e-7 / (1 - 1e-7)) = -16.118095 (as
// float32 evaluates it).  -DCLV_BCE_SYMMETRIC_CLIP builds the exact-arithmetic form (+-16.118095) instead.
#ifdef CLV_BCE_SYMMETRIC_CLIP
constexpr float BCE_CLIP_LO = -16.11809555f, BCE_CLIP_HI = 16.11809555f;
#else
constexpr float BCE_CLIP_LO = -16.11809555f, BCE_CLIP_HI = 15.94238503f;
#endif

// ---- device math ------------------------------------------------------------
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// tanh(x) = 1 - 2/(1 + 2^(2x log2 e)); v_mul + v_exp_f32 + v_add + v_rcp_f32 + v_fma, abs error ~2e-7.  No clamp:
// 2^(+big) = inf -> rcp = 0 -> 1, 2^(-big) = 0 -> rcp(1) = 1 -> -1.
__device__ __forceinline__ float fast_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.885390082f);
  return 1.0f - 2.0f * fast_rcp(e + 1.0f);
}

__device__ __forceinline__ float sigmoidf_(float x) {
  float xc = fminf(fmaxf(x, -30.0f), 30.0f);
  return fast_rcp(1.0f + __expf(-xc));
}

// ---- fp32 = three bf16 pieces, exactly (wgrad_bf16.hip, out_head_bf16.hip, lstm_mx.hip) -----------------------------
// x = p0 + p1 + p2: v_cvt_pk_bf16_f32 rounds a pair to nearest even and packs it (the 4 bytes an MFMA fragment wants), a
// piece as a float is its 16 bits shifted up, and the residual x - bf16(x) is exact in fp32: 11 instructions per pair.
typedef __bf16 clv_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned bf16_pack2(float lo, float hi) {
  const clv_bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void bf16_split_pair(float a, float b, unsigned (&piece)[3]) {
  piece[0] = bf16_pack2(a, b);
  const float ra = a - __builtin_bit_cast(float, piece[0] << 16), rb = b - __builtin_bit_cast(float, piece[0] & 0xffff0000u);
  piece[1] = bf16_pack2(ra, rb);
  piece[2] = bf16_pack2(ra - __builtin_bit_cast(float, piece[1] << 16), rb - __builtin_bit_cast(float, piece[1] & 0xffff0000u));
}
// The residual in ONE instruction: v_dot2c_f32_bf16 computes d += a.lo * b.lo + a.hi * b.hi on bf16 pairs, so with
// b = (-1, 0) or (0, -1) and d = x it subtracts one half of the packed pair from x; the result is exactly representable, so
// the instruction's own rounding cannot matter (tools/probes/dot2_split.hip: bitwise equal to shift + subtract on 2M random
// pairs, denormals included; an Inf / NaN in one half reaches its partner as 0 * Inf).  The constants sit in scalar
// registers: as literals the compiler encodes (-1, 0) as the inline constant -1.0, which the instruction does not read as
// that pair (same probe: every low half wrong).  7 instructions per pair -- but the dot product is NOT a 4-cycle vector
// instruction next to MFMAs: in wgrad_bf16's producers (which split while the other waves of the SIMD issue MFMAs) it made
// the kernel 10-18 % slower (35.9 -> 39.6 us at 32768 rows, 225 -> 266 us at 262144; profiles/r04_out_head_log.txt), so only
// code that splits outside MFMA phases uses it.
__device__ __forceinline__ unsigned sreg_const(unsigned v) {
  unsigned r;
  asm("s_mov_b32 %0, %1" : "=s"(r) : "i"(v));
  return r;
}
__device__ __forceinline__ float minus_bf16_lo(float x, unsigned pair) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(clv_bf16x2, pair), __builtin_bit_cast(clv_bf16x2, sreg_const(0x0000BF80u)), x, false);
}
__device__ __forceinline__ float minus_bf16_hi(float x, unsigned pair) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(clv_bf16x2, pair), __builtin_bit_cast(clv_bf16x2, sreg_const(0xBF800000u)), x, false);
}
__device__ __forceinline__ void bf16_split_pair_dot2(float a, float b, unsigned (&piece)[3]) {
  piece[0] = bf16_pack2(a, b);
  const float ra = minus_bf16_lo(a, piece[0]), rb = minus_bf16_hi(b, piece[0]);
  piece[1] = bf16_pack2(ra, rb);
  piece[2] = bf16_pack2(minus_bf16_lo(ra, piece[1]), minus_bf16_hi(rb, piece[1]));
}

// Keras 2.0.0 hard_sigmoid: clip(0.2*x + 0.5, 0, 1)
__device__ __forceinline__ float hard_sigmoid(float z) { return fminf(fmaxf(0.2f * z + 0.5f, 0.0f), 1.0f); }
// derivative; TF's clip passes the gradient at ties
__device__ __forceinline__ float hard_sigmoid_grad(float z) {
  float y = 0.2f * z + 0.5f;
  return (y >= 0.0f && y <= 1.0f) ? 0.2f : 0.0f;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace clv
