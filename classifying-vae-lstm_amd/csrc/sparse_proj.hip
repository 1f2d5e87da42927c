// sparse_proj.hip -- input projection of piano-roll frames: out[r,:] = sum_k X[r,k] * K[k,:]   (gfx950)
//
// The LSTM input projections x_t . K_x (cl_vrnn/model.py:193-196, 218-226 as one [B*T,88] x [88,352] product
// per LSTM) multiply frames that are ~4 % nonzero (0.0443 note density): a dense GEMM spends 22 of
// every 23 FMAs on zeros and is bound by its own operand traffic.  Here the kernel K [nx,N] (124 KB for
// 88 x 352) stays in LDS for the lifetime of a persistent workgroup and each wave adds the kernel rows of its
// frame's nonzero inputs.  Exact for any float input (a dense frame just costs nx row additions); summation
// order = ascending input index, like the dense product.
#include "common.h"

namespace clv {

constexpr int SP_NW = 16;          // waves per workgroup, each one owns whole frames
constexpr int SP_NT = SP_NW * 64;
constexpr int SP_XMAX = 128;       // max inputs per frame (2 per lane)
constexpr int SP_NC = 6;           // output columns per lane (N <= 384)

struct SparseProjArgs {
  int R, nx, N, ldx, ldo;
  const float* X;      // [R, ldx]
  const float* K;      // [nx, N]
  float* out;          // [R, ldo]
};

// A wave handles a frame on its own: lane k holds inputs k and k+64, two ballots give the nonzero sets as
// scalar masks, and a scalar bit-scan loop adds the listed kernel rows (LDS) into the lane's 6 output columns.
// No lists, no LDS traffic besides the kernel rows, no barriers after the kernel is staged.
__global__ __launch_bounds__(SP_NT) void sparse_proj_kernel(SparseProjArgs a) {
  extern __shared__ __attribute__((aligned(16))) float Kl[];            // [nx][N]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  {
    const int nv = a.nx * a.N / 4;
    const float4* src = reinterpret_cast<const float4*>(a.K);
    float4* dst = reinterpret_cast<float4*>(Kl);
    for (int i0 = tid; i0 < nv; i0 += 4 * SP_NT) {      // 4 loads in flight per thread
      float4 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = src[min(i0 + q * SP_NT, nv - 1)];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (i0 + q * SP_NT < nv) dst[i0 + q * SP_NT] = v[q];
    }
  }
  __syncthreads();
  const int per = (a.R + gridDim.x - 1) / gridDim.x;
  const int first = blockIdx.x * per, last = min(a.R, first + per);
  int colo[SP_NC];
#pragma unroll
  for (int c = 0; c < SP_NC; ++c) colo[c] = min(lane + 64 * c, a.N - 1);
  const int k0 = min(lane, a.nx - 1), k1 = min(lane + 64, a.nx - 1);
  const bool v0 = lane < a.nx, v1 = lane + 64 < a.nx;
  float nx0, nx1;
  {
    const float* fp = a.X + (size_t)min(first + wave, a.R - 1) * a.ldx;
    nx0 = fp[k0]; nx1 = fp[k1];
  }
  for (int f = first + wave; f < last; f += SP_NW) {
    const float fx0 = nx0, fx1 = nx1;
    {                                                     // next frame of this wave (clamped, unconditional)
      const float* fp = a.X + (size_t)min(f + SP_NW, a.R - 1) * a.ldx;
      nx0 = fp[k0]; nx1 = fp[k1];
    }
    unsigned long long m0 = __ballot(v0 && fx0 != 0.f), m1 = __ballot(v1 && fx1 != 0.f);
    float acc[SP_NC];
#pragma unroll
    for (int c = 0; c < SP_NC; ++c) acc[c] = 0.f;
    while (m0) {                                          // scalar loop over the notes that are on
      const int k = __builtin_ctzll(m0);
      m0 &= m0 - 1;
      const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fx0), k));
      const float* kr = Kl + k * a.N;
#pragma unroll
      for (int c = 0; c < SP_NC; ++c) acc[c] = fmaf(v, kr[colo[c]], acc[c]);
    }
    while (m1) {
      const int k = __builtin_ctzll(m1);
      m1 &= m1 - 1;
      const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fx1), k));
      const float* kr = Kl + (k + 64) * a.N;
#pragma unroll
      for (int c = 0; c < SP_NC; ++c) acc[c] = fmaf(v, kr[colo[c]], acc[c]);
    }
    float* op = a.out + (size_t)f * a.ldo;
#pragma unroll
    for (int c = 0; c < SP_NC; ++c)
      if (lane + 64 * c < a.N) op[lane + 64 * c] = acc[c];
  }
}

}  // namespace clv

extern "C" size_t clv_sparse_proj_lds_bytes(int nx, int N) { return (size_t)nx * N * sizeof(float); }

extern "C" int clv_sparse_proj_supported(int nx, int N) {
  return nx >= 1 && nx <= clv::SP_XMAX && N >= 1 && N <= clv::SP_NC * 64 && (nx * N) % 4 == 0 &&
         clv_sparse_proj_lds_bytes(nx, N) <= 150 * 1024;
}

extern "C" int clv_sparse_proj(int R, int nx, int N, const float* X, int ldx, const float* K, float* out, int ldo,
                               void* stream) {
  using namespace clv;
  if (R <= 0 || !X || !K || !out || ldx < nx || ldo < N || !clv_sparse_proj_supported(nx, N)) return CLV_EINVAL;
  if (((uintptr_t)K) % 16 != 0) return CLV_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sparse_proj_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  SparseProjArgs a{R, nx, N, ldx, ldo, X, K, out};
  const int wgs = R < 256 * SP_NW ? (R + SP_NW - 1) / SP_NW : 256;     // one persistent workgroup per CU
  ProfScope p("sparse_proj", s);
  hipLaunchKernelGGL(sparse_proj_kernel, dim3(wgs), dim3(SP_NT), clv_sparse_proj_lds_bytes(nx, N), s, a);
  return launch_status();
}
