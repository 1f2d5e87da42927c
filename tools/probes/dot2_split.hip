// probe: is  a - bf16(a)  exact through v_dot2c_f32_bf16 (one instruction) on gfx950?  Compared bitwise with shift + subtract.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  const bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}
__global__ void k(const float* p, float* ref, float* got, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float a = p[2 * i], b = p[2 * i + 1];
  const unsigned p0 = pack2(a, b);
  ref[2 * i] = a - __builtin_bit_cast(float, p0 << 16);
  ref[2 * i + 1] = b - __builtin_bit_cast(float, p0 & 0xffff0000u);
  const bf16x2 v = __builtin_bit_cast(bf16x2, p0);
  unsigned ulo = 0x0000BF80u, uhi = 0xBF800000u;
  asm volatile("" : "+v"(ulo), "+v"(uhi));            // registers, not inline constants
  const bf16x2 klo = __builtin_bit_cast(bf16x2, ulo), khi = __builtin_bit_cast(bf16x2, uhi);
  got[2 * i] = __builtin_amdgcn_fdot2_f32_bf16(v, klo, a, false);
  got[2 * i + 1] = __builtin_amdgcn_fdot2_f32_bf16(v, khi, b, false);
}
int main() {
  const int n = 1 << 20;
  float* h = (float*)malloc(2 * n * 4);
  srand(1);
  for (int i = 0; i < 2 * n; ++i) {
    unsigned u = ((unsigned)rand() << 16) ^ (unsigned)rand();
    if (i % 3 == 0) { float f = ((rand() % 20001) - 10000) * 1e-4f; memcpy(&u, &f, 4); }      // ordinary magnitudes
    if (((u >> 23) & 0xff) == 0xff) u &= 0x7fffffffu >> 1;                                      // no inf / nan
    memcpy(&h[i], &u, 4);
  }
  float *d, *r, *g;
  hipMalloc(&d, 2 * n * 4); hipMalloc(&r, 2 * n * 4); hipMalloc(&g, 2 * n * 4);
  hipMemcpy(d, h, 2 * n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, r, g, n);
  float* hr = (float*)malloc(2 * n * 4); float* hg = (float*)malloc(2 * n * 4);
  hipMemcpy(hr, r, 2 * n * 4, hipMemcpyDeviceToHost); hipMemcpy(hg, g, 2 * n * 4, hipMemcpyDeviceToHost);
  long bad = 0, bad_normal = 0, bad_par[2] = {0, 0};
  for (int i = 0; i < 2 * n; ++i)
    if (memcmp(&hr[i], &hg[i], 4)) {
      ++bad; ++bad_par[i & 1];
      if (fabsf(hr[i]) > 1.2e-38f || fabsf(hg[i]) > 1.2e-38f) {
        if (bad_normal++ < 5) printf("  a=%a ref=%a got=%a\n", h[i], hr[i], hg[i]);
      }
    }
  printf("dot2 split: %ld of %d differ bitwise (%ld low halves, %ld high halves), %ld of them outside the denormal range\n", bad, 2 * n, bad_par[0], bad_par[1], bad_normal);
  return 0;
}
