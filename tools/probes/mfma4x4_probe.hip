// Layout and issue rate of v_mfma_f32_4x4x1_16B_f32 on gfx950 (standalone: hipcc --offload-arch=gfx950 -O3 -o probe ...).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void layout(float* out) {
  const int l = threadIdx.x;
  const float a = (float)(l + 1), b = (float)(100 * (l + 1));
  f32x4 d = {0.f, 0.f, 0.f, 0.f};
  d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[r * 64 + l] = d[r];
}

__global__ void rate(long long* cyc, float* sink, int n) {
  const int l = threadIdx.x & 63;
  f32x4 d0 = {0, 0, 0, 0}, d1 = d0, d2 = d0, d3 = d0;
  float a = (float)l, b = 1.0f + (float)l * 1e-3f;
  const long long t0 = clock64();
  for (int i = 0; i < n; ++i) {
    d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d0, 0, 0, 0);
    d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d1, 0, 0, 0);
    d2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d2, 0, 0, 0);
    d3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d3, 0, 0, 0);
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = d0[0] + d1[1] + d2[2] + d3[3];
}

__global__ void rate_dep(long long* cyc, float* sink, int n) {     // one dependent accumulator chain
  const int l = threadIdx.x & 63;
  f32x4 d0 = {0, 0, 0, 0};
  float a = (float)l, b = 1.0f + (float)l * 1e-3f;
  const long long t0 = clock64();
  for (int i = 0; i < 4 * n; ++i) d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d0, 0, 0, 0);
  const long long t1 = clock64();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = d0[0];
}

int main() {
  float* out; hipMalloc(&out, 256 * 4);
  layout<<<1, 64>>>(out);
  float h[256]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  // hypothesis: D[r] in lane l = A[lane 4*(l/4) + r] * B[lane l]
  int ok = 1;
  for (int r = 0; r < 4; ++r)
    for (int l = 0; l < 64; ++l) {
      const float want = (float)(4 * (l / 4) + r + 1) * (float)(100 * (l + 1));
      if (h[r * 64 + l] != want) ok = 0;
    }
  printf("layout D[r][lane l] = A[lane 4*(l/4)+r] * B[lane l]: %s\n", ok ? "YES" : "NO");
  if (!ok) for (int r = 0; r < 4; ++r) { for (int l = 0; l < 8; ++l) printf("%10.0f ", h[r * 64 + l]); printf("\n"); }
  long long* cyc; float* sink; hipMalloc(&cyc, 8 * 1024); hipMalloc(&sink, 4 * 1024 * 1024);
  const int n = 2000;
  for (int waves = 1; waves <= 8; waves *= 2) {
    rate<<<256, 64 * waves>>>(cyc, sink, n);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    rate_dep<<<256, 64 * waves>>>(cyc, sink, n);
    long long c2; hipMemcpy(&c2, cyc, 8, hipMemcpyDeviceToHost);
    printf("%d wave(s)/CU: %.2f shader cycles per MFMA per wave (4 accumulators), %.2f (one dependent chain)\n", waves,
           (double)c / (4.0 * n), (double)c2 / (4.0 * n));
  }
  return 0;
}
