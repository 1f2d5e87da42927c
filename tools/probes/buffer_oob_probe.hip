// Does the raw-buffer range check of gfx950 include the SGPR offset?  (standalone)
// A descriptor of N floats over a buffer of 2N floats (second half = 7.0): loads with (voffset, soffset) on either side.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__global__ void k(float* p, int n, float* out) {
  rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p, 0, n * 4, 0x00020000);
  const int l = threadIdx.x;
  auto ld = [&](int voff, int soff) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0)); };
  out[0 * 64 + l] = ld(l * 4, 0);                  // in range
  out[1 * 64 + l] = ld(l * 4, n * 4);              // voffset in range, soffset = num_records
  out[2 * 64 + l] = ld(n * 4 + l * 4, 0);          // voffset out of range
  out[3 * 64 + l] = ld(l * 4, (n - 32) * 4);       // straddles: lanes 32.. beyond with soffset counted
  out[4 * 64 + l] = ld(0x80000000u + l * 4, 0);    // far out of range
}
int main() {
  const int n = 64;
  float h[2 * n]; for (int i = 0; i < 2 * n; ++i) h[i] = i < n ? (float)(i + 1) : 7.0f;
  float *p, *o; hipMalloc(&p, sizeof(h)); hipMalloc(&o, 5 * 64 * 4); hipMemcpy(p, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 64>>>(p, n, o);
  float r[5 * 64]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  const char* names[5] = {"in range", "soffset = num_records", "voffset >= num_records", "soffset = N-32 (lanes 32+ beyond)", "voffset 0x80000000"};
  for (int j = 0; j < 5; ++j) printf("%-36s lane0 %g lane31 %g lane32 %g lane63 %g\n", names[j], r[j * 64], r[j * 64 + 31], r[j * 64 + 32], r[j * 64 + 63]);
  return 0;
}
