// Do two 1024-thread workgroups share a CU on gfx950?  512 workgroups that spin ~20 us each, 256 CUs: with two per CU all of
// them start at once, with one per CU the second half starts when the first ends.  Variants: highest VGPR touched (clobber),
// highest SGPR touched, dynamic LDS bytes.   hipcc -O3 --offload-arch=gfx950 -o /tmp/coreside tools/probes/coreside_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

__device__ unsigned long long g_rec[4096][3];

template <int V, int S>
__global__ __launch_bounds__(1024) void spin(int us) {
  extern __shared__ float lds[];
  if (V >= 32) asm volatile("v_mov_b32 v31, 0" ::: "v31");
  if (V >= 64) asm volatile("v_mov_b32 v61, 0" ::: "v61");
  if (V >= 72) asm volatile("v_mov_b32 v70, 0" ::: "v70");
  if (S >= 80) asm volatile("s_mov_b32 s70, 0" ::: "s70");
  if (S >= 90) asm volatile("s_mov_b32 s84, 0" ::: "s84");
  if (S >= 100) asm volatile("s_mov_b32 s95, 0" ::: "s95");
  unsigned long long t0, t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  lds[threadIdx.x] = 1.f;
  do { asm volatile("s_sleep 8\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); } while (t - t0 < (unsigned long long)us * 100);
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    g_rec[blockIdx.x][0] = t0; g_rec[blockIdx.x][1] = t; g_rec[blockIdx.x][2] = hw;
  }
}

template <int V, int S>
static void run(const char* name, int lds, int threads = 1024) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(spin<V, S>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int n = 512;
  for (int it = 0; it < 2; ++it) {
    hipLaunchKernelGGL((spin<V, S>), dim3(n), dim3(threads), lds, 0, 20);
    hipDeviceSynchronize();
  }
  std::vector<unsigned long long> r(4096 * 3);
  hipMemcpyFromSymbol(r.data(), HIP_SYMBOL(g_rec), sizeof(unsigned long long) * 4096 * 3);
  unsigned long long t0 = ~0ull;
  for (int i = 0; i < n; ++i) t0 = std::min(t0, r[3 * i]);
  int late = 0;
  double last = 0;
  for (int i = 0; i < n; ++i) { const double s = (r[3 * i] - t0) / 100.0; late += s > 10.0; last = std::max(last, (r[3 * i + 1] - t0) / 100.0); }
  hipFuncAttributes fa;
  hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(spin<V, S>));
  printf("%-28s threads %4d lds %6d  regs %3d  -> %3d of 512 workgroups started late, all done at %5.1f us\n", name, threads, lds, fa.numRegs, late, last);
}

int main() {
  run<0, 0>("small", 4096);
  run<64, 0>("vgpr 62", 4096);
  run<72, 0>("vgpr 71", 4096);
  run<64, 80>("vgpr 62 sgpr ~71", 4096);
  run<64, 90>("vgpr 62 sgpr ~85", 4096);
  run<64, 100>("vgpr 62 sgpr ~96", 4096);
  run<64, 0>("vgpr 62, lds 62 KB", 62 * 1024);
  run<64, 0>("vgpr 62, lds 76 KB", 76 * 1024);
  run<64, 0>("vgpr 62, lds 79 KB", 79 * 1024);
  run<64, 80>("vgpr 62 sgpr ~71 lds 76 KB", 76 * 1024);
  run<0, 0>("small, 512 threads", 4096, 512);
  return 0;
}
