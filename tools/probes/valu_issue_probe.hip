// How fast does ONE SIMD of gfx950 issue vector instructions when 1, 2 or 3 waves share it?
// (standalone: hipcc --offload-arch=gfx950 -O3 -o valu_issue_probe valu_issue_probe.hip)
// One workgroup of 4*W waves (W waves per SIMD), each wave runs a loop of 64 instructions of one kind on independent
// registers (or one dependent chain) and reports shader cycles per instruction.  Kinds: v_fma_f32, v_pk_fma_f32,
// v_add_f32, v_exp_f32, a dependent v_fma chain, and a mix like the pair kernels' step (44 pk_fma + 40 plain).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ void probe(long long* cyc, float* sink, int n) {
  const int l = threadIdx.x;
  float a0 = l * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = p0 + 1.f, p5 = p1 + 1.f, p6 = p2 + 1.f, p7 = p3 + 1.f;
  const float m = 0.999f, b = 1e-6f;
  const f2 pm = {m, m}, pb = {b, b};
  __syncthreads();
  const long long t0 = clock64();
  for (int i = 0; i < n; ++i) {
    if (KIND == 0) {        // 64 independent-ish v_fma_f32 (8 chains)
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                        "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(b));)
    } else if (KIND == 1) { // 64 v_pk_fma_f32 (8 chains)
      REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                        "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm), "v"(pb));)
    } else if (KIND == 2) { // 64 v_add_f32
      REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                        "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
    } else if (KIND == 3) { // 64 v_exp_f32
      REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                        "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (KIND == 4) { // 64 dependent v_fma_f32 (one chain)
      REP64(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(m), "v"(b));)
    } else if (KIND == 5) { // mix: 44 pk_fma + 40 plain (20 fma + 20 add), independent
      REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                        "v_pk_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %10, %11\n v_fma_f32 %6, %6, %10, %11\n v_add_f32 %7, %7, %11\n"
                        "v_add_f32 %5, %5, %11\n v_fma_f32 %6, %6, %10, %11"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(pm), "v"(pb), "v"(m), "v"(b));)
    } else if (KIND == 6) { // 64 dependent v_pk_fma (one chain)
      REP64(asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(pm), "v"(pb));)
    } else if (KIND == 7) { // 32 x (v_fma + s_nop 0)
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n s_nop 0\n v_fma_f32 %1, %1, %8, %9\n s_nop 0\n v_fma_f32 %2, %2, %8, %9\n s_nop 0\n v_fma_f32 %3, %3, %8, %9\n s_nop 0"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(b));)
    } else if (KIND == 8) { // 32 x (v_fma + s_add)
      int sdummy = i;
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n s_add_i32 %10, %10, 1\n v_fma_f32 %1, %1, %8, %9\n s_add_i32 %10, %10, 1\n v_fma_f32 %2, %2, %8, %9\n s_add_i32 %10, %10, 1\n v_fma_f32 %3, %3, %8, %9\n s_add_i32 %10, %10, 1"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(b), "s"(sdummy));)
    } else if (KIND == 9) { // 64 v_add_f32_dpp quad_perm (dependent on previous via same reg chain of 8)
      REP8(asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                        "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                        "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                        "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    }
  }
  const long long t1 = clock64();
  if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0[0] + p1[1] + p2[0] + p3[1] + p4[0] + p5[1] + p6[0] + p7[1];
}

template <int KIND>
void run(const char* name, int per_iter, long long* cyc, float* sink) {
  const int n = 500;
  printf("%-44s", name); fflush(stdout);
  for (int w = 1; w <= 3; ++w) {
    probe<KIND><<<1, 256 * w>>>(cyc, sink, n);
    long long c[12];
    hipMemcpy(c, cyc, sizeof(long long) * 4 * w, hipMemcpyDeviceToHost);
    long long mx = 0;
    for (int i = 0; i < 4 * w; ++i) mx = c[i] > mx ? c[i] : mx;
    // cycles of SIMD time per instruction = slowest wave's cycles / (instructions per wave * waves per SIMD)
    hipError_t e = hipDeviceSynchronize(); if (e != hipSuccess) { printf(" ERR %s", hipGetErrorString(e)); fflush(stdout); return; }
    printf("  %dw/SIMD: %5.2f cyc/instr/wave, %5.2f SIMD-cyc/instr", w, (double)mx / (n * per_iter), (double)mx / (n * per_iter * w));
  }
  printf("\n"); fflush(stdout);
}

int main() {
  long long* cyc; float* sink;
  hipMalloc(&cyc, 8 * 64); hipMalloc(&sink, 4 * 4096);
  run<0>("v_fma_f32 (8 chains)", 64, cyc, sink);
  run<1>("v_pk_fma_f32 (8 chains)", 64, cyc, sink);
  run<2>("v_add_f32 (8 chains)", 64, cyc, sink);
  run<3>("v_exp_f32 (8 chains)", 64, cyc, sink);
  run<4>("v_fma_f32 dependent chain", 64, cyc, sink);
  run<6>("v_pk_fma_f32 dependent chain", 64, cyc, sink);
  run<5>("mix 40 pk_fma + 40 plain per 80", 80, cyc, sink);
  run<7>("v_fma + s_nop 0 pairs (per pair)", 32, cyc, sink);
  run<8>("v_fma + s_add pairs (per pair)", 32, cyc, sink);
  run<9>("v_add_f32_dpp quad_perm (8 chains)", 64, cyc, sink);
  return 0;
}
