// Dev tool (not part of the product): times lstm_fwd_kernel with one phase removed at a time
// (cdna_hip_programming.md 7, "Ablate").  Results of ABL != 0 variants are wrong by design.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/lstm_ablate.hip -o tools/lstm_ablate && tools/lstm_ablate
#include <vector>
#include "../classifying-vae-lstm_amd/csrc/lstm.hip"

namespace clv {
void prof_begin(const char*, hipStream_t) {}
void prof_end(hipStream_t) {}
bool prof_on() { return false; }
}

template <int KS, int ABL>
float run(const clv::LstmFwdArgs& a, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i)
    hipLaunchKernelGGL((clv::lstm_fwd_kernel<KS, 1, 0, true, ABL>), dim3(a.B), dim3(clv::Geo<KS>::NT), 0, 0, a);
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i)
    hipLaunchKernelGGL((clv::lstm_fwd_kernel<KS, 1, 0, true, ABL>), dim3(a.B), dim3(clv::Geo<KS>::NT), 0, 0, a);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.f / reps;
}

template <int KS, int R>
float run_r(clv::LstmFwdArgs a, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i)
    hipLaunchKernelGGL((clv::lstm_fwd_kernel<KS, R, 0, true, 0>), dim3(a.B / R), dim3(clv::Geo<KS>::NT), 0, 0, a);
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i)
    hipLaunchKernelGGL((clv::lstm_fwd_kernel<KS, R, 0, true, 0>), dim3(a.B / R), dim3(clv::Geo<KS>::NT), 0, 0, a);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.f / reps;
}

int main() {
  const int B = 1024, T = 128;
  float *xproj, *rb, *U, *hs, *cs, *gates;
  hipMalloc(&xproj, (size_t)B * T * 352 * 4); hipMalloc(&gates, (size_t)B * T * 352 * 4);
  hipMalloc(&rb, B * 352 * 4); hipMalloc(&U, 88 * 352 * 4);
  hipMalloc(&hs, (size_t)B * T * 88 * 4); hipMalloc(&cs, (size_t)B * T * 88 * 4);
  std::vector<float> h((size_t)B * T * 352);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0.5f * sinf(0.37f * i);
  hipMemcpy(xproj, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(rb, h.data(), B * 352 * 4, hipMemcpyHostToDevice);
  for (int i = 0; i < 88 * 352; ++i) h[i] = 0.1f * cosf(0.11f * i);
  hipMemcpy(U, h.data(), 88 * 352 * 4, hipMemcpyHostToDevice);
  clv::LstmFwdArgs a{256, T, xproj, rb, U, nullptr, nullptr, hs, cs, gates, nullptr, nullptr};
  {
    printf("rows per workgroup, 256 workgroups each (us per launch):\n");
    clv::LstmFwdArgs b1 = a, b2 = a, b4 = a;
    b1.B = 256; b2.B = 512; b4.B = 1024;
    printf("  KS=8: R=1 %7.2f  R=2 %7.2f  R=4 %7.2f\n", run_r<8, 1>(b1, 20), run_r<8, 2>(b2, 20), run_r<8, 4>(b4, 20));
    printf("  KS=4: R=1 %7.2f  R=2 %7.2f  R=4 %7.2f\n", run_r<4, 1>(b1, 20), run_r<4, 2>(b2, 20), run_r<4, 4>(b4, 20));
  }
  const int reps = 20;
  const char* names[] = {"full", "no global stores", "no gate math", "no FMAs", "no barrier", "no xproj loads", "no DPP reduce"};
  printf("lstm_fwd B=%d T=%d  (us per launch; ns per step)\n", B, T);
#define ROW(KS, ABL) { clv::LstmFwdArgs b = a; b.T = 32; float u0 = run<KS, ABL>(b, reps); float us = run<KS, ABL>(a, reps); \
    printf("  KS=%d %-18s T=128 %8.2f us   T=32 %7.2f us   slope %7.1f ns/step   intercept %6.2f us\n", KS, names[ABL], us, u0, (us - u0) * 1000 / 96, u0 - (us - u0) / 96 * 32); }
  ROW(8, 0) ROW(8, 1) ROW(8, 2) ROW(8, 3) ROW(8, 4) ROW(8, 5) ROW(8, 6)
  ROW(4, 0) ROW(4, 1) ROW(4, 2) ROW(4, 3) ROW(4, 4) ROW(4, 5) ROW(4, 6)
  return 0;
}
