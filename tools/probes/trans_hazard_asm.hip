// The gfx940/gfx950 "trans forwarding hazard" behind an inline-asm statement (standalone reproducer, round 5).
//
// A VALU instruction that reads the VGPR a transcendental instruction (v_rcp_f32, v_exp_f32, ...) has just written needs
// one wait state in between.  LLVM's hazard recognizer inserts the s_nop for instructions the compiler emits; it does not
// look inside an asm statement.  Until round 5 csrc/lstm_common.h held `asm("v_cndmask_b32_e64 ...")` as its lane select,
// and with SLP vectorisation on, the sigmoid-gate instance of lstm_pair_fwd_kernel scheduled `v_rcp_f32 v97` directly in
// front of the asm select reading v97: wrong LSTM states (the Makefile then carried -fno-slp-vectorize "for correctness").
//
// kernel `bad`  : v_rcp_f32 and the consuming v_cndmask back to back inside ONE asm block (what the scheduler produced)
// kernel `good` : the same select written as inverse_ballot(mask) ? a : b -- the compiler's own v_cndmask, s_nop inserted
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o trans_hazard_asm trans_hazard_asm.hip && ./trans_hazard_asm
// Prints how many of the 64 lanes of each kernel differ from 1/x; `bad` > 0 shows the hazard on the hardware at hand (a
// part that forwards in time prints 0 for both -- the ISA only promises `good`).
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void bad(const float* x, float* out) {
  const int l = threadIdx.x;
  unsigned long long m = __builtin_amdgcn_ballot_w64((l & 1) == 0);
  float v = x[l], stale = -1.0f, r;
  // r = (even lane) ? rcp(v) : stale, the select issued right behind the rcp
  asm volatile("v_rcp_f32_e32 %1, %1\n\tv_cndmask_b32_e64 %0, %2, %1, %3" : "=v"(r), "+v"(v) : "v"(stale), "s"(m));
  out[l] = r;
}

__global__ void good(const float* x, float* out) {
  const int l = threadIdx.x;
  unsigned long long m = __builtin_amdgcn_ballot_w64((l & 1) == 0);
  asm volatile("" : "+s"(m));                       // opaque mask, like clv::lane_mask()
  const float v = __builtin_amdgcn_rcpf(x[l]);
  out[l] = __builtin_amdgcn_inverse_ballot_w64(m) ? v : -1.0f;
}

int main() {
  float h[64], r[64];
  for (int i = 0; i < 64; ++i) h[i] = 2.0f + i;
  float *x, *o;
  hipMalloc(&x, sizeof(h)); hipMalloc(&o, sizeof(r));
  hipMemcpy(x, h, sizeof(h), hipMemcpyHostToDevice);
  for (int which = 0; which < 2; ++which) {
    if (which == 0) bad<<<1, 64>>>(x, o); else good<<<1, 64>>>(x, o);
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    int wrong = 0;
    for (int i = 0; i < 64; ++i) {
      const float want = (i & 1) ? -1.0f : 1.0f / h[i];
      if (!(r[i] > want - 1e-6f && r[i] < want + 1e-6f)) ++wrong;
    }
    printf("%-4s: %d of 64 lanes wrong (lane 0: got %g, 1/x = %g)\n", which ? "good" : "bad", wrong, r[0], 1.0f / h[0]);
  }
  return 0;
}
