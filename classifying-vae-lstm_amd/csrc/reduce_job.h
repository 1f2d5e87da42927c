// reduce_job.h -- a pending split-K reduction (shared by gemm.hip and out_head.hip)
#pragma once
#include "common.h"

namespace clv {

constexpr int MAX_PROB = 4;

// One pending split-K reduction: everything the epilogue needs, small enough that a table of them
// travels as kernel arguments (clv_reduce_job in the C ABI is this struct, opaque).
struct ReduceProb { float* C; int ldc; int row0; };
struct ReduceJob {
  const float* partial;    // [splits][M][N] raw partial sums
  int M, N, splits, nprob;
  float alpha, beta;
  const float* bias;
  const float* aux;
  int act;
  int pad_;                // set by the reduce launchers: 1 = 4 outputs per thread (16-byte slab loads)
  ReduceProb prob[MAX_PROB];   // nprob == 0: prob[0] is the single output
};
static_assert(sizeof(ReduceJob) <= sizeof(clv_reduce_job), "clv_reduce_job too small");

int launch_reduce(const ReduceJob& j, hipStream_t s);     // gemm.hip: one reduction, now

}  // namespace clv
