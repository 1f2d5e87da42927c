// optim.hip -- Adam with weight normalisation over a flat parameter buffer
// (restates utils/weightnorm.py:75-178 of the reference; plain Keras Adam when
// weightnorm == 0).  HBM-bound: ~7 floats moved per parameter per step.
//
// Work is cut into units of <= 64 rows of one tensor (all its columns).  The two
// per-column reductions (||V||^2 and sum g.V before the update, ||V'||^2 after)
// go through deterministic partial slabs, not atomics:
//   K1 stats    : partial (sum V^2, sum g.V) per unit and column
//   K2 columns  : per column: ||V||, grad_g, Adam on g  -> column scalars
//   K3 update   : grad_V, Adam on V, W <- V', partial sum V'^2 ; biases: plain Adam
//   K4 columns2 : per column: s' = g'/||V'|| -> s ; advances `iterations`
//   K5 rescale  : W <- s'.V'
// The column kernels sum a tensor's per-unit partials with 64 columns x 4 unit-lanes per block and
// several loads in flight (a serial loop over hW's 352 units would be latency-bound).
#include "common.h"

namespace clv {

#ifndef CLV_ADAM_UNIT_ROWS
#define CLV_ADAM_UNIT_ROWS 16
#endif
constexpr int UNIT_ROWS = CLV_ADAM_UNIT_ROWS;
// fused small-tensor kernel: 16 row lanes x 9 rows = matrices of <= 144 rows (round 3: 9, so that cl_vrnn's decoder input
// kernel at latent_dim 32 -- 88 + 32 + 10 = 130 rows -- is a small tensor and the hW kernel stays the ONLY tall one: the
// two-launch form of clv_adam_wn_step then applies at configuration 5 as well)
constexpr int SM_RL = 16, SM_RMAX = 9;
constexpr int SM_ROWS = SM_RL * SM_RMAX;

struct AdamUnit {
  int64_t offset;       // element offset of the tensor
  int64_t col_offset;   // tensor's offset into s/mg/vg
  int32_t row0, nrows, cols;
  int32_t is_matrix;
  int32_t part_off;     // this unit's offset into the partial slabs
  int32_t small;        // tensor handled by wn_small_kernel when weight norm is on (the chain kernels skip it)
};
struct AdamCol {
  int64_t col_global;   // index into s/mg/vg
  int32_t part_base;    // first unit's partial offset for this tensor
  int32_t nunits;
  int32_t cols;
  int32_t col_local;
};
struct AdamHyper {
  float lr, b1, b2, eps;
  int weightnorm;
  int step_t;                 // used when iterations == nullptr
  const int32_t* iterations;  // device counter (Keras `iterations`), t = *iterations + 1
};

__device__ __forceinline__ float adam_lr_t(const AdamHyper& h) {
  // CLV_STEP_ADVANCED: the counter was advanced by the step that produced the gradients, it already holds t
  const int t = h.iterations ? (*h.iterations + (h.step_t == CLV_STEP_ADVANCED ? 0 : 1)) : h.step_t;
  return h.lr * sqrtf(1.f - powf(h.b2, (float)t)) / (1.f - powf(h.b1, (float)t));
}

__global__ __launch_bounds__(256) void wn_stats_kernel(const AdamUnit* units, const float* params, const float* grads,
                                                       const float* s, float* partA, float* partB) {
  const AdamUnit un = units[blockIdx.x];
  if (!un.is_matrix || un.small) return;
  __shared__ float ra[4][64], rbb[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  for (int c0 = 0; c0 < un.cols; c0 += 64) {
    const int col = c0 + cx;
    float a = 0.f, b = 0.f;
    if (col < un.cols) {
      const float inv_s = 1.f / s[un.col_offset + col];
      float pv[UNIT_ROWS / 4], gv[UNIT_ROWS / 4];
#pragma unroll
      for (int i = 0; i < UNIT_ROWS / 4; ++i) {
        const int r = ry + 4 * i;
        const size_t o = un.offset + (size_t)(un.row0 + min(r, un.nrows - 1)) * un.cols + col;
        pv[i] = params[o]; gv[i] = grads[o];
      }
#pragma unroll
      for (int i = 0; i < UNIT_ROWS / 4; ++i) {
        if (ry + 4 * i < un.nrows) {
          const float V = pv[i] * inv_s;
          a += V * V;
          b += gv[i] * V;
        }
      }
    }
    ra[ry][cx] = a; rbb[ry][cx] = b;
    __syncthreads();
    if (ry == 0 && col < un.cols) {
      partA[un.part_off + col] = ra[0][cx] + ra[1][cx] + ra[2][cx] + ra[3][cx];
      partB[un.part_off + col] = rbb[0][cx] + rbb[1][cx] + rbb[2][cx] + rbb[3][cx];
    }
    __syncthreads();
  }
}

// block = 4 consecutive global matrix columns x 64 unit-lanes (only tall tensors reach these kernels: hW/kernel has
// 704 partial rows per column); returns this lane's share of the column sums
constexpr int CL = 64;     // unit-lanes per column
constexpr int CPB = 4;     // columns per block
__device__ __forceinline__ void col_partial_sums(const AdamCol& c, const float* pa, const float* pb, int zy,
                                                 float& a, float& b) {
  const float* qa = pa + c.part_base + c.col_local;
  const float* qb = pb ? pb + c.part_base + c.col_local : nullptr;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
  int k = zy;
  // 12 (24 with qb) independent loads in flight: hW/kernel at config 3 has 704 unit partials per column, 11 per lane --
  // with 4 in flight that was three dependent round trips to L2 / HBM (7.6 us for a 2.5 MB reduction)
  for (; k + 11 * CL < c.nunits; k += 12 * CL) {
    float ta[12], tb[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      ta[i] = qa[(size_t)(k + i * CL) * c.cols];
      tb[i] = qb ? qb[(size_t)(k + i * CL) * c.cols] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 12; i += 4) {
      a0 += ta[i]; a1 += ta[i + 1]; a2 += ta[i + 2]; a3 += ta[i + 3];
      b0 += tb[i]; b1 += tb[i + 1]; b2 += tb[i + 2]; b3 += tb[i + 3];
    }
  }
  {   // the remainder (< 12 per lane) in one round as well: clamped loads, masked sums
    float ta[12], tb[12];
    const int last = c.nunits - 1;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int kk = min(k + i * CL, last);
      ta[i] = qa[(size_t)kk * c.cols];
      tb[i] = qb ? qb[(size_t)kk * c.cols] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 12; i += 4) {
      a0 += (k + i * CL < c.nunits) ? ta[i] : 0.f;             a1 += (k + (i + 1) * CL < c.nunits) ? ta[i + 1] : 0.f;
      a2 += (k + (i + 2) * CL < c.nunits) ? ta[i + 2] : 0.f;   a3 += (k + (i + 3) * CL < c.nunits) ? ta[i + 3] : 0.f;
      b0 += (k + i * CL < c.nunits) ? tb[i] : 0.f;             b1 += (k + (i + 1) * CL < c.nunits) ? tb[i + 1] : 0.f;
      b2 += (k + (i + 2) * CL < c.nunits) ? tb[i + 2] : 0.f;   b3 += (k + (i + 3) * CL < c.nunits) ? tb[i + 3] : 0.f;
    }
  }
  a0 += a2; a1 += a3; b0 += b2; b1 += b3;
  a = a0 + a1; b = b0 + b1;
}
__device__ __forceinline__ float lanes16_sum(float (*red)[CPB], int zy, int cx, float v) {
  red[zy][cx] = v;
  __syncthreads();
  float t = 0.f;
  if (zy == 0)
#pragma unroll
    for (int i = 0; i < CL; ++i) t += red[i][cx];
  __syncthreads();
  return t;
}

// colscal[4*j + {0,1,2,3}] = {1/s, grad_g/||V||, s, new_g}
__global__ __launch_bounds__(256) void wn_cols_kernel(int n_cols, const AdamCol* cols, const float* partA,
                                                      const float* partB, const float* s, float* mg, float* vg,
                                                      float* colscal, AdamHyper h) {
  __shared__ float red[CL][CPB];
  const int cx = threadIdx.x & (CPB - 1), zy = threadIdx.x / CPB;
  const int j = blockIdx.x * CPB + cx;
  float a = 0.f, b = 0.f;
  AdamCol c;
  bool skip = true;
  if (j < n_cols) { c = cols[j]; skip = c.nunits < 0; if (!skip) col_partial_sums(c, partA, partB, zy, a, b); }
  a = lanes16_sum(red, zy, cx, a);
  b = lanes16_sum(red, zy, cx, b);
  if (zy != 0 || j >= n_cols || skip) return;
  const float lr_t = adam_lr_t(h);
  const float sc = s[c.col_global];
  const float Vn = sqrtf(a);
  const float gparam = sc * Vn;
  const float grad_g = b / Vn;
  const float mgn = h.b1 * mg[c.col_global] + (1.f - h.b1) * grad_g;
  const float vgn = h.b2 * vg[c.col_global] + (1.f - h.b2) * grad_g * grad_g;
  mg[c.col_global] = mgn;
  vg[c.col_global] = vgn;
  colscal[4 * j + 0] = 1.f / sc;
  colscal[4 * j + 1] = grad_g / Vn;
  colscal[4 * j + 2] = sc;
  colscal[4 * j + 3] = gparam - lr_t * mgn / (sqrtf(vgn) + h.eps);
}

// s' = g'/||V'|| per column -> colscal[4j+0] (reused) and the persistent s; advances `iterations`
__global__ __launch_bounds__(256) void wn_cols2_kernel(int n_cols, const AdamCol* cols, const float* partC, float* s,
                                                       float* colscal, int32_t* iterations, float* vn2) {
  __shared__ float red[CL][CPB];
  if (blockIdx.x == 0 && threadIdx.x == 0 && iterations) *iterations += 1;
  const int cx = threadIdx.x & (CPB - 1), zy = threadIdx.x / CPB;
  const int j = blockIdx.x * CPB + cx;
  float a = 0.f, b = 0.f;
  AdamCol c;
  bool skip = true;
  if (j < n_cols) { c = cols[j]; skip = c.nunits < 0; if (!skip) col_partial_sums(c, partC, nullptr, zy, a, b); }
  a = lanes16_sum(red, zy, cx, a);
  if (zy != 0 || j >= n_cols || skip) return;
  const float snew = colscal[4 * j + 3] / sqrtf(a);
  colscal[4 * j + 0] = snew;
  s[c.col_global] = snew;
  if (vn2) vn2[c.col_global] = a;          // ||V'||^2: after the rescale V = W / s' = V', so this is the next step's sum V^2
}

__device__ __forceinline__ void wn_update_body(const AdamUnit& un, int unit_idx, float* params, const float* grads,
                                               float* m, float* v, const float* colscal,
                                               const int32_t* colidx0, float* partC, const AdamHyper& h) {
  if (un.small && h.weightnorm == CLV_OPT_ADAM_WN) return;   // done by the small-tensor blocks
  const float lr_t = adam_lr_t(h);
  if (h.weightnorm == CLV_OPT_RMSPROP) {  // Keras RMSprop: a = rho a + (1 - rho) g^2 ; p -= lr g / (sqrt(a) + eps)
    const int n = un.nrows * un.cols;
    const size_t base = un.offset + (size_t)un.row0 * un.cols;
    for (int i = threadIdx.x; i < n; i += 256) {
      const float g = grads[base + i];
      const float an = h.b2 * v[base + i] + (1.f - h.b2) * g * g;
      v[base + i] = an;
      params[base + i] -= h.lr * g / (sqrtf(an) + h.eps);
    }
    return;
  }
  if (!un.is_matrix || !h.weightnorm) {   // plain Adam (biases; everything when weightnorm is off)
    const int n = un.nrows * un.cols;
    const size_t base = un.offset + (size_t)un.row0 * un.cols;
    for (int i = threadIdx.x; i < n; i += 256) {
      const float g = grads[base + i];
      const float mn = h.b1 * m[base + i] + (1.f - h.b1) * g;
      const float vn = h.b2 * v[base + i] + (1.f - h.b2) * g * g;
      m[base + i] = mn; v[base + i] = vn;
      params[base + i] -= lr_t * mn / (sqrtf(vn) + h.eps);
    }
    return;
  }
  __shared__ float rc[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int cbase = colidx0[unit_idx];     // index of this tensor's column 0 in colscal
  for (int c0 = 0; c0 < un.cols; c0 += 64) {
    const int col = c0 + cx;
    float acc = 0.f;
    if (col < un.cols) {
      const float inv_s = colscal[4 * (cbase + col) + 0], gov = colscal[4 * (cbase + col) + 1];
      const float sc = colscal[4 * (cbase + col) + 2];
      float pv[UNIT_ROWS / 4], gv[UNIT_ROWS / 4], mv[UNIT_ROWS / 4], vv[UNIT_ROWS / 4];
#pragma unroll
      for (int i = 0; i < UNIT_ROWS / 4; ++i) {
        const int r = ry + 4 * i;
        const size_t o = un.offset + (size_t)(un.row0 + min(r, un.nrows - 1)) * un.cols + col;
        pv[i] = params[o]; gv[i] = grads[o]; mv[i] = m[o]; vv[i] = v[o];
      }
#pragma unroll
      for (int i = 0; i < UNIT_ROWS / 4; ++i) {
        const int r = ry + 4 * i;
        if (r < un.nrows) {
          const size_t o = un.offset + (size_t)(un.row0 + r) * un.cols + col;
          const float V = pv[i] * inv_s;
          const float gV = sc * (gv[i] - gov * V);
          const float mn = h.b1 * mv[i] + (1.f - h.b1) * gV;
          const float vn = h.b2 * vv[i] + (1.f - h.b2) * gV * gV;
          m[o] = mn; v[o] = vn;
          const float Vp = V - lr_t * mn / (sqrtf(vn) + h.eps);
          params[o] = Vp;
          acc += Vp * Vp;
        }
      }
    }
    rc[ry][cx] = acc;
    __syncthreads();
    if (ry == 0 && col < un.cols) partC[un.part_off + col] = rc[0][cx] + rc[1][cx] + rc[2][cx] + rc[3][cx];
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void wn_rescale_kernel(const AdamUnit* units, float* params, const float* colscal,
                                                         const int32_t* colidx0) {
  const AdamUnit un = units[blockIdx.x];
  if (!un.is_matrix || un.small) return;
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int cbase = colidx0[blockIdx.x];
  for (int c0 = 0; c0 < un.cols; c0 += 64) {
    const int col = c0 + cx;
    if (col >= un.cols) continue;
    const float snew = colscal[4 * (cbase + col) + 0];
    for (int r = ry; r < un.nrows; r += 4) {
      const size_t o = un.offset + (size_t)(un.row0 + r) * un.cols + col;
      params[o] *= snew;
    }
  }
}

// ---------------------------------------------------------------------------
// Matrices of <= 144 rows (every tensor of cl_vae; everything but hW/kernel in cl_vrnn) and the biases: the whole
// Adam-WN update of 16 columns in ONE workgroup -- both column reductions stay in LDS, parameters and gradients are
// read once -- instead of five launches with partial slabs in between.
// ---------------------------------------------------------------------------
struct SmallItem {
  int64_t offset;       // tensor's element offset
  int64_t col_offset;   // tensor's offset into s/mg/vg
  int32_t rows, cols, col0, is_matrix;
};
__device__ __forceinline__ void wn_small_body(const SmallItem& it, float* params, const float* grads, float* m,
                                              float* v, float* mg, float* vg, float* s, const AdamHyper& h) {
  const float lr_t = adam_lr_t(h);
  if (!it.is_matrix) {                                  // bias: plain Adam over its elements
    const int n = it.rows * it.cols;
    for (int i = threadIdx.x; i < n; i += 256) {
      const size_t o = it.offset + i;
      const float g = grads[o];
      const float mn = h.b1 * m[o] + (1.f - h.b1) * g;
      const float vn = h.b2 * v[o] + (1.f - h.b2) * g * g;
      m[o] = mn; v[o] = vn;
      params[o] -= lr_t * mn / (sqrtf(vn) + h.eps);
    }
    return;
  }
  __shared__ float red[SM_RL][16], tot[2][16];
  const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int col = it.col0 + cx;
  const bool live = col < it.cols;
  const int cc = min(col, it.cols - 1);
  const float sc = s[it.col_offset + cc];
  const float inv_s = 1.f / sc;
  float pv[SM_RMAX], gv[SM_RMAX], mv[SM_RMAX], vv[SM_RMAX];
#pragma unroll
  for (int i = 0; i < SM_RMAX; ++i) {                   // unconditional (clamped) loads, all in flight
    const size_t o = it.offset + (size_t)min(ry + SM_RL * i, it.rows - 1) * it.cols + cc;
    pv[i] = params[o]; gv[i] = grads[o]; mv[i] = m[o]; vv[i] = v[o];
  }
  float a = 0.f, b = 0.f;
#pragma unroll
  for (int i = 0; i < SM_RMAX; ++i)
    if (ry + SM_RL * i < it.rows) {
      const float V = pv[i] * inv_s;
      pv[i] = V;
      a += V * V;
      b += gv[i] * V;
    }
  auto colsum = [&](float x, int slot) {               // sum over the 16 row lanes, result to every lane of the column
    red[ry][cx] = x;
    __syncthreads();
    if (ry == 0) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < SM_RL; ++i) t += red[i][cx];
      tot[slot][cx] = t;
    }
    __syncthreads();
    return tot[slot][cx];
  };
  a = colsum(a, 0);
  b = colsum(b, 1);
  const float Vn = sqrtf(a);
  const float gparam = sc * Vn;
  const float grad_g = b / Vn;
  const size_t cg = it.col_offset + cc;
  const float mgn = h.b1 * mg[cg] + (1.f - h.b1) * grad_g;
  const float vgn = h.b2 * vg[cg] + (1.f - h.b2) * grad_g * grad_g;
  const float gnew = gparam - lr_t * mgn / (sqrtf(vgn) + h.eps);
  const float gov = grad_g / Vn;
  float c2 = 0.f;
#pragma unroll
  for (int i = 0; i < SM_RMAX; ++i)
    if (ry + SM_RL * i < it.rows) {
      const float gV = sc * (gv[i] - gov * pv[i]);
      const float mn = h.b1 * mv[i] + (1.f - h.b1) * gV;
      const float vn = h.b2 * vv[i] + (1.f - h.b2) * gV * gV;
      mv[i] = mn; vv[i] = vn;
      const float Vp = pv[i] - lr_t * mn / (sqrtf(vn) + h.eps);
      pv[i] = Vp;
      c2 += Vp * Vp;
    }
  c2 = colsum(c2, 0);
  const float snew = gnew / sqrtf(c2);
  if (live) {
#pragma unroll
    for (int i = 0; i < SM_RMAX; ++i)
      if (ry + SM_RL * i < it.rows) {
        const size_t o = it.offset + (size_t)(ry + SM_RL * i) * it.cols + col;
        m[o] = mv[i]; v[o] = vv[i];
        params[o] = snew * pv[i];
      }
    if (ry == 0) { mg[cg] = mgn; vg[cg] = vgn; s[cg] = snew; }
  }
}

// K3 and the small tensors in one launch: blocks [0, n_units) update a unit of a tall matrix (or plain Adam), blocks
// [n_units, n_units + n_small) do the whole Adam-WN of one small tensor.  The small-tensor blocks are a latency chain
// of their own (~7 us); next to the update blocks they cost nothing.
__global__ __launch_bounds__(256) void wn_update_kernel(int n_units, const AdamUnit* units, const SmallItem* items,
                                                        float* params, const float* grads, float* m, float* v,
                                                        float* mg, float* vg, float* s, const float* colscal,
                                                        const int32_t* colidx0, float* partC, AdamHyper h) {
  if ((int)blockIdx.x >= n_units) {
    wn_small_body(items[blockIdx.x - n_units], params, grads, m, v, mg, vg, s, h);
    return;
  }
  wn_update_body(units[blockIdx.x], (int)blockIdx.x, params, grads, m, v, colscal, colidx0, partC, h);
}

__global__ void adam_bump_kernel(int32_t* iterations) { *iterations += 1; }

// ---------------------------------------------------------------------------
// Two-launch Adam-WN of ONE tall matrix (cl_vrnn's hW/kernel: 87 % of the parameters) when both column sums of the
// first pass are known BEFORE the optimizer runs (clv_adam_wn_step):
//   sum_r V^2    = ||V||^2 of the previous step's result (kept per column in vn2: the rescale leaves W = s' V', so the
//                  next step's V is this step's V');
//   sum_r g.V    = (1/s) sum_r g[r,c] W[r,c], and with g = X^T dH (the layer's own gradient) that is
//                  (1/s) sum_b (X W)[b,c] dH[b,c]: a sum over the BATCH of pre-activation x upstream gradient, which the
//                  kernel that forms g has in LDS anyway (clv_sparse_outer's gdot) -- no pass over the 4 MB of W and g.
// Launch 1 (wn_fast_update): every block derives its columns' scalars itself and updates 64 rows; block 0 also steps the
// per-column Adam state of g.  Launch 2 (wn_fast_rescale): every block sums the 176 partial ||V'||^2 rows of its
// columns (62 KB, L2-resident), rescales its 64 rows; block 0 stores s', ||V'||^2 and advances `iterations`.
// Measured at config 3 (rocprofv3): 12.5 + 6.5 us against 34.6 us for the classic chain; 128-row blocks (88 of them):
// 16.2 + 5.8 (too few CUs stream), 64-row blocks of 256 threads: 17.2 + 11.3.
// The classic chain needs five launches (stats, columns, update, columns, rescale).
// ---------------------------------------------------------------------------
#ifndef CLV_FAST_ROWS
#define CLV_FAST_ROWS 64
#endif
constexpr int FAST_ROWS = CLV_FAST_ROWS;       // rows per workgroup: 176 workgroups for hW/kernel, i.e. 176 partial rows for the second launch
                                               // (round 4, -DCLV_FAST_ROWS: 48 rows = 235 workgroups 15.5 + 6.8 us, 32 = 352: 15.4 + 8.9, against 12.6 + 6.0)
constexpr int FAST_NT = 1024;       // 64 column-pair lanes x 16 row lanes
struct AdamFast {
  int64_t offset, col_offset;
  int rows, cols, nunits;
  const float* gdot;      // [cols] sum_b (X W)[b,c] dH[b,c]
  float* vn2;             // per-column ||V||^2 (global column layout like s / mg / vg)
  float* gnew;            // [cols] scratch: the new gain, launch 1 -> launch 2
  float* partC;           // [nunits][cols]
};

__device__ __forceinline__ void wn_fast_update_body(const AdamFast& f, int unit, float* params, const float* grads, float* m,
                                                    float* v, float* mg, float* vg, const float* s, const AdamHyper& h) {
  __shared__ float cs[3][128];
  __shared__ float2 red[16][64];
  const int tid = threadIdx.x, cx = tid & 63, ry = tid >> 6;
  const float lr_t = adam_lr_t(h);
  for (int c = tid; c < f.cols; c += FAST_NT) {             // this block's copy of the column scalars
    const size_t cg = f.col_offset + c;
    const float sc = s[cg], a = f.vn2[cg];
    const float Vn = sqrtf(a), inv_s = 1.f / sc;
    const float grad_g = f.gdot[c] * inv_s / Vn;           // sum g.V / ||V||
    cs[0][c] = inv_s; cs[1][c] = grad_g / Vn; cs[2][c] = sc;
    if (unit == 0) {                                        // Adam on the gain (utils/weightnorm.py:109-121), once per column
      const float mgn = h.b1 * mg[cg] + (1.f - h.b1) * grad_g;
      const float vgn = h.b2 * vg[cg] + (1.f - h.b2) * grad_g * grad_g;
      mg[cg] = mgn; vg[cg] = vgn;
      f.gnew[c] = sc * Vn - lr_t * mgn / (sqrtf(vgn) + h.eps);
    }
  }
  __syncthreads();
  const int row0 = unit * FAST_ROWS, nrows = min(FAST_ROWS, f.rows - row0), n2 = f.cols / 2;
  float2 acc = make_float2(0.f, 0.f);
  if (cx < n2) {
    // a thread owns a column pair and every 16th row: its 8 rows x 4 arrays are 32 eight-byte loads in flight at once
    const float2 inv_s = make_float2(cs[0][2 * cx], cs[0][2 * cx + 1]), gov = make_float2(cs[1][2 * cx], cs[1][2 * cx + 1]);
    const float2 sc = make_float2(cs[2][2 * cx], cs[2][2 * cx + 1]);
    float2 pv[FAST_ROWS / 16], gv[FAST_ROWS / 16], mv[FAST_ROWS / 16], vv[FAST_ROWS / 16];
#pragma unroll
    for (int i = 0; i < FAST_ROWS / 16; ++i) {
      const size_t o = f.offset + (size_t)(row0 + min(ry + 16 * i, nrows - 1)) * f.cols + 2 * cx;
      pv[i] = *reinterpret_cast<const float2*>(params + o); gv[i] = *reinterpret_cast<const float2*>(grads + o);
      mv[i] = *reinterpret_cast<const float2*>(m + o); vv[i] = *reinterpret_cast<const float2*>(v + o);
    }
    auto one = [&](float p, float g, float m0, float v0, float is, float go, float s0, float& mo, float& vo, float& a2) {
      const float V = p * is;
      const float gV = s0 * (g - go * V);
      mo = h.b1 * m0 + (1.f - h.b1) * gV;
      vo = h.b2 * v0 + (1.f - h.b2) * gV * gV;
      const float Vp = V - lr_t * mo / (sqrtf(vo) + h.eps);
      a2 += Vp * Vp;
      return Vp;
    };
#pragma unroll
    for (int i = 0; i < FAST_ROWS / 16; ++i) {
      const int r = ry + 16 * i;
      if (r < nrows) {
        const size_t o = f.offset + (size_t)(row0 + r) * f.cols + 2 * cx;
        float2 mo, vo, po;
        po.x = one(pv[i].x, gv[i].x, mv[i].x, vv[i].x, inv_s.x, gov.x, sc.x, mo.x, vo.x, acc.x);
        po.y = one(pv[i].y, gv[i].y, mv[i].y, vv[i].y, inv_s.y, gov.y, sc.y, mo.y, vo.y, acc.y);
        *reinterpret_cast<float2*>(m + o) = mo;
        *reinterpret_cast<float2*>(v + o) = vo;
        *reinterpret_cast<float2*>(params + o) = po;
      }
    }
  }
  red[ry][cx] = acc;
  __syncthreads();
  if (ry == 0 && cx < n2) {
    float2 t = make_float2(0.f, 0.f);
#pragma unroll
    for (int w = 0; w < 16; ++w) { t.x += red[w][cx].x; t.y += red[w][cx].y; }
    *reinterpret_cast<float2*>(f.partC + (size_t)unit * f.cols + 2 * cx) = t;
  }
}

#ifndef CLV_ADAM_FLAT
#define CLV_ADAM_FLAT 1      // 0: the float2 / row-lane form (round 3) for A/B runs
#endif
// The same pass with the tile taken as what it is in memory -- 64 rows x cols floats, CONTIGUOUS in each of the four arrays --
// and walked 16 bytes per lane (round 4: 12.6 -> 11.0 us at configuration 3; 3 / 4 float4 per thread on fewer threads: 11.8 / 12.6): thread t owns the float4s t and t + 1024 of the tile (cols % 4 == 0: a float4 never crosses a
// row, its columns are (4 t) % cols ..+3).  The per-column sums of V'^2 go through an LDS copy of the tile's squares.
__device__ __forceinline__ void wn_fast_update_body_flat(const AdamFast& f, int unit, float* params, const float* grads, float* m,
                                                         float* v, float* mg, float* vg, const float* s, const AdamHyper& h) {
  __shared__ __attribute__((aligned(16))) float cs[3][128];
  __shared__ __attribute__((aligned(16))) float sq[FAST_ROWS * 128];          // V'^2 of the tile, [row][col]; then [8][cols] partial column sums at its front
  const int tid = threadIdx.x;
  const float lr_t = adam_lr_t(h);
  for (int c = tid; c < f.cols; c += FAST_NT) {
    const size_t cg = f.col_offset + c;
    const float sc = s[cg], a = f.vn2[cg];
    const float Vn = sqrtf(a), inv_s = 1.f / sc;
    const float grad_g = f.gdot[c] * inv_s / Vn;
    cs[0][c] = inv_s; cs[1][c] = grad_g / Vn; cs[2][c] = sc;
    if (unit == 0) {
      const float mgn = h.b1 * mg[cg] + (1.f - h.b1) * grad_g;
      const float vgn = h.b2 * vg[cg] + (1.f - h.b2) * grad_g * grad_g;
      mg[cg] = mgn; vg[cg] = vgn;
      f.gnew[c] = sc * Vn - lr_t * mgn / (sqrtf(vgn) + h.eps);
    }
  }
  const int row0 = unit * FAST_ROWS, nrows = min(FAST_ROWS, f.rows - row0);
  const int n4 = nrows * f.cols / 4;
  const size_t o0 = f.offset + (size_t)row0 * f.cols;
#ifndef CLV_ADAM_FLAT_PT
#define CLV_ADAM_FLAT_PT 2
#endif
  constexpr int PT = CLV_ADAM_FLAT_PT;                    // float4s per thread
  constexpr int NTA = (FAST_ROWS * 128 / 4 + PT - 1) / PT < FAST_NT ? (FAST_ROWS * 128 / 4 + PT - 1) / PT : FAST_NT;      // threads that move data
  float4 pv[PT], gv[PT], mv[PT], vv[PT];
#pragma unroll
  for (int k = 0; k < PT; ++k) {                 // all loads of the thread in flight (clamped; masked below)
    const size_t o = o0 + 4 * (size_t)min(tid + NTA * k, n4 - 1);
    pv[k] = *reinterpret_cast<const float4*>(params + o); gv[k] = *reinterpret_cast<const float4*>(grads + o);
    mv[k] = *reinterpret_cast<const float4*>(m + o); vv[k] = *reinterpret_cast<const float4*>(v + o);
  }
  __syncthreads();                               // the column scalars are in LDS
#pragma unroll
  for (int k = 0; k < PT; ++k) {
    const int i = tid + NTA * k;
    if (tid < NTA && i < n4) {
      const int c0 = (4 * i) % f.cols;
      const float pin[4] = {pv[k].x, pv[k].y, pv[k].z, pv[k].w}, gin[4] = {gv[k].x, gv[k].y, gv[k].z, gv[k].w};
      const float min_[4] = {mv[k].x, mv[k].y, mv[k].z, mv[k].w}, vin[4] = {vv[k].x, vv[k].y, vv[k].z, vv[k].w};
      // the four columns' scalars as three 16-byte LDS reads (c0 is a multiple of 4): twelve 4-byte reads at a lane stride of
      // 4 floats put lanes l and l + 16 on one bank (SQ counters, round 4: half of this kernel's LDS cycles were conflicts)
      const float4 c_is = *reinterpret_cast<const float4*>(&cs[0][c0]), c_gg = *reinterpret_cast<const float4*>(&cs[1][c0]),
                   c_sc = *reinterpret_cast<const float4*>(&cs[2][c0]);
      const float is4[4] = {c_is.x, c_is.y, c_is.z, c_is.w}, gg4[4] = {c_gg.x, c_gg.y, c_gg.z, c_gg.w}, sc4[4] = {c_sc.x, c_sc.y, c_sc.z, c_sc.w};
      float po[4], mo[4], vo[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float V = pin[c] * is4[c];
        const float gV = sc4[c] * (gin[c] - gg4[c] * V);
        mo[c] = h.b1 * min_[c] + (1.f - h.b1) * gV;
        vo[c] = h.b2 * vin[c] + (1.f - h.b2) * gV * gV;
        po[c] = V - lr_t * mo[c] / (sqrtf(vo[c]) + h.eps);
      }
      const size_t o = o0 + 4 * (size_t)i;
      *reinterpret_cast<float4*>(m + o) = make_float4(mo[0], mo[1], mo[2], mo[3]);
      *reinterpret_cast<float4*>(v + o) = make_float4(vo[0], vo[1], vo[2], vo[3]);
      *reinterpret_cast<float4*>(params + o) = make_float4(po[0], po[1], po[2], po[3]);
      *reinterpret_cast<float4*>(sq + 4 * i) = make_float4(po[0] * po[0], po[1] * po[1], po[2] * po[2], po[3] * po[3]);
    }
  }
  __syncthreads();
  // column sums of the squares: 8 row lanes x cols column lanes, then the 8 partials
  const int cx = tid % f.cols, rl = tid / f.cols;
  float part = 0.f;
  if (rl < 8)
    for (int r = rl; r < nrows; r += 8) part += sq[r * f.cols + cx];
  __syncthreads();
  if (rl < 8) sq[rl * f.cols + cx] = part;
  __syncthreads();
  if (tid < f.cols) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) t += sq[w * f.cols + tid];
    f.partC[(size_t)unit * f.cols + tid] = t;
  }
}

__global__ __launch_bounds__(FAST_NT) void wn_fast_update_kernel(AdamFast f, const SmallItem* items, float* params,
                                                                 const float* grads, float* m, float* v, float* mg, float* vg,
                                                                 float* s, AdamHyper h) {
  if ((int)blockIdx.x >= f.nunits) {
    if (threadIdx.x >= 256) return;           // the small-tensor blocks are 256-thread work: the other waves leave at once
    wn_small_body(items[blockIdx.x - f.nunits], params, grads, m, v, mg, vg, s, h);
    return;
  }
#if CLV_ADAM_FLAT
  if (f.cols % 4 == 0 && f.offset % 4 == 0 && 8 * f.cols <= FAST_NT) {
    wn_fast_update_body_flat(f, (int)blockIdx.x, params, grads, m, v, mg, vg, s, h);
    return;
  }
#endif
  wn_fast_update_body(f, (int)blockIdx.x, params, grads, m, v, mg, vg, s, h);
}

__global__ __launch_bounds__(FAST_NT) void wn_fast_rescale_kernel(AdamFast f, float* params, float* s, int32_t* iterations) {
  __shared__ __attribute__((aligned(16))) float2 red[16][64], snew[64];
  const int tid = threadIdx.x, cx = tid & 63, ry = tid >> 6, unit = blockIdx.x, n2 = f.cols / 2;
  if (unit == 0 && tid == 0 && iterations) *iterations += 1;
#if CLV_ADAM_FLAT
  // the tile's rows (contiguous: see wn_fast_update_body_flat) are requested first: they do not depend on the column sums
  const bool flat = f.cols % 4 == 0 && f.offset % 4 == 0;
  const int frow0 = unit * FAST_ROWS, fn4 = min(FAST_ROWS, f.rows - frow0) * f.cols / 4;
  float* ftile = params + f.offset + (size_t)frow0 * f.cols;
  float4 fpv[2];
  if (flat) {
#pragma unroll
    for (int k = 0; k < 2; ++k) fpv[k] = *reinterpret_cast<const float4*>(ftile + 4 * (size_t)min(tid + FAST_NT * k, fn4 - 1));
  }
#endif
  float2 a = make_float2(0.f, 0.f);
  if (cx < n2)
    for (int k = ry; k < f.nunits; k += 16) {
      const float2 t = *reinterpret_cast<const float2*>(f.partC + (size_t)k * f.cols + 2 * cx);
      a.x += t.x; a.y += t.y;
    }
  red[ry][cx] = a;
  __syncthreads();
  if (ry == 0 && cx < n2) {
    float2 t = make_float2(0.f, 0.f);
#pragma unroll
    for (int w = 0; w < 16; ++w) { t.x += red[w][cx].x; t.y += red[w][cx].y; }
    const float2 sn = make_float2(f.gnew[2 * cx] / sqrtf(t.x), f.gnew[2 * cx + 1] / sqrtf(t.y));
    snew[cx] = sn;
    if (unit == 0) {
      *reinterpret_cast<float2*>(s + f.col_offset + 2 * cx) = sn;
      *reinterpret_cast<float2*>(f.vn2 + f.col_offset + 2 * cx) = t;
    }
  }
  __syncthreads();
#if CLV_ADAM_FLAT
  if (flat) {
    const float* sf = reinterpret_cast<const float*>(snew);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = tid + FAST_NT * k;
      if (i < fn4) {
        const int c0 = (4 * i) % f.cols;
        const float4 sn4 = *reinterpret_cast<const float4*>(sf + c0);      // one 16-byte read (see wn_fast_update_body_flat)
        *reinterpret_cast<float4*>(ftile + 4 * (size_t)i) = make_float4(fpv[k].x * sn4.x, fpv[k].y * sn4.y, fpv[k].z * sn4.z, fpv[k].w * sn4.w);
      }
    }
    return;
  }
#endif
  const int row0 = unit * FAST_ROWS, nrows = min(FAST_ROWS, f.rows - row0);
  if (cx < n2) {
    const float2 sn = snew[cx];
    float2 pv[FAST_ROWS / 16];
#pragma unroll
    for (int i = 0; i < FAST_ROWS / 16; ++i)
      pv[i] = *reinterpret_cast<const float2*>(params + f.offset + (size_t)(row0 + min(ry + 16 * i, nrows - 1)) * f.cols + 2 * cx);
#pragma unroll
    for (int i = 0; i < FAST_ROWS / 16; ++i)
      if (ry + 16 * i < nrows)
        *reinterpret_cast<float2*>(params + f.offset + (size_t)(row0 + ry + 16 * i) * f.cols + 2 * cx) =
            make_float2(pv[i].x * sn.x, pv[i].y * sn.y);
  }
}

struct PlanCounts { int n_units, n_cols, n_part, n_small, n_big; };

static bool is_small(const clv_param_desc& t) { return !t.is_matrix || t.rows <= SM_ROWS; }
static PlanCounts plan_counts(const clv_param_desc* t, int n) {
  PlanCounts c{0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const int units = (t[i].rows + UNIT_ROWS - 1) / UNIT_ROWS;
    c.n_units += units;
    if (t[i].is_matrix) { c.n_cols += t[i].cols; c.n_part += units * t[i].cols; }
    if (is_small(t[i])) c.n_small += t[i].is_matrix ? (t[i].cols + 15) / 16 : 1;
    else c.n_big += 1;
  }
  return c;
}

}  // namespace clv

using namespace clv;

// device blob: AdamUnit[n_units] | AdamCol[n_cols] | int32 colidx0[n_units] | SmallItem[n_small]
extern "C" size_t clv_adam_wn_plan_bytes(const clv_param_desc* host_table, int n_tensors) {
  if (!host_table || n_tensors <= 0) return 0;
  PlanCounts c = plan_counts(host_table, n_tensors);
  return align_up(sizeof(AdamUnit) * c.n_units, 16) + align_up(sizeof(AdamCol) * (c.n_cols > 0 ? c.n_cols : 1), 16) +
         align_up(sizeof(int32_t) * c.n_units, 16) + align_up(sizeof(SmallItem) * (c.n_small > 0 ? c.n_small : 1), 16);
}

extern "C" int clv_adam_wn_plan_build(const clv_param_desc* host_table, int n_tensors, void* host_blob) {
  if (!host_table || n_tensors <= 0 || !host_blob) return CLV_EINVAL;
  PlanCounts c = plan_counts(host_table, n_tensors);
  char* p = (char*)host_blob;
  AdamUnit* units = (AdamUnit*)p;
  p += align_up(sizeof(AdamUnit) * c.n_units, 16);
  AdamCol* cols = (AdamCol*)p;
  p += align_up(sizeof(AdamCol) * (c.n_cols > 0 ? c.n_cols : 1), 16);
  int32_t* colidx0 = (int32_t*)p;
  p += align_up(sizeof(int32_t) * c.n_units, 16);
  SmallItem* small = (SmallItem*)p;
  int ui = 0, ci = 0, part = 0, si = 0;
  for (int i = 0; i < n_tensors; ++i) {
    const clv_param_desc& t = host_table[i];
    if (t.rows <= 0 || t.cols <= 0) return CLV_EINVAL;
    if (is_small(t)) {
      if (t.is_matrix)
        for (int c0 = 0; c0 < t.cols; c0 += 16) small[si++] = SmallItem{t.offset, t.col_offset, t.rows, t.cols, c0, 1};
      else small[si++] = SmallItem{t.offset, 0, t.rows, t.cols, 0, 0};
    }
    const int nun = (t.rows + UNIT_ROWS - 1) / UNIT_ROWS;
    const int part_base = part;
    for (int k = 0; k < nun; ++k) {
      AdamUnit& u = units[ui];
      u.offset = t.offset; u.col_offset = t.col_offset;
      u.row0 = k * UNIT_ROWS;
      u.nrows = (t.rows - u.row0) < UNIT_ROWS ? (t.rows - u.row0) : UNIT_ROWS;
      u.cols = t.cols; u.is_matrix = t.is_matrix;
      u.part_off = t.is_matrix ? part : 0;
      u.small = is_small(t) ? 1 : 0;
      colidx0[ui] = t.is_matrix ? ci : 0;
      if (t.is_matrix) part += t.cols;
      ++ui;
    }
    if (t.is_matrix) {
      for (int cidx = 0; cidx < t.cols; ++cidx) {
        AdamCol& c2 = cols[ci + cidx];
        c2.col_global = t.col_offset + cidx;
        c2.part_base = part_base; c2.nunits = is_small(t) ? -1 : nun; c2.cols = t.cols; c2.col_local = cidx;
      }
      ci += t.cols;
    }
  }
  return CLV_OK;
}

extern "C" size_t clv_adam_wn_workspace_bytes(const clv_param_desc* host_table, int n_tensors) {
  if (!host_table || n_tensors <= 0) return 0;
  PlanCounts c = plan_counts(host_table, n_tensors);
  return (size_t)(3 * c.n_part + 4 * c.n_cols + 16) * sizeof(float);
}

extern "C" int clv_adam_wn_step(const clv_param_desc* host_table, int n_tensors, const void* plan_dev,
                                   float* params, const float* grads, float* m, float* v,
                                   float* mg, float* vg, float* s,
                                   int32_t* iterations_dev, int step_t, float lr, float beta1, float beta2, float eps,
                                   int weightnorm, const clv_adam_known_sums* known, void* ws, size_t ws_bytes, void* stream) {
  if (!host_table || n_tensors <= 0 || !plan_dev || !params || !grads || !m || !v) return CLV_EINVAL;
  float* vn2 = known ? known->vnorm2 : nullptr;
  if (weightnorm < 0 || weightnorm > CLV_OPT_RMSPROP) return CLV_EINVAL;
  if (weightnorm == CLV_OPT_ADAM_WN && (!mg || !vg || !s)) return CLV_EINVAL;
  PlanCounts c = plan_counts(host_table, n_tensors);
  if (!ws || ws_bytes < clv_adam_wn_workspace_bytes(host_table, n_tensors)) return CLV_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const char* p = (const char*)plan_dev;
  const AdamUnit* units = (const AdamUnit*)p;
  p += align_up(sizeof(AdamUnit) * c.n_units, 16);
  const AdamCol* cols = (const AdamCol*)p;
  p += align_up(sizeof(AdamCol) * (c.n_cols > 0 ? c.n_cols : 1), 16);
  const int32_t* colidx0 = (const int32_t*)p;
  p += align_up(sizeof(int32_t) * c.n_units, 16);
  const SmallItem* small = (const SmallItem*)p;
  float* partA = (float*)ws;
  float* partB = partA + c.n_part;
  float* partC = partB + c.n_part;
  float* colscal = partC + c.n_part;
  AdamHyper h{lr, beta1, beta2, eps, weightnorm, step_t, iterations_dev};
  // step_t == -1 with a device counter: read it, leave it alone (another call of the same step, on another subset of the
  // tensors, advances it)
  int32_t* bump = (iterations_dev && (step_t == -1 || step_t == CLV_STEP_ADVANCED)) ? nullptr : iterations_dev;
  ProfScope pr("adam_wn_step", st);
  const bool wn = weightnorm == CLV_OPT_ADAM_WN && c.n_cols > 0;
  if (known && known->use && wn) {
    // the two-launch form: exactly one tall matrix, whose sum g.V the caller brings and whose ||V||^2 vn2 holds
    const int ti = known->tensor;
    if (ti < 0 || ti >= n_tensors || c.n_big != 1 || is_small(host_table[ti]) || !known->gdot || !vn2) return CLV_EINVAL;
    const clv_param_desc& t = host_table[ti];
    if (t.cols > 128 || t.cols % 2 || t.offset % 2 || t.col_offset % 2) return CLV_EINVAL;      // 8-byte accesses
    const int nunits = (t.rows + FAST_ROWS - 1) / FAST_ROWS;
    if ((size_t)nunits * t.cols > (size_t)c.n_part) return CLV_EWORKSPACE;      // partC lives where the classic chain keeps its own
    AdamFast f{t.offset, t.col_offset, t.rows, t.cols, nunits, known->gdot, vn2, colscal, partC};
    hipLaunchKernelGGL(wn_fast_update_kernel, dim3(nunits + c.n_small), dim3(FAST_NT), 0, st, f, small, params, grads, m, v, mg,
                       vg, s, h);
    hipLaunchKernelGGL(wn_fast_rescale_kernel, dim3(nunits), dim3(FAST_NT), 0, st, f, params, s, bump);
    return launch_status();
  }
  const bool chain = !wn || c.n_big > 0;      // tall matrices (partial slabs), or plain Adam for everything
  const int n_small = wn ? c.n_small : 0;     // small matrices and biases: whole update in their own blocks of K3
  if (wn && chain) {
    hipLaunchKernelGGL(wn_stats_kernel, dim3(c.n_units), dim3(256), 0, st, units, params, grads, s, partA, partB);
    hipLaunchKernelGGL(wn_cols_kernel, dim3((c.n_cols + CPB - 1) / CPB), dim3(256), 0, st, c.n_cols, cols, partA, partB, s,
                       mg, vg, colscal, h);
  }
  {
    const int n_units = chain ? c.n_units : 0;
    if (n_units + n_small > 0)
      hipLaunchKernelGGL(wn_update_kernel, dim3(n_units + n_small), dim3(256), 0, st, n_units, units, small, params, grads, m,
                         v, mg, vg, s, colscal, colidx0, partC, h);
  }
  if (wn && chain) {
    hipLaunchKernelGGL(wn_cols2_kernel, dim3((c.n_cols + CPB - 1) / CPB), dim3(256), 0, st, c.n_cols, cols, partC, s, colscal,
                       bump, vn2);
    hipLaunchKernelGGL(wn_rescale_kernel, dim3(c.n_units), dim3(256), 0, st, units, params, colscal, colidx0);
  } else if (bump) {
    hipLaunchKernelGGL(adam_bump_kernel, dim3(1), dim3(1), 0, st, bump);
  }
  return launch_status();
}
