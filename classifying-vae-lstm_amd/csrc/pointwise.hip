// pointwise.hip -- reparameterisation, the four loss terms and their gradients,
// deterministic reductions.  All HBM-bound row kernels: one thread per row for the
// tiny label/latent heads (C <= 32, L <= 64), one wave per row for the 88-note
// Bernoulli NLL (wave64 reduction), partial-slab column sums for bias gradients.
#include <string.h>

#include "common.h"

namespace clv {

constexpr float EPS_K = 1e-7f;                 // keras.backend._EPSILON
constexpr float W2_SHIFT = 1e-10f;             // cl_vae/model.py:208
constexpr int MAXC = 32;

// ------------------------------------------------------------------ label --
__global__ void label_fwd_kernel(int B, int C, const float* mean, const float* logvar, int ld_in,
                                 const float* eps, const float* onehot, float prior,
                                 float* w, float* rowloss) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int C1 = C - 1;
  float e[MAXC];
  float S = 1.f;   // exp(0) of the appended zero
  float klw = 0.f;
  const float ep = __expf(prior);
  for (int j = 0; j < C1; ++j) {
    const float m = mean[(size_t)b * ld_in + j], lv = logvar[(size_t)b * ld_in + j];
    const float sd = expf(0.5f * lv);
    e[j] = expf(m + sd * eps[(size_t)b * C1 + j]);
    S += e[j];
    klw += 1.f - prior + lv - sd * sd / ep - m * m / ep;
  }
  e[C1] = 1.f;
  const float invS = 1.f / S;
  float wv[MAXC];
  float qs = 0.f;
  int amax = 0, tmax = 0;
  float wbest = -1.f, tbest = -1.f;
  for (int j = 0; j < C; ++j) {
    wv[j] = e[j] * invS;
    w[(size_t)b * C + j] = wv[j];
    qs += wv[j] + W2_SHIFT;
    if (wv[j] > wbest) { wbest = wv[j]; amax = j; }
    const float tj = onehot ? onehot[(size_t)b * C + j] : 0.f;
    if (tj > tbest) { tbest = tj; tmax = j; }
  }
  if (rowloss) {
    float wrec = 0.f;
    if (onehot) {
      for (int j = 0; j < C; ++j) {
        const float n = (wv[j] + W2_SHIFT) / qs;
        const float nc = fminf(fmaxf(n, EPS_K), 1.f - EPS_K);
        wrec -= onehot[(size_t)b * C + j] * logf(nc);
      }
    }
    rowloss[(size_t)b * 3 + 0] = -0.5f * klw;
    rowloss[(size_t)b * 3 + 1] = (float)C1 * wrec;
    rowloss[(size_t)b * 3 + 2] = (onehot && amax == tmax) ? 1.f : 0.f;
  }
}

__global__ void label_bwd_kernel(int B, int C, const float* mean, const float* logvar, int ld_in,
                                 const float* eps, const float* onehot, const float* w, const float* dw,
                                 float prior, float class_weight, float w_kl_weight, float inv_b,
                                 float* dmean, float* dlogvar, int ld_out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int C1 = C - 1;
  const float ep = __expf(prior);
  // d(w_rec)/dw through renormalise + clip
  float wv[MAXC], d[MAXC];
  float qs = 0.f;
  for (int j = 0; j < C; ++j) { wv[j] = w[(size_t)b * C + j]; qs += wv[j] + W2_SHIFT; }
  float dn[MAXC];
  float dot = 0.f;
  for (int j = 0; j < C; ++j) {
    const float n = (wv[j] + W2_SHIFT) / qs;
    const bool inside = (n >= EPS_K) && (n <= 1.f - EPS_K);
    const float nc = fminf(fmaxf(n, EPS_K), 1.f - EPS_K);
    dn[j] = inside ? -(float)C1 * onehot[(size_t)b * C + j] / nc : 0.f;
    dot += dn[j] * n;
  }
  float dsum = 0.f;
  for (int j = 0; j < C; ++j) {
    const float drec = (dn[j] - dot) / qs;
    d[j] = dw[(size_t)b * C + j] + class_weight * inv_b * drec;
    dsum += d[j] * wv[j];
  }
  for (int j = 0; j < C1; ++j) {
    const float ds = wv[j] * (d[j] - dsum);          // softmax backward, appended zero dropped
    const float m = mean[(size_t)b * ld_in + j], lv = logvar[(size_t)b * ld_in + j];
    const float sd = expf(0.5f * lv);
    dmean[(size_t)b * ld_out + j] = ds + w_kl_weight * inv_b * (m / ep);
    dlogvar[(size_t)b * ld_out + j] = ds * eps[(size_t)b * C1 + j] * 0.5f * sd
                                      + w_kl_weight * inv_b * (-0.5f * (1.f - sd * sd / ep));
  }
}

// ------------------------------------------------------------------ gauss --
// one thread per (row, latent) element, LP = next power of two >= L lanes per row (coalesced for any L <= 64);
// the row's KL term is a shuffle reduction over those LP lanes.
template <int LP>
__global__ __launch_bounds__(256) void gauss_fwd_kernel(int R, int L, const float* zargs, const float* eps, float* z,
                                                        int ldz, float* rowkl) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t r = gid / LP;
  const int j = (int)(gid % LP);
  float term = 0.f;
  if (r < R && j < L) {
    const float m = zargs[r * 2 * L + j], lv = zargs[r * 2 * L + L + j];
    const float sd = expf(0.5f * lv);
    z[r * ldz + j] = m + sd * eps[r * L + j];
    term = 1.f + lv - m * m - sd * sd;
  }
#pragma unroll
  for (int o = LP / 2; o > 0; o >>= 1) term += __shfl_xor(term, o, 64);
  if (rowkl && r < R && j == 0) rowkl[r] = -0.5f * term;
}

__global__ __launch_bounds__(256) void gauss_bwd_kernel(int R, int L, const float* zargs, const float* eps,
                                                        const float* dz, int lddz, float kl_scale, float* dzargs) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (int64_t)R * L) return;
  const int64_t r = gid / L;
  const int j = (int)(gid % L);
  const float m = zargs[r * 2 * L + j], lv = zargs[r * 2 * L + L + j];
  const float sd = expf(0.5f * lv);
  const float d = dz[r * lddz + j];
  dzargs[r * 2 * L + j] = d + kl_scale * m;
  dzargs[r * 2 * L + L + j] = d * eps[r * L + j] * 0.5f * sd - 0.5f * kl_scale * (1.f - sd * sd);
}

// ------------------------------------------------------- Bernoulli NLL (BCE) --
// one wave per row, 4 rows per 256-thread block
__global__ __launch_bounds__(256) void bernoulli_nll_kernel(int R, int D, const float* logits, const float* y, int ldy,
                                                            float scale, float* rownll, float* dlogits) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= R) return;
  float acc = 0.f;
  for (int j = lane; j < D; j += 64) {
    const float a = logits[(size_t)row * D + j];
    const float t = y[(size_t)row * ldy + j];
    const float l = fminf(fmaxf(a, BCE_CLIP_LO), BCE_CLIP_HI);
    // softplus(l) = max(l,0) + log(1 + e^-|l|); e = e^-|l| in (1e-7, 1] so 1+e is exact enough for v_log_f32
    const float e = __expf(-fabsf(l));
    const float sp = fmaxf(l, 0.f) + __logf(1.f + e);
    acc += sp - l * t;
    if (dlogits) {
      const bool inside = (a >= BCE_CLIP_LO) && (a <= BCE_CLIP_HI);
      const float r1 = fast_rcp(1.f + e);
      const float sg = l >= 0.f ? r1 : e * r1;          // sigmoid(l) from the same exponential
      dlogits[(size_t)row * D + j] = inside ? scale * (sg - t) : 0.f;
    }
  }
  acc = wave_sum(acc);
  if (lane == 0 && rownll) rownll[row] = acc;
}

// ------------------------------------------------------------- reductions --
// single block, fixed order => deterministic
__global__ __launch_bounds__(1024) void sum_strided_kernel(int n, const float* x, int stride, float scale, float* out) {
  __shared__ float part[16];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) acc += x[(size_t)i * stride];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < 16; ++i) t += part[i];
    out[0] = t * scale;
  }
}

struct SumArgs { const float* x[5]; int n[5]; int stride[5]; };
// out[k] = mean of x[k] (n[k] strided elements); one block per term
__global__ __launch_bounds__(1024) void loss_sums_kernel(SumArgs a, float* out) {
  __shared__ float part[16];
  const int k = blockIdx.x;
  const float* x = a.x[k];
  const int n = a.n[k], st = a.stride[k];
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
  int i = threadIdx.x;
  if (st == 1 && ((uintptr_t)x) % 16 == 0) {        // contiguous term: float4 loads, 4 in flight per thread
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const int n4 = n / 4;
    int j = threadIdx.x;
    for (; j + 3072 < n4; j += 4096) {
      const float4 a = x4[j], b = x4[j + 1024], c = x4[j + 2048], d = x4[j + 3072];
      acc0 += (a.x + a.y) + (a.z + a.w); acc1 += (b.x + b.y) + (b.z + b.w);
      acc2 += (c.x + c.y) + (c.z + c.w); acc3 += (d.x + d.y) + (d.z + d.w);
    }
    for (; j < n4; j += 1024) { const float4 a = x4[j]; acc0 += (a.x + a.y) + (a.z + a.w); }
    i = 4 * n4 + threadIdx.x;                        // tail elements
  }
  for (; i + 3072 < n; i += 4096) {
    acc0 += x[(size_t)i * st]; acc1 += x[(size_t)(i + 1024) * st];
    acc2 += x[(size_t)(i + 2048) * st]; acc3 += x[(size_t)(i + 3072) * st];
  }
  for (; i < n; i += 1024) acc0 += x[(size_t)i * st];
  float acc = wave_sum((acc0 + acc1) + (acc2 + acc3));
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int j = 0; j < 16; ++j) t += part[j];
    out[k] = t / (float)n;
  }
}

// column sums, stage 1: block (64 cols x 4 row-lanes) handles a chunk of rows
constexpr int CS_ROWS = 64;
__global__ __launch_bounds__(256) void colsum_partial_kernel(int M, int N, const float* X, int ldx, float* partial) {
  __shared__ float red[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cx;
  const int r0 = blockIdx.y * CS_ROWS, r1 = min(M, r0 + CS_ROWS);
  float acc = 0.f;
  if (col < N)
    for (int r = r0 + ry; r < r1; r += 4) acc += X[(size_t)r * ldx + col];
  red[ry][cx] = acc;
  __syncthreads();
  if (ry == 0 && col < N) partial[(size_t)blockIdx.y * N + col] = red[0][cx] + red[1][cx] + red[2][cx] + red[3][cx];
}
// stage 2: 64 columns x 4 chunk-lanes per block, 4 loads in flight per thread
__global__ __launch_bounds__(256) void colsum_final_kernel(int N, int chunks, const float* partial, float beta, float* out) {
  __shared__ float red[4][64];
  const int cx = threadIdx.x & 63, zy = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cx;
  float acc = 0.f;
  if (col < N) {
    const float* p = partial + col;
    int c = zy;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (; c + 12 < chunks; c += 16) {
      a0 += p[(size_t)(c + 0) * N]; a1 += p[(size_t)(c + 4) * N];
      a2 += p[(size_t)(c + 8) * N]; a3 += p[(size_t)(c + 12) * N];
    }
    for (; c < chunks; c += 4) a0 += p[(size_t)c * N];
    acc = (a0 + a1) + (a2 + a3);
  }
  red[zy][cx] = acc;
  __syncthreads();
  if (zy == 0 && col < N)
    out[col] = (beta != 0.f ? beta * out[col] : 0.f) + ((red[0][cx] + red[1][cx]) + (red[2][cx] + red[3][cx]));
}

// few rows (a batch of row vectors): one launch, 64 columns x 16 row-lanes per block
__global__ __launch_bounds__(1024) void colsum_small_kernel(int M, int N, const float* X, int ldx, float beta, float* out) {
  __shared__ float red[16][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cx;
  float a0 = 0.f, a1 = 0.f;
  if (col < N) {
    int r = ry;
    for (; r + 16 < M; r += 32) { a0 += X[(size_t)r * ldx + col]; a1 += X[(size_t)(r + 16) * ldx + col]; }
    if (r < M) a0 += X[(size_t)r * ldx + col];
  }
  red[ry][cx] = a0 + a1;
  __syncthreads();
  if (ry == 0 && col < N) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[i][cx];
    out[col] = (beta != 0.f ? beta * out[col] : 0.f) + t;
  }
}

__global__ void axpy_kernel(int64_t n, float alpha, const float* x, float* y) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] += alpha * x[i];
}

// dpre = dy * act'(.) written in terms of the activation's OUTPUT y: relu -> [y > 0], sigmoid -> y (1 - y)
__global__ void act_grad_kernel(int64_t n, int act, const float* y, const float* dy, float* dpre) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = y[i], g = dy[i];
  dpre[i] = act == CLV_ACT_RELU ? (v > 0.f ? g : 0.f) : act == CLV_ACT_SIGMOID ? g * v * (1.f - v) : g;
}

// out[r, :] = src[idx[r], :]  (row gather: mini-batch assembly from the HBM-resident data set).  A source row
// is a sequence of `chunk`-float pieces (frames); piece j of row r lands at out + (r*pieces + j)*out_ld, so
// the history frames can be written straight into the [Xp | Z] decoder-input buffer (out_ld > chunk).
template <bool VEC>
__global__ void gather_rows_kernel(int64_t rows, int64_t row_elems, const float* src, const int64_t* idx, float* out,
                                   int64_t chunk, int64_t out_ld) {
  constexpr int W = VEC ? 4 : 1;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nw = row_elems / W;
  if (i >= rows * nw) return;
  const int64_t r = i / nw, c = (i % nw) * W;
  const int64_t piece = c / chunk, within = c % chunk;
  const float* sp = src + idx[r] * row_elems + c;
  float* dp = out + (r * (row_elems / chunk) + piece) * out_ld + within;
  if (VEC) *reinterpret_cast<float4*>(dp) = *reinterpret_cast<const float4*>(sp);
  else *dp = *sp;
}

// up to 3 gathers that share the row index list (current frames, history frames, labels of a mini-batch) in one
// launch; idx == nullptr: rows row0 .. row0+rows-1 (staging a contiguous batch)
// a source row starts at element table[idx] * stride + offset (table optional: then idx itself; stride defaults to
// row_elems): overlapping windows of a frame store are rows of stride one frame
struct GatherSeg { const float* src; float* out; const int64_t* table; int64_t row_elems, out_ld, stride, offset;
                   int chunk, pieces, vec, u8; unsigned char* notes; };
struct GatherArgs { GatherSeg seg[4]; int nseg; int64_t rows; const int64_t* idx; int64_t row0; int nlist; int list_seg[4];
                    // batch cursor (clv_gather_rows_multi): the launch reads the device step counter and takes batch
                    // j = (step - step0) mod period: rows j * cur_stride + cur_offset .. of the row list
                    const int32_t* step_dev; int32_t step0, period; int64_t cur_stride, cur_offset; };
__device__ __forceinline__ int64_t gather_base(const GatherArgs& a) {
  if (!a.step_dev) return 0;
  int j = (*a.step_dev - a.step0) % a.period;
  j = j < 0 ? j + a.period : j;
  return (int64_t)j * a.cur_stride + a.cur_offset;
}

// Note lists (clv_gather_rows_multi, notes_out): frame p of output row r -> notes[(r * pieces + p) * CLV_NOTE_ROW ..]: the
// indices of the bytes that are not zero, then CLV_NOTE_NONE up to the end of the row (at least 8 of them: a reader that
// walks the row 4 bytes at a time always meets the terminator).  One wave per frame: lanes 0..21 take 4 bytes each, the
// position of a note in the list is a prefix count over four ballots (the order of a list is byte-major, not
// ascending: the consumer sums kernel rows, any order will do).
__device__ __forceinline__ void gather_note_lists(const GatherArgs& a, const GatherSeg& sg) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int piece = blockIdx.x * 4 + wave;
  if (piece >= sg.pieces) return;
  const int64_t base = gather_base(a);
  for (int64_t r = blockIdx.y; r < a.rows; r += gridDim.y) {
    int64_t sr = a.idx ? a.idx[base + r] : a.row0 + base + r;
    if (sg.table) sr = sg.table[sr];
    const unsigned char* sp = reinterpret_cast<const unsigned char*>(sg.src) + sr * sg.stride + sg.offset + (int64_t)piece * sg.chunk;
    const unsigned v = lane < sg.chunk / 4 ? *reinterpret_cast<const unsigned*>(sp + 4 * lane) : 0u;
    unsigned char* out = sg.notes + ((size_t)r * sg.pieces + piece) * CLV_NOTE_ROW;
    int before = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool on = ((v >> (8 * j)) & 255u) != 0u;
      const unsigned long long m = __ballot(on);
      if (on) out[before + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned char)(4 * lane + j);
      before += __popcll(m);
    }
    for (int p = before + lane; p < CLV_NOTE_ROW; p += 64) out[p] = (unsigned char)CLV_NOTE_NONE;
  }
}
// grid = (work items of a row / (256 GATHER_IPT), rows, segments): no 64-bit divisions per element (they cost more than the copy)
constexpr int GATHER_IPT = 4;      // (8: 10.5 us at configuration 3 against 9.05, the same at configuration 5)
__global__ __launch_bounds__(256) void gather_multi_kernel(GatherArgs a) {
  if ((int)blockIdx.z >= a.nseg) {            // the z-slices behind the copies build note lists
    const int li = blockIdx.z - a.nseg;
    int k = a.list_seg[0];
#pragma unroll
    for (int q = 1; q < 4; ++q) k = li == q ? a.list_seg[q] : k;
    GatherSeg sg = a.seg[0];
#pragma unroll
    for (int q = 1; q < 4; ++q) if (k == q) sg = a.seg[q];
    gather_note_lists(a, sg);
    return;
  }
  const int si = blockIdx.z;
  const float* src = a.seg[0].src; float* out = a.seg[0].out;
  const int64_t* table = a.seg[0].table;
  int64_t row_elems = a.seg[0].row_elems, out_ld = a.seg[0].out_ld, stride = a.seg[0].stride, offset = a.seg[0].offset;
  int chunk = a.seg[0].chunk, pieces = a.seg[0].pieces, vec = a.seg[0].vec, u8 = a.seg[0].u8;
#pragma unroll
  for (int k = 1; k < 4; ++k)
    if (si == k) { src = a.seg[k].src; out = a.seg[k].out; table = a.seg[k].table; row_elems = a.seg[k].row_elems;
                   out_ld = a.seg[k].out_ld; stride = a.seg[k].stride; offset = a.seg[k].offset;
                   chunk = a.seg[k].chunk; pieces = a.seg[k].pieces; vec = a.seg[k].vec; u8 = a.seg[k].u8; }
  const int W = vec == 2 ? 8 : (vec ? 4 : 1);      // work item = 4 elements (aligned segments) or one; 8 bytes of a byte copy
  // GATHER_IPT items per thread, 256 apart: the row's source offset is a chain of dependent loads (step counter -> row list
  // -> window table), paid once per block, and a thread's loads are issued together (round 4: one item per thread made the
  // launch 8448 blocks of one chain + one load each at configuration 3: 11.9 us for 30 MB)
  unsigned cs[GATHER_IPT];
#pragma unroll
  for (int k = 0; k < GATHER_IPT; ++k) cs[k] = ((blockIdx.x * GATHER_IPT + k) * 256u + threadIdx.x) * (unsigned)W;
  if ((int64_t)cs[0] >= row_elems) return;
  const int64_t base = gather_base(a);
  for (int64_t r = blockIdx.y; r < a.rows; r += gridDim.y) {
    int64_t sr = a.idx ? a.idx[base + r] : a.row0 + base + r;
    if (table) sr = table[sr];
    const int64_t s0 = sr * stride + offset;       // first element of the source row
    if (vec == 2) {                                // bytes as they are, 8 per work item (a batch that stays uint8; out_ld in bytes)
      uint2 v[GATHER_IPT];
#pragma unroll
      for (int k = 0; k < GATHER_IPT; ++k) {
        const unsigned c = (int64_t)cs[k] < row_elems ? cs[k] : cs[0];
        v[k] = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned char*>(src) + s0 + c);
      }
#pragma unroll
      for (int k = 0; k < GATHER_IPT; ++k) {
        const unsigned c = cs[k];
        if ((int64_t)c < row_elems) {
          unsigned piece = 0, within = c;
          if (pieces > 1) { piece = c / (unsigned)chunk; within = c - piece * (unsigned)chunk; }
          *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(out) + (r * pieces + piece) * out_ld + within) = v[k];
        }
      }
      continue;
    }
    if (u8 && vec) {                               // uint8 store (binary piano-roll frames): 4 bytes in, one float4 out
      unsigned int v[GATHER_IPT];
#pragma unroll
      for (int k = 0; k < GATHER_IPT; ++k) {
        const unsigned c = (int64_t)cs[k] < row_elems ? cs[k] : cs[0];
        v[k] = *reinterpret_cast<const unsigned int*>(reinterpret_cast<const unsigned char*>(src) + s0 + c);
      }
#pragma unroll
      for (int k = 0; k < GATHER_IPT; ++k) {
        const unsigned c = cs[k];
        if ((int64_t)c < row_elems) {
          unsigned piece = 0, within = c;
          if (pieces > 1) { piece = c / (unsigned)chunk; within = c - piece * (unsigned)chunk; }
          if (u8 == 2) {                           // ... or the four bytes as they are (a batch that stays uint8; out_ld in bytes)
            *reinterpret_cast<unsigned int*>(reinterpret_cast<unsigned char*>(out) + (r * pieces + piece) * out_ld + within) = v[k];
            continue;
          }
          *reinterpret_cast<float4*>(out + (r * pieces + piece) * out_ld + within) =
              make_float4((float)(v[k] & 255u), (float)((v[k] >> 8) & 255u), (float)((v[k] >> 16) & 255u), (float)(v[k] >> 24));
        }
      }
      continue;
    }
#pragma unroll
    for (int k = 0; k < GATHER_IPT; ++k) {
      const unsigned c = cs[k];
      if ((int64_t)c >= row_elems) break;
      unsigned piece = 0, within = c;
      if (pieces > 1) { piece = c / (unsigned)chunk; within = c - piece * (unsigned)chunk; }
      float* dp = out + (r * pieces + piece) * out_ld + within;
      if (u8) {
        *dp = (float)*(reinterpret_cast<const unsigned char*>(src) + s0 + c);
      } else {
        const float* sp = src + s0 + c;
        if (vec) *reinterpret_cast<float4*>(dp) = *reinterpret_cast<const float4*>(sp);
        else *dp = *sp;
      }
    }
  }
}

__global__ void bernoulli_sample_kernel(int64_t n, const float* p, const float* u, float* x) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = (u[i] <= p[i]) ? 1.f : 0.f;
}

}  // namespace clv

using namespace clv;

extern "C" int clv_label_fwd(int B, int C, const float* mean, const float* logvar, int ld_in,
                             const float* eps, const float* onehot, float prior_logvar,
                             float* w, float* rowloss, void* stream) {
  if (B <= 0 || C < 2 || C > MAXC || !mean || !logvar || !eps || !w) return CLV_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  ProfScope p("label_fwd", s);
  hipLaunchKernelGGL(label_fwd_kernel, dim3((B + 63) / 64), dim3(64), 0, s, B, C, mean, logvar, ld_in, eps, onehot,
                     prior_logvar, w, rowloss);
  return launch_status();
}

extern "C" int clv_label_bwd(int B, int C, const float* mean, const float* logvar, int ld_in,
                             const float* eps, const float* onehot, const float* w, const float* dw,
                             float prior_logvar, float class_weight, float w_kl_weight, float inv_b,
                             float* dmean, float* dlogvar, int ld_out, void* stream) {
  if (B <= 0 || C < 2 || C > MAXC || !mean || !logvar || !eps || !onehot || !w || !dw || !dmean || !dlogvar)
    return CLV_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  ProfScope p("label_bwd", s);
  hipLaunchKernelGGL(label_bwd_kernel, dim3((B + 63) / 64), dim3(64), 0, s, B, C, mean, logvar, ld_in, eps, onehot, w,
                     dw, prior_logvar, class_weight, w_kl_weight, inv_b, dmean, dlogvar, ld_out);
  return launch_status();
}

extern "C" int clv_gauss_fwd(int R, int L, const float* zargs, const float* eps, float* z, int ldz,
                             float* rowkl, void* stream) {
  if (R <= 0 || L <= 0 || L > 64 || !zargs || !eps || !z) return CLV_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  ProfScope p("gauss_fwd", s);
  int lp = 1;
  while (lp < L) lp <<= 1;
  const unsigned blocks = (unsigned)(((int64_t)R * lp + 255) / 256);
#define GAUSS_CASE(LPV) case LPV: hipLaunchKernelGGL(gauss_fwd_kernel<LPV>, dim3(blocks), dim3(256), 0, s, R, L, zargs, eps, z, ldz, rowkl); break;
  switch (lp) { GAUSS_CASE(1) GAUSS_CASE(2) GAUSS_CASE(4) GAUSS_CASE(8) GAUSS_CASE(16) GAUSS_CASE(32) GAUSS_CASE(64) }
#undef GAUSS_CASE
  return launch_status();
}

extern "C" int clv_gauss_bwd(int R, int L, const float* zargs, const float* eps, const float* dz, int lddz,
                             float kl_scale, float* dzargs, void* stream) {
  if (R <= 0 || L <= 0 || !zargs || !eps || !dz || !dzargs) return CLV_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  ProfScope p("gauss_bwd", s);
  hipLaunchKernelGGL(gauss_bwd_kernel, dim3((unsigned)(((int64_t)R * L + 255) / 256)), dim3(256), 0, s, R, L, zargs, eps,
                     dz, lddz, kl_scale, dzargs);
  return launch_status();
}

extern "C" int clv_bernoulli_nll(int R, int D, const float* logits, const float* y, int ldy, float scale,
                                 float* rownll, float* dlogits, void* stream) {
  if (R <= 0 || D <= 0 || !logits || !y) return CLV_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  ProfScope p("bernoulli_nll", s);
  hipLaunchKernelGGL(bernoulli_nll_kernel, dim3((R + 3) / 4), dim3(256), 0, s, R, D, logits, y, ldy, scale, rownll,
                     dlogits);
  return launch_status();
}

extern "C" int clv_sum_strided(int n, const float* x, int stride, float scale, float* out, void* stream) {
  if (n <= 0 || !x || !out) return CLV_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  ProfScope p("sum_strided", s);
  hipLaunchKernelGGL(sum_strided_kernel, dim3(1), dim3(1024), 0, s, n, x, stride, scale, out);
  return launch_status();
}

extern "C" size_t clv_colsum_workspace_bytes(int M, int N) {
  const int chunks = (M + CS_ROWS - 1) / CS_ROWS;
  return (size_t)chunks * N * sizeof(float);
}

extern "C" int clv_colsum_f32(int M, int N, const float* X, int ldx, float beta, float* out,
                              void* ws, size_t ws_bytes, void* stream) {
  if (M <= 0 || N <= 0 || !X || !out) return CLV_EINVAL;
  const int chunks = (M + CS_ROWS - 1) / CS_ROWS;
  if (!ws || ws_bytes < (size_t)chunks * N * sizeof(float)) return CLV_EWORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  ProfScope p("colsum", s);
  if (M <= 1024) {
    hipLaunchKernelGGL(colsum_small_kernel, dim3((N + 63) / 64), dim3(1024), 0, s, M, N, X, ldx, beta, out);
    return launch_status();
  }
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((N + 63) / 64, chunks), dim3(256), 0, s, M, N, X, ldx, (float*)ws);
  hipLaunchKernelGGL(colsum_final_kernel, dim3((N + 63) / 64), dim3(256), 0, s, N, chunks, (const float*)ws, beta, out);
  return launch_status();
}

extern "C" int clv_bernoulli_sample(int64_t n, const float* p, const float* u, float* x, void* stream) {
  if (n <= 0 || !p || !u || !x) return CLV_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  ProfScope pr("bernoulli_sample", s);
  hipLaunchKernelGGL(bernoulli_sample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, p, u, x);
  return launch_status();
}

namespace clv {
// out[r, c] = beta * out[r, c] + X[r, c] * m(U[r / T, c]),  m(u) = (u >= rate) / (1 - rate): see clv_dropout_rows
__global__ __launch_bounds__(256) void dropout_rows_kernel(int64_t total, int T, int n, const float* X, int ldx, const float* U,
                                                           int ldu, float rate, float inv_keep, float beta, float* out, int ldo) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int64_t r = e / n;
  const int c = (int)(e - r * n);
  const float m = U[(r / T) * ldu + c] >= rate ? inv_keep : 0.f;
  const float v = X[r * ldx + c] * m;
  float* o = out + r * ldo + c;
  *o = beta != 0.f ? fmaf(beta, *o, v) : v;
}
}  // namespace clv

extern "C" int clv_dropout_rows(int R, int T, int n, const float* X, int ldx, const float* U, int ldu, float rate, float beta,
                                float* out, int ldo, void* stream) {
  using namespace clv;
  if (R <= 0 || T <= 0 || n <= 0 || !X || !U || !out || ldx < n || ldu < n || ldo < n || !(rate >= 0.f && rate < 1.f)) return CLV_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  ProfScope pr("dropout_rows", s);
  const int64_t total = (int64_t)R * n;
  hipLaunchKernelGGL(dropout_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, total, T, n, X, ldx, U, ldu, rate,
                     1.f / (1.f - rate), beta, out, ldo);
  return launch_status();
}

extern "C" int clv_axpy(int64_t n, float alpha, const float* x, float* y, void* stream) {
  if (n <= 0 || !x || !y) return CLV_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  ProfScope pr("axpy", s);
  hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, alpha, x, y);
  return launch_status();
}

extern "C" int clv_act_grad(int64_t n, int act, const float* y, const float* dy, float* dpre, void* stream) {
  if (n <= 0 || !y || !dy || !dpre) return CLV_EINVAL;
  if (act != CLV_ACT_NONE && act != CLV_ACT_RELU && act != CLV_ACT_SIGMOID) return CLV_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  ProfScope p("act_grad", s);
  hipLaunchKernelGGL(act_grad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, act, y, dy, dpre);
  return launch_status();
}

extern "C" int clv_gather_rows(int64_t rows, int64_t row_elems, const float* src, const int64_t* idx, float* out,
                               int64_t chunk, int64_t out_ld, void* stream) {
  if (rows <= 0 || row_elems <= 0 || !src || !idx || !out) return CLV_EINVAL;
  if (chunk <= 0) { chunk = row_elems; out_ld = row_elems; }
  if (row_elems % chunk != 0 || out_ld < chunk) return CLV_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  ProfScope pr("gather_rows", s);
  const bool vec = chunk % 4 == 0 && out_ld % 4 == 0 && ((uintptr_t)src % 16 == 0) && ((uintptr_t)out % 16 == 0);
  const int64_t n = rows * (vec ? row_elems / 4 : row_elems);
  if (vec)
    hipLaunchKernelGGL(gather_rows_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, rows, row_elems, src,
                       idx, out, chunk, out_ld);
  else
    hipLaunchKernelGGL(gather_rows_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, rows, row_elems,
                       src, idx, out, chunk, out_ld);
  return launch_status();
}

extern "C" int clv_gather_rows_multi(int64_t rows, const int64_t* idx, int64_t row0, int nseg,
                                            const void* const* src, const int32_t* src_u8, float* const* out,
                                            const int64_t* row_elems, const int64_t* chunk, const int64_t* out_ld,
                                            const int64_t* src_stride, const int64_t* src_offset,
                                            const int64_t* const* src_table, unsigned char* const* notes_out,
                                            const clv_batch_cursor* cursor, void* stream) {
  if (rows <= 0 || nseg < 1 || nseg > 4 || !src || !out || !row_elems || !chunk || !out_ld) return CLV_EINVAL;
  if (cursor && (!cursor->step_dev || cursor->period < 1)) return CLV_EINVAL;
  GatherArgs a;
  memset(&a, 0, sizeof(a));
  a.nseg = nseg; a.rows = rows; a.idx = idx; a.row0 = row0;
  if (cursor) { a.step_dev = cursor->step_dev; a.step0 = cursor->step0; a.period = cursor->period;
                a.cur_stride = cursor->stride; a.cur_offset = cursor->offset; }
  int64_t maxw = 0, maxl = 0;
  for (int k = 0; k < nseg; ++k) {
    if (!src[k] || !out[k] || row_elems[k] <= 0 || row_elems[k] >= (1ll << 31)) return CLV_EINVAL;
    const int64_t ch = chunk[k] > 0 ? chunk[k] : row_elems[k];
    if (row_elems[k] % ch != 0) return CLV_EINVAL;
    const int64_t ld = chunk[k] > 0 ? out_ld[k] : row_elems[k];
    const int u8 = src_u8 ? (src_u8[k] == 2 ? 2 : (src_u8[k] != 0)) : 0;      // 2: a uint8 source copied to a uint8 output
    const int64_t stride = (src_stride && src_stride[k] > 0) ? src_stride[k] : row_elems[k];
    const int64_t offset = src_offset ? src_offset[k] : 0;
    const int64_t salign = u8 ? 4 : 16, ealign = u8 ? 4 : 4;       // bytes / elements a vector access needs
    const int vec = row_elems[k] % 4 == 0 && ch % 4 == 0 && ld % 4 == 0 && ((uintptr_t)src[k]) % salign == 0 &&
                    stride % ealign == 0 && offset % ealign == 0 && ((uintptr_t)out[k]) % (u8 == 2 ? 4 : 16) == 0;
    if (u8 == 2 && !vec) return CLV_EINVAL;                      // the byte copy moves dwords only
    // ... or 8 bytes at a time where everything is 8-byte aligned (88-byte frames are): the launch is bound by the chain of
    // dependent loads in front of every block (step counter -> row list -> window table), so fewer, fatter blocks
    const int wide = u8 == 2 && row_elems[k] % 8 == 0 && ch % 8 == 0 && ld % 8 == 0 && stride % 8 == 0 && offset % 8 == 0 &&
                     ((uintptr_t)src[k]) % 8 == 0 && ((uintptr_t)out[k]) % 8 == 0;
    a.seg[k] = GatherSeg{(const float*)src[k], out[k], src_table ? src_table[k] : nullptr, row_elems[k], ld, stride, offset,
                         (int)ch, (int)(row_elems[k] / ch), wide ? 2 : vec, u8, notes_out ? notes_out[k] : nullptr};
    int64_t w = row_elems[k] / (wide ? 8 : (vec ? 4 : 1));
    if (a.seg[k].notes) {      // byte frames of at most 88 notes, read 4 bytes per lane
      if (!u8 || ch > CLV_NOTE_NONE || ch % 4 || stride % 4 || offset % 4 || ((uintptr_t)src[k]) % 4) return CLV_EINVAL;
      a.list_seg[a.nlist++] = k;
      const int64_t wl = (a.seg[k].pieces + 3) / 4 * 256;      // a wave per frame, 4 frames per block
      maxl = wl > maxl ? wl : maxl;
    }
    maxw = w > maxw ? w : maxw;
  }
  hipStream_t s = (hipStream_t)stream;
  ProfScope p("gather_rows", s);
  const int64_t gx_copy = (maxw + 256 * GATHER_IPT - 1) / (256 * GATHER_IPT), gx_list = (maxl + 255) / 256;
  hipLaunchKernelGGL(gather_multi_kernel, dim3((unsigned)(gx_copy > gx_list ? gx_copy : gx_list), (unsigned)(rows < 65535 ? rows : 65535),
                                               nseg + a.nlist), dim3(256), 0, s, a);
  return launch_status();
}

extern "C" int clv_loss_sums(const float* x0, int n0, int s0, const float* x1, int n1, int s1, const float* x2, int n2,
                             int s2, const float* x3, int n3, int s3, const float* x4, int n4, int s4, float* out,
                             void* stream) {
  if (!x0 || !x1 || !x2 || !x3 || !x4 || !out || n0 <= 0 || n1 <= 0 || n2 <= 0 || n3 <= 0 || n4 <= 0) return CLV_EINVAL;
  SumArgs a{{x0, x1, x2, x3, x4}, {n0, n1, n2, n3, n4}, {s0, s1, s2, s3, s4}};
  hipStream_t s = (hipStream_t)stream;
  ProfScope p("loss_sums", s);
  hipLaunchKernelGGL(loss_sums_kernel, dim3(5), dim3(1024), 0, s, a, out);
  return launch_status();
}
