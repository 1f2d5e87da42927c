// label_head.hip -- the cl_vrnn label path of one batch row per workgroup.
//
// Forward (cl_vrnn/model.py:175-193 + the W rows of both LSTM kernels):
//   Wargs = hW.K_a + b_a ; W = softmax([mean + exp(lv/2)*eps, 0]) ; kl_w, w_rec, hit ;
//   rowbias_enc = W.K_enc[D:D+C] + b_enc ; rowbias_dec = W.K_dec[off:off+C] + b_dec
// Backward: dW = dzsum_dec.K_dec_w^T + dzsum_enc.K_enc_w^T ; label backward ; dWargs ;
//   dhW = (dWargs.K_a^T) * (hW > 0)
// Each of these is a handful of 88..352-long dot products per row: far too small for a GEMM launch
// each (a launch costs ~5 us), so one workgroup does the whole chain for its row out of LDS.
#include "common.h"
#include "lstm_pair_pack.h"
#include "philox.h"
#include "reduce_job.h"
#include "label_bwd_row.h"

namespace clv {

struct LabelFwdArgs {
  int B, D, C, G4;
  const float* hW;          // [B,D]
  __device__ float* hW_out() const { return const_cast<float*>(hW); }
  const float* Ka;          // Wargs kernel [D, 2(C-1)]
  const float* ba;          // [2(C-1)]
  float* eps;               // [B,C-1]: read, or drawn here and written (noise.on)
  struct { int on; uint32_t k0, k1, stream, step; uint64_t first; const int32_t* step_dev; } noise;
  const float* onehot;      // [B,C] or null
  float prior;
  const float* Kenc_w;      // [C,G4] rows of the encoder kernel that multiply W
  const float* benc;        // [G4]
  const float* Kdec_w;      // [C,G4]
  const float* bdec;
  float* wargs;             // [B,2(C-1)]
  float* W;                 // [B,C]
  float* rowloss;           // [B,3]
  float* rb_enc;            // [B,G4]
  float* rb_dec;            // [B,G4]
};

// label path of row b from its hW activations in s_h (LDS); NT threads take part
template <int NT>
__device__ __forceinline__ void label_fwd_row(const LabelFwdArgs& a, int b, int tid, const float* s_h, float* s_wargs, float* s_w,
                                              long long oh_row = -1) {      // row of a.onehot that belongs to batch row b (-1: b)
  const int C1 = a.C - 1, NA = 2 * C1;
  // the row's small vectors go to LDS in one round trip: the serial part below (thread 0) would otherwise pay an L2
  // round trip per element
  __shared__ float s_eps[LH_MAXC], s_oh[LH_MAXC];
  if (tid >= NT - 64 && tid - (NT - 64) < a.C) {
    const int j = tid - (NT - 64);
    float e = 0.f;
    if (j < C1) {
      if (a.noise.on) {          // the value clv_philox_normal writes for this element; kept for the backward pass
        e = philox_normal_at(a.noise.first + (uint64_t)b * C1 + j, a.noise.k0, a.noise.k1, a.noise.stream,
                             a.noise.step + (a.noise.step_dev ? (uint32_t)*a.noise.step_dev : 0u));
        a.eps[(size_t)b * C1 + j] = e;
      } else {
        e = a.eps[(size_t)b * C1 + j];
      }
    }
    s_eps[j] = e;
    s_oh[j] = a.onehot ? a.onehot[(size_t)(oh_row < 0 ? b : oh_row) * a.C + j] : 0.f;
  }
  {
    // Wargs = hW . K_a + b_a: [D] x [D, NA] with NA <= 62.  All NT threads take part: thread = (column, k-slice), every
    // thread's few kernel elements are requested at once (one L2 round trip for the layer; a column per thread walks D
    // rows in dependent batches), partial sums meet in LDS.
    __shared__ float s_part[16][2 * LH_MAXC];
    constexpr int KSL = NT / 64 < 16 ? NT / 64 : 16;       // k-slices: one per wave (6 or 16)
    const int col = tid & 63, sl = tid >> 6;
    if (sl < KSL) {
      float kv[(128 + KSL - 1) / KSL];
      const int cc = min(col, NA - 1);
#pragma unroll
      for (int i = 0; i < (128 + KSL - 1) / KSL; ++i) kv[i] = a.Ka[(size_t)min(sl + KSL * i, a.D - 1) * NA + cc];
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < (128 + KSL - 1) / KSL; ++i) acc = fmaf(sl + KSL * i < a.D ? s_h[sl + KSL * i] : 0.f, kv[i], acc);
      if (col < NA) s_part[sl][col] = acc;
    }
    __syncthreads();
    if (tid < NA) {
      float acc = a.ba[tid];
#pragma unroll
      for (int i = 0; i < KSL; ++i) acc += s_part[i][tid];
      s_wargs[tid] = acc;
      a.wargs[(size_t)b * NA + tid] = acc;
    }
  }
  __syncthreads();
  if (tid == 0) {
    float e[LH_MAXC];
    float S = 1.f, klw = 0.f;
    const float ep = __expf(a.prior);
    for (int j = 0; j < C1; ++j) {
      const float m = s_wargs[j], lv = s_wargs[C1 + j];
      const float sd = expf(0.5f * lv);
      e[j] = expf(m + sd * s_eps[j]);
      S += e[j];
      klw += 1.f - a.prior + lv - sd * sd / ep - m * m / ep;
    }
    e[C1] = 1.f;
    const float invS = 1.f / S;
    float qs = 0.f, wbest = -1.f, tbest = -1.f;
    int amax = 0, tmax = 0;
    for (int j = 0; j < a.C; ++j) {
      const float w = e[j] * invS;
      s_w[j] = w;
      a.W[(size_t)b * a.C + j] = w;
      qs += w + LW2;
      if (w > wbest) { wbest = w; amax = j; }
      const float tj = s_oh[j];
      if (tj > tbest) { tbest = tj; tmax = j; }
    }
    float wrec = 0.f;
    if (a.onehot)
      for (int j = 0; j < a.C; ++j) {
        const float n = (s_w[j] + LW2) / qs;
        wrec -= s_oh[j] * logf(fminf(fmaxf(n, LEPS_K), 1.f - LEPS_K));
      }
    a.rowloss[(size_t)b * 3 + 0] = -0.5f * klw;
    a.rowloss[(size_t)b * 3 + 1] = (float)C1 * wrec;
    a.rowloss[(size_t)b * 3 + 2] = (a.onehot && amax == tmax) ? 1.f : 0.f;
  }
  __syncthreads();
  for (int c = tid; c < a.G4; c += NT) {
    float e = a.benc[c], d = a.bdec[c];
    for (int j0 = 0; j0 < a.C; j0 += 8) {          // 16 loads in flight
      float ke[8], kd[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int j = min(j0 + q, a.C - 1);
        ke[q] = a.Kenc_w[(size_t)j * a.G4 + c];
        kd[q] = a.Kdec_w[(size_t)j * a.G4 + c];
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float w = j0 + q < a.C ? s_w[j0 + q] : 0.f;
        e = fmaf(w, ke[q], e);
        d = fmaf(w, kd[q], d);
      }
    }
    a.rb_enc[(size_t)b * a.G4 + c] = e;
    a.rb_dec[(size_t)b * a.G4 + c] = d;
  }
}

__global__ __launch_bounds__(LH_T) void vrnn_label_fwd_kernel(LabelFwdArgs a) {
  __shared__ float s_h[128], s_wargs[2 * LH_MAXC], s_w[LH_MAXC];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid < a.D) s_h[tid] = a.hW[(size_t)b * a.D + tid];
  __syncthreads();
  label_fwd_row<LH_T>(a, b, tid, s_h, s_wargs, s_w);
}

// The same with the hW Dense layer in front (cl_vrnn/model.py:174-176): hW = relu(flat(X_b) . K_h + b_h) over the
// row's nonzero inputs (the window is ~4 % notes), as in sparse_dense_kernel -- 16 waves scan 64-input chunks, ballot,
// and add the listed kernel rows (float2 per lane, 4 loads in flight) -- then the label path without leaving the
// workgroup.  hW is also written out (the backward pass needs it).
struct LabelFwdXArgs {
  LabelFwdArgs l;          // l.hW is the OUTPUT here
  const float* X;          // [B, ldx]
  const float* Kh;         // [nx, D]
  const float* bh;         // [D]
  int nx, ldx;
  PairPackArgs pack;       // out != null: also write the pair kernels' weight pack
  // part != null: X . Kh arrives as split-K partial sums [splits][B][D] (dense_window_fwd_bf16_kernel, outer_bf16.hip: the
  // dense product on the bf16 matrix cores for byte-valued frames); the workgroup sums its row's, X / Kh are not read
  const float* part; int splits;
  // stage.on: the workgroup of batch row b ASSEMBLES the row first (clv_label_stage: what the batch gather launch does for
  // the training step -- byte frames of the data set -> float rows of X and of the history buffer, the label row), reads the
  // bytes for its own scan, and X / the history frames / the labels are there for every later launch of the step.  The
  // mini-batch assembly then is no launch of its own.
  struct Stage {
    int on;
    const unsigned char* cur; const unsigned char* hist;       // byte stores (hist may be null)
    long long cur_stride, cur_offset, hist_stride, hist_offset, row0;
    const long long* cur_table; const long long* hist_table; const long long* idx;
    const int* step_dev; int step0, period; long long cur_s, cur_o;      // clv_batch_cursor
    float* X; float* Xh; int hist_chunk; long long hist_ld;    // history frame p of row b at Xh + (b * pieces + p) * hist_ld
    const float* w_src; float* w_out;
    unsigned char* X8; unsigned char* Xh8;      // not null: the rows are copied as BYTES ([B, nx] each) instead of widened into X / Xh
  } stage;
};
// the workgroup of batch row b (of a.B such workgroups in the launch)
__device__ __forceinline__ void label_fwd_x_block(const LabelFwdXArgs& ax, const int b, float2 (*part)[64], float* s_h, float* s_wargs,
                                                  float* s_w) {
  const LabelFwdArgs& a = ax.l;
  const int tid = threadIdx.x, lane = tid & 63;
  if (ax.pack.out) {
    // by-product: the pair LSTM kernels' lane-order weight pack (clv_lstm_pair_pack), a few elements per thread; the pair
    // forward kernel is the next launch but one and nothing in this kernel reads the pack
    for (int i = b * 1024 + tid; i < PK_TOTAL; i += a.B * 1024)
      ax.pack.out[i] = pair_pack_element(i, ax.pack.L, ax.pack.U_e, ax.pack.U_d, ax.pack.Kz, ax.pack.Wz,
                                         [](const float* p) { return *p; });
  }
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n2 = a.D / 2;
  const int lc = min(lane, n2 - 1);
  const float2* K2 = reinterpret_cast<const float2*>(ax.Kh);
  const float* xr = ax.X + (size_t)b * ax.ldx;
  const unsigned char* xbytes = nullptr;         // stage.on: the row's current frames as bytes
  long long oh_row = -1;
  if (ax.stage.on) {
    const LabelFwdXArgs::Stage& g = ax.stage;
    long long base = 0;
    if (g.step_dev) {
      int j = (*g.step_dev - g.step0) % g.period;
      j = j < 0 ? j + g.period : j;
      base = (long long)j * g.cur_s + g.cur_o;
    }
    const long long sr = g.idx ? g.idx[base + b] : g.row0 + base + b;
    oh_row = sr;
    xbytes = g.cur + (g.cur_table ? g.cur_table[sr] : sr) * g.cur_stride + g.cur_offset;
    // bytes -> floats, 4 at a time (nx and the frame length are multiples of 4; the stores are 4-byte aligned: checked by the
    // launcher); requested before the scan, nothing below waits for the stores.  X8: the bytes as they are (round 6: every
    // later launch of the step reads frames as bytes) -- a quarter of the stores
    const unsigned char* hb = g.hist ? g.hist + (g.hist_table ? g.hist_table[sr] : sr) * g.hist_stride + g.hist_offset : nullptr;
    if (g.X8) {
      unsigned* xo8 = reinterpret_cast<unsigned*>(g.X8 + (size_t)b * ax.nx);
      unsigned* ho8 = reinterpret_cast<unsigned*>(g.Xh8 + (size_t)b * ax.nx);
      for (int c = 4 * tid; c < ax.nx; c += 4 * 1024) {
        xo8[c >> 2] = *reinterpret_cast<const unsigned*>(xbytes + c);
        if (hb) ho8[c >> 2] = *reinterpret_cast<const unsigned*>(hb + c);
      }
    } else {
      float* xo = g.X + (size_t)b * ax.ldx;
      for (int c = 4 * tid; c < ax.nx; c += 4 * 1024) {
        const unsigned v = *reinterpret_cast<const unsigned*>(xbytes + c);
        *reinterpret_cast<float4*>(xo + c) = make_float4((float)(v & 255u), (float)((v >> 8) & 255u), (float)((v >> 16) & 255u), (float)(v >> 24));
      }
      if (hb) {
        const int pieces = ax.nx / g.hist_chunk;
        for (int c = 4 * tid; c < ax.nx; c += 4 * 1024) {
          const unsigned v = *reinterpret_cast<const unsigned*>(hb + c);
          const int p = c / g.hist_chunk, w = c - p * g.hist_chunk;
          *reinterpret_cast<float4*>(g.Xh + ((size_t)b * pieces + p) * g.hist_ld + w) =
              make_float4((float)(v & 255u), (float)((v >> 8) & 255u), (float)((v >> 16) & 255u), (float)(v >> 24));
        }
      }
    }
    if (g.w_out && tid < a.C) g.w_out[(size_t)b * a.C + tid] = g.w_src[(size_t)sr * a.C + tid];
  }
  float2 acc = make_float2(0.f, 0.f);
  const int nchunk = (ax.nx + 63) / 64;
  // Two chunks of 64 inputs per iteration: their kernel rows (up to 4 each per round) are requested together, so a
  // wave pays one L2 round trip per PAIR of chunks (a window row has 176 chunks, 11 per wave).  (Four chunks per iteration,
  // 16 rows in flight per lane: 74 registers instead of 62, i.e. one workgroup per CU instead of two, and slower -- 24.9 us
  // against 22.8 at configuration 3, 131 against 119 at configuration 5.)
  // The chunks are requested a round ahead as they lie in memory -- a byte as the aligned dword that holds it (the rows start on
  // 4-byte boundaries) -- and made floats where they are USED: a conversion (or a branch) at the load makes the compiler wait
  // for it on the spot, and the round ahead hid nothing (rounds 3-5: two L2 round trips per iteration instead of one).
  // Unconditional: past the row's last chunk the last one is read again and not used.
  typedef const __attribute__((address_space(1))) unsigned* gwords;
  auto xload = [&](int ch) -> unsigned {
    const int i = min(min(ch, nchunk - 1) * 64 + lane, ax.nx - 1);
    return xbytes ? ((gwords)xbytes)[i >> 2] : __builtin_bit_cast(unsigned, xr[i]);
  };
  const int xsh = 8 * (lane & 3);          // (the clamped lanes of a row's last chunk are masked below)
  auto xval = [&](unsigned raw) { return xbytes ? (float)((raw >> xsh) & 0xffu) : __builtin_bit_cast(float, raw); };
  unsigned xa = xload(wave), xb = xload(wave + 16);
  for (int ch = wave; ch < nchunk; ch += 32) {
    const float x0 = xval(xa), x1 = xval(xb);
    const int j0 = ch * 64, j1 = (ch + 16) * 64;
    xa = xload(ch + 32); xb = xload(ch + 48);
    unsigned long long m0 = __ballot(j0 + lane < ax.nx && x0 != 0.f);
    unsigned long long m1 = __ballot(ch + 16 < nchunk && j1 + lane < ax.nx && x1 != 0.f);
    while (m0 | m1) {
      int kk[8];
      float vv[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        unsigned long long& m = q < 4 ? m0 : m1;
        const bool on = m != 0;
        const int bit = on ? __builtin_ctzll(m) : 0;
        m = on ? (m & (m - 1)) : 0;
        kk[q] = (q < 4 ? j0 : j1) + bit;
        const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, q < 4 ? x0 : x1), bit));
        vv[q] = on ? v : 0.f;
      }
      float2 kr[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) kr[q] = K2[(size_t)min(kk[q], ax.nx - 1) * n2 + lc];
#pragma unroll
      for (int q = 0; q < 8; ++q) { acc.x = fmaf(vv[q], kr[q].x, acc.x); acc.y = fmaf(vv[q], kr[q].y, acc.y); }
    }
  }
  part[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && lane < n2) {
    float2 t = make_float2(ax.bh[2 * lane], ax.bh[2 * lane + 1]);
#pragma unroll
    for (int w = 0; w < 16; ++w) { t.x += part[w][lane].x; t.y += part[w][lane].y; }
    t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f);
    s_h[2 * lane] = t.x; s_h[2 * lane + 1] = t.y;
    float* op = a.hW_out() + (size_t)b * a.D + 2 * lane;
    op[0] = t.x; op[1] = t.y;
  }
  __syncthreads();
  label_fwd_row<1024>(a, b, tid, s_h, s_wargs, s_w, oh_row);
}

__global__ __launch_bounds__(1024) void vrnn_label_fwd_x_kernel(LabelFwdXArgs ax) {
  __shared__ float2 part[16][64];
  __shared__ float s_h[128], s_wargs[2 * LH_MAXC], s_w[LH_MAXC];
  label_fwd_x_block(ax, (int)blockIdx.x, part, s_h, s_wargs, s_w);
}

// ---------------------------------------------------------------------------------------------------------------------------
// The label launch with the LSTMs' frame projections as its second half (round 6): x_t . K_x of the encoder and
// x_{t-1} . K_x of the decoder (cl_vrnn/model.py:193-196, 218-226; sparse_proj.hip has the product and its reasons) for the
// SAME mini-batch rows, read from the byte stores through the same clv_label_stage, so the product does not wait for the
// label workgroups' copy of the batch.  Why one launch: the label path is a chain of L2 round trips (waves parked 3/4 of
// the time), the projection is bound by its 92 MB of stores, and as two launches of a graph they run one after the other
// (24 + 23 us at configuration 3; as parallel branches of the graph too -- hipGraph puts them on one queue; as two streams
// they DO overlap: profiles/r06_front_overlap.txt).  Here workgroups 0..B-1 are the label rows and the rest projection
// workgroups; both kinds fit a CU together (16 waves each, <= 64 registers, 13 + 62 KB of LDS), which is why a projection
// workgroup holds HALF of the kernel's columns (176 of 352: 62 KB) where sparse_proj_kernel holds all of them.
// Results: bit for bit those of vrnn_label_fwd_x_kernel and sparse_proj_kernel (same sums in the same order).
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int FP_RMAX = 64;        // batch rows per projection workgroup
constexpr int FP_NC = 4;           // output columns per lane, consecutive (half of N <= 256: 176 = 44 lanes x 4)
#ifndef FP_PF
#define FP_PF 4                    // frames of a wave whose bytes are in flight
#endif
#ifndef FP_ROUND
#define FP_ROUND 1                 // notes per round of LDS reads
#endif
#ifndef FP_ABL
#define FP_ABL 0                   // measurement builds (wrong results): 1 = no output stores, 2 = no note loop, 3 = frames not loaded after the first
#endif
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
struct FrameProjArgs {
  int T, N, ldo, wgs, nset;        // wgs: workgroups per (projection, column half)
  const float* K[2];               // [D, N]: 0 = over the current frames, 1 = over the history frames
  float* out[2];                   // [B*T, ldo]
};

// -DFRONT_STAMPS (tools/front_timeline.py): where and when each workgroup ran -- HW_ID / XCC_ID and the 100 MHz clock at its
// start and end
#ifdef FRONT_STAMPS
__device__ unsigned long long g_front_wg[2048][6];
#define FRONT_STAMP(k)                                                                                               \
  do {                                                                                                               \
    __syncthreads();                                                                                                 \
    if (threadIdx.x == 0 && blockIdx.x < 2048) {                                                                     \
      unsigned long long t__;                                                                                        \
      unsigned hw__, xcc__;                                                                                          \
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__));                                          \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw__));                                             \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc__));                                           \
      g_front_wg[blockIdx.x][k < 2 ? k : k + 2] = t__;                                                                \
      g_front_wg[blockIdx.x][2] = hw__; g_front_wg[blockIdx.x][3] = xcc__;                                           \
    }                                                                                                                \
  } while (0)
#else
#define FRONT_STAMP(k)
#endif

__device__ __forceinline__ void frame_proj_block(const LabelFwdXArgs& ax, const FrameProjArgs& fp, const int u, float* Kl,
                                                 const unsigned char** rowp) {
  const LabelFwdXArgs::Stage& g = ax.stage;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = u / fp.wgs, w = u - grp * fp.wgs;
  const bool second = grp >= 2;
  const int nf = ax.l.D, B = ax.l.B;
  const int nch = fp.N / 2, c0 = (grp & 1) * nch;
  const float* K = second ? fp.K[1] : fp.K[0];
  float* out = second ? fp.out[1] : fp.out[0];
  {   // this half of the kernel's columns -> LDS [nf][nch]
    const int q4 = nch / 4, nv = nf * q4;
    float4* dst = reinterpret_cast<float4*>(Kl);
    for (int i0 = tid; i0 < nv; i0 += 4 * 1024) {       // 4 loads in flight per thread
      float4 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = min(i0 + q * 1024, nv - 1), k = e / q4;
        v[q] = *reinterpret_cast<const float4*>(K + (size_t)k * fp.N + c0 + 4 * (e - k * q4));
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (i0 + q * 1024 < nv) dst[i0 + q * 1024] = v[q];
    }
  }
  // this workgroup's batch rows b = w, w + wgs, ..: where their frames are (the label workgroups resolve the same addresses)
  const int nrows = w < B ? (B - w + fp.wgs - 1) / fp.wgs : 0;
  if (tid < nrows) {
    const int b = w + tid * fp.wgs;
    long long base = 0;
    if (g.step_dev) {
      int j = (*g.step_dev - g.step0) % g.period;
      j = j < 0 ? j + g.period : j;
      base = (long long)j * g.cur_s + g.cur_o;
    }
    const long long sr = g.idx ? g.idx[base + b] : g.row0 + base + b;
    rowp[tid] = second ? g.hist + (g.hist_table ? g.hist_table[sr] : sr) * g.hist_stride + g.hist_offset
                       : g.cur + (g.cur_table ? g.cur_table[sr] : sr) * g.cur_stride + g.cur_offset;
  }
  __syncthreads();
  FRONT_STAMP(2);                                         // (slot 4: the kernel half and the row addresses are in LDS)
  const int nq = nrows * fp.T;                            // frames of this workgroup, row after row
  if (nq == 0) return;
  // lane l < nch / 4 owns columns 4 l .. 4 l + 3: one 16-byte LDS read per note, four FMAs, one 16-byte store per frame (the
  // kernel is bound by the vector instructions it issues, 16 + 16 waves on a CU: tools/front_timeline.py with FP_ROUND > 1)
  const int colb = min(lane, nch / 4 - 1) * 4;
  const unsigned vo = lane < nch / 4 ? (unsigned)lane * 16u : 0x80000000u;
  const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)((size_t)B * fp.T * fp.ldo * 4), 0x00020000);
  const int k0 = min(lane, nf - 1), k1 = min(lane + 64, nf - 1);
  const bool v0 = lane < nf, v1 = lane + 64 < nf;
  // a frame's two inputs of this lane, as the BYTES they are: widened where they are used (a conversion here makes the
  // compiler wait for the load at once, which is what sparse_proj_kernel's byte instance did until round 6)
  // (loaded as the aligned dword that holds the byte -- frames start on 4-byte boundaries -- and extracted at the use)
  typedef const __attribute__((address_space(1))) unsigned* gwords;
  const int w0 = k0 >> 2, w1 = k1 >> 2, sh0 = 8 * (k0 & 3), sh1 = 8 * (k1 & 3), nfw = nf / 4;
  // The loop below is bound by the NUMBER of instructions a wave issues (one per four cycles at best; FP_ROUND / two register
  // sets, see the note loop), so what is uniform stays in scalar registers and moves by additions: a row's address is
  // fetched from LDS when the row changes, the store offset advances with the frame.
  auto rowbase = [&](int r) -> gwords {
    const unsigned long long p = (unsigned long long)rowp[min(r, nrows - 1)];
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
    return (gwords)(((unsigned long long)hi << 32) | lo);
  };
  const gwords base0 = rowbase(0);
  auto frame = [&](gwords base, int t, unsigned& x0, unsigned& x1) {
    gwords bp = base + t * nfw;
    x0 = bp[w0]; x1 = bp[w1];
  };
  // This wave's frames are q = wave, wave + 16, .. of the nq.  A frame's bytes are requested FP_PF of the wave's frames ahead:
  // with one frame ahead (sparse_proj_kernel) an iteration lasts one global load (~1.2 us measured, for ~0.3 us of work).
  int ri = wave / fp.T, t = wave - ri * fp.T;             // (row, frame) being computed ...
  int ra = ri, ta = t;                                    // ... and being requested
  gwords base_a = rowbase(ra);
  auto advance_a = [&]() {
    ta += 16;
    if (ta >= fp.T) {
      do { ta -= fp.T; ++ra; } while (ta >= fp.T);
      base_a = rowbase(ra);
    }
  };
  // byte offset of the computed frame's outputs: row (w + ri wgs) T + t of [.., ldo], this workgroup's column half
  const unsigned so_step = 16u * (unsigned)fp.ldo * 4u, so_wrap = (unsigned)(fp.wgs - 1) * (unsigned)fp.T * (unsigned)fp.ldo * 4u;
  unsigned so = (unsigned)(((w + ri * fp.wgs) * fp.T + t) * fp.ldo + c0) * 4u;
  auto advance_c = [&]() {
    t += 16; so += so_step;
    while (t >= fp.T) { t -= fp.T; so += so_wrap; }
  };
  unsigned qx0[FP_PF], qx1[FP_PF];
#pragma unroll
  for (int j = 0; j < FP_PF; ++j) {
    const bool ok = wave + 16 * j < nq;                   // (unconditional loads, clamped to frame 0 of the workgroup: a load
    frame(ok ? base_a : base0, ok ? ta : 0, qx0[j], qx1[j]);      // under a condition is waited for where it is issued)
    // (as many stores -- out of range, dropped -- as a loop iteration has: the loop's first wait is the minimum over both ways
    // into the loop of the operations issued since the request, and without these that minimum is the prologue's)
    __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{0u, 0u, 0u, 0u}, r_out, (int)0x80000000u, 0, 0);
    advance_a();
  }
  for (int qb = wave; qb < nq; qb += 16 * FP_PF) {
#pragma unroll
    for (int j = 0; j < FP_PF; ++j) {
      const int q = qb + 16 * j;
      const bool live = q < nq;                           // (no break: the slots keep their registers)
      const float fx0 = (float)((qx0[j] >> sh0) & 0xffu), fx1 = (float)((qx1[j] >> sh1) & 0xffu);
      const bool more = q + 16 * FP_PF < nq;
      if (FP_ABL != 3) frame(more ? base_a : base0, more ? ta : 0, qx0[j], qx1[j]);
      advance_a();
      unsigned long long m0 = __ballot(live && v0 && fx0 != 0.f), m1 = __ballot(live && v1 && fx1 != 0.f);
      float acc[FP_NC];
#pragma unroll
      for (int c = 0; c < FP_NC; ++c) acc[c] = 0.f;
      // the notes that are on, ascending (the sums are sparse_proj_kernel's, term by term), one LDS round trip each.  What was
      // tried on this loop, co-running with the label rows (projection workgroups' duration, tools/front_timeline.py): FP_ROUND
      // notes per round with the absent ones predicated off, 32 -> 39 (2) / 50 us (4); the next note's row requested before
      // the current one is used (two register sets), 32 -> 34 us.  A wave issues at most one instruction per four cycles and
      // every variant that hides latency adds instructions: the loop is bound by its instruction count.
      auto rounds = [&](unsigned long long m, const float fx, const int kbase) {
        if (FP_ROUND == 1) {
          while (m) {
            const int k = __builtin_ctzll(m);
            asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(k));          // m &= ~(1 << k): one instruction, not three
            const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fx), k));
            const float4 k4 = *reinterpret_cast<const float4*>(Kl + (kbase + k) * nch + colb);
            acc[0] = fmaf(v, k4.x, acc[0]); acc[1] = fmaf(v, k4.y, acc[1]); acc[2] = fmaf(v, k4.z, acc[2]); acc[3] = fmaf(v, k4.w, acc[3]);
          }
          return;
        }
        while (m) {
          float v[FP_ROUND], kv[FP_ROUND][FP_NC];
          bool on[FP_ROUND];
#pragma unroll
          for (int r = 0; r < FP_ROUND; ++r) {
            on[r] = m != 0;
            const int k = on[r] ? __builtin_ctzll(m) : 0;
            m &= m - 1;                                    // (0 stays 0)
            v[r] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fx), k));
            const float4 k4 = *reinterpret_cast<const float4*>(Kl + (kbase + k) * nch + colb);
            kv[r][0] = k4.x; kv[r][1] = k4.y; kv[r][2] = k4.z; kv[r][3] = k4.w;
          }
#pragma unroll
          for (int r = 0; r < FP_ROUND; ++r)
#pragma unroll
            for (int c = 0; c < FP_NC; ++c) acc[c] = (r == 0 || on[r]) ? fmaf(v[r], kv[r][c], acc[c]) : acc[c];
        }
      };
      if (FP_ABL != 2) {
        rounds(m0, fx0, 0);
        rounds(m1, fx1, 64);
      } else {
        acc[0] = fx0; acc[1] = fx1;
      }
      // buffer stores, lanes without a column (and frames beyond the last) at an out-of-range offset: no branch around a
      // store, so the number of memory operations between a request and its use is fixed and the wait in front of the use
      // is a counted one (behind `if`s the compiler must assume the stores were skipped and waits for the newest of them)
      __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{__builtin_bit_cast(unsigned, acc[0]), __builtin_bit_cast(unsigned, acc[1]),
                                                      __builtin_bit_cast(unsigned, acc[2]), __builtin_bit_cast(unsigned, acc[3])},
                                             r_out, (int)((live && FP_ABL != 1) ? vo : 0x80000000u), (int)so, 0);
      advance_c();
    }
  }
}

// Two of these workgroups must fit a CU (a label row beside a projection workgroup): 16 waves each = 8 per SIMD, which the
// hardware admits up to 64 VGPRs and ~80 SGPRs per wave (800 SGPRs per SIMD: tools/probes/coreside_probe.hip -- at 83 the
// second workgroup waits for the first, measured) -- hence the SGPR cap (the compiler moves what does not fit to a VGPR's lanes).
__global__ __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(72))) void vrnn_front_kernel(LabelFwdXArgs ax, FrameProjArgs fp) {
  __shared__ float2 part[16][64];
  __shared__ float s_h[128], s_wargs[2 * LH_MAXC], s_w[LH_MAXC];
  __shared__ const unsigned char* rowp[FP_RMAX];
  extern __shared__ __attribute__((aligned(16))) float fp_kl[];
  FRONT_STAMP(0);
  if ((int)blockIdx.x < ax.l.B) label_fwd_x_block(ax, (int)blockIdx.x, part, s_h, s_wargs, s_w);
  else frame_proj_block(ax, fp, (int)blockIdx.x - ax.l.B, fp_kl, rowp);
  FRONT_STAMP(1);
}

// The same with X . Kh handed in as split-K partial sums (ax.part: dense_window_fwd_bf16_kernel, outer_bf16.hip).  Summing a
// row's chunks is a few loads per lane, so this workgroup is the label path's size (LH_T threads: five fit a CU, and the
// 1024 rows of configuration 5 are resident at once; as a mode of the 1024-thread kernel above they took two rounds, 32 us).
__global__ __launch_bounds__(LH_T) void vrnn_label_fwd_parts_kernel(LabelFwdXArgs ax) {
  constexpr int NWV = LH_T / 64;
  __shared__ float2 part[NWV][64];
  __shared__ float s_h[128], s_wargs[2 * LH_MAXC], s_w[LH_MAXC];
  const LabelFwdArgs& a = ax.l;
  const int tid = threadIdx.x, lane = tid & 63;
  if (ax.pack.out) {
    for (int i = blockIdx.x * LH_T + tid; i < PK_TOTAL; i += gridDim.x * LH_T)
      ax.pack.out[i] = pair_pack_element(i, ax.pack.L, ax.pack.U_e, ax.pack.U_d, ax.pack.Kz, ax.pack.Wz,
                                         [](const float* p) { return *p; });
  }
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x;
  const int n2 = a.D / 2;
  const int lc = min(lane, n2 - 1);
  float2 acc = make_float2(0.f, 0.f);
  const float2* pp = reinterpret_cast<const float2*>(ax.part + (size_t)b * a.D) + lc;
  const size_t cs = (size_t)a.B * a.D / 2;                 // float2 per chunk
  for (int c0 = wave; c0 < ax.splits; c0 += 8 * NWV) {     // wave w sums the chunks w, w + 6, ..: 8 loads of a lane in flight
    float2 v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = pp[(size_t)min(c0 + NWV * q, ax.splits - 1) * cs];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float mk = c0 + NWV * q < ax.splits ? 1.f : 0.f;
      acc.x = fmaf(v[q].x, mk, acc.x); acc.y = fmaf(v[q].y, mk, acc.y);
    }
  }
  part[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && lane < n2) {
    float2 t = make_float2(ax.bh[2 * lane], ax.bh[2 * lane + 1]);
#pragma unroll
    for (int w = 0; w < NWV; ++w) { t.x += part[w][lane].x; t.y += part[w][lane].y; }
    t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f);
    s_h[2 * lane] = t.x; s_h[2 * lane + 1] = t.y;
    float* op = a.hW_out() + (size_t)b * a.D + 2 * lane;
    op[0] = t.x; op[1] = t.y;
  }
  __syncthreads();
  label_fwd_row<LH_T>(a, b, tid, s_h, s_wargs, s_w);
}

__global__ __launch_bounds__(LH_T) void vrnn_label_bwd_kernel(LabelBwdArgs a) {
  label_bwd_row<LH_T>(a, (int)blockIdx.x, (int)threadIdx.x);
}

}  // namespace clv

using namespace clv;

extern "C" int clv_vrnn_label_fwd(int B, int D, int C, int G4, const float* hW, const float* Ka, const float* ba,
                                  const float* eps, const float* onehot, float prior_logvar,
                                  const float* Kenc_w, const float* benc, const float* Kdec_w, const float* bdec,
                                  float* wargs, float* W, float* rowloss, float* rb_enc, float* rb_dec, void* stream) {
  if (B <= 0 || D <= 0 || D > 128 || C < 2 || C > LH_MAXC || G4 <= 0) return CLV_EINVAL;
  if (!hW || !Ka || !ba || !eps || !Kenc_w || !benc || !Kdec_w || !bdec || !wargs || !W || !rowloss || !rb_enc || !rb_dec)
    return CLV_EINVAL;
  LabelFwdArgs a{B, D, C, G4, hW, Ka, ba, const_cast<float*>(eps), {0, 0, 0, 0, 0, 0, nullptr}, onehot, prior_logvar, Kenc_w, benc,
                 Kdec_w, bdec, wargs, W, rowloss, rb_enc, rb_dec};
  hipStream_t s = (hipStream_t)stream;
  ProfScope p("vrnn_label_fwd", s);
  hipLaunchKernelGGL(vrnn_label_fwd_kernel, dim3(B), dim3(LH_T), 0, s, a);
  return launch_status();
}

extern "C" size_t clv_vrnn_label_bwd_workspace_bytes(int B, int D, int C) {
  return (size_t)B * (D + 1) * 2 * (C - 1) * sizeof(float);
}

extern "C" int clv_vrnn_label_bwd(int B, int D, int C, int G4, const float* dzsum_enc, const float* dzsum_dec,
                                     const float* Kenc_w, const float* Kdec_w, const float* wargs, const float* eps,
                                     const float* onehot, const float* W, const float* hW, const float* Ka,
                                     float prior_logvar, float class_weight, float w_kl_weight, float inv_b,
                                     float* dwargs, float* dhW, float* dKa, float* dba, void* ws, size_t ws_bytes,
                                     clv_reduce_job* job, void* stream) {
  if (job) memset(job, 0, sizeof(*job));
  if (B <= 0 || D <= 0 || D > 128 || C < 2 || C > LH_MAXC || G4 <= 0) return CLV_EINVAL;
  if (!dzsum_enc || !dzsum_dec || !Kenc_w || !Kdec_w || !wargs || !eps || !onehot || !W || !hW || !Ka || !dwargs || !dhW)
    return CLV_EINVAL;
  const bool wg = dKa != nullptr;
  if (wg && (!dba || !ws || ws_bytes < clv_vrnn_label_bwd_workspace_bytes(B, D, C))) return CLV_EWORKSPACE;
  LabelBwdArgs a{B, D, C, G4, dzsum_enc, dzsum_dec, Kenc_w, Kdec_w, wargs, eps, onehot, W, hW, Ka,
                 prior_logvar, class_weight, w_kl_weight, inv_b, dwargs, dhW, wg ? (float*)ws : nullptr};
  hipStream_t s = (hipStream_t)stream;
  {
    ProfScope p("vrnn_label_bwd", s);
    hipLaunchKernelGGL(vrnn_label_bwd_kernel, dim3(B), dim3(LH_T), 0, s, a);
  }
  int st = launch_status();
  if (st || !wg) return st;
  ReduceJob j;
  memset(&j, 0, sizeof(j));
  const int NA = 2 * (C - 1);
  j.partial = (const float*)ws;
  j.M = D + 1; j.N = NA; j.splits = B; j.nprob = 2;
  j.alpha = 1.f; j.beta = 0.f; j.act = CLV_ACT_NONE;
  j.prob[0] = ReduceProb{dKa, NA, 0};
  j.prob[1] = ReduceProb{dba, NA, D};
  if (job && B > 1) memcpy(job, &j, sizeof(j));
  else st = launch_reduce(j, s);
  return st;
}

extern "C" int clv_vrnn_label_fwd_x_proj_supported(int B, int D, int nx, int T, int N);

static int label_fwd_x_launch(int B, int D, int C, int G4, const float* X, int ldx, int nx, const float* Kh, const float* part,
                              int splits, const clv_label_stage* stage, const float* bh, float* hW_out, const float* Ka, const float* ba,
                              float* eps, const float* onehot, float prior_logvar,
                              const float* Kenc_w, const float* benc, const float* Kdec_w, const float* bdec,
                              float* wargs, float* W, float* rowloss, float* rb_enc, float* rb_dec,
                              const clv_noise_draw* noise, const clv_pair_pack_src* pack, const clv_frame_proj* proj, void* stream) {
  if (pack && (pack->H != LH || !clv_lstm_pair_supported(pack->H, pack->L) || !pack->U_enc || !pack->U_dec || !pack->Kz ||
               !pack->Wz || !pack->pack || ((uintptr_t)pack->pack) % 16))
    return CLV_EINVAL;
  if (B <= 0 || D <= 0 || D > 128 || D % 2 != 0 || C < 2 || C > LH_MAXC || G4 <= 0) return CLV_EINVAL;
  if (part ? (splits <= 0 || ((uintptr_t)part) % 8 != 0)
           : (nx <= 0 || ldx < nx || (!X && !(stage && stage->X8)) || !Kh || ((uintptr_t)Kh) % 8 != 0)) return CLV_EINVAL;
  if (stage) {        // the batch assembly inside the launch: byte stores, 4 bytes / 4 floats at a time
    const clv_label_stage& g = *stage;
    if (part || !g.cur || nx % 4 || ldx % 4 || ((uintptr_t)g.cur) % 4 || g.cur_stride % 4 || g.cur_offset % 4) return CLV_EINVAL;
    if (g.hist && (((uintptr_t)g.hist) % 4 || g.hist_stride % 4 || g.hist_offset % 4)) return CLV_EINVAL;
    if (g.X8) {       // the byte batch: [B, nx] rows of the current and (with hist) of the history frames
      if (((uintptr_t)g.X8) % 4 || (g.hist && (!g.Xh8 || ((uintptr_t)g.Xh8) % 4))) return CLV_EINVAL;
    } else {
      if (!g.X || ((uintptr_t)g.X) % 16) return CLV_EINVAL;
      if (g.hist && (!g.Xh || g.hist_chunk <= 0 || g.hist_chunk % 4 || nx % g.hist_chunk || g.hist_ld % 4 || g.hist_ld < g.hist_chunk ||
                     ((uintptr_t)g.Xh) % 16))
        return CLV_EINVAL;
    }
    if ((g.w_out != nullptr) != (g.w_src != nullptr)) return CLV_EINVAL;
    if (g.cursor.step_dev && g.cursor.period < 1) return CLV_EINVAL;
  }
  if (!bh || !hW_out || !Ka || !ba || !eps || !Kenc_w || !benc || !Kdec_w || !bdec || !wargs || !W || !rowloss ||
      !rb_enc || !rb_dec)
    return CLV_EINVAL;
  LabelFwdXArgs a{{B, D, C, G4, hW_out, Ka, ba, eps, {0, 0, 0, 0, 0, 0, nullptr}, onehot, prior_logvar, Kenc_w, benc, Kdec_w, bdec,
                   wargs, W, rowloss, rb_enc, rb_dec}, X, Kh, bh, nx, ldx, {0, nullptr, nullptr, nullptr, nullptr, nullptr}, part, splits, {}};
  if (stage) {
    const clv_label_stage& g = *stage;
    a.stage = LabelFwdXArgs::Stage{1, g.cur, g.hist, (long long)g.cur_stride, (long long)g.cur_offset, (long long)g.hist_stride,
                                   (long long)g.hist_offset, (long long)g.row0, (const long long*)g.cur_table,
                                   (const long long*)g.hist_table, (const long long*)g.idx, g.cursor.step_dev, g.cursor.step0,
                                   g.cursor.period, (long long)g.cursor.stride, (long long)g.cursor.offset, g.X, g.Xh, g.hist_chunk,
                                   (long long)g.hist_ld, g.w_src, g.w_out, g.X8, g.Xh8};
    if (g.w_src) a.l.onehot = g.w_src;           // the label path reads the row's labels where they come from
  }
  if (pack) a.pack = PairPackArgs{pack->L, pack->U_enc, pack->U_dec, pack->Kz, pack->Wz, reinterpret_cast<float4*>(pack->pack)};
  if (noise) {
    a.l.noise.on = 1; a.l.noise.k0 = (uint32_t)noise->seed; a.l.noise.k1 = (uint32_t)(noise->seed >> 32);
    a.l.noise.stream = noise->stream; a.l.noise.step = noise->step; a.l.noise.first = noise->first;
    a.l.noise.step_dev = noise->step_dev;
  }
  hipStream_t s = (hipStream_t)stream;
  if (proj) {         // the frame projections ride along: vrnn_front_kernel
    if (!clv_vrnn_label_fwd_x_proj_supported(B, D, nx, proj->T, proj->N) || !stage || !stage->X8 || part || proj->ldo < proj->N ||
        (size_t)B * proj->T * proj->ldo * 4 >= ((size_t)1 << 31) || proj->ldo % 4 || ((uintptr_t)proj->out_cur) % 16 || ((uintptr_t)proj->out_hist) % 16 ||
        !proj->K_cur || !proj->out_cur || ((uintptr_t)proj->K_cur) % 16 || (proj->K_hist && (!proj->out_hist || !stage->hist ||
        ((uintptr_t)proj->K_hist) % 16)))
      return CLV_EINVAL;
    FrameProjArgs fp;
    memset(&fp, 0, sizeof(fp));
    fp.T = proj->T; fp.N = proj->N; fp.ldo = proj->ldo; fp.nset = proj->K_hist ? 2 : 1;
    fp.K[0] = proj->K_cur; fp.out[0] = proj->out_cur; fp.K[1] = proj->K_hist; fp.out[1] = proj->out_hist;
    // one projection workgroup per CU in all (each CU then holds one label row and one projection workgroup), no more of
    // them than batch rows, no more than FP_RMAX rows each
    const int groups = 2 * fp.nset;
    int wgs = 256 / groups;
    wgs = wgs > B ? B : wgs;
    const int need = (B + FP_RMAX - 1) / FP_RMAX;
    wgs = wgs < need ? need : wgs;
    fp.wgs = wgs;
    size_t lds = (size_t)D * (proj->N / 2) * sizeof(float);
#ifdef FRONT_STAMPS
    if (const char* e = getenv("CLV_EXP_FRONT_LDS")) lds = (size_t)atoi(e);      // placement experiments (wrong results below the real size)
#endif
    if (int e = clv::allow_dynamic_lds(reinterpret_cast<const void*>(vrnn_front_kernel), 64 * 1024)) return e;
    ProfScope p("vrnn_front", s);
    hipLaunchKernelGGL(vrnn_front_kernel, dim3(B + groups * wgs), dim3(1024), lds, s, a, fp);
    return launch_status();
  }
  ProfScope p("vrnn_label_fwd", s);
  if (part) hipLaunchKernelGGL(vrnn_label_fwd_parts_kernel, dim3(B), dim3(LH_T), 0, s, a);
  else hipLaunchKernelGGL(vrnn_label_fwd_x_kernel, dim3(B), dim3(1024), 0, s, a);
  return launch_status();
}

#ifdef FRONT_STAMPS
extern "C" int clv_debug_front_wg(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(clv::g_front_wg), sizeof(unsigned long long) * 2048 * 6);
}
#endif

// proj of clv_vrnn_label_fwd_x: frames of D <= 128 bytes (a multiple of 4), T of them per batch row (nx = T * D), N = 4H output columns in two
// halves of <= 256, half a kernel (D * N / 2 floats) in 64 KB of LDS
extern "C" int clv_vrnn_label_fwd_x_proj_supported(int B, int D, int nx, int T, int N) {
  return B >= 1 && D >= 4 && D <= 128 && D % 4 == 0 && T >= 1 && (long long)T * D == nx && N >= 8 && N % 8 == 0 && N / 2 <= 64 * clv::FP_NC &&      /* (and ldo, the outputs 16-byte aligned: the launcher) */
         (size_t)D * (N / 2) * sizeof(float) <= 64 * 1024 && (size_t)B * T * N * sizeof(float) < ((size_t)1 << 31);      // (one buffer descriptor per output)
}

extern "C" int clv_vrnn_label_fwd_x(int B, int D, int C, int G4, const float* X, int ldx, int nx, const float* Kh,
                                    const clv_label_stage* stage,
                                    const float* bh, float* hW_out, const float* Ka, const float* ba,
                                    float* eps, const float* onehot, float prior_logvar,
                                    const float* Kenc_w, const float* benc, const float* Kdec_w, const float* bdec,
                                    float* wargs, float* W, float* rowloss, float* rb_enc, float* rb_dec,
                                    const clv_noise_draw* noise, const clv_pair_pack_src* pack, const clv_frame_proj* proj,
                                    void* stream) {
  if (stage)      // the launch assembles its mini-batch rows itself: X = stage->X is an output, the labels come from stage->w_src
    return label_fwd_x_launch(B, D, C, G4, stage->X, ldx, nx, Kh, nullptr, 0, stage, bh, hW_out, Ka, ba, eps, stage->w_src,
                              prior_logvar, Kenc_w, benc, Kdec_w, bdec, wargs, W, rowloss, rb_enc, rb_dec, noise, pack, proj, stream);
  if (proj) return CLV_EINVAL;       // the projections read the byte stores a stage names
  return label_fwd_x_launch(B, D, C, G4, X, ldx, nx, Kh, nullptr, 0, nullptr, bh, hW_out, Ka, ba, eps, onehot, prior_logvar, Kenc_w, benc,
                            Kdec_w, bdec, wargs, W, rowloss, rb_enc, rb_dec, noise, pack, nullptr, stream);
}

extern "C" int clv_vrnn_label_fwd_parts(int B, int D, int C, int G4, const float* part, int splits,
                                        const float* bh, float* hW_out, const float* Ka, const float* ba,
                                        float* eps, const float* onehot, float prior_logvar,
                                        const float* Kenc_w, const float* benc, const float* Kdec_w, const float* bdec,
                                        float* wargs, float* W, float* rowloss, float* rb_enc, float* rb_dec,
                                        const clv_noise_draw* noise, const clv_pair_pack_src* pack, void* stream) {
  if (!part) return CLV_EINVAL;
  return label_fwd_x_launch(B, D, C, G4, nullptr, 0, 0, nullptr, part, splits, nullptr, bh, hW_out, Ka, ba, eps, onehot, prior_logvar, Kenc_w,
                            benc, Kdec_w, bdec, wargs, W, rowloss, rb_enc, rb_dec, noise, pack, nullptr, stream);
}

