// vae_fused.hip -- one whole cl_vae training step (forward, sampling, four losses, backward) in ONE launch.
//
// cl_vae is 33 k parameters: as separate layers it is ~45 launches of a few microseconds each, i.e. launch-
// bound.  Here a workgroup owns RB batch rows and walks the entire graph of cl_vae/model.py:136-219 with its
// activations resident in LDS; per-row sampling / loss math runs on the first RB threads.  Weight gradients are
// produced per workgroup (K = RB rows of the batch) into a slab laid out like the flat parameter buffer and
// summed by a second tiny launch, so the result is deterministic (no float atomics).
//
// The kernel is a latency chain (12 dependent products, ~40 workgroups on 256 CUs), so it is written for latency:
//   * the graph is a TABLE of 12 stages (6 forward layers, 6 backward) built by the launcher; the kernel is one loop
//     over it with a single product body, a single weight-gradient body and the per-row hooks between stages --
//     the code is small enough to stay in the instruction cache instead of being ~70 KB executed once;
//   * a stage's weights go from L2 into one wave's MFMA B registers (v_mfma_f32_16x16x4_f32, exact fp32) ONE STAGE
//     AHEAD, with unconditional clamped loads (a predicated load turns into a branch with a wait behind it), and
//     the barriers do not drain them;
//   * concatenated Dense inputs ([x, w] and [w, x_prev, z]) are physically adjacent in LDS and every buffer is
//     zero beyond its width, so an A operand is one ds_read at a constant offset; all A reads of a product are
//     issued before its first MFMA;
//   * all global inputs (frames, labels, noise) are staged once at the start; the noise can be drawn in the
//     kernel (Philox, same values as clv_philox_normal2).
// Dims: D, H, Hc <= 96; C, L <= 16.
#include "common.h"
#include "philox.h"

namespace clv {

#ifndef VAB
#define VAB 0      // timing ablations (wrong results): 1 no gradient stores, 2 no gradient MFMAs, 3 no gradients, 5 no product MFMAs
#endif

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
typedef short s16x4v __attribute__((ext_vector_type(4)));

// BF16 = true (clv_vae_step_opts.bf16): every Dense product and every weight-gradient product rounds its two operands
// to bf16 and accumulates in fp32 on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16 / 16x16x16) -- 8x the fp32 MFMA
// rate; sampling, losses and the per-row backward math stay fp32.  The k-slot of an operand register is the same in
// both modes (slot t of step j is k = 32 j + 4 t + q), so loads and LDS layout do not change.
__device__ __forceinline__ bf16x8v pack8(const float* v) {
  return (bf16x8v){(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3], (__bf16)v[4], (__bf16)v[5], (__bf16)v[6], (__bf16)v[7]};
}
__device__ __forceinline__ s16x4v pack4(const float* v) {
  typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
  const bf16x4v b = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
  return __builtin_bit_cast(s16x4v, b);
}

constexpr int VRB = 16;    // batch rows per workgroup (one MFMA row tile)
constexpr int VL = 132;    // LDS row stride of every buffer: == 4 mod 64, so the 16 rows x 4 k of an MFMA A read, the
                           // 16 columns x 4 rows of a product's output and the (row = 4 q + m) reads of the weight
                           // gradients each touch 64 different banks; >= 16 + 96 + 16 (the widest concatenated
                           // input); ONE stride keeps every LDS offset of the bodies an instruction immediate
constexpr int VNW = 8;     // waves per workgroup: one 16-column tile of an 88-wide layer per wave
constexpr int VNT = VNW * 64;
constexpr int VNSTAGE = 12;
constexpr float VEPS_K = 1e-7f, VW2 = 1e-10f;

// One stage of the graph.  Product: out = epilogue(A[16 x K] . B), A in LDS, B(k, c) = W[k*ldw + c] or, with VF_NT,
// W[c*ldw + k] (the backward product DY . W^T).  VF_PAIR: waves 0 and 1 compute two separate narrow products of the
// same A (index 0 / 1 below), otherwise wave i owns columns [16 i, 16 i + 16) of product 0.
// Weight gradient (backward stages): G[g_off][ga_K x K] = GA[16 x ga_K]^T . A, bias gradient = column sums of A.
enum { VF_MM = 1, VF_NT = 2, VF_PAIR = 4, VF_RELU = 8, VF_ACC = 16, VF_MASK = 32, VF_BIAS = 64 };
struct VStage {
  int flags, a_off, K, ldw;
  int w_off[2], ncol[2], o_off[2];
  int b_off, mask_off;
  int ga_off, ga_K;                // ga_K = 0: no weight gradient
  int tiling;                      // (ceil(K / 16) << 24) | ceil(65536 / ceil(K / 16)): tile index -> (k tile, n tile)
  int g_off, gb_off;
};

struct VaeArgs {
  int B, D, H, Hc, C, L, use_xp;
  const float* x; const float* xp; const float* onehot;    // [B,D] [B,D] [B,C]
  const float* y;                                          // [B,D] reconstruction target (NULL: x itself)
  float* eps_w; float* eps_z;                              // [B,C-1] [B,L]: read, or written when draw != 0
  const float* P;                                          // flat parameters
  float prior, class_weight, kl_weight, w_kl_weight;
  int need_grads, draw;
  uint32_t k0, k1, stream_w, stream_z, step;               // Philox key / streams / step of the in-kernel draw
  const int32_t* step_dev;
  uint64_t first_w, first_z;                               // stream index of eps_w[0], eps_z[0]
  float* slab; long slab_stride;                           // [n_wg][slab_stride] partial gradients
  float* logits; float* w_out; float* wargs_out; float* zargs_out;   // [B,D] [B,C] [B,2(C-1)] [B,2L] (for predict/tests)
  float* rownll; float* rowkl; float* rowloss;             // [B] [B] [B,3]
  // stage.on: the kernel assembles its own mini-batch rows (clv_label_stage: byte frames of the data set chosen by the device
  // step counter / row list / window table) instead of reading x / xp / onehot; x_out / xp_out / w_out (may be null) receive
  // the rows as the gather launch would have left them
  struct Stage {
    int on;
    const unsigned char* cur; const unsigned char* hist;
    long long cur_stride, cur_offset, hist_stride, hist_offset, row0;
    const long long* cur_table; const long long* hist_table; const long long* idx;
    const int* step_dev; int step0, period; long long cur_s, cur_o;
    const float* w_src;
    float* x_out; float* xp_out; float* w_out;
  } stage;
  VStage st[VNSTAGE];
};

// LDS map (floats): 17 buffers of 16 rows
struct VMap {
  int XC, DC, HW, Hh, HD, LG, G1, G2, WARGS, ZARGS, DWV, DWARGS, DZARGS, DZ, EW, EZ, OH, total;
  __host__ __device__ VMap() {
    int o = 0;
    int* f[17] = {&XC, &DC, &HW, &Hh, &HD, &LG, &G1, &G2, &WARGS, &ZARGS, &DWV, &DWARGS, &DZARGS, &DZ, &EW, &EZ, &OH};
    for (int i = 0; i < 17; ++i) { *f[i] = o; o += VRB * VL; }
    total = (o + 3) & ~3;
  }
};

// -DCLV_VAE_STAMPS: workgroup 0 records the 100 MHz wall clock at every stage boundary (tools/vae_stamps.py)
#ifdef CLV_VAE_STAMPS
__device__ unsigned long long g_vae_stamps[64];
#define VSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_vae_stamps[i] = wall_clock64(); } while (0)
// inside stages 2, 3 (forward) and 6, 7 (backward): slots 20 + 4 * {0, 1, 2, 3} + point
#define VSTAMPI(si, k) do { if ((si) == 2 || (si) == 3 || (si) == 6 || (si) == 7) \
    VSTAMP(20 + 4 * (((si) & 1) + ((si) >= 6 ? 2 : 0)) + (k)); } while (0)
#else
#define VSTAMP(i) do { } while (0)
#define VSTAMPI(si, k) do { } while (0)
#endif

// a barrier that does not drain the weight loads in flight (__syncthreads() waits for vmcnt(0))
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// One wave's weight operand of one stage: KS k-steps of 4 cover the longest product of the graph (K = C + D + L of
// the decoder): KS = 24 while that is <= 96, else 32.  What a clamped element multiplies is either a zero A operand
// or lands in a column nobody stores.
//
// The loads and the wait for them are written by hand.  The compiler's own wait-count bookkeeping sees the
// weight-gradient stores of the previous stage in the same counter as these loads and, not knowing their relative
// order, makes the first MFMA of a stage wait for the prefetch issued a moment ago -- which is the latency this
// scheme exists to hide.  Loads return in order among themselves, so "at most KS + 1 operations outstanding" right
// after the next stage's KS + 1 loads were issued means this stage's have all landed, whatever the stores do.
// (A register the compiler does not know to be pending must not be moved or spilled before frag_wait: check
// tools/kres.py -- no scratch -- after touching this kernel.)
template <int KS> struct WFrag { float v[KS]; float bias; };

__device__ __forceinline__ void wload(float& dst, const float* base, unsigned byte_off) {
  asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(byte_off), "s"(base));
}

#define VTIE4(f, b) "+v"(f.v[b]), "+v"(f.v[b + 1]), "+v"(f.v[b + 2]), "+v"(f.v[b + 3])
// every register of f has landed after this; newer = a prefetch of KS + 1 loads was issued after f's
template <int KS>
__device__ __forceinline__ void frag_wait(WFrag<KS>& f, bool newer) {
  static_assert(KS == 24 || KS == 32, "");
  // the branch holds an operand-less wait only: with tied operands inside it the compiler copies the registers on one
  // side BEFORE the wait
  if (!newer) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (KS == 24) asm volatile("s_waitcnt vmcnt(25)" : VTIE4(f, 0), VTIE4(f, 4), VTIE4(f, 8));
  else asm volatile("s_waitcnt vmcnt(33)" : VTIE4(f, 0), VTIE4(f, 4), VTIE4(f, 8));
  asm volatile("" : VTIE4(f, 12), VTIE4(f, 16), VTIE4(f, 20), "+v"(f.bias));
  if (KS == 32) asm volatile("" : VTIE4(f, KS - 8), VTIE4(f, KS - 4));
}

__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// does this wave own a tile of the stage's product?
__device__ __forceinline__ bool stage_has_tile(const VStage& s, int wave) {
  if (!(s.flags & VF_MM)) return false;
  if (s.flags & VF_PAIR) return wave < 2 && (wave == 1 ? s.ncol[1] : s.ncol[0]) > 0;
  return wave * 16 < s.ncol[0];
}

template <int KS>
__device__ __forceinline__ void stage_load(WFrag<KS>& f, const VStage& s, const float* P) {
  const int lane = threadIdx.x & 63, wave = wave_id();
  const int r = lane & 15, q = lane >> 4;
  const bool pair = s.flags & VF_PAIR;
  const bool second = pair && wave == 1;
  const int ncol = second ? s.ncol[1] : s.ncol[0];
  const int cc = min((pair ? 0 : wave) * 16 + r, ncol - 1);
  const int sk = (s.flags & VF_NT) ? 1 : s.ldw, sc = (s.flags & VF_NT) ? s.ldw : 1;
  const float* base = P + (second ? s.w_off[1] : s.w_off[0]);
  unsigned o = 4u * (unsigned)(cc * sc + q * sk);
  const unsigned omax = 4u * (unsigned)(cc * sc + (s.K - 1) * sk), ostep = 16u * (unsigned)sk;
#pragma unroll
  for (int i = 0; i < KS; ++i) { wload(f.v[i], base, min(o, omax)); o += ostep; }
  wload(f.bias, P + ((s.flags & VF_BIAS) ? s.b_off : 0), 4u * (unsigned)cc);     // ignored without VF_BIAS
}

// out = epilogue(A . B) for this wave's tile; B in registers once frag_wait returns
template <int KS, bool BF16>
__device__ __forceinline__ void stage_mm(const VStage& s, WFrag<KS>& f, float* lds, bool newer) {
  const int lane = threadIdx.x & 63, wave = wave_id();
  const int r = lane & 15, q = lane >> 4;
  const bool pair = s.flags & VF_PAIR;
  const bool second = pair && wave == 1;
  const int ncol = second ? s.ncol[1] : s.ncol[0];
  const int tile = pair ? 0 : wave;
  frag_wait(f, newer);
  const float* ap = lds + s.a_off + r * VL + q;
  float av[KS];
#pragma unroll
  for (int g = 0; g < KS / 8; ++g)
    if (32 * g < s.K) {
#pragma unroll
      for (int i = 8 * g; i < 8 * g + 8; ++i) av[i] = ap[4 * i];
    }
  __builtin_amdgcn_sched_barrier(0);
  f32x4v acc = (f32x4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < KS / 8; ++g)
    if (32 * g < s.K) {
#if VAB == 5
      for (int i = 8 * g; i < 8 * g + 8; ++i) acc[i & 3] += av[i] * f.v[i];
#else
      if (BF16) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pack8(av + 8 * g), pack8(f.v + 8 * g), acc, 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 8 * g; i < 8 * g + 8; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], f.v[i], acc, 0, 0, 0);
      }
#endif
    }
  const int col = tile * 16 + r;
  if (col < ncol) {
    float* out = lds + (second ? s.o_off[1] : s.o_off[0]) + q * 4 * VL + col;
    const float* mk = lds + s.mask_off + q * 4 * VL + col;
    const float bias = (s.flags & VF_BIAS) ? f.bias : 0.f;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      float v = acc[reg] + bias;
      if (s.flags & VF_ACC) v += out[reg * VL];
      if (s.flags & VF_RELU) v = fmaxf(v, 0.f);
      if (s.flags & VF_MASK) v = mk[reg * VL] > 0.f ? v : 0.f;
      out[reg * VL] = v;
    }
  }
}

// this workgroup's share of the stage's kernel and bias gradients: G = GA^T . DY (DY = the stage's A operand).
// A wave takes tiles wave, wave + 8, ...; the operands of all its tiles are read before the first MFMA.  The MFMA
// computes the TRANSPOSED tile (DY^T . GA), so a lane ends up with four consecutive columns of one gradient row: one
// 16-byte store per tile into the slab, whose rows are padded to a multiple of four columns (ceil4(N)).
// Batch row of k-step m for lane group q: 4 q + m (any bijection does; this one is bank-conflict-free).
template <int MT, bool BF16>
__device__ __forceinline__ void stage_wgrad(const VStage& s, const float* lds, float* G) {
  const int lane = threadIdx.x & 63, wave = wave_id();
  const int r = lane & 15, q = lane >> 4;
  const int N = s.K, Kin = s.ga_K, Np = (N + 3) & ~3;
  const int nts = (int)((unsigned)s.tiling >> 24), magic = s.tiling & 0xffffff;
  const int ntiles = ((Kin + 15) >> 4) * nts;
  const float* ga = lds + s.ga_off + 4 * q * VL;
  const float* dy = lds + s.a_off + 4 * q * VL;
  float av[MT][4], bv[MT][4];
  int k0[MT], n0[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    const int tc = min(wave + VNW * j, ntiles - 1);
    const int kt = (tc * magic) >> 16;
    k0[j] = kt * 16; n0[j] = (tc - kt * nts) * 16;
    const int ka = min(k0[j] + r, Kin - 1), nb = min(n0[j] + r, N - 1);      // clamped: duplicates are not stored / land
#pragma unroll                                                                // in the slab's padding columns
    for (int m = 0; m < 4; ++m) {
      av[j][m] = dy[m * VL + nb];
      bv[j][m] = ga[m * VL + ka];
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    if (wave + VNW * j < ntiles) {
      f32x4v acc = (f32x4v){0.f, 0.f, 0.f, 0.f};
#if VAB == 2
      for (int m = 0; m < 4; ++m) acc[m] += av[j][m] * bv[j][m];
#else
      if (BF16) {
        acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pack4(av[j]), pack4(bv[j]), acc, 0, 0, 0);
      } else {
#pragma unroll
        for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][m], bv[j][m], acc, 0, 0, 0);
      }
#endif
      const int row = k0[j] + r, col = n0[j] + 4 * q;
#if VAB == 1
      if (row < Kin && col < Np && acc[0] == 12345.678f)
#else
      if (row < Kin && col < Np)
#endif
        *reinterpret_cast<f32x4v*>(G + s.g_off + row * Np + col) = acc;
    }
  }
  for (int c = threadIdx.x; c < N; c += VNT) {
    float t = 0.f;
#pragma unroll
    for (int rr = 0; rr < VRB; ++rr) t += lds[s.a_off + rr * VL + c];
    G[s.gb_off + c] = t;
  }
}

template <int KS, bool BF16>
__global__ __launch_bounds__(VNT) void vae_fused_kernel(VaeArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const VMap M;
  const int tid = threadIdx.x;
  const int row0 = blockIdx.x * VRB;
  const int nvalid = min(VRB, a.B - row0);
  const int D = a.D, C = a.C, L = a.L, C1 = a.C - 1, NA = 2 * (a.C - 1), NZ = 2 * a.L;
  const int xo = a.use_xp ? D : 0;
  const float* P = a.P;
  const float inv_b = 1.f / (float)a.B;
  constexpr int MT = KS == 24 ? 5 : 6;       // weight-gradient tiles per wave: 36 (48) tiles of a 96 (128) x 96 kernel
  VSTAMP(0);

  // the first stage's weights are on their way while the inputs are staged
  WFrag<KS> f0, f1;
  if (stage_has_tile(a.st[0], wave_id())) stage_load(f0, a.st[0], P);
  // ---- LDS: zero everything (products rely on zeros beyond a buffer's width), then the inputs; the frames are
  // requested first so that the zero fill runs under their latency -------------------------------------------------
  constexpr int FPT = (VRB * 96 + VNT - 1) / VNT;          // frame elements per thread
  float fx[FPT], fp[FPT], fy[FPT];
  __shared__ long long s_cur[VRB], s_hist[VRB], s_sr[VRB];
  if (a.stage.on) {
    // the workgroup's rows of the mini-batch: source row (step counter -> row list), byte offsets of its frames
    if (tid < VRB) {
      const VaeArgs::Stage& g = a.stage;
      long long base = 0;
      if (g.step_dev) {
        int j = (*g.step_dev - g.step0) % g.period;
        j = j < 0 ? j + g.period : j;
        base = (long long)j * g.cur_s + g.cur_o;
      }
      const int b = row0 + min(tid, nvalid - 1);
      const long long sr = g.idx ? g.idx[base + b] : g.row0 + base + b;
      s_sr[tid] = sr;
      s_cur[tid] = (g.cur_table ? g.cur_table[sr] : sr) * g.cur_stride + g.cur_offset;
      s_hist[tid] = g.hist ? (g.hist_table ? g.hist_table[sr] : sr) * g.hist_stride + g.hist_offset : 0;
    }
    __syncthreads();
    // The bytes are requested here and made floats BEHIND the zero fill (round 6: converted at the load, each of a thread's
    // frame elements was waited for where it was requested -- three round trips in a row, the history frames' loads behind
    // a condition one more each: 3 us of the launch).  All loads unconditional: without history frames the current ones
    // are read twice.
    const bool has_hist = a.use_xp && a.stage.hist;
    const unsigned char* hsrc = has_hist ? a.stage.hist : a.stage.cur;
    unsigned ux[FPT], up[FPT];
#pragma unroll
    for (int u = 0; u < FPT; ++u) {
      const int i = min(tid + u * VNT, nvalid * D - 1);
      const int r = i / D, c = i - r * D;
      ux[u] = a.stage.cur[s_cur[r] + c];
      up[u] = hsrc[(has_hist ? s_hist[r] : s_cur[r]) + c];
    }
    for (int i = tid; i < M.total / 4; i += VNT) reinterpret_cast<float4*>(lds)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < FPT; ++u) {
      fx[u] = (float)ux[u];
      fp[u] = has_hist ? (float)up[u] : 0.f;
      fy[u] = 0.f;
      const int i = tid + u * VNT;
      if (i < nvalid * D) {
        const size_t g = (size_t)row0 * D + i;
        if (a.stage.x_out) a.stage.x_out[g] = fx[u];
        if (a.stage.xp_out && a.use_xp) a.stage.xp_out[g] = fp[u];
      }
    }
  } else {
#pragma unroll
    for (int u = 0; u < FPT; ++u) {
      const int i = min(tid + u * VNT, nvalid * D - 1);
      const size_t g = (size_t)row0 * D + i;
      fx[u] = a.x[g];
      fp[u] = a.use_xp ? a.xp[g] : 0.f;
      fy[u] = a.y ? a.y[g] : 0.f;
    }
  }
  if (!a.stage.on)
    for (int i = tid; i < M.total / 4; i += VNT) reinterpret_cast<float4*>(lds)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  lds_barrier();
#pragma unroll
  for (int u = 0; u < FPT; ++u) {
    const int i = tid + u * VNT;
    if (i < nvalid * D) {
      const int r = i / D, c = i - r * D;
      lds[M.XC + r * VL + c] = fx[u];
      if (a.use_xp) lds[M.DC + r * VL + C + c] = fp[u];
      if (a.y) lds[M.G2 + r * VL + c] = fy[u];            // the target waits in G2 (free until the backward pass)
    }
  }
  if (a.stage.on) {
    for (int i = tid; i < nvalid * C; i += VNT) {
      const int r = i / C, c = i - r * C;
      const float v = a.stage.w_src[(size_t)s_sr[r] * C + c];
      lds[M.OH + r * VL + c] = v;
      if (a.stage.w_out) a.stage.w_out[(size_t)(row0 + r) * C + c] = v;
    }
  } else if (a.onehot)
    for (int i = tid; i < nvalid * C; i += VNT) {
      const int r = i / C, c = i - r * C;
      lds[M.OH + r * VL + c] = a.onehot[(size_t)(row0 + r) * C + c];
    }
  {
    const uint32_t stp = a.step + ((a.draw && a.step_dev) ? (uint32_t)*a.step_dev : 0u);
    for (int i = tid; i < nvalid * (C1 + L); i += VNT) {
      const bool zs = i >= nvalid * C1;
      const int e = zs ? i - nvalid * C1 : i, w_ = zs ? L : C1;
      const int r = e / w_, c = e - r * w_;
      const size_t g = (size_t)(row0 + r) * w_ + c;
      float* ep = zs ? a.eps_z : a.eps_w;
      float v;
      if (a.draw) {
        v = philox_normal_at((zs ? a.first_z : a.first_w) + g, a.k0, a.k1, zs ? a.stream_z : a.stream_w, stp);
        ep[g] = v;
      } else {
        v = ep[g];
      }
      lds[(zs ? M.EZ : M.EW) + r * VL + c] = v;
    }
  }
  lds_barrier(); VSTAMP(1);

  float* G = a.slab + (size_t)blockIdx.x * a.slab_stride;
  const int nst = a.need_grads ? VNSTAGE : VNSTAGE / 2;

  // one stage: request the next stage's weights (into fn), multiply with this stage's (fc), weight gradients.
  // The table entries live in scalar registers and are fetched two stages ahead (sa: this stage, sb: the next,
  // sc: the one after, arriving), so no stage starts by waiting on its own description.
  // A wave requests and waits for weights only in stages where it owns a tile.
  const int wave = wave_id();
  auto run_stage = [&](int si, const VStage& sa, const VStage& sb, WFrag<KS>& fc, WFrag<KS>& fn) {
    const bool newer = si + 1 < nst && stage_has_tile(sb, wave);
    VSTAMPI(si, 0);
    if (newer) stage_load(fn, sb, P);
    VSTAMPI(si, 1);
    if (stage_has_tile(sa, wave)) stage_mm<KS, BF16>(sa, fc, lds, newer);
    VSTAMPI(si, 2);
#if VAB != 3
    if (sa.ga_K > 0) stage_wgrad<MT, BF16>(sa, lds, G);
#endif
    VSTAMPI(si, 3);
    lds_barrier(); VSTAMP(2 + si);
  };

  VStage sa = a.st[0], sb = a.st[1];
#pragma unroll 1
  for (int si = 0; si < nst; si += 2) {
    VStage sc = a.st[min(si + 2, VNSTAGE - 1)];
    run_stage(si, sa, sb, f0, f1);
    sa = a.st[min(si + 3, VNSTAGE - 1)];
    run_stage(si + 1, sb, sc, f1, f0);
    sb = sa; sa = sc;
    // ---- per-row work between stages (every hook follows an odd stage) --------------------------------------
    if (si == 0) {
      // w ~ logistic-normal, label losses (:146-157,198-206)
      if (tid < VRB) {
        const int r = tid;
        const float* wargs = lds + M.WARGS + r * VL;
        const float* oh = lds + M.OH + r * VL;
        float* wv = lds + M.DC + r * VL;                  // w opens the decoder input ...
        float* wx = lds + M.XC + r * VL + D;              // ... and follows x in the latent encoder's
        float S = 1.f, klw = 0.f;
        const float ep = __expf(a.prior);
        for (int j = 0; j < C1; ++j) {
          const float m = wargs[j], lv = wargs[C1 + j];
          const float sd = expf(0.5f * lv);
          const float e = expf(m + sd * lds[M.EW + r * VL + j]);
          wv[j] = e;
          S += e;
          klw += 1.f - a.prior + lv - sd * sd / ep - m * m / ep;
        }
        wv[C1] = 1.f;
        const float invS = 1.f / S;
        float qs = 0.f, wbest = -1.f, tbest = -1.f;
        int amax = 0, tmax = 0;
        for (int j = 0; j < C; ++j) {
          const float wj = wv[j] * invS;
          wv[j] = wj; wx[j] = wj;
          qs += wj + VW2;
          if (wj > wbest) { wbest = wj; amax = j; }
          if (oh[j] > tbest) { tbest = oh[j]; tmax = j; }
        }
        if (r < nvalid) {
          float wrec = 0.f;
          if (a.onehot || a.stage.on)
            for (int j = 0; j < C; ++j)
              wrec -= oh[j] * logf(fminf(fmaxf((wv[j] + VW2) / qs, VEPS_K), 1.f - VEPS_K));
          a.rowloss[(size_t)(row0 + r) * 3 + 0] = -0.5f * klw;
          a.rowloss[(size_t)(row0 + r) * 3 + 1] = (float)C1 * wrec;
          a.rowloss[(size_t)(row0 + r) * 3 + 2] = ((a.onehot || a.stage.on) && amax == tmax) ? 1.f : 0.f;
          for (int j = 0; j < C; ++j) a.w_out[(size_t)(row0 + r) * C + j] = wv[j];
          for (int j = 0; j < NA; ++j) a.wargs_out[(size_t)(row0 + r) * NA + j] = wargs[j];
        }
      }
    } else if (si == 2) {
      // z = mu + sigma eps, KL (:170-174,195-196)
      if (tid < VRB) {
        const int r = tid;
        const float* zargs = lds + M.ZARGS + r * VL;
        float* zv = lds + M.DC + r * VL + C + xo;
        float kl = 0.f;
        for (int j = 0; j < L; ++j) {
          const float m = zargs[j], lv = zargs[L + j];
          const float sd = expf(0.5f * lv);
          zv[j] = m + sd * lds[M.EZ + r * VL + j];
          kl += 1.f + lv - m * m - sd * sd;
        }
        if (r < nvalid) {
          a.rowkl[row0 + r] = -0.5f * kl;
          for (int j = 0; j < NZ; ++j) a.zargs_out[(size_t)(row0 + r) * NZ + j] = zargs[j];
        }
      }
    } else if (si == 4) {
      // Bernoulli NLL on the logits (Keras clip semantics) and its gradient, in place
      const int lane = tid & 63, wave = tid >> 6;
      for (int r = wave; r < VRB; r += VNW) {
        float acc = 0.f;
        for (int j = lane; j < D; j += 64) {
          const float av = lds[M.LG + r * VL + j];
          const float t = a.y ? lds[M.G2 + r * VL + j] : lds[M.XC + r * VL + j];
          if (a.y) lds[M.G2 + r * VL + j] = 0.f;          // G2 is a zero-padded product operand again
          if (r < nvalid && a.logits) a.logits[(size_t)(row0 + r) * D + j] = av;
          const float l = fminf(fmaxf(av, BCE_CLIP_LO), BCE_CLIP_HI);
          const float e = __expf(-fabsf(l));
          acc += fmaxf(l, 0.f) + __logf(1.f + e) - l * t;
          const float r1 = fast_rcp(1.f + e);
          const float sg = l >= 0.f ? r1 : e * r1;
          const bool inside = (av >= BCE_CLIP_LO) && (av <= BCE_CLIP_HI);
          lds[M.LG + r * VL + j] = (inside && r < nvalid) ? inv_b * (sg - t) : 0.f;
        }
        acc = wave_sum(acc);
        if (lane == 0 && r < nvalid) a.rownll[row0 + r] = acc;
      }
    } else if (si == 6) {
      // gaussian head backward
      if (tid < VRB) {
        const int r = tid;
        const float ks = a.kl_weight * inv_b;
        const float* zargs = lds + M.ZARGS + r * VL;
        float* dza = lds + M.DZARGS + r * VL;
        for (int j = 0; j < L; ++j) {
          const float m = zargs[j], lv = zargs[L + j];
          const float sd = expf(0.5f * lv);
          const float d = lds[M.DZ + r * VL + j];
          const float ez = lds[M.EZ + r * VL + j];
          dza[j] = r < nvalid ? d + ks * m : 0.f;
          dza[L + j] = r < nvalid ? d * ez * 0.5f * sd - 0.5f * ks * (1.f - sd * sd) : 0.f;
        }
      }
    } else if (si == 8) {
      // label head backward
      if (tid < VRB) {
        const int r = tid;
        const float ep = __expf(a.prior);
        const float* wv = lds + M.DC + r * VL;
        const float* wargs = lds + M.WARGS + r * VL;
        const float* oh = lds + M.OH + r * VL;
        float* dwa = lds + M.DWARGS + r * VL;
        float dn[16], d[16];
        float qs = 0.f, dot = 0.f, dsum = 0.f;
        for (int j = 0; j < C; ++j) qs += wv[j] + VW2;
        for (int j = 0; j < C; ++j) {
          const float n = (wv[j] + VW2) / qs;
          const bool inside = (n >= VEPS_K) && (n <= 1.f - VEPS_K);
          const float nc = fminf(fmaxf(n, VEPS_K), 1.f - VEPS_K);
          dn[j] = inside ? -(float)C1 * oh[j] / nc : 0.f;
          dot += dn[j] * n;
        }
        for (int j = 0; j < C; ++j) {
          d[j] = lds[M.DWV + r * VL + j] + a.class_weight * inv_b * ((dn[j] - dot) / qs);
          dsum += d[j] * wv[j];
        }
        for (int j = 0; j < C1; ++j) {
          const float ds = wv[j] * (d[j] - dsum);
          const float m = wargs[j], lv = wargs[C1 + j];
          const float sd = expf(0.5f * lv);
          const float ew = lds[M.EW + r * VL + j];
          dwa[j] = r < nvalid ? ds + a.w_kl_weight * inv_b * (m / ep) : 0.f;
          dwa[C1 + j] = r < nvalid ? ds * ew * 0.5f * sd + a.w_kl_weight * inv_b * (-0.5f * (1.f - sd * sd / ep)) : 0.f;
        }
      }
    }
    if (si + 2 < nst) { lds_barrier(); }
    VSTAMP(14 + si / 2);
  }
}

// The slab of one workgroup: the 12 tensors in host_offsets12 order, each row padded to a multiple of four columns
// (so the gradient tiles are stored 16 bytes at a time), each tensor starting on a multiple of four floats.
struct VSlabMap {
  int flat_off[12], slab_off[12], rows[12], cols[12];
  unsigned magic[12];              // ceil(2^32 / ceil4(cols)): slab index -> row
  int stride;
};
static VSlabMap vae_slab_map(int D, int H, int Hc, int C, int L, int use_xp, const int64_t* o) {
  const int NA = 2 * (C - 1), NZ = 2 * L, KD = C + (use_xp ? D : 0) + L;
  const int rows[12] = {D, 1, Hc, 1, D + C, 1, H, 1, KD, 1, H, 1};
  const int cols[12] = {Hc, Hc, NA, NA, H, H, NZ, NZ, H, H, D, D};
  VSlabMap m{};
  int off = 0;
  for (int t = 0; t < 12; ++t) {
    const int np = (cols[t] + 3) & ~3;
    m.flat_off[t] = o ? (int)o[t] : 0; m.slab_off[t] = off; m.rows[t] = rows[t]; m.cols[t] = cols[t];
    m.magic[t] = (unsigned)((0x100000000ULL + np - 1) / np);
    off += rows[t] * np;
  }
  m.stride = off;
  return m;
}

// grads = sum of the slabs (un-padded into the flat layout); five more blocks reduce the per-row losses to their
// batch means, and the step counter advances here when the caller asks for it (one launch instead of three)
struct VaeTail {
  const float* slab; int nslabs; float* grads; unsigned nblk_grads;
  const float* rownll; const float* rowkl; const float* rowloss; int B; float* means;   // means NULL: no loss blocks
  int32_t* bump;
  VSlabMap map;
};
__global__ __launch_bounds__(256) void slab_sum_kernel(VaeTail t) {
  if (blockIdx.x >= t.nblk_grads) {
    __shared__ float part[4];
    const int k = blockIdx.x - t.nblk_grads;            // 0 nll, 1 kl_z, 2..4 the rowloss columns
    const float* x = k == 0 ? t.rownll : k == 1 ? t.rowkl : t.rowloss + (k - 2);
    const int st = k < 2 ? 1 : 3;
    float acc = 0.f;
    for (int i = threadIdx.x; i < t.B; i += 256) acc += x[(size_t)i * st];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) t.means[k] = ((part[0] + part[1]) + (part[2] + part[3])) / (float)t.B;
    return;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && t.bump) *t.bump += 1;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= t.map.stride) return;
  int seg = 0;
#pragma unroll
  for (int k = 1; k < 12; ++k) seg = j >= t.map.slab_off[k] ? k : seg;
  int so = 0, fo = 0, nc = 1; unsigned mg = 0;
#pragma unroll
  for (int k = 0; k < 12; ++k)
    if (seg == k) { so = t.map.slab_off[k]; fo = t.map.flat_off[k]; nc = t.map.cols[k]; mg = t.map.magic[k]; }
  const int local = j - so, np = (nc + 3) & ~3;
  const int row = (int)__umulhi((unsigned)local, mg), col = local - row * np;
  if (col >= nc) return;                                 // padding column
  const float* slab = t.slab + j;
  const size_t n = (size_t)t.map.stride;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int s = 0;
  for (; s + 3 < t.nslabs; s += 4) {
    a0 += slab[(size_t)s * n]; a1 += slab[(size_t)(s + 1) * n];
    a2 += slab[(size_t)(s + 2) * n]; a3 += slab[(size_t)(s + 3) * n];
  }
  for (; s < t.nslabs; ++s) a0 += slab[(size_t)s * n];
  t.grads[fo + row * nc + col] = (a0 + a1) + (a2 + a3);
}

}  // namespace clv

using namespace clv;


#ifdef CLV_VAE_STAMPS
extern "C" int clv_debug_vae_stamps(unsigned long long* host_out, int n) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(clv::g_vae_stamps), sizeof(unsigned long long) * (n < 64 ? n : 64));
}
#endif

extern "C" int clv_vae_fused_supported(int D, int H, int Hc, int C, int L) {
  return D > 0 && D <= 96 && H > 0 && H <= 96 && Hc > 0 && Hc <= 96 && C >= 2 && C <= 16 && L > 0 && L <= 16;
}

extern "C" size_t clv_vae_fused_workspace_bytes(int B, int D, int H, int Hc, int C, int L, int use_x_prev) {
  if (!clv_vae_fused_supported(D, H, Hc, C, L) || B <= 0) return 0;
  return (size_t)((B + VRB - 1) / VRB) * (size_t)vae_slab_map(D, H, Hc, C, L, use_x_prev, nullptr).stride * sizeof(float);
}

// the 12 stages of cl_vae/model.py:136-219 and its backward pass, as the kernel's table
static void vae_build_stages(VStage* st, int D, int H, int Hc, int C, int L, int use_xp, const int64_t* o,
                             const VSlabMap& sm) {
  const VMap M;
  const int hw_k = (int)o[0], hw_b = (int)o[1], wa_k = (int)o[2], wa_b = (int)o[3], h_k = (int)o[4], h_b = (int)o[5],
            za_k = (int)o[6], za_b = (int)o[7], d_k = (int)o[8], d_b = (int)o[9], x_k = (int)o[10], x_b = (int)o[11];
  const int NA = 2 * (C - 1), NZ = 2 * L, xo = use_xp ? D : 0, KD = C + xo + L;
  auto fwd = [&](int a_off, int K, int w, int N, int b, int relu, int out) {
    VStage s{};
    s.flags = VF_MM | VF_BIAS | (relu ? VF_RELU : 0);
    s.a_off = a_off; s.K = K; s.ldw = N;
    s.w_off[0] = s.w_off[1] = w; s.ncol[0] = N; s.ncol[1] = 0; s.o_off[0] = s.o_off[1] = out;
    s.b_off = b;
    return s;
  };
  // backward stage on DY [16 x N]: (optional) product(s) DY . W^T and the gradients of the layer that produced DY
  auto bwd = [&](int dy, int N, int ga, int ga_K, int g, int gb) {
    VStage s{};
    const int nts = (N + 15) / 16;
    s.flags = VF_NT; s.a_off = dy; s.K = N; s.ldw = N;
    s.ga_off = ga; s.ga_K = ga_K; s.tiling = (nts << 24) | ((65536 + nts - 1) / nts);
    s.g_off = g; s.gb_off = gb;
    return s;
  };
  auto prod = [&](VStage& s, int which, int w, int ncol, int out) {
    s.flags |= VF_MM; s.w_off[which] = w; s.ncol[which] = ncol; s.o_off[which] = out;
    if (which == 0) { s.w_off[1] = w; s.ncol[1] = 0; s.o_off[1] = out; }
  };
  auto mask = [&](VStage& s, int m) { s.flags |= VF_MASK; s.mask_off = m; };
  st[0] = fwd(M.XC, D, hw_k, Hc, hw_b, 1, M.HW);                         // h_w = relu(x W)            (:141)
  st[1] = fwd(M.HW, Hc, wa_k, NA, wa_b, 0, M.WARGS);                     // [w_mean, w_log_var]        (:142-143)
  st[2] = fwd(M.XC, D + C, h_k, H, h_b, 1, M.Hh);                        // h = relu([x, w] W)         (:160-167)
  st[3] = fwd(M.Hh, H, za_k, NZ, za_b, 0, M.ZARGS);                      // [z_mean, z_log_var]        (:168-169)
  st[4] = fwd(M.DC, KD, d_k, H, d_b, 1, M.HD);                           // h_dec = relu([w, xp, z] W) (:177-186)
  st[5] = fwd(M.HD, H, x_k, D, x_b, 0, M.LG);                            // logits                     (:187-188)
  // (weights are read at their flat offsets, gradients are written at the slab's)
  st[6] = bwd(M.LG, D, M.HD, H, sm.slab_off[10], sm.slab_off[11]);                               // x_decoded_mean backward
  prod(st[6], 0, x_k, H, M.G1); mask(st[6], M.HD);                       //   G1 = d h_dec
  st[7] = bwd(M.G1, H, M.DC, KD, sm.slab_off[8], sm.slab_off[9]);                              // decoder_h backward
  st[7].flags |= VF_PAIR;
  prod(st[7], 0, d_k, C, M.DWV);                                         //   dL/dw (decoder part)
  prod(st[7], 1, d_k + (C + xo) * H, L, M.DZ);                           //   dL/dz
  st[8] = bwd(M.DZARGS, NZ, M.Hh, H, sm.slab_off[6], sm.slab_off[7]);                        // zargs backward
  prod(st[8], 0, za_k, H, M.G2); mask(st[8], M.Hh);                      //   G2 = d h
  st[9] = bwd(M.G2, H, M.XC, D + C, sm.slab_off[4], sm.slab_off[5]);                           // h backward
  st[9].flags |= VF_PAIR | VF_ACC;
  prod(st[9], 0, h_k + D * H, C, M.DWV);                                 //   dL/dw += encoder part
  st[10] = bwd(M.DWARGS, NA, M.HW, Hc, sm.slab_off[2], sm.slab_off[3]);                      // wargs backward
  prod(st[10], 0, wa_k, Hc, M.G1); mask(st[10], M.HW);                   //   G1 = d h_w
  st[11] = bwd(M.G1, Hc, M.XC, D, sm.slab_off[0], sm.slab_off[1]);                           // h_w backward (no product)
}

static int vae_step_impl(int B, int D, int H, int Hc, int C, int L, int use_x_prev,
                                     const float* x, const float* xp, const float* target, const float* onehot,
                                     const clv_label_stage* stage, float* eps_w, float* eps_z,
                                     const float* params, const int64_t* host_offsets12, long n_params,
                                     float prior_logvar, float class_weight, float kl_weight, float w_kl_weight,
                                     int need_grads, float* grads, void* ws, size_t ws_bytes,
                                     float* logits, float* w_out, float* wargs_out, float* zargs_out,
                                     float* rownll, float* rowkl, float* rowloss, const clv_vae_step_opts* opts,
                                     void* stream) {
  if (!clv_vae_fused_supported(D, H, Hc, C, L) || B <= 0 || n_params <= 0 || n_params > 0x7fffffffL) return CLV_EINVAL;
  if (!eps_w || !eps_z || !params || !host_offsets12 || !w_out || !wargs_out || !zargs_out || !rownll || !rowkl || !rowloss)
    return CLV_EINVAL;
  if (stage) {
    if (!stage->cur || !stage->w_src || (use_x_prev && !stage->hist) || (stage->cursor.step_dev && stage->cursor.period < 1))
      return CLV_EINVAL;
  } else {
    if (!x || (use_x_prev && !xp) || (need_grads && !onehot)) return CLV_EINVAL;
  }
  if (need_grads && !grads) return CLV_EINVAL;
  const int nwg = (B + VRB - 1) / VRB;
  const VSlabMap sm = vae_slab_map(D, H, Hc, C, L, use_x_prev, host_offsets12);
  if (need_grads && (!ws || ws_bytes < (size_t)nwg * sm.stride * sizeof(float))) return CLV_EWORKSPACE;
  VaeArgs a{};
  a.B = B; a.D = D; a.H = H; a.Hc = Hc; a.C = C; a.L = L; a.use_xp = use_x_prev;
  a.x = x; a.xp = xp; a.onehot = onehot; a.y = target == x ? nullptr : target;
  if (stage) {
    const clv_label_stage& g = *stage;
    a.y = nullptr;
    a.stage = VaeArgs::Stage{1, g.cur, g.hist, (long long)g.cur_stride, (long long)g.cur_offset, (long long)g.hist_stride,
                             (long long)g.hist_offset, (long long)g.row0, (const long long*)g.cur_table, (const long long*)g.hist_table,
                             (const long long*)g.idx, g.cursor.step_dev, g.cursor.step0, g.cursor.period, (long long)g.cursor.stride,
                             (long long)g.cursor.offset, g.w_src, g.X, g.Xh, g.w_out};
  }
  a.eps_w = eps_w; a.eps_z = eps_z; a.P = params;
  a.prior = prior_logvar; a.class_weight = class_weight; a.kl_weight = kl_weight; a.w_kl_weight = w_kl_weight;
  a.need_grads = need_grads;
  if (opts && opts->draw) {
    a.draw = 1; a.k0 = (uint32_t)opts->noise_seed; a.k1 = (uint32_t)(opts->noise_seed >> 32);
    a.stream_w = opts->stream_w; a.stream_z = opts->stream_z; a.step = opts->step; a.step_dev = opts->step_dev;
    a.first_w = opts->first_w; a.first_z = opts->first_z;
  }
  a.slab = (float*)ws; a.slab_stride = sm.stride;
  a.logits = logits; a.w_out = w_out; a.wargs_out = wargs_out; a.zargs_out = zargs_out;
  a.rownll = rownll; a.rowkl = rowkl; a.rowloss = rowloss;
  vae_build_stages(a.st, D, H, Hc, C, L, use_x_prev, host_offsets12, sm);
  hipStream_t s = (hipStream_t)stream;
  {
    ProfScope p("vae_fused_step", s);
    const bool k24 = C + (use_x_prev ? D : 0) + L <= 96 && D + C <= 96;     // the longest products fit 24 k-steps
    const bool bf16 = opts && opts->bf16;
    auto k = bf16 ? (k24 ? vae_fused_kernel<24, true> : vae_fused_kernel<32, true>)
                  : (k24 ? vae_fused_kernel<24, false> : vae_fused_kernel<32, false>);
    const size_t lds = (size_t)VMap().total * sizeof(float);
    if (int e = allow_dynamic_lds(reinterpret_cast<const void*>(k), (int)lds)) return e;
    hipLaunchKernelGGL(k, dim3(nwg), dim3(VNT), lds, s, a);
  }
  int st = launch_status();
  float* means = opts ? opts->loss_means : nullptr;
  int32_t* bump = (opts && need_grads) ? opts->bump_iterations : nullptr;
  if (st || (!need_grads && !means)) return st;
  {
    ProfScope p("vae_slab_sum", s);
    VaeTail t{(const float*)ws, nwg, grads, need_grads ? (unsigned)((sm.stride + 255) / 256) : 0u,
              rownll, rowkl, rowloss, B, means, bump, sm};
    hipLaunchKernelGGL(slab_sum_kernel, dim3(t.nblk_grads + (means ? 5u : 0u)), dim3(256), 0, s, t);
  }
  return launch_status();
}

extern "C" int clv_vae_fused_step(int B, int D, int H, int Hc, int C, int L, int use_x_prev,
                                  const float* x, const float* xp, const float* target, const float* onehot,
                                  const clv_label_stage* stage, float* eps_w, float* eps_z,
                                  const float* params, const int64_t* host_offsets12, long n_params,
                                  float prior_logvar, float class_weight, float kl_weight, float w_kl_weight,
                                  int need_grads, float* grads, void* ws, size_t ws_bytes,
                                  float* logits, float* w_out, float* wargs_out, float* zargs_out,
                                  float* rownll, float* rowkl, float* rowloss, const clv_vae_step_opts* opts,
                                  void* stream) {
  if (stage)      // the workgroups assemble their own rows: x / xp / target / onehot are not read
    return vae_step_impl(B, D, H, Hc, C, L, use_x_prev, nullptr, nullptr, nullptr, nullptr, stage, eps_w, eps_z, params, host_offsets12,
                         n_params, prior_logvar, class_weight, kl_weight, w_kl_weight, need_grads, grads, ws, ws_bytes, logits, w_out,
                         wargs_out, zargs_out, rownll, rowkl, rowloss, opts, stream);
  return vae_step_impl(B, D, H, Hc, C, L, use_x_prev, x, xp, target, onehot, nullptr, eps_w, eps_z, params, host_offsets12, n_params,
                       prior_logvar, class_weight, kl_weight, w_kl_weight, need_grads, grads, ws, ws_bytes, logits, w_out, wargs_out,
                       zargs_out, rownll, rowkl, rowloss, opts, stream);
}
