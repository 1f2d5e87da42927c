// sparse_proj.hip -- input projection of piano-roll frames: out[r,:] = sum_k X[r,k] * K[k,:]   (gfx950)
//
// The LSTM input projections x_t . K_x (cl_vrnn/model.py:193-196, 218-226 as one [B*T,88] x [88,352] product
// per LSTM) multiply frames that are ~4 % nonzero (0.0443 note density): a dense GEMM spends 22 of
// every 23 FMAs on zeros and is bound by its own operand traffic.  Here the kernel K [nx,N] (124 KB for
// 88 x 352) stays in LDS for the lifetime of a persistent workgroup and each wave adds the kernel rows of its
// frame's nonzero inputs.  Exact for any float input (a dense frame just costs nx row additions); summation
// order = ascending input index, like the dense product.
#include "common.h"
#include <type_traits>

namespace clv {

constexpr int SP_NW = 16;          // waves per workgroup, each one owns whole frames
constexpr int SP_NT = SP_NW * 64;
constexpr int SP_XMAX = 128;       // max inputs per frame (2 per lane)
constexpr int SP_NC = 6;           // output columns per lane (N <= 384)

struct SparseProjSet {
  int nx, ldx;
  const void* X;       // [R, ldx] float, or uint8 (XU8 kernel: ldx in bytes)
  const float* K;      // [nx, N]
  float* out;          // [R, ldo]
};
struct SparseProjArgs {
  int R, N, ldo, wgs;  // wgs: workgroups per projection (a launch carries one or two projections over the same R frames)
  SparseProjSet set[2];
};

// A wave handles a frame on its own: lane k holds inputs k and k+64, two ballots give the nonzero sets as
// scalar masks, and a scalar bit-scan loop adds the listed kernel rows (LDS) into the lane's 6 output columns.
// No lists, no LDS traffic besides the kernel rows, no barriers after the kernel is staged.
// (Round 3, tried: four consecutive columns per lane -- 2 ds_read_b128 per note and 2 float4 stores per frame instead of 6
// + 6 dword accesses: 34.3 us against 32.6 at configuration 3 (HIP events), 202 against 196 at configuration 5.  The dword
// form's 256-byte wave stores are what the write path likes.)
template <bool XU8>
__global__ __launch_bounds__(SP_NT) void sparse_proj_kernel(SparseProjArgs g) {
  extern __shared__ __attribute__((aligned(16))) float Kl[];            // [nx][N]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // second projection of the launch: its workgroups start as the first one's retire, so the two tails overlap
  const bool second = (int)blockIdx.x >= g.wgs;
  struct { int R, nx, N, ldx, ldo; const void* X; const float* K; float* out; } a;
  a.R = g.R; a.N = g.N; a.ldo = g.ldo;
  a.nx = second ? g.set[1].nx : g.set[0].nx; a.ldx = second ? g.set[1].ldx : g.set[0].ldx;
  a.X = second ? g.set[1].X : g.set[0].X; a.K = second ? g.set[1].K : g.set[0].K;
  a.out = second ? g.set[1].out : g.set[0].out;
  const int bid = (int)blockIdx.x - (second ? g.wgs : 0);
  {
    const int nv = a.nx * a.N / 4;
    const float4* src = reinterpret_cast<const float4*>(a.K);
    float4* dst = reinterpret_cast<float4*>(Kl);
    for (int i0 = tid; i0 < nv; i0 += 4 * SP_NT) {      // 4 loads in flight per thread
      float4 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = src[min(i0 + q * SP_NT, nv - 1)];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (i0 + q * SP_NT < nv) dst[i0 + q * SP_NT] = v[q];
    }
  }
  __syncthreads();
  const int per = (a.R + g.wgs - 1) / g.wgs;
  const int first = bid * per, last = min(a.R, first + per);
  int colo[SP_NC];
#pragma unroll
  for (int c = 0; c < SP_NC; ++c) colo[c] = min(lane + 64 * c, a.N - 1);
  const int k0 = min(lane, a.nx - 1), k1 = min(lane + 64, a.nx - 1);
  const bool v0 = lane < a.nx, v1 = lane + 64 < a.nx;
  // a frame's two inputs of this lane, as loaded (byte frames: the bytes, widened where they are USED -- a byte IS its float
  // value; converted here, the compiler waits for the load at once and the request one frame ahead hides nothing: round 6)
  typedef typename std::conditional<XU8, unsigned, float>::type raw_t;
  auto frame = [&](int f, raw_t& x0, raw_t& x1) {
    const size_t o = (size_t)min(f, a.R - 1) * a.ldx;
    if constexpr (XU8) { const unsigned char* bp = static_cast<const unsigned char*>(a.X) + o; x0 = bp[k0]; x1 = bp[k1]; }
    else { const float* fp = static_cast<const float*>(a.X) + o; x0 = fp[k0]; x1 = fp[k1]; }
  };
  raw_t nx0, nx1;
  frame(first + wave, nx0, nx1);
  for (int f = first + wave; f < last; f += SP_NW) {
    const float fx0 = (float)nx0, fx1 = (float)nx1;
    frame(f + SP_NW, nx0, nx1);                           // next frame of this wave (clamped, unconditional)
    unsigned long long m0 = __ballot(v0 && fx0 != 0.f), m1 = __ballot(v1 && fx1 != 0.f);
    float acc[SP_NC];
#pragma unroll
    for (int c = 0; c < SP_NC; ++c) acc[c] = 0.f;
    while (m0) {                                          // scalar loop over the notes that are on
      const int k = __builtin_ctzll(m0);
      m0 &= m0 - 1;
      const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fx0), k));
      const float* kr = Kl + k * a.N;
#pragma unroll
      for (int c = 0; c < SP_NC; ++c) acc[c] = fmaf(v, kr[colo[c]], acc[c]);
    }
    while (m1) {
      const int k = __builtin_ctzll(m1);
      m1 &= m1 - 1;
      const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fx1), k));
      const float* kr = Kl + (k + 64) * a.N;
#pragma unroll
      for (int c = 0; c < SP_NC; ++c) acc[c] = fmaf(v, kr[colo[c]], acc[c]);
    }
    float* op = a.out + (size_t)f * a.ldo;
#pragma unroll
    for (int c = 0; c < SP_NC; ++c)
      if (lane + 64 * c < a.N) op[lane + 64 * c] = acc[c];
  }
}

// ---------------------------------------------------------------------------
// Dense layer over a whole sparse window: out[r,:] = act(sum_j X[r,j] K[j,:] + bias), nx ~ 10^4 (the label
// path's hW layer reads the flattened window, cl_vrnn/model.py:174-176).  K stays in HBM/L2; a workgroup owns
// one row, its 16 waves scan 64-input chunks, ballot the nonzeros and add the listed kernel rows (one float2 per
// lane, N <= 128 and even), 4 row loads in flight; the 16 partial sums meet in LDS.
// ---------------------------------------------------------------------------
struct SparseDenseArgs {
  int R, nx, N, ldx, ldo, act;
  const float* X;
  const float* K;      // [nx, N]
  const float* bias;   // [N] or null
  float* out;
};

__global__ __launch_bounds__(1024) void sparse_dense_kernel(SparseDenseArgs a) {
  __shared__ float2 part[16][64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = blockIdx.x;
  const int n2 = a.N / 2;
  const int lc = min(lane, n2 - 1);
  const float2* K2 = reinterpret_cast<const float2*>(a.K);
  const float* xr = a.X + (size_t)r * a.ldx;
  float2 acc = make_float2(0.f, 0.f);
  const int nchunk = (a.nx + 63) / 64;
  float xn = 0.f;
  if (wave < nchunk) xn = xr[min(wave * 64 + lane, a.nx - 1)];
  for (int ch = wave; ch < nchunk; ch += 16) {
    const float x = xn;
    const int j0 = ch * 64;
    if (ch + 16 < nchunk) xn = xr[min((ch + 16) * 64 + lane, a.nx - 1)];
    unsigned long long m = __ballot(j0 + lane < a.nx && x != 0.f);
    while (m) {                                   // up to 4 kernel rows in flight
      int kk[4];
      float vv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool on = m != 0;
        const int bit = on ? __builtin_ctzll(m) : 0;
        m = on ? (m & (m - 1)) : 0;
        kk[q] = j0 + bit;
        const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), bit));
        vv[q] = on ? v : 0.f;
      }
      float2 kr[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) kr[q] = K2[(size_t)min(kk[q], a.nx - 1) * n2 + lc];
#pragma unroll
      for (int q = 0; q < 4; ++q) { acc.x = fmaf(vv[q], kr[q].x, acc.x); acc.y = fmaf(vv[q], kr[q].y, acc.y); }
    }
  }
  part[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && lane < n2) {
    float2 t = make_float2(0.f, 0.f);
#pragma unroll
    for (int w = 0; w < 16; ++w) { t.x += part[w][lane].x; t.y += part[w][lane].y; }
    if (a.bias) { t.x += a.bias[2 * lane]; t.y += a.bias[2 * lane + 1]; }
    if (a.act == CLV_ACT_RELU) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); }
    float* op = a.out + (size_t)r * a.ldo + 2 * lane;
    op[0] = t.x; op[1] = t.y;
  }
}

// ---------------------------------------------------------------------------
// Its weight gradient: dK[j,:] = sum_b X[b,j] G[b,:]  (X [Bn,nx] sparse, G [Bn,N] dense, N <= 128 even).
// A workgroup owns 64 consecutive inputs j; per block of 128 batch rows it stages X[b, j0:j0+64] (transposed
// access through a padded LDS tile) and G[b,:] in LDS, and each wave walks the nonzero b of its 4 inputs.
// Every output row is written once: no split-K slabs, no reduce pass.
// ---------------------------------------------------------------------------
constexpr int SO_BB = 128;         // batch rows per staged block
constexpr int SO_JT = 64;          // inputs per workgroup
constexpr int SO_XS = SO_JT + 1;   // padded tile stride: a column read hits 64 different banks
struct SparseOuterArgs {
  int Bn, nx, N, ldx, ldg, ldo;
  const float* X;
  const float* G;
  float* out;          // [nx, ldo]
  float* colsum;       // [N] = sum_b G[b,:] (the layer's bias gradient) or null; written by workgroup 0
  // gdot [N] = sum_b (Hact[b,c] - hbias[c]) G[b,c] (or null): Hact = relu(X.K + hbias) is the layer's output and G its
  // relu-masked upstream gradient, so this is sum_b (X.K)[b,c] G[b,c] = sum_j K[j,c] dK[j,c]: the weight-norm optimizer's
  // sum g.W per column without a pass over K and dK (clv_adam_wn_step).  Formed by one extra workgroup of the launch.
  const float* Hact; const float* hbias; float* gdot; int ldh;
};

__global__ __launch_bounds__(1024) void sparse_outer_kernel(SparseOuterArgs a) {
  extern __shared__ __attribute__((aligned(16))) float so_lds[];
  float* Xt = so_lds;                               // [SO_BB][SO_XS]
  float2* Gl = reinterpret_cast<float2*>(so_lds + SO_BB * SO_XS);          // [SO_BB][N/2]   (SO_BB * SO_XS is even)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j0 = blockIdx.x * SO_JT;
  const int n2 = a.N / 2;
  const int lc = min(lane, n2 - 1);
  float2 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = make_float2(0.f, 0.f);
  float2 csum = make_float2(0.f, 0.f);
  if (a.gdot && (int)blockIdx.x == (int)gridDim.x - 1) {
    // the extra workgroup behind the tiles: gdot[c] = sum_b (Hact[b,c] - hbias[c]) G[b,c].  It runs next to the tile
    // workgroups (fewer of them than CUs), every thread takes one column pair and every RL-th batch row with all of
    // its loads in flight, the row lanes meet in LDS.
    constexpr int RL = 16;                       // row lanes (x 64 column-pair lanes = 1024 threads)
    float2* redl = reinterpret_cast<float2*>(so_lds);      // [RL][64]
    const int rl = wave;
    float2 acc = make_float2(0.f, 0.f);
    if (lane < n2) {
      const float2 hb = make_float2(a.hbias[2 * lane], a.hbias[2 * lane + 1]);
      for (int b0 = rl; b0 < a.Bn; b0 += 8 * RL) {
        float2 hv[8], gv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int bb = min(b0 + i * RL, a.Bn - 1);
          hv[i] = *reinterpret_cast<const float2*>(a.Hact + (size_t)bb * a.ldh + 2 * lane);
          gv[i] = *reinterpret_cast<const float2*>(a.G + (size_t)bb * a.ldg + 2 * lane);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float mk = b0 + i * RL < a.Bn ? 1.f : 0.f;
          acc.x = fmaf((hv[i].x - hb.x) * mk, gv[i].x, acc.x);
          acc.y = fmaf((hv[i].y - hb.y) * mk, gv[i].y, acc.y);
        }
      }
    }
    redl[rl * 64 + lane] = acc;
    __syncthreads();
    if (wave == 0 && lane < n2) {
      float2 t = make_float2(0.f, 0.f);
#pragma unroll
      for (int w = 0; w < RL; ++w) { t.x += redl[w * 64 + lane].x; t.y += redl[w * 64 + lane].y; }
      a.gdot[2 * lane] = t.x; a.gdot[2 * lane + 1] = t.y;
    }
    return;
  }
  // The next block of batch rows is fetched into registers while this one is walked (the kernel is two or three round
  // trips to L2 and a handful of FMAs: the fetch of block 2 under the walk of block 1 is a third of its time)
  constexpr int XR = SO_BB * SO_JT / 1024, GR = SO_BB * 64 / 1024;      // values per thread: 8 of X, up to 8 pairs of G
  float xr[XR];
  float2 gr[GR];
  auto fetch = [&](int b0) {
    const int nb = min(SO_BB, a.Bn - b0);
#pragma unroll
    for (int i = 0; i < XR; ++i) {
      const int e = tid + 1024 * i, bb = e / SO_JT, jj = e % SO_JT;
      const bool ok = bb < nb && j0 + jj < a.nx;
      const float v = a.X[(size_t)(b0 + min(bb, nb - 1)) * a.ldx + min(j0 + jj, a.nx - 1)];
      xr[i] = v * (ok ? 1.f : 0.f);
    }
#pragma unroll
    for (int i = 0; i < GR; ++i) {
      const int e = min(tid + 1024 * i, SO_BB * n2 - 1), bb = e / n2, c = e % n2;
      const float2 v = *reinterpret_cast<const float2*>(a.G + (size_t)(b0 + min(bb, nb - 1)) * a.ldg + 2 * c);
      const float mk = bb < nb ? 1.f : 0.f;
      gr[i] = make_float2(v.x * mk, v.y * mk);
    }
  };
  fetch(0);
  for (int b0 = 0; b0 < a.Bn; b0 += SO_BB) {
#pragma unroll
    for (int i = 0; i < XR; ++i) {
      const int e = tid + 1024 * i;
      Xt[(e / SO_JT) * SO_XS + e % SO_JT] = xr[i];
    }
#pragma unroll
    for (int i = 0; i < GR; ++i)
      if (tid + 1024 * i < SO_BB * n2) Gl[tid + 1024 * i] = gr[i];
    if (b0 + SO_BB < a.Bn) fetch(b0 + SO_BB);
    __syncthreads();
    if (a.colsum && blockIdx.x == 0 && wave == 15 && lane < n2) {      // bias gradient: column sums of the staged G block
      float2 t = make_float2(0.f, 0.f);
      for (int bb = 0; bb < SO_BB; bb += 4) {
        const float2 g0 = Gl[bb * n2 + lane], g1 = Gl[(bb + 1) * n2 + lane], g2 = Gl[(bb + 2) * n2 + lane], g3 = Gl[(bb + 3) * n2 + lane];
        t.x += (g0.x + g1.x) + (g2.x + g3.x);
        t.y += (g0.y + g1.y) + (g2.y + g3.y);
      }
      csum.x += t.x; csum.y += t.y;
    }

#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int jj = wave + 16 * i;                 // this wave's i-th input of the tile
      const float x0 = Xt[lane * SO_XS + jj], x1 = Xt[(lane + 64) * SO_XS + jj];
      unsigned long long m0 = __ballot(x0 != 0.f), m1 = __ballot(x1 != 0.f);
      while (m0) {
        const int bit = __builtin_ctzll(m0);
        m0 &= m0 - 1;
        const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x0), bit));
        const float2 g2 = Gl[bit * n2 + lc];
        acc[i].x = fmaf(v, g2.x, acc[i].x); acc[i].y = fmaf(v, g2.y, acc[i].y);
      }
      while (m1) {
        const int bit = __builtin_ctzll(m1);
        m1 &= m1 - 1;
        const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x1), bit));
        const float2 g2 = Gl[(bit + 64) * n2 + lc];
        acc[i].x = fmaf(v, g2.x, acc[i].x); acc[i].y = fmaf(v, g2.y, acc[i].y);
      }
    }
    __syncthreads();
  }
  if (a.colsum && blockIdx.x == 0 && wave == 15 && lane < n2) {
    a.colsum[2 * lane] = csum.x;
    a.colsum[2 * lane + 1] = csum.y;
  }

#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = j0 + wave + 16 * i;
    if (j < a.nx && lane < n2) {
      float* op = a.out + (size_t)j * a.ldo + 2 * lane;
      op[0] = acc[i].x; op[1] = acc[i].y;
    }
  }
}

}  // namespace clv

extern "C" size_t clv_sparse_proj_lds_bytes(int nx, int N) { return (size_t)nx * N * sizeof(float); }

extern "C" int clv_sparse_proj_supported(int nx, int N) {
  return nx >= 1 && nx <= clv::SP_XMAX && N >= 1 && N <= clv::SP_NC * 64 && (nx * N) % 4 == 0 &&
         clv_sparse_proj_lds_bytes(nx, N) <= 150 * 1024;
}

static int sparse_proj_launch(int R, int N, int ldo, int nset, const clv::SparseProjSet* sets, bool x_u8, hipStream_t s) {
  using namespace clv;
  int nxmax = 0;
  for (int i = 0; i < nset; ++i) {
    const SparseProjSet& p = sets[i];
    if (!p.X || !p.K || !p.out || p.ldx < p.nx || !clv_sparse_proj_supported(p.nx, N) || ((uintptr_t)p.K) % 16 != 0)
      return CLV_EINVAL;
    nxmax = p.nx > nxmax ? p.nx : nxmax;
  }
  if (R <= 0 || ldo < N) return CLV_EINVAL;
  const void* kern = x_u8 ? reinterpret_cast<const void*>(sparse_proj_kernel<true>) : reinterpret_cast<const void*>(sparse_proj_kernel<false>);
  if (int e = clv::allow_dynamic_lds(kern, 150 * 1024)) return e;
  SparseProjArgs a;
  memset(&a, 0, sizeof(a));
  a.R = R; a.N = N; a.ldo = ldo;
  // one persistent workgroup per CU IN ALL: 256 / nset per projection.  (Round 4: with 256 per projection every CU staged its
  // 124 KB of kernel twice, once per projection, for 8 frames per wave each: 26.2 us at configuration 3 against 23.6; 192: 29.7,
  // 64: 36.3)
  const int cap = 256 / (nset > 0 ? nset : 1);
  a.wgs = R < cap * SP_NW ? (R + SP_NW - 1) / SP_NW : cap;
  for (int i = 0; i < nset; ++i) a.set[i] = sets[i];
  ProfScope p("sparse_proj", s);
  if (x_u8) hipLaunchKernelGGL(sparse_proj_kernel<true>, dim3(nset * a.wgs), dim3(SP_NT), clv_sparse_proj_lds_bytes(nxmax, N), s, a);
  else hipLaunchKernelGGL(sparse_proj_kernel<false>, dim3(nset * a.wgs), dim3(SP_NT), clv_sparse_proj_lds_bytes(nxmax, N), s, a);
  return launch_status();
}

extern "C" int clv_sparse_proj(int R, int nx, int N, const void* X, int x_u8, int ldx, const float* K, float* out, int ldo,
                               void* stream) {
  const clv::SparseProjSet set{nx, ldx, X, K, out};
  return sparse_proj_launch(R, N, ldo, 1, &set, x_u8 != 0, (hipStream_t)stream);
}

extern "C" int clv_sparse_proj2(int R, int N, int ldo, int x_u8, int nx0, const void* X0, int ldx0, const float* K0, float* out0,
                                int nx1, const void* X1, int ldx1, const float* K1, float* out1, void* stream) {
  const clv::SparseProjSet sets[2] = {{nx0, ldx0, X0, K0, out0}, {nx1, ldx1, X1, K1, out1}};
  return sparse_proj_launch(R, N, ldo, 2, sets, x_u8 != 0, (hipStream_t)stream);
}

extern "C" int clv_sparse_dense_supported(int N) { return N >= 2 && N <= 128 && N % 2 == 0; }

extern "C" int clv_sparse_dense(int R, int nx, int N, const float* X, int ldx, const float* K, const float* bias, int act,
                                float* out, int ldo, void* stream) {
  using namespace clv;
  if (R <= 0 || nx <= 0 || !X || !K || !out || ldx < nx || ldo < N || !clv_sparse_dense_supported(N)) return CLV_EINVAL;
  if (act != CLV_ACT_NONE && act != CLV_ACT_RELU) return CLV_EINVAL;
  if (((uintptr_t)K) % 8 != 0) return CLV_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  SparseDenseArgs a{R, nx, N, ldx, ldo, act, X, K, bias, out};
  ProfScope p("sparse_dense", s);
  hipLaunchKernelGGL(sparse_dense_kernel, dim3(R), dim3(1024), 0, s, a);
  return launch_status();
}

extern "C" int clv_sparse_outer(int Bn, int nx, int N, const float* X, int ldx, const float* G, int ldg, float* out, int ldo,
                                   float* colsum, const float* Hact, int ldh, const float* hbias, float* gdot, void* stream) {
  using namespace clv;
  if (gdot && (!Hact || !hbias || ldh < N || ldh % 2 != 0 || ((uintptr_t)Hact) % 8 != 0)) return CLV_EINVAL;
  if (Bn <= 0 || nx <= 0 || !X || !G || !out || ldx < nx || ldg < N || ldo < N || !clv_sparse_dense_supported(N))
    return CLV_EINVAL;
  if (((uintptr_t)G) % 8 != 0 || ldg % 2 != 0) return CLV_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = (size_t)(SO_BB * SO_XS + SO_BB * N) * sizeof(float);
  if (int e = clv::allow_dynamic_lds(reinterpret_cast<const void*>(sparse_outer_kernel), 150 * 1024)) return e;
  SparseOuterArgs a{Bn, nx, N, ldx, ldg, ldo, X, G, out, colsum, Hact, hbias, gdot, ldh};
  ProfScope p("sparse_outer", s);
  hipLaunchKernelGGL(sparse_outer_kernel, dim3((nx + SO_JT - 1) / SO_JT + (gdot ? 1 : 0)), dim3(1024), lds, s, a);
  return launch_status();
}
