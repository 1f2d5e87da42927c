/* clvae.h -- C ABI of libclvae_hip.so: the MI355X (gfx950) hot path of
 * mobeets/classifying-vae-lstm (cl_vae / cl_vrnn training + sampling arithmetic).
 *
 * The reference has no FFI: its arithmetic runs inside Keras 2.0.0 / TF 1.0.1
 * behind code/cl_vae/model.py and code/cl_vrnn/model.py.  This header is the
 * boundary a maintainer would bind (ctypes stub in INTEGRATION.md); each entry
 * point cites the reference lines it replaces (paths relative to
 * /root/reference/code).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller unless it is marked
 *     "host"; tensors are row-major, contiguous, float32 unless stated;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it and
 *     nothing synchronises (graph-capture safe, no hidden allocation);
 *   - scratch memory comes from the caller: ask `*_workspace_bytes`, pass `ws`;
 *   - return value: 0 on success, a negative CLV_E* code or a positive
 *     hipError_t otherwise; no exceptions cross the ABI;
 *   - no global state except the opt-in profiler (clv_prof_*).
 *
 * Weight layouts are Keras': Dense kernel [in,out], bias [out]; LSTM kernel
 * [in,4H], recurrent_kernel [H,4H], bias [4H], gate blocks i,f,c,o.
 */
#ifndef CLVAE_H
#define CLVAE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CLV_OK          0
#define CLV_EINVAL     (-1)   /* bad argument / unsupported shape          */
#define CLV_EWORKSPACE (-2)   /* workspace too small                       */
#define CLV_ENOGPU     (-3)   /* no gfx950 device visible                  */

#define CLV_ACT_NONE     0
#define CLV_ACT_RELU     1
#define CLV_ACT_SIGMOID  2
#define CLV_ACT_MASKPOS  3   /* out = acc * (aux[m,n] > 0): relu' through a saved activation */

#define CLV_GATE_HARD_SIGMOID 0   /* Keras 2.0.0 default recurrent_activation */
#define CLV_GATE_SIGMOID      1

/* The sources of clv_lstm_pair_pack, for a kernel that writes the pack as a by-product of its own launch
 * (clv_vrnn_label_fwd_x: it runs right before the pair forward kernel). */
typedef struct clv_pair_pack_src {
  int H, L;
  const float *U_enc, *U_dec, *Kz, *Wz;
  float* pack;
} clv_pair_pack_src;

/* In-kernel noise: a kernel that takes `const clv_noise_draw* noise` draws its standard-normal eps itself when noise is
 * not NULL -- element e of the launch's eps tensor gets the clv_philox_normal value of (seed, step + *step_dev, stream,
 * index first + e), i.e. bit for bit what clv_philox_normal / clv_philox_normal2 would have written -- and WRITES it to
 * the eps buffer (the backward pass reads it there); noise == NULL: eps is read.  first = the global index of the
 * launch's element 0 (data-parallel ranks draw at their global sample indices). */
typedef struct clv_noise_draw {
  uint64_t seed, first;
  uint32_t stream, step;
  const int32_t* step_dev;
} clv_noise_draw;

/* Which mini-batch a staging launch assembles, chosen by a DEVICE step counter (clv_gather_rows_multi,
 * clv_vrnn_label_fwd_x): batch j = (*step_dev - step0) mod period, rows idx[j * stride + offset + r]. */
typedef struct clv_batch_cursor {
  const int32_t* step_dev;
  int32_t step0, period;
  int64_t stride, offset;
} clv_batch_cursor;

/* clv_vrnn_label_fwd_x(stage != NULL): the step's MINI-BATCH ASSEMBLY inside the launch (cl_vae/train.py:66-71, the host-side slicing of
 * Model.fit; what clv_gather_rows_multi does as a launch of its own): the workgroup of batch row b resolves its source
 * row sr = idx[base + b] (idx NULL: row0 + base + b; base from the batch cursor), converts the row's byte frames into the float
 * rows every later launch of the step reads --
 *   X[b, :nx]                         = cur  + (cur_table  ? cur_table[sr]  : sr) * cur_stride  + cur_offset   (bytes)
 *   Xh + (b * nx / hist_chunk + p) * hist_ld, hist_chunk floats per frame p
 *                                     = hist + (hist_table ? hist_table[sr] : sr) * hist_stride + hist_offset  (hist NULL: none)
 *   w_out[b, :C] = w_src[sr, :C]      (both NULL: none; the label path reads w_src[sr] either way)
 * -- and scans the bytes itself.  nx, hist_chunk, hist_ld, the strides and offsets are multiples of 4; stores 4-byte, X / Xh
 * 16-byte aligned; X and onehot of the call are then not read (the labels are w_src).  The assembly then costs no launch
 * (configuration 3: 8.8 us). */
typedef struct clv_label_stage {
  const uint8_t* cur; const uint8_t* hist;
  int64_t cur_stride, cur_offset, hist_stride, hist_offset, row0;
  const int64_t* cur_table; const int64_t* hist_table; const int64_t* idx;
  clv_batch_cursor cursor;                 /* step_dev NULL: no cursor */
  float* X; float* Xh; int32_t hist_chunk; int64_t hist_ld;
  const float* w_src; float* w_out;
  uint8_t* X8; uint8_t* Xh8;               /* clv_vrnn_label_fwd_x only; not NULL: the rows are copied as BYTES into X8 / Xh8 ([B, nx]
                                            * each) INSTEAD of widened into X / Xh (which may then be NULL): the batch of a step whose
                                            * later launches read frames as bytes (x_u8 / y_u8 / CLV_FRAMES_U8) */
} clv_label_stage;

/* clv_vrnn_label_fwd_x(proj != NULL; needs stage with X8): the launch ALSO forms the LSTMs' frame projections of the same
 * mini-batch (clv_sparse_proj2's product, bit for bit) -- out_cur[b * T + t, :N] = frame t of the current row b . K_cur
 * ([D, N]), out_hist likewise over the history rows and K_hist (NULL: none) -- in workgroups of their own that read the byte
 * stores through the stage and run beside the label rows' workgroups on the same CUs (the label path waits on L2 round trips,
 * the projection on its stores: configuration 3, 24 + 23 us as two launches, 33 us as one).  A frame is D bytes, nx = T * D;
 * clv_vrnn_label_fwd_x_proj_supported says which shapes.  cl_vrnn/model.py:193-196, 218-226. */
typedef struct clv_frame_proj {
  int32_t T, N, ldo;
  const float* K_cur; float* out_cur;
  const float* K_hist; float* out_hist;
} clv_frame_proj;

/* ABI version = CLV_ABI_VERSION of the header the library was built from.  It changes whenever an existing entry point
 * changes its argument list or the size / meaning of a buffer (a caller built against an older header would still resolve
 * the symbol): the binding compares it at load time and refuses a mismatch.
 *   100  rounds 1-2
 *   300  round 3: clv_lstm_pair_fwd / _bwd / clv_vrnn_label_fwd_x took new trailing pointers; the pair kernels' aux_* buffers
 *        are [B*T, 2, H] (kcarry, kc), no longer [B*T, H] cell states
 *   400  round 4: additions only (the large-batch sequence kernels, the batch cursor, the paired kernel-gradient launch, the
 *        dense bf16 products of the hW layer, mini-batch assembly inside the label / cl_vae launches)
 *   500  round 5: the coef / aux buffers between clv_lstm_mx_fwd and clv_lstm_mx_bwd are unit-major records ([B*T,H,4] and
 *        [B*T,H,2]; same sizes); clv_lstm_seq_fwd / _bwd take any H <= 1024; + clv_dropout_rows
 *   600  round 6: ONE entry point per operation -- the *_ex / *_staged / *_means / *_notes / *_cursor variants took over the base
 *        names (their argument lists) and the thin wrappers are gone; frames may be passed as BYTES (x_u8 / y_u8 /
 *        CLV_FRAMES_U8 of lstm_mx_fwd, lstm_wgrad, out_head_train, dense_window_fwd_bf16, dense_outer_bf16; src_u8 == 2 of the
 *        gather); clv_latent_head_fwd draws its own eps (noise); removed: the 4x4x1 f32-MFMA sequence forward
 *        (lstm_seq_fwd_z) and the in-kernel input gather of the single-LSTM forward (lstm_seq_fwd_x), which no shipped path
 *        launched */
#define CLV_ABI_VERSION 600
int clv_version(void);
/* number of visible HIP devices whose arch is gfx950 (0 => the product must fail loudly) */
int clv_device_count(void);
const char* clv_error_string(int code);

/* ------------------------------------------------------------------ GEMM --
 * C[M,N] = act(alpha * op(A)[M,K] . op(B)[K,N] + bias[N] + beta * C)
 * op(A) = A (A is [M,K], lda) or A^T (A is [K,M], lda) when transa != 0; same for B.
 * fp32 in / fp32 accumulate on v_mfma_f32_16x16x4_f32 (bit-exact fma chains).
 * `aux` (ld = ldc) is only read for CLV_ACT_MASKPOS.  split_k > 1 needs
 * ws >= clv_gemm_workspace_bytes(M,N,split_k).
 * Replaces every keras.layers.Dense / TimeDistributed(Dense) / LSTM input
 * projection matmul and their gradients: cl_vae/model.py:141-143,160-167,
 * 184-186; cl_vrnn/model.py:174-175,196-209,225-234.
 * split_k <= 0 lets the library choose (clv_gemm_auto_split); the workspace must then hold
 * clv_gemm_workspace_bytes(M, N, clv_gemm_auto_split(M, N, K)). */
int clv_gemm_auto_split(int M, int N, int K);
size_t clv_gemm_workspace_bytes(int M, int N, int split_k);

/* Grouped weight-gradient GEMM: C_p[M_p,N] = (beta ? beta*C_p : 0) + op(A_p)^T . B for up to 4 problems
 * that share B [K,N] -- one pass over dz yields every kernel gradient of an LSTM
 * (x^T.dz, h_{t-1}^T.dz, z^T.dz) or dW and db of a Dense layer.  A_p is [K, M_p] row-major (lda).
 * a_shift/a_zero_period: row k of A_p is taken from row k - a_shift and is zero when
 * k % a_zero_period == 0 (h_{t-1}: shift 1, period T, zero initial state).  ones == 1: A_p is an
 * implicit row of ones, M_p must be 1 (column sums of B = bias gradient).  ones == 2: A_p has M_p - 1 real
 * columns and row M_p - 1 of C_p is that column sum -- a Dense layer's kernel and bias gradient as ONE problem
 * when the bias follows the kernel in the flat gradient buffer (A 16-byte aligned, lda % 4 == 0, no shift).
 * Replaces the weight-gradient half of K.gradients() for cl_vrnn/model.py:196-199,225-228. */
typedef struct clv_gemm_prob {
  const float* A; int32_t lda; int32_t M;
  float* C; int32_t ldc;
  int32_t a_shift; int32_t a_zero_period; int32_t ones;
} clv_gemm_prob;
int clv_gemm_grouped_auto_split(const clv_gemm_prob* host_probs, int nprob, int N, int K);
size_t clv_gemm_grouped_workspace_bytes(const clv_gemm_prob* host_probs, int nprob, int N, int split_k);
/* Two such grouped products with few output rows (<= 16 in total each) over a short K (<= 4096) and DIFFERENT B
 * operands in one launch: the label rows and the bias of both LSTM input-kernel gradients of cl_vrnn
 * (B = sum_t dz [batch,4H] of the encoder / of the decoder, K = batch).  ones in {0, 1}; beta = 0. */
int clv_gemm_grouped_tn_small2(const clv_gemm_prob* probs0, int nprob0, const float* B0,
                               const clv_gemm_prob* probs1, int nprob1, const float* B1,
                               int N, int K, int ldb, void* stream);

/* clv_gemm_f32 (the plain product documented at the top of this section) and clv_gemm_grouped_tn.  Deferred split-K
 * reduction: job == NULL: the product is finished by the call.  job != NULL and the product was split: the partial slabs stay
 * in `ws` and *job describes the pending reduction (+ epilogue) instead of launching it; `ws` must then stay untouched until
 * clv_splitk_reduce_multi has run.  A backward pass queues all of its weight-gradient products this way and finishes them
 * with ONE reduce launch (up to 16 pending jobs; jobs that needed no split are skipped).  Summation order per output is fixed,
 * so results are bit-identical either way. */
typedef struct { unsigned char opaque[160]; } clv_reduce_job;
int clv_gemm_f32(int transa, int transb, int M, int N, int K, float alpha,
                          const float* A, int lda, const float* B, int ldb,
                          float beta, float* C, int ldc,
                          const float* bias, int act, const float* aux,
                          int split_k, void* ws, size_t ws_bytes, clv_reduce_job* job, void* stream);
int clv_gemm_grouped_tn(const clv_gemm_prob* probs, int nprob, int N, int K,
                                 const float* B, int ldb, float beta,
                                 int split_k, void* ws, size_t ws_bytes, clv_reduce_job* job, void* stream);
/* clv_splitk_reduce_multi: the pending reductions in one launch, with up to five strided means computed by extra blocks
 * (means_out[i] = mean of the n[i] elements x[i][k * stride[i]]; x / n / stride: host arrays of n_terms entries, device pointers
 * in x): a training step takes its loss terms here instead of in a clv_loss_sums launch of its own -- and with up to two
 * few-row products riding in the launch as well: C[r, :N] = sum_k A[k, r] B[k, :N] for r < rows
 * (rows + (bias_row != NULL) <= 16), bias_row[:N] = column sums of B -- the label rows and the bias of an LSTM
 * input-kernel gradient over K = batch rows of sum_t dz; both products must have the same N. */
typedef struct clv_skinny_product {
  const float* A; int lda, rows;      /* A [K, lda]: the first `rows` columns are the products' rows */
  const float* B; int ldb, N, K;
  float* C; int ldc;
  float* bias_row;                    /* may be NULL */
} clv_skinny_product;
int clv_splitk_reduce_multi(const clv_reduce_job* jobs, int njobs, const float* const* x, const int* n,
                               const int* stride, int n_terms, float* means_out,
                               const clv_skinny_product* riders, int n_riders, void* stream);

/* Every kernel gradient of one LSTM in one pass over dz [K,N], N = 4H = 352, K = B*T (cl_vrnn/model.py:196-199,
 * 225-228; replaces the grouped f32-MFMA product for these shapes):
 *   dKx [nx,N] = X^T . dz            X [K,ldx]: the input frames (first nx columns)
 *   dU  [nh,N] = H'^T . dz           H'_k = H[k - h_shift] (h of the previous step), zero where k % h_zero_period == 0
 *   dKz [nz,N] = Z^T . dz            Z [K,ldz]: the latent columns of the decoder input; nz = 0: absent
 * computed on the bf16 matrix cores from exact piece products: every fp32 operand is split into three bf16 pieces
 * (x = p0 + p1 + p2 exactly), a product of two pieces is exact in fp32 and the partial products accumulate in fp32.  Of the
 * 3 x 3 piece pairs of an H or Z value and a dz value, the three below 2^-25 of the product (less than the rounding of one
 * fp32 multiply) are left out since round 4: 6 bf16 MFMAs instead of 8 f32 MFMAs at 1/16 of the rate.  x_exact_bf16 != 0
 * promises that every X value is exactly representable in bf16 (0/1 piano-roll frames, any uint8): those rows then need
 * one piece, and their products with all three dz pieces are exact.  The products leave as split-K slabs: ws >= clv_lstm_wgrad_workspace_bytes,
 * and like the GEMMs' deferred form (job) the final sums (C = beta*C + sum) are formed by the reduction, now (job == NULL) or by
 * clv_splitk_reduce_multi.  Limits (clv_lstm_wgrad_supported): N == 352; nx <= 96, nh <= 96, nz <= 32; nx, nh,
 * ldx, ldh, lddz multiples of 4, 16-byte aligned bases; more than 96 rows of H and Z together, or more than 8 rows
 * of Z, need x_exact_bf16 (the wide form of the kernel has no room for three pieces of X).
 * x_exact_bf16 == CLV_FRAMES_U8 (2): X IS the frames as bytes -- uint8 rows, ldx in bytes, 4-byte aligned -- i.e. the frame
 * store's own format: the training step of the large-batch path never widens its frames to float (round 6). */
#define CLV_FRAMES_F32       0   /* float rows, any values */
#define CLV_FRAMES_F32_EXACT 1   /* float rows whose values are exactly bf16 numbers */
#define CLV_FRAMES_U8        2   /* uint8 rows */
int clv_lstm_wgrad_supported(int N, int nx, int nh, int nz, int x_exact_bf16);
/* split_scale (1..8): that many times as many, proportionally shorter row ranges (and slabs).  1 = one workgroup per CU,
 * the fastest grid on an idle GPU; 2 is what the data-parallel step uses: the gradient all-reduce's kernel holds a few
 * CUs while these products run, and a grid of exactly one workgroup per CU would then need a whole second round. */
size_t clv_lstm_wgrad_workspace_bytes(int K, int N, int nx, int nh, int nz, int split_scale);
int clv_lstm_wgrad(int K, int N, const void* X, int ldx, int nx, int x_exact_bf16,
                      const float* H, int ldh, int nh, int h_shift, int h_zero_period,
                      const float* Z, int ldz, int nz, const float* dz, int lddz,
                      float* dKx, int ld_kx, float* dU, int ld_u, float* dKz, int ld_kz, float beta,
                      int split_scale, void* ws, size_t ws_bytes, clv_reduce_job* job, void* stream);
/* Two such products in ONE launch (the encoder's and the decoder's of a cl_vrnn step, cl_vrnn/model.py:196-199 and
 * 225-228, whose dz both exist once the backward pass is through): the arguments of clv_lstm_wgrad per problem, each
 * with its own slab workspace (>= clv_lstm_wgrad_pair_workspace_bytes: row ranges are twice as long as in the single
 * launch, so there are half as many slabs) and its own reduction job (NULL: reduced at once).  The two problems must have
 * the same K, N and x_exact_bf16 and take the same form of the kernel (clv_lstm_wgrad_pair_supported). */
typedef struct clv_wgrad_problem {
  int32_t K, N;
  const void* X; int32_t ldx, nx, x_exact_bf16;      /* x_exact_bf16 == CLV_FRAMES_U8: X holds bytes, ldx counts bytes */
  const float* H; int32_t ldh, nh, h_shift, h_zero_period;
  const float* Z; int32_t ldz, nz;
  const float* dz; int32_t lddz;
  float* dKx; int32_t ld_kx;
  float* dU; int32_t ld_u;
  float* dKz; int32_t ld_kz;
  float beta;
  void* ws; size_t ws_bytes;
} clv_wgrad_problem;
int clv_lstm_wgrad_pair_supported(const clv_wgrad_problem* p, const clv_wgrad_problem* q);
size_t clv_lstm_wgrad_pair_workspace_bytes(int K, int N, int nx, int nh, int nz, int split_scale);
int clv_lstm_wgrad_pair(const clv_wgrad_problem* p, const clv_wgrad_problem* q, int split_scale,
                        clv_reduce_job* job_p, clv_reduce_job* job_q, void* stream);


/* column sums: out[N] = (beta ? out : 0) + sum_m X[m, n]   (bias gradients) */
size_t clv_colsum_workspace_bytes(int M, int N);
int clv_colsum_f32(int M, int N, const float* X, int ldx, float beta, float* out,
                   void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ LSTM --
 * Persistent sequence kernels, one workgroup per batch row.  H == 88 (the reference's default --intermediate_dim,
 * cl_vrnn/train.py:90): recurrent kernel U [H,4H] resident in registers for all T steps (csrc/lstm.hip).  Any other
 * 1 <= H <= 1024 (clv_lstm_seq_fwd and clv_lstm_seq_bwd only): the same contract with U streamed from L2 every step
 * (csrc/lstm_any.hip) -- correct for whatever LSTM(intermediate_dim) cl_vrnn/model.py:196-199,225-228 builds, not fast.
 *   z_t = xproj[b,t,:] + rowbias[b,:] + h_{t-1} . U
 *   i,f,o = gate_act(z_i,z_f,z_o); g = tanh(z_c); c_t = f*c_{t-1} + i*g; h_t = o*tanh(c_t)
 * fwd writes hs[B,T,H], cs[B,T,H] and gates[B,T,4H] = (z_i, z_f, tanh(z_c), z_o)
 * (what BPTT needs); h0/c0 may be NULL (zero state) and hT/cT may be NULL.
 * bwd consumes dhs[B,T,H] (dL/dh_t from the layers above), overwrites
 * gates with dz[B,T,4H] (dL/dz_t) and writes dzsum[B,4H] = sum_t dz.
 * Replaces keras.layers.LSTM (K.rnn loop) at cl_vrnn/model.py:196-199,225-228
 * and the stateful step models at :122-125,149-152. */
int clv_lstm_seq_fwd(int B, int T, int H, int gate_act,
                     const float* xproj, const float* rowbias, const float* U,
                     const float* h0, const float* c0,
                     float* hs, float* cs, float* gates, float* hT, float* cT,
                     void* stream);
/* clv_lstm_seq_bwd + dZ_t = dz_t . Kz^T (Kz [nz,4H]: the z rows of decoder_h/kernel; dZ: B*T rows of stride lddz;
 * nz <= 40) inside the same launch: two more waves whose "units" are latents.  The decoder's backward pass without the
 * [B*T,4H] x [4H,nz] product as a launch of its own and without reading dz a second time
 * (cl_vrnn/model.py:218-228 under K.gradients). */
int clv_lstm_seq_bwd_z(int B, int T, int H, int gate_act,
                       const float* U, const float* dhs, const float* cs, const float* c0,
                       float* gates_inout_dz, float* dzsum, const float* Kz, int nz, float* dZ, int lddz, void* stream);

int clv_lstm_seq_bwd(int B, int T, int H, int gate_act,
                     const float* U, const float* dhs, const float* cs, const float* c0,
                     float* gates_inout_dz, float* dzsum, void* stream);

/* ----------------------------------------- LSTM, large batches: bf16-split MFMA --
 * The training pass of one LSTM layer (cl_vrnn/model.py:196-199 encoder_h, :225-228 decoder_h; Keras LSTM under
 * K.rnn / K.gradients) for batches of >= 4 rows per CU (BASELINE configuration 5: 1024 rows per GPU), csrc/lstm_mx.hip.
 * A workgroup owns four batch rows; the recurrent product runs on v_mfma_f32_16x16x32_bf16 with EXACT fp32 products
 * (an fp32 number = three bf16 pieces; the pieces of h sit in the MFMA's N dimension, so the nine piece products cost
 * three MFMAs per tile and k-step), accumulated in fp32.  The per-step inputs are multiplied INSIDE the kernel:
 *   z_t = X[b,t,:nx] . Kx + Z[b,t,:nz] . Kz + rowbias[b,:] + h_{t-1} . U
 * X: B*T rows of stride ldx (piano-roll frames; any float values are handled exactly, cost grows with the nonzeros of a
 * frame: Kx [nx,4H] stays in LDS and only the rows of the notes that are on are added), nx <= 96, nx == 0: no frames;
 * x_u8 != 0: X holds the frames as BYTES (uint8 rows, ldx in bytes) -- the frame store's own format, no float copy of a batch;
 * Z: B*T rows of stride ldz, nz <= 32 latent inputs times Kz [nz,4H] as one more MFMA k-step, nz == 0: none.
 * No [B*T,4H] projection buffer exists (clv_sparse_proj + clv_lstm_seq_fwd read and write one).  Zero initial state.
 * Outputs: hs [B*T,H] and the backward pass's coefficients (the quantities of clv_lstm_pair_fwd) as UNIT-MAJOR records,
 * one 16-byte and one 8-byte access per unit and step (private to this pair of entry points; 16-byte aligned buffers):
 *   coef [B*T,H,4] = (ki, kf, kg, ko) = (g i', c_{t-1} f', i g', tanh(c) o'),  aux [B*T,H,2] = (kcarry, kc) = (f, o (1 - tanh(c)^2)).
 * clv_lstm_mx_bwd: BPTT from dhs [B*T,H]: dc += dh kc; dz = (dc ki, dc kf, dc kg, dh ko); dc *= kcarry, with
 * dh = dhs_t + dz_{t+1} . U^T on the same matrix cores; coef is overwritten in place with dz [B*T,4H] (gate-major, the
 * layout of the LSTM's kernel columns: what clv_lstm_wgrad_bf16 reads), dzsum [B,4H] =
 * sum_t dz.  nz > 0: dZ_t = dz_t . Kz^T as well (dZ: B*T rows of stride lddz), by one or two more waves.
 * clv_lstm_mx_supported: H == 88, nx <= 96, nz <= 32 and a batch the engine hands to these kernels (>= 768 rows;
 * CLV_LSTM_MX=1 / 0 forces / forbids them for measurements and tests). */
int clv_lstm_mx_supported(int B, int H, int nx, int nz);
int clv_lstm_mx_fwd(int B, int T, int H, int gate_act,
                    const void* X, int x_u8, int ldx, int nx, const float* Kx,
                    const float* Z, int ldz, int nz, const float* Kz,
                    const float* rowbias, const float* U,
                    float* hs, float* coef, float* aux, void* stream);
int clv_lstm_mx_bwd(int B, int T, int H, const float* U, const float* dhs, const float* aux,
                    float* coef_inout_dz, float* dzsum, const float* Kz, int nz, float* dZ, int lddz, void* stream);

/* ------------------------------------------ cl_vrnn: both LSTMs in one launch --
 * cl_vrnn/model.py:193-228 as ONE persistent kernel: encoder LSTM, the fused latent head
 * [Z_mean | Z_log_var] (Wz [H,2L], bz [2L]), z = mean + exp(log_var/2)*eps, its KL term, and the
 * decoder LSTM, whose input projection gets z_t . Kz (Kz [L,4H] = the z rows of decoder_h/kernel) added
 * inside the kernel.  A workgroup owns one batch row; the encoder chain and the decoder chain (two steps
 * behind) run in different waves of the same CU, so the pair costs about one sequence kernel.
 *   gates_enc : in  x_t.K_x [B,T,4H]          out the backward pass's gate coefficients (ki, kf, kg, ko) =
 *                                             (g i', c_{t-1} f', i g', tanh(c) o'): dz_{i,f,g} = dc k, dz_o = dh ko
 *   gates_dec : in  x_{t-1}.K_x (dec_has_xproj != 0; else ignored)   out likewise
 *   aux_enc / aux_dec [B*T,2,H]: out (kcarry, kc) = (f, o (1 - tanh(c)^2)): dc += dh kc, dc_{t-1} = dc kcarry.
 *     (These six numbers per unit and step are all the backward pass needs of the forward pass; the cell states
 *     themselves are not stored.  The format is private to the clv_lstm_pair_fwd / _bwd pair.)
 *   rowbias_* : [B,4H] per-row bias (W.K_w + b)
 *   zargs [B*T,2L], Z: B*T rows of stride ldz, klterm [B*T,L] = L * KL_l (the mean over ALL entries is
 *   the per-frame KL, which is what clv_loss_sums computes).
 * notes_enc != NULL: the input projections are formed INSIDE the kernel from note lists (clv_gather_rows_multi, notes_out:
 *   notes_enc [B*T, CLV_NOTE_ROW] of the frames x_t, notes_dec of x_{t-1}) and the kernels' frame rows Kx_enc / Kx_dec
 *   [88,4H] (rows 0..87 of encoder_h/kernel, decoder_h/kernel); gates_* are then outputs only and no projection launch
 *   (clv_sparse_proj / GEMM) is needed.  Binary frames only (a list says which notes are on, not how loud).
 * Initial states are zero (training windows: cl_vrnn/model.py builds stateless LSTMs for training).
 * clv_lstm_pair_supported: H == 88 and 1 <= L <= 8 (the forward kernel alone carries L <= 16); otherwise
 * use the separate kernels.
 *
 * clv_lstm_pair_bwd is the matching backward pass: decoder BPTT, dZ_t = dz_dec_t . Kz^T, the latent
 * head's backward (dzargs = [dZ + kl*mean | dZ*eps*sd/2 - kl*(1 - sd^2)/2], kl = kl_scale, written to
 * dzargs [B*T,2L] for the head's weight gradient), dh_enc_t = dzargs_t . Wz^T and encoder BPTT, one
 * launch; gates_* are overwritten in place with dz and dzsum_* [B,4H] = sum_t dz like clv_lstm_seq_bwd.
 * dhs_dec [B,T,H] is the decoder's upstream gradient (from the output head).
 * hs_enc != NULL: the latent head's weight gradient rides along -- dWz [H,2L] = hs_enc^T . dzargs and dbz [2L] = column sums
 * of dzargs (cl_vrnn/model.py:201-210, the two TimeDistributed Dense heads as one [H,2L] kernel), accumulated per batch
 * row in the chains' spare lanes into ws (>= clv_lstm_pair_bwd_workspace_bytes) and summed over the rows by the pending
 * reduction `job` (NULL: summed at once), like a split-K product's slabs; no GEMM launch, no second read of hs_enc. */
int clv_lstm_pair_supported(int H, int L);
/* Both passes read the recurrent kernels U_enc / U_dec [H,4H], Kz and (forward) Wz from `pack`: every workgroup needs
 * each weight exactly once, one value per lane, and clv_lstm_pair_pack lays them out in that order (one wave load = 1 KB
 * contiguous instead of four 64-byte pieces of four rows).  Run it after every weight update, before the forward pass;
 * pack holds clv_lstm_pair_pack_floats() floats, 16-byte aligned. */
size_t clv_lstm_pair_pack_floats(void);
int clv_lstm_pair_pack(int H, int L, const float* U_enc, const float* U_dec, const float* Kz, const float* Wz,
                       float* pack, void* stream);
int clv_lstm_pair_fwd(int B, int T, int H, int L, int gate_act,
                      float* gates_enc, const float* rowbias_enc,
                      float* gates_dec, int dec_has_xproj, const float* rowbias_dec,
                      const float* pack, const float* bz, float* eps,
                      float* hs_enc, float* aux_enc, float* hs_dec, float* aux_dec,
                      float* zargs, float* Z, int ldz, float* klterm,
                      const unsigned char* notes_enc, const float* Kx_enc,
                      const unsigned char* notes_dec, const float* Kx_dec,
                      const clv_noise_draw* noise, void* stream);
size_t clv_lstm_pair_bwd_workspace_bytes(int B, int H, int L);
/* ... with the label path's backward (clv_vrnn_label_bwd: same arithmetic, same outputs) as the epilogue of every
 * workgroup: row b's sum_t dz of both LSTMs is what the label backward of row b reads, so it needs no launch of its own
 * (cl_vrnn/model.py:174-191, 244-252 under K.gradients).  label == NULL: clv_lstm_pair_bwd.  dKa / dba / ws / job as in
 * clv_vrnn_label_bwd (dKa == NULL: no layer gradient). */
typedef struct clv_label_bwd_rider {
  int D, C;
  const float *Kenc_w, *Kdec_w;         /* [C,352] the kernel rows that multiply W */
  const float *wargs, *eps, *onehot, *W, *hW, *Ka;
  float prior_logvar, class_weight, w_kl_weight, inv_b;
  float *dwargs, *dhW;
  float *dKa, *dba;
  void* ws; size_t ws_bytes;
  clv_reduce_job* job;
} clv_label_bwd_rider;
int clv_lstm_pair_bwd(int B, int T, int H, int L, int gate_act, float kl_scale,
                         const float* pack, const float* Wz,
                         const float* dhs_dec, const float* aux_dec, const float* aux_enc,
                         float* gates_dec_inout_dz, float* gates_enc_inout_dz,
                         float* dzsum_dec, float* dzsum_enc,
                         const float* zargs, const float* eps, float* dzargs,
                         const float* hs_enc, float* dWz, float* dbz, void* ws, size_t ws_bytes, clv_reduce_job* job,
                         const clv_label_bwd_rider* label, void* stream);

/* ------------------------------------------------ cl_vrnn generation, persistent --
 * cl_vrnn/model.py:9-60 (generate_sample's frame loop) for N independent sequences, one workgroup per
 * sequence for its whole length, one launch: per frame encoder LSTM step on [x_{t-1}, w], latent head,
 * z = mean + exp(log_var/2)*eps (z_prior != 0: z = eps), decoder LSTM step on [x_{t-1}, z, w],
 * x_hat = sigmoid(head), x_t = [u <= x_hat].  Frames t < S are teacher-forced from x_seed [N,S,D];
 * the nsteps sampled frames go to Xs [N,nsteps,D]; xhat [N,S+nsteps,D] (optional) receives every frame's
 * probabilities.  eps = the clv_philox_normal value for (seed, step t, stream 0, index n*L+l), u = the
 * clv_philox_uniform value for (seed, step t, stream 1, index n*D+j).  Kernel pieces are the row blocks of the
 * Keras tensors: encoder_h/kernel = [Kx_enc (D rows) ; Kw_enc (C rows)], decoder_h/kernel = [Kx_dec (D rows,
 * NULL without use_x_prev) ; Kz (L rows) ; Kw_dec (C rows)], Wz = [Z_mean | Z_log_var] kernel [H,2L].
 * clv_vrnn_generate_supported: D == H == 88, L <= 16, C <= 32. */
int clv_vrnn_generate_supported(int D, int H, int L, int C);
int clv_vrnn_generate(int N, int S, int nsteps, int D, int H, int L, int C, int gate_act, int z_prior,
                      uint64_t seed, const float* x_seed, const float* w,
                      const float* Kx_enc, const float* Kw_enc, const float* b_enc, const float* U_enc,
                      const float* Wz, const float* bz,
                      const float* Kx_dec, const float* Kz, const float* Kw_dec, const float* b_dec,
                      const float* U_dec, const float* Wo, const float* bo,
                      float* Xs, float* xhat, void* stream);

/* ------------------------------------------------- cl_vae generation, persistent --
 * cl_vae/model.py:9-42 (generate_sample's frame loop) for N independent sequences, one workgroup per sequence for its
 * whole length, one launch: per frame  h = relu([x_prev | w] . Kh + bh), [z_mean | z_log_var] = h . Kz + bz,
 * z = mean + exp(log_var/2)*eps (z_prior != 0: z = eps), h_d = relu([w | x_prev_t | z] . Kd + bd) (the history rows only
 * with use_x_prev; x_prev_t = the frame BEFORE x_prev, :38-40), x_hat = sigmoid(h_d . Ko + bo), x_t = [u <= x_hat].
 * x_seed [N,D] is both x_prev and x_prev_t of frame 0.  eps = the clv_philox_normal value for (seed, step t, stream 0,
 * index n*L+l), u = the clv_philox_uniform value for (seed, step t, stream 1, index n*D+j).  Kernels in Keras layout:
 * Kh = h/kernel [D+C,H] (frame rows, label rows), Kz = the fused head [H,2L], Kd = decoder_h/kernel [C+(D)+L,H]
 * (label rows, history rows, latent rows), Ko = x_decoded_mean/kernel [H,D].  Xs [N,nsteps,D]; xhat (optional) the
 * probabilities.  clv_vae_generate_supported: D == H == 88, L <= 32, C <= 32 (models with hidden layers). */
int clv_vae_generate_supported(int D, int H, int L, int C);
int clv_vae_generate(int N, int nsteps, int D, int H, int L, int C, int use_x_prev, int z_prior, uint64_t seed,
                     const float* x_seed, const float* w, const float* Kh, const float* bh, const float* Kz,
                     const float* bz, const float* Kd, const float* bd, const float* Ko, const float* bo,
                     float* Xs, float* xhat, void* stream);

/* ------------------------------------------------------------ pointwise --
 * logistic-normal label sample + its two losses, one thread per row:
 *   w = softmax([mean + exp(lv/2)*eps, 0]); kl_w, w_rec = (C-1)*CCE(onehot, w+1e-10), hit
 * wargs is [B, 2(C-1)] = [mean | log_var] (cl_vrnn "Wargs") or two separate
 * pointers with ld (cl_vae heads).  rowloss[B,3] = (kl_w, w_rec, hit).
 * cl_vae/model.py:146-157,198-206; cl_vrnn/model.py:183-191,244-252. */
int clv_label_fwd(int B, int C, const float* mean, const float* logvar, int ld_in,
                  const float* eps, const float* onehot, float prior_logvar,
                  float* w, float* rowloss, void* stream);
/* backward of the above: dw[B,C] is dL/dw from the layers below (RepeatVector sums
 * already applied); adds class_weight*inv_b*d(w_rec) and w_kl_weight*inv_b*d(kl_w);
 * writes dmean/dlogvar (ld_out). */
int clv_label_bwd(int B, int C, const float* mean, const float* logvar, int ld_in,
                  const float* eps, const float* onehot, const float* w, const float* dw,
                  float prior_logvar, float class_weight, float w_kl_weight, float inv_b,
                  float* dmean, float* dlogvar, int ld_out, void* stream);

/* One whole cl_vae step in one launch (+ one slab-sum launch for the weight gradients): forward of
 * cl_vae/model.py:141-188 with injected eps, the four losses (:190-206) as per-row arrays
 * (rownll[B], rowkl[B], rowloss[B,3] = kl_w, w_rec, hit), and -- when need_grads -- the gradient of
 * (sum_rows vae + kl_weight*kl_z + w_kl_weight*kl_w + class_weight*w_rec) / B into `grads` (flat layout).
 * `params` is the flat parameter buffer with the two head pairs stored fused ([in, 2n] kernels:
 * w_mean|w_log_var and z_mean|z_log_var); host_offsets12 = element offsets of
 * {h_w, wargs, h, zargs, decoder_h, x_decoded_mean} x {kernel, bias}.  Activations live in LDS, each layer's
 * weights are requested from L2 one layer ahead into MFMA operands.  Limits: D, H, Hc <= 96; C, L <= 16 (clv_vae_fused_supported).
 * ws >= clv_vae_fused_workspace_bytes(B, D, H, Hc, C, L, use_x_prev).  logits may be NULL.  target [B,D] is what the decoder output is
 * scored against: NULL or x for the auto-encoder, the next frame under --predict_next (cl_vae/train.py:15,66). */
int clv_vae_fused_supported(int D, int H, int Hc, int C, int L);
size_t clv_vae_fused_workspace_bytes(int B, int D, int H, int Hc, int C, int L, int use_x_prev);

/* opts (may be NULL): the three small launches around the step folded in (a cl_vae training step is launch-bound):
 *   draw != 0        the kernel draws eps_w / eps_z itself -- the values clv_philox_normal2(eps_w, B*(C-1), noise_seed,
 *                    step, step_dev, stream_w, first_w, eps_z, B*L, ..., stream_z, first_z) would have written -- and stores
 *                    them into eps_w / eps_z (which are outputs then);
 *   loss_means       [5] batch means of rownll, rowkl and the three rowloss columns, by the slab-sum launch
 *                    (replaces a clv_loss_sums call); NULL: not computed;
 *   bump_iterations  device step counter incremented once by the slab-sum launch (need_grads only), so that the
 *                    optimizer call that follows can run with step_t = CLV_STEP_ADVANCED; NULL: left alone.
 *   bf16 != 0        the Dense products and the weight-gradient products round their operands to bf16 and accumulate in
 *                    fp32 on the bf16 matrix cores (BASELINE configuration 2: "bf16 ... encoder/decoder MFMA kernels
 *                    only"); sampling, losses and the optimizer stay fp32.  Measured against the fp64 oracle in
 *                    tests/test_gpu_models.py::test_cl_vae_bf16_step_tolerance.
 * opts == NULL: eps is read, no loss means, no counter bump, fp32 products. */
typedef struct clv_vae_step_opts {
  int draw;
  uint32_t stream_w, stream_z, step;
  uint64_t noise_seed, first_w, first_z;
  const int32_t* step_dev;
  float* loss_means;
  int32_t* bump_iterations;
  int bf16;
} clv_vae_step_opts;

/* The whole cl_vrnn label path of a batch row in one launch (one workgroup per row):
 * fwd: Wargs = hW.K_a + b_a; W = logistic-normal sample; (kl_w, w_rec, hit) -> rowloss[B,3];
 *      rb_enc = W.K_enc_w + b_enc, rb_dec = W.K_dec_w + b_dec  -- the per-row LSTM biases that carry
 *      the RepeatVector(W) columns of the LSTM inputs (K_*_w = the C kernel rows that multiply W).
 * bwd: dW = dzsum_dec.K_dec_w^T + dzsum_enc.K_enc_w^T (RepeatVector sums), label backward with the
 *      weighted w_rec / kl_w terms, dwargs[B,2(C-1)], dhW = (dwargs.K_a^T)*(hW > 0).
 * D <= 128, C <= 32.  cl_vrnn/model.py:175-193,218-222,244-252. */
int clv_vrnn_label_fwd(int B, int D, int C, int G4, const float* hW, const float* Ka, const float* ba,
                       const float* eps, const float* onehot, float prior_logvar,
                       const float* Kenc_w, const float* benc, const float* Kdec_w, const float* bdec,
                       float* wargs, float* W, float* rowloss, float* rb_enc, float* rb_dec, void* stream);
/* clv_vrnn_label_fwd with the hW Dense layer in front of it, same workgroup: hW[b,:] = relu(X[b,:nx] . Kh + bh) over
 * the nonzero inputs of the row (cl_vrnn/model.py:174-176; X = the flattened window, ~4 % notes), written to hW_out
 * [B,D] for the backward pass, then the label path as above.  D even. */
int clv_vrnn_label_fwd_x(int B, int D, int C, int G4, const float* X, int ldx, int nx, const float* Kh,
                         const clv_label_stage* stage,      /* NULL, or the mini-batch assembly inside the launch: clv_label_stage */
                         const float* bh, float* hW_out, const float* Ka, const float* ba,
                         float* eps, const float* onehot, float prior_logvar,
                         const float* Kenc_w, const float* benc, const float* Kdec_w, const float* bdec,
                         float* wargs, float* W, float* rowloss, float* rb_enc, float* rb_dec,
                         const clv_noise_draw* noise, const clv_pair_pack_src* pack,
                         const clv_frame_proj* proj,        /* NULL, or the LSTMs' frame projections in the same launch: clv_frame_proj */
                         void* stream);
int clv_vrnn_label_fwd_x_proj_supported(int B, int D, int nx, int T, int N);
/* The hW layer's product as a DENSE one on the bf16 matrix cores, for inputs that are exactly representable in bf16 (the
 * caller's promise: 0/1 piano-roll frames, any uint8 value): part[c][b][:N] = sum over the inputs i of chunk c of
 * X[b,i] K[i,:], c < clv_dense_window_fwd_bf16_splits(Bn, nx) (split-K: the output is only [Bn,N]); X is one bf16 piece, K
 * three, the piece products are exact and accumulate in fp32.  clv_vrnn_label_fwd_parts is clv_vrnn_label_fwd_x with that
 * product handed in: the workgroup of a batch row sums its row of the chunks, adds bh, applies the relu and goes on with the
 * label path.  From a few hundred batch rows on this replaces the note-walking gather (every kernel row is then fetched
 * once per workgroup instead of once per batch row that has the note).  N <= 96; N, ldx, ldk multiples of 4, nx of 8;
 * 16-byte aligned X and K; arrays below 2 GiB.  x_u8 != 0: X holds the frames as BYTES (uint8, ldx in bytes, 4-byte
 * aligned) -- the frame store's own format, widened inside the kernel (clv_frames_u8 below).  cl_vrnn/model.py:174-176. */
int clv_dense_window_fwd_bf16_supported(int Bn, int nx, int N, int ldx, int ldk);
int clv_dense_window_fwd_bf16_splits(int Bn, int nx);
size_t clv_dense_window_fwd_bf16_workspace_bytes(int Bn, int nx, int N);
int clv_dense_window_fwd_bf16(int Bn, int nx, int N, const void* X, int x_u8, int ldx, const float* K, int ldk, float* part,
                              size_t part_bytes, void* stream);
int clv_vrnn_label_fwd_parts(int B, int D, int C, int G4, const float* part, int splits,
                             const float* bh, float* hW_out, const float* Ka, const float* ba,
                             float* eps, const float* onehot, float prior_logvar,
                             const float* Kenc_w, const float* benc, const float* Kdec_w, const float* bdec,
                             float* wargs, float* W, float* rowloss, float* rb_enc, float* rb_dec,
                             const clv_noise_draw* noise, const clv_pair_pack_src* pack, void* stream);
/* stage != NULL: the mini-batch assembly inside the step (cl_vae/train.py:66-71): every workgroup resolves its 16 batch rows
 * (clv_label_stage with rows of D bytes: cur = the frames x, hist = the previous frames x_prev, w_src = the labels) and reads
 * their bytes itself -- x / xp / target / onehot are not read (no separate target: the auto-encoder); stage->X / Xh / w_out (each
 * may be NULL) receive the rows as a gather launch would have left them.  The cl_vae training step is then three launches:
 * this one, the slab sum, Adam. */
int clv_vae_fused_step(int B, int D, int H, int Hc, int C, int L, int use_x_prev,
                       const float* x, const float* xp, const float* target, const float* onehot,
                       const clv_label_stage* stage, float* eps_w, float* eps_z,
                       const float* params, const int64_t* host_offsets12, long n_params,
                       float prior_logvar, float class_weight, float kl_weight, float w_kl_weight,
                       int need_grads, float* grads, void* ws, size_t ws_bytes,
                       float* logits, float* w_out, float* wargs_out, float* zargs_out,
                       float* rownll, float* rowkl, float* rowloss, const clv_vae_step_opts* opts, void* stream);
/* dKa != NULL -- the Wargs layer's own gradient rides along: dKa [D,2(C-1)] = hW^T . dwargs, dba = column sums of
 * dwargs, as per-row outer products in ws (>= clv_vrnn_label_bwd_workspace_bytes) summed by the pending reduction `job`
 * (NULL: at once), like a split-K product's slabs: no GEMM launch over K = batch. */
size_t clv_vrnn_label_bwd_workspace_bytes(int B, int D, int C);
int clv_vrnn_label_bwd(int B, int D, int C, int G4, const float* dzsum_enc, const float* dzsum_dec,
                          const float* Kenc_w, const float* Kdec_w, const float* wargs, const float* eps,
                          const float* onehot, const float* W, const float* hW, const float* Ka,
                          float prior_logvar, float class_weight, float w_kl_weight, float inv_b,
                          float* dwargs, float* dhW, float* dKa, float* dba, void* ws, size_t ws_bytes,
                          clv_reduce_job* job, void* stream);

/* gaussian reparameterisation rows: zargs[R, 2L] = [mean | log_var];
 * z[R, ldz] (written at column offset 0..L-1) = mean + exp(lv/2)*eps; rowkl[R] = KL(N(mean,exp(lv))||N(0,1)).
 * cl_vae/model.py:170-174,193-196; cl_vrnn/model.py:212-216,236-239. */
int clv_gauss_fwd(int R, int L, const float* zargs, const float* eps, float* z, int ldz,
                  float* rowkl, void* stream);
/* dzargs[R,2L] = [dz + kl_scale*mean | dz*eps*0.5*exp(lv/2) - 0.5*kl_scale*(1-exp(lv))] */
int clv_gauss_bwd(int R, int L, const float* zargs, const float* eps, const float* dz, int lddz,
                  float kl_scale, float* dzargs, void* stream);

/* Bernoulli NLL on logits with Keras' epsilon-clip semantics (A.3), one wave per row:
 * rownll[R] = sum_j softplus(l) - l*y, l = clip(a, -16.118095, +15.942385): the logits at which the FLOAT32 Keras
 * path clips p = sigmoid(a) to [float32(1e-7), float32(1 - 1e-7) = 1 - 2^-23] (the upper point is log(2^23 - 1); a
 * library built with -DCLV_BCE_SYMMETRIC_CLIP uses the exact-arithmetic +-log((1-1e-7)/1e-7) = +-16.118095 instead);
 * dlogits[R,D] = scale * (sigmoid(l) - y) * [a inside the clip]   (dlogits may alias logits,
 * or be NULL for loss only).  cl_vae/model.py:190-191; cl_vrnn/model.py:241-242. */
int clv_bernoulli_nll(int R, int D, const float* logits, const float* y, int ldy, float scale,
                      float* rownll, float* dlogits, void* stream);
/* The output head with that loss fused into the GEMM epilogue (cl_vrnn/model.py:229-242 in one launch):
 *   a = A[M,K].B[K,N] + bias;  logits[M,N] = a (optional, row stride ldc);  rownll[M] and dlogits[M,N] (row stride
 *   ldc) as clv_bernoulli_nll computes them from a and the targets Y (row stride ldy).  N <= 176. */
int clv_gemm_bce_f32(int M, int N, int K, const float* A, int lda, const float* B, int ldb, const float* bias,
                     const float* Y, int ldy, float scale, float* logits, float* dlogits, int ldc,
                     float* rownll, void* stream);

/* The output head of cl_vrnn in one training pass (H == D == 88): logits = hs.Wo + bo, the Bernoulli NLL and its
 * gradient exactly as clv_gemm_bce_f32, then dhs = dl.Wo^T (upstream gradient of the decoder BPTT) and
 * dWo = hs^T.dl, dbo = sum_r dl, with dl kept on chip (cl_vrnn/model.py:229-234, 241-242 and their K.gradients).
 * hs [R,88] and Wo 16-byte aligned; logits / dlogits may be NULL (not stored).  dWo/dbo leave as one partial slab per
 * workgroup in `ws`: with job == NULL they are reduced at once, otherwise *job receives the pending reduction for
 * clv_splitk_reduce_multi.  Replaces clv_gemm_bce_f32 + clv_gemm_f32 (NT) + clv_gemm_grouped_tn for this layer.
 * y_u8 != 0: the targets are the frames as BYTES (uint8, ldy in bytes and a multiple of 4, Y 4-byte aligned). */
int clv_out_head_train_supported(int H, int D);
size_t clv_out_head_train_workspace_bytes(int R);
int clv_out_head_train(int R, int H, int D, const float* hs, const float* Wo, const float* bo,
                       const void* Y, int y_u8, int ldy, float scale, float* logits, float* rownll, float* dlogits,
                       float* dhs, float* dWo, float* dbo, void* ws, size_t ws_bytes, clv_reduce_job* job,
                       void* stream);

/* The latent head of cl_vrnn outside the pair kernels (H == 88, latent_dim <= 32; cl_vrnn/model.py:200-216, 243 and their
 * K.gradients), one launch per pass (csrc/latent_head.hip):
 *   forward:  zargs = hs.Wz + bz [R,2L] = (mean | log_var), Z[r, :L] = mean + exp(log_var/2) * eps (row stride ldz),
 *             rowkl[r] = -0.5 sum_l (1 + log_var - mean^2 - exp(log_var)) (may be NULL); noise != NULL: eps [R,L] is drawn in
 *             the kernel (clv_noise_draw: element r * L + l) and written for the backward pass, no Philox launch
 *             -- replaces clv_gemm_f32 + clv_gauss_fwd;
 *   backward: dzargs = (dZ + kl_scale*mean | dZ*eps*sd/2 - kl_scale*(1 - sd^2)/2) (stored only when dzargs != NULL),
 *             dhs = dzargs.Wz^T [R,88], dWz = hs^T.dzargs, dbz = sum_r dzargs -- replaces clv_gauss_bwd + clv_gemm_f32 (NT)
 *             + clv_gemm_grouped_tn.  dWz/dbz leave as one partial slab per workgroup in `ws`: with job == NULL they are
 *             reduced at once, otherwise *job receives the pending reduction for clv_splitk_reduce_multi.
 * hs 16-byte aligned. */
int clv_latent_head_supported(int H, int L);
size_t clv_latent_head_bwd_workspace_bytes(int R, int L);
int clv_latent_head_fwd(int R, int H, int L, const float* hs, const float* Wz, const float* bz, float* eps,
                        float* zargs, float* Z, int ldz, float* rowkl, const clv_noise_draw* noise, void* stream);
int clv_latent_head_bwd(int R, int H, int L, const float* hs, const float* Wz, const float* zargs, const float* eps,
                        const float* dZ, int lddz, float kl_scale, float* dzargs, float* dhs, float* dWz, float* dbz,
                        void* ws, size_t ws_bytes, clv_reduce_job* job, void* stream);

/* dpre[i] = dy[i] * act'(pre[i]) from the activation's output y: CLV_ACT_RELU -> [y > 0], CLV_ACT_SIGMOID -> y(1-y),
 * CLV_ACT_NONE -> copy.  Backward of a Dense layer used on its own (the torch module face, clvae_amd/nn.py); the
 * training engines fold this mask into the epilogue of the neighbouring GEMM instead (CLV_ACT_MASKPOS). */
int clv_act_grad(int64_t n, int act, const float* y, const float* dy, float* dpre, void* stream);
/* y[i] += alpha * x[i]  (epoch running sums of the loss scalars stay on the device) */
int clv_axpy(int64_t n, float alpha, const float* x, float* y, void* stream);

/* Input projection of sparse frames: out[r, 0:N] = sum_k X[r,k] * K[k,:] for r < R, X [R,ldx] (nx used
 * columns), K [nx,N] (16-byte aligned), out rows of stride ldo.  The LSTM input projections of cl_vrnn
 * (cl_vrnn/model.py:193-196, 218-226) multiply piano-roll frames that are ~4 % nonzero; this keeps K in LDS
 * and adds only the kernel rows of a frame's nonzero inputs.  Exact for any float input (cost grows with
 * the number of nonzeros); clv_sparse_proj_supported: nx <= 128, N <= 384, nx*N*4 <= 150 KB.  x_u8 != 0: X holds the frames
 * as bytes (uint8 rows, ldx in bytes). */
int clv_sparse_proj_supported(int nx, int N);
size_t clv_sparse_proj_lds_bytes(int nx, int N);
int clv_sparse_proj(int R, int nx, int N, const void* X, int x_u8, int ldx, const float* K, float* out, int ldo, void* stream);
/* two such projections over the same R frames (the two LSTMs of cl_vrnn) in one launch */
int clv_sparse_proj2(int R, int N, int ldo, int x_u8, int nx0, const void* X0, int ldx0, const float* K0, float* out0,
                     int nx1, const void* X1, int ldx1, const float* K1, float* out1, void* stream);

/* The same idea for a Dense layer over a whole flattened window (cl_vrnn's hW layer, model.py:174-176;
 * nx = seq_length*88 inputs, ~4 % nonzero): out[r,:N] = act(sum_j X[r,j] K[j,:] + bias), act in {none, relu};
 * and for its kernel gradient dK[j,:N] = sum_b X[b,j] G[b,:] (every output row written once: no split-K
 * slabs).  N even and <= 128 (clv_sparse_dense_supported). */
int clv_sparse_dense_supported(int N);
int clv_sparse_dense(int R, int nx, int N, const float* X, int ldx, const float* K, const float* bias, int act,
                     float* out, int ldo, void* stream);
/* gdot != NULL also returns gdot[c] = sum_b (Hact[b,c] - hbias[c]) G[b,c], Hact [Bn,ldh] = the layer's relu output
 * and G its (relu-masked) upstream gradient: = sum_j K[j,c] dK[j,c], the weight-norm optimizer's sum g.W per column
 * without a pass over K and dK (clv_adam_wn_step). */
int clv_sparse_outer(int Bn, int nx, int N, const float* X, int ldx, const float* G, int ldg, float* out, int ldo,
                        float* colsum, const float* Hact, int ldh, const float* hbias, float* gdot, void* stream);

/* The same kernel gradient, dK[j,:N] = sum_b X[b,j] G[b,:] (+ colsum, gdot as in clv_sparse_outer), DENSE on the bf16
 * matrix cores for inputs that are exactly representable in bf16 -- the caller's promise: 0/1 piano-roll frames, any uint8
 * value.  X is then one bf16 piece, G three (an fp32 number is exactly the sum of three bf16 numbers), the piece products are
 * exact and accumulate in fp32: the products of the fp32 path in another summation order.  The kernel streams X once
 * (csrc/outer_bf16.hip); it replaces the note-walking kernel from a few hundred batch rows on (cl_vrnn/model.py:174-176, the
 * hW layer's kernel gradient).  N <= 96, N, nx, ldx, ldg multiples of 4, 16-byte aligned X and G, arrays below 2 GiB.
 * x_u8 != 0: X holds bytes (uint8, ldx in bytes, 4-byte aligned). */
int clv_dense_outer_bf16_supported(int Bn, int nx, int N, int ldx, int ldg);
int clv_dense_outer_bf16(int Bn, int nx, int N, const void* X, int x_u8, int ldx, const float* G, int ldg, float* out, int ldo,
                         float* colsum, const float* Hact, int ldh, const float* hbias, float* gdot, void* stream);

/* out[r, :] = src[idx[r], :] for r < rows; idx is a device int64 array (mini-batch assembly from the
 * HBM-resident data set; replaces the host-side slicing of Model.fit, cl_vae/train.py:66-71).
 * A row is row_elems/chunk pieces of `chunk` floats (frames); piece j of row r is written at
 * out + (r*pieces + j)*out_ld.  chunk <= 0 means one contiguous piece (out_ld = row_elems). */
int clv_gather_rows(int64_t rows, int64_t row_elems, const float* src, const int64_t* idx, float* out,
                    int64_t chunk, int64_t out_ld, void* stream);
/* Up to 4 such gathers that share one row list in a single launch (current frames, history frames, labels and --
 * under --predict_next -- target frames of a mini-batch).  idx == NULL takes the consecutive rows row0 .. row0+rows-1 (staging a contiguous batch).
 * Per segment k: src[k] rows of row_elems[k] elements, written like clv_gather_rows with chunk[k] / out_ld[k].
 * Source row r of segment k starts at element t * src_stride[k] + src_offset[k], t = src_table[k] ? src_table[k][i] : i,
 * i = idx ? idx[r] : row0 + r (src_stride / src_offset / src_table may be NULL: stride = row_elems[k], offset 0, no table).
 * With stride = one frame the rows are overlapping windows of a frame store (songs are kept once, as
 * uint8, and the sliding windows of utils/pianoroll.py:49-71 are never materialised).
 * src_u8 (may be NULL): src_u8[k] != 0 marks a uint8 source (binary piano-roll frames kept as bytes in HBM,
 * a quarter of the footprint and of the gather's read traffic); the output is float -- unless
 * src_u8[k] == 2: then out[k] is a uint8 buffer too (out_ld[k] in bytes) and the rows are copied as bytes: the batch of
 * the large-batch training step, whose kernels read frames as bytes (x_u8 / CLV_FRAMES_U8).  Needs row_elems, chunk, out_ld,
 * stride and offset multiples of 4 and 4-byte aligned bases. */
/* notes_out != NULL: the same launch also writes NOTE LISTS for the segments whose notes_out[k] is not NULL
 * (uint8 sources of binary frames, chunk[k] <= 88 and a multiple of 4): frame p of output row r gets CLV_NOTE_ROW bytes
 * at notes_out[k] + (r * pieces + p) * CLV_NOTE_ROW -- the indices of its nonzero bytes (any order), then CLV_NOTE_NONE
 * up to the end of the row.  clv_lstm_pair_fwd gathers the LSTM input projections x_t . K_x from such lists. */
#define CLV_NOTE_ROW 96
#define CLV_NOTE_NONE 88
/* ... with a BATCH CURSOR read on the device: the launch takes batch j = (*step_dev - step0) mod period, i.e. the rows
 * i = (idx ? idx[base + r] : row0 + base + r), base = j * stride + offset, r < rows.  step_dev is the optimizer's
 * `iterations` counter (clv_adam_wn_step advances it at the end of a step), so the mini-batch assembly of
 * Model.fit (cl_vae/train.py:66-71: one contiguous slice of the shuffled index per step) becomes a node of the step's
 * hipGraph: a step is ONE graph launch, nothing is staged from the host.  cursor == NULL: the rows idx[r] / row0 + r. */
/* (clv_batch_cursor is declared near the top of this header, next to clv_noise_draw) */
int clv_gather_rows_multi(int64_t rows, const int64_t* idx, int64_t row0, int nseg,
                                 const void* const* src, const int32_t* src_u8, float* const* out,
                                 const int64_t* row_elems, const int64_t* chunk, const int64_t* out_ld,
                                 const int64_t* src_stride, const int64_t* src_offset,
                                 const int64_t* const* src_table, unsigned char* const* notes_out,
                                 const clv_batch_cursor* cursor, void* stream);

/* the five loss scalars of a step in one launch: out[k] = scale[k] * sum_{i<n[k]} x[k][i*stride[k]], k < 5
 * (vae, kl_z, kl_w, w_rec, acc means; fixed summation order => deterministic). */
int clv_loss_sums(const float* x0, int n0, int s0, const float* x1, int n1, int s1, const float* x2, int n2, int s2,
                  const float* x3, int n3, int s3, const float* x4, int n4, int s4, float* out, void* stream);

/* deterministic sum of n floats with stride: out[0] = scale * sum_i x[i*stride] */
int clv_sum_strided(int n, const float* x, int stride, float scale, float* out, void* stream);

/* ---------------------------------------------------------------- Adam-WN --
 * Adam with weight normalisation over a flat parameter buffer
 * (utils/weightnorm.py:75-178).  `table` is a device array of n_tensors
 * clv_param_desc; matrices (rows>1 in the sense ndim>1) get the weight-norm
 * reparameterisation over all axes but the last, vectors plain Adam.
 * State: m, v (flat, same layout as params); mg, vg, s (flat per-column layout,
 * s initialised to 1).  step_t = iterations + 1 (1 on the first call).
 * weightnorm == 0 gives plain Keras Adam ('adam'). */
typedef struct clv_param_desc {
  int64_t offset;      /* element offset into params / grads / m / v          */
  int32_t rows;        /* product of all axes but the last (1 for a bias)     */
  int32_t cols;        /* last axis                                           */
  int64_t col_offset;  /* element offset into mg / vg / s (unused for biases) */
  int32_t is_matrix;   /* ndim > 1                                            */
  int32_t pad_;
} clv_param_desc;
/* plan: a device-resident work table derived from the tensor table.  Build it once on
 * the host (clv_adam_wn_plan_build into a host blob of clv_adam_wn_plan_bytes), copy the
 * blob to the device, pass the device pointer as plan_dev on every step. */
size_t clv_adam_wn_plan_bytes(const clv_param_desc* host_table, int n_tensors);
int clv_adam_wn_plan_build(const clv_param_desc* host_table, int n_tensors, void* host_blob);
size_t clv_adam_wn_workspace_bytes(const clv_param_desc* host_table, int n_tensors);
/* iterations_dev (device int32, may be NULL): Keras' `iterations` variable; when given,
 * t = *iterations_dev + 1 is read on the device and the counter is advanced by the call
 * (so a captured graph can be replayed); otherwise t = step_t.  iterations_dev with step_t == -1: the
 * counter is read but NOT advanced -- a step may be split over several calls on disjoint tensor subsets (each with its
 * own table and plan over the same flat buffers), of which only the last one advances the counter.
 * iterations_dev with step_t == CLV_STEP_ADVANCED: the counter already holds t (it was advanced by the launch that
 * produced the gradients, clv_vae_fused_step's bump_iterations) and is left alone. */
#define CLV_STEP_ADVANCED (-2)
/* `weightnorm` selects the update rule: */
#define CLV_OPT_ADAM     0   /* plain Keras Adam on every tensor                                                   */
#define CLV_OPT_ADAM_WN  1   /* utils/weightnorm.py:75-143: matrices per output column as g V/||V||, biases plain   */
#define CLV_OPT_RMSPROP  2   /* Keras RMSprop (the 'rmsprop' optimizer string, cl_vae/train.py:83): a = rho a +     */
                             /* (1 - rho) g^2, p -= lr g / (sqrt(a) + eps); rho = beta2, `v` holds a, `m` is unused  */
/* clv_adam_wn_step: `known` (may be NULL = clv_adam_wn_step) concerns the ONE tall matrix of the table (more than 144
 * rows: cl_vrnn's hW/kernel) under CLV_OPT_ADAM_WN:
 *   vnorm2 [n columns, laid out like s]: every call keeps ||V||^2 per column of the tall matrix there (the rescale leaves
 *     W = s' V', so the next step's sum V^2 is this step's ||V'||^2);
 *   use != 0: both column sums of the first pass are known -- sum V^2 from vnorm2 (valid after any earlier call that was
 *     given vnorm2 and no write to the parameters since) and sum g.V = gdot[c] / s[c] with gdot[c] = sum_b (X W)[b,c]
 *     dH[b,c] (clv_sparse_outer's gdot: the layer's pre-activation times its upstream gradient, summed over the batch) --
 *     and the update runs in TWO launches instead of five, with no pass over W and g for the statistics.
 * Not for gradients that were averaged across ranks after gdot was formed. */
typedef struct clv_adam_known_sums {
  int tensor;            /* index of the tall matrix in host_table */
  int use;
  const float* gdot;     /* [cols] */
  float* vnorm2;
} clv_adam_known_sums;
int clv_adam_wn_step(const clv_param_desc* host_table, int n_tensors, const void* plan_dev,
                        float* params, const float* grads, float* m, float* v,
                        float* mg, float* vg, float* s,
                        int32_t* iterations_dev, int step_t, float lr, float beta1, float beta2, float eps,
                        int weightnorm, const clv_adam_known_sums* known, void* ws, size_t ws_bytes, void* stream);

/* -------------------------------------------------------------------- RNG --
 * Counter-based Philox4x32-10, key = (seed_lo, seed_hi), counter =
 * ((index>>2)_lo, (index>>2)_hi, stream_id, step): element i of a draw is a pure
 * function of (seed, step, stream_id, first_index + i), so 1/2/4/8-GPU runs see the
 * same noise for the same global sample.  Normal = Box-Muller (words 0,1 -> cos,sin;
 * words 2,3 -> cos,sin).  The effective step is step + *step_dev when step_dev
 * (device int32, e.g. the Adam `iterations` counter) is not NULL.
 * Replaces K.random_normal (cl_vae/model.py:152,172; cl_vrnn/model.py:185,214)
 * and np.random.rand in sample_x (cl_vae/model.py:44-45; cl_vrnn/model.py:62-63). */
int clv_philox_normal(float* out, int64_t n, uint64_t seed, uint32_t step, const int32_t* step_dev,
                      uint32_t stream_id, uint64_t first_index, void* stream);
int clv_philox_uniform(float* out, int64_t n, uint64_t seed, uint32_t step, const int32_t* step_dev,
                       uint32_t stream_id, uint64_t first_index, void* stream);
/* Two normal draws (same seed/step, their own stream ids and first indices) in one launch: the label noise and
 * the latent noise of a training step.  Values are identical to two clv_philox_normal calls. */
int clv_philox_normal2(float* out0, int64_t n0, uint32_t stream_id0, uint64_t first_index0,
                       float* out1, int64_t n1, uint32_t stream_id1, uint64_t first_index1,
                       uint64_t seed, uint32_t step, const int32_t* step_dev, void* stream);
/* *counter += v on the device (advances the Philox step between replays of a captured sampling step) */
int clv_i32_add(int32_t* counter, int32_t v, void* stream);
/* x[i] = (u[i] <= p[i]) ? 1 : 0   -- sample_x */
int clv_bernoulli_sample(int64_t n, const float* p, const float* u, float* x, void* stream);
/* Input dropout of an LSTM in the training phase (get_model(dropout=p): cl_vrnn/model.py:164,198,227; Keras 2.0.0
 * recurrent.py, implementation 0: one mask per gate and sample, constant over the time steps, K.dropout(ones, p) =
 * floor(1 - p + u) / (1 - p)):   out[r, c] = beta * out[r, c] + X[r, c] * m(U[r / T, c]),   m(u) = (u >= rate) / (1 - rate),
 * r < R, c < n.  U holds UNIFORMS (clv_philox_uniform), one row per sample; T = rows of X per sample.  Forward: X = the
 * inputs of a gate's projection; backward: X = that gate's share of dL/d(input), beta = 1 sums the four gates. */
int clv_dropout_rows(int R, int T, int n, const float* X, int ldx, const float* U, int ldu, float rate, float beta,
                     float* out, int ldo, void* stream);

/* ----------------------------------------------------------------- graphs --
 * thin wrappers so a host without HIP bindings can capture a step once and
 * replay it (launch-bound inner loops: SURVEY.md 7.1 step 8). */
int clv_graph_begin_capture(void* stream);
int clv_graph_end_capture(void* stream, void** graph_exec_out);
int clv_graph_launch(void* graph_exec, void* stream);
int clv_graph_destroy(void* graph_exec);

/* --------------------------------------------------------------- profiler --
 * opt-in per-kernel HIP-event timing on the launch stream (bench.py roofline):
 * enable -> every kernel launched through this library is bracketed by events;
 * clv_prof_collect synchronises, and fills up to `cap` records. */
typedef struct clv_prof_record {
  char name[48];
  int32_t launches;
  float total_ms;
} clv_prof_record;
int clv_prof_enable(int on);
int clv_prof_collect(clv_prof_record* host_out, int cap);
/* records an empty bracket ("event_pair": two event records, no launch between them) when the profiler is on: the time the
 * events themselves add to a bracketed launch */
int clv_prof_empty_scope(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CLVAE_H */
