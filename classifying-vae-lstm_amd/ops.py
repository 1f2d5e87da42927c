"""Thin torch-tensor wrappers over the C ABI (include/clvae.h).

torch is used for device memory and streams only; every computation below is a
HIP kernel of libclvae_hip.so.  All tensors must be contiguous float32 CUDA
tensors (views with a row stride are passed as (tensor, ld)).
"""
import ctypes as C

import torch

from . import _lib
from ._lib import ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_MASKPOS, check  # noqa: F401


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Workspace:
    """A growable byte buffer on the device handed to ops that need scratch."""

    def __init__(self, device, nbytes=1 << 20):
        self.device = device
        self.buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)

    def ensure(self, nbytes):
        if self.buf.numel() < nbytes:
            self.buf = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=self.device)
        return self.buf


class ReduceQueue:
    """Pending split-K reductions of one backward pass (the `job` argument of the products + clv_splitk_reduce_multi).
    Every queued product keeps its partial slabs in its own scratch buffer until flush(); the buffers are
    kept per queue slot, so a step that queues the same products in the same order reuses the same memory
    (and a captured hipGraph sees fixed pointers)."""
    MAX_JOBS = 16

    def __init__(self, device):
        self.device = device
        self.jobs = (_lib.ReduceJob * self.MAX_JOBS)()
        self.slots = [None] * self.MAX_JOBS
        self.n = 0

    def scratch(self, nbytes):
        if self.n >= self.MAX_JOBS:
            raise _lib.ClvError("ReduceQueue: more than %d pending reductions" % self.MAX_JOBS)
        buf = self.slots[self.n]
        if buf is None or buf.numel() < nbytes:
            buf = self.slots[self.n] = torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=self.device)
        return buf

    def next_job(self):
        j = C.byref(self.jobs, self.n * C.sizeof(_lib.ReduceJob))
        self.n += 1
        return j

    def next_job_addr(self):
        """the next slot's address (for a job pointer that travels inside a struct)"""
        a = C.addressof(self.jobs) + self.n * C.sizeof(_lib.ReduceJob)
        self.n += 1
        return a

    def flush(self, means=None, out=None, skinny=None):
        """Run the pending reductions in one launch.  means: up to five (tensor, n, stride) terms whose means go to
        out[k] from extra blocks of the same launch (a step's loss terms: no launch of their own).  skinny: up to two
        dict(A, lda, rows, B, ldb, N, K, C, ldc, bias_row) few-row products A[:, :rows]^T B riding along as well."""
        if means or skinny or self.n:
            means = means or []
            k = len(means)
            xs = (C.c_void_p * max(k, 1))(*[_ptr(t) for t, _, _ in means])
            ns = (C.c_int * max(k, 1))(*[int(n) for _, n, _ in means])
            st = (C.c_int * max(k, 1))(*[int(s) for _, _, s in means])
            skinny = skinny or []
            riders = (_lib.SkinnyProduct * max(len(skinny), 1))()
            for i, p in enumerate(skinny):
                riders[i] = _lib.SkinnyProduct(_ptr(p['A']), p['lda'], p['rows'], _ptr(p['B']), p['ldb'], p['N'], p['K'],
                                               _ptr(p['C']), p['ldc'], _ptr(p.get('bias_row')))
            check(_lib.lib().clv_splitk_reduce_multi(self.jobs, self.n, xs if k else None, ns if k else None, st if k else None, k,
                                                     _ptr(out) if k else None, riders if skinny else None, len(skinny), _stream()),
                  "clv_splitk_reduce_multi")
        self.n = 0


def gemm(A, B, C_out, M, N, K, ta=False, tb=False, lda=None, ldb=None, ldc=None, alpha=1.0, beta=0.0,
         bias=None, act=ACT_NONE, aux=None, split_k=None, ws=None, defer=None):
    """C[M,N] = act(alpha*op(A).op(B) + bias + beta*C); A/B/C are tensors (possibly offset views).
    defer: a ReduceQueue -- a split product leaves its reduction (and epilogue) pending until defer.flush()."""
    L = _lib.lib()
    lda = lda if lda is not None else (M if ta else K)
    ldb = ldb if ldb is not None else (K if tb else N)
    ldc = ldc if ldc is not None else N
    if split_k is None:
        split_k = L.clv_gemm_auto_split(M, N, K)
    wsp, wsb, job = None, 0, None
    if split_k > 1:
        need = L.clv_gemm_workspace_bytes(M, N, split_k)
        buf = defer.scratch(need) if defer is not None else ws.ensure(need)
        wsp, wsb = _ptr(buf), buf.numel()
        job = defer.next_job() if defer is not None else None
    check(L.clv_gemm_f32(int(ta), int(tb), M, N, K, float(alpha), _ptr(A), lda, _ptr(B), ldb, float(beta),
                                  _ptr(C_out), ldc, _ptr(bias), act, _ptr(aux), split_k, wsp, wsb, job, _stream()),
          "clv_gemm_f32")


def gemm_grouped_tn(probs, N, K, B, ws, ldb=None, beta=0.0, split_k=None, defer=None, split_scale=1):
    """probs: list of dict(A=tensor|None, lda, M, C=tensor, ldc, shift=0, zero_period=0, ones=0|1|2);
    ones=1: implicit row of ones (M == 1); ones=2: row M-1 is an implicit row of ones appended to A's M-1 columns;
    C_p[M_p,N] = A_p^T . B for every problem in one launch (all share B [K,N])."""
    L = _lib.lib()
    arr = (_lib.GemmProb * len(probs))()
    for i, p in enumerate(probs):
        A = p.get('A')
        arr[i] = _lib.GemmProb(A.data_ptr() if A is not None else None, p.get('lda', p['M']), p['M'],
                               p['C'].data_ptr(), p.get('ldc', N), p.get('shift', 0), p.get('zero_period', 0),
                               int(p.get('ones', 0)))
    n = len(probs)
    if split_k is None:
        split_k = L.clv_gemm_grouped_auto_split(arr, n, N, K)
        if split_k > 1:
            split_k *= int(split_scale)      # finer grid: more, shorter workgroups (see VrnnEngine.fine_grid)
    wsp, wsb, job = None, 0, None
    if split_k > 1:
        need = L.clv_gemm_grouped_workspace_bytes(arr, n, N, split_k)
        buf = defer.scratch(need) if defer is not None else ws.ensure(need)
        wsp, wsb = _ptr(buf), buf.numel()
        job = defer.next_job() if defer is not None else None
    check(L.clv_gemm_grouped_tn(arr, n, N, K, _ptr(B), ldb if ldb is not None else N, float(beta), split_k,
                                         wsp, wsb, job, _stream()), "clv_gemm_grouped_tn")


def lstm_wgrad_supported(N, nx, nh, nz, x_exact_bf16):
    return bool(_lib.lib().clv_lstm_wgrad_supported(N, nx, nh, nz, int(bool(x_exact_bf16))))


def lstm_wgrad(K, N, X, ldx, nx, x_exact_bf16, H, ldh, nh, T, Z, ldz, nz, dz, dKx, dU, dKz, ws, defer=None, beta=0.0,
               split_scale=1):
    """All kernel gradients of one LSTM in one pass over dz (split-bf16 products, 6 of the 9 piece pairs: fp32-rounding
    accuracy; csrc/wgrad_bf16.hip):
    dKx = X^T dz, dU = H'^T dz (H' = hs shifted by one step, zero at window starts: period T), dKz = Z^T dz.
    split_scale: that many times as many, shorter row ranges (2 under the data-parallel step: see include/clvae.h)."""
    L = _lib.lib()
    need = L.clv_lstm_wgrad_workspace_bytes(K, N, nx, nh, nz, int(split_scale))
    buf = defer.scratch(need) if defer is not None else ws.ensure(need)
    job = defer.next_job() if defer is not None else None
    check(L.clv_lstm_wgrad(K, N, _ptr(X), ldx, nx, _frames_mode(X, x_exact_bf16), _ptr(H), ldh, nh, 1, T,
                              _ptr(Z), ldz, nz, _ptr(dz), N, _ptr(dKx), N, _ptr(dU), N, _ptr(dKz), N, float(beta),
                              int(split_scale), _ptr(buf), buf.numel(), job, _stream()), "clv_lstm_wgrad")


def _wgrad_problem(K, N, X, ldx, nx, x_exact_bf16, H, ldh, nh, T, Z, ldz, nz, dz, dKx, dU, dKz, beta=0.0):
    w = _lib.WgradProblem()
    w.K, w.N = K, N
    w.X, w.ldx, w.nx, w.x_exact_bf16 = _ptr(X), ldx, nx, _frames_mode(X, x_exact_bf16)
    w.H, w.ldh, w.nh, w.h_shift, w.h_zero_period = _ptr(H), ldh, nh, 1, T
    w.Z, w.ldz, w.nz = _ptr(Z), ldz, nz
    w.dz, w.lddz = _ptr(dz), N
    w.dKx, w.ld_kx, w.dU, w.ld_u, w.dKz, w.ld_kz = _ptr(dKx), N, _ptr(dU), N, _ptr(dKz), N
    w.beta = float(beta)
    return w


def lstm_wgrad_pair_supported(p, q):
    """p, q: the argument tuples of lstm_wgrad up to dKz (K, N, X, ldx, nx, x_exact_bf16, H, ldh, nh, T, Z, ldz, nz, dz, dKx,
    dU, dKz): can both products run as ONE launch (clv_lstm_wgrad_pair)?"""
    a, b = _wgrad_problem(*p), _wgrad_problem(*q)
    return bool(_lib.lib().clv_lstm_wgrad_pair_supported(C.byref(a), C.byref(b)))


def lstm_wgrad_pair(p, q, ws, defer=None, split_scale=1):
    """Both LSTMs' kernel gradients of a step in one launch (csrc/wgrad_bf16.hip: row ranges twice as long, half as many
    slabs); p, q as in lstm_wgrad_pair_supported.  ws: a pair of Workspaces when the reductions are not deferred."""
    L = _lib.lib()
    probs, jobs = [], []
    for i, t in enumerate((p, q)):
        w = _wgrad_problem(*t)
        need = L.clv_lstm_wgrad_pair_workspace_bytes(w.K, w.N, w.nx, w.nh, w.nz, int(split_scale))
        buf = defer.scratch(need) if defer is not None else ws[i].ensure(need)
        jobs.append(defer.next_job() if defer is not None else None)
        w.ws, w.ws_bytes = _ptr(buf), buf.numel()
        probs.append(w)
    check(L.clv_lstm_wgrad_pair(C.byref(probs[0]), C.byref(probs[1]), int(split_scale), jobs[0], jobs[1], _stream()),
          "clv_lstm_wgrad_pair")


def gemm_bce(A, B, bias, Y, scale, logits, dlogits, rownll, M, N, K, lda=None, ldb=None, ldy=None, ldc=None):
    """Output head + Bernoulli NLL in one launch: logits = A.B + bias, rownll, dlogits = scale*(sigmoid - Y)."""
    check(_lib.lib().clv_gemm_bce_f32(M, N, K, _ptr(A), lda if lda is not None else K, _ptr(B),
                                      ldb if ldb is not None else N, _ptr(bias), _ptr(Y), ldy if ldy is not None else N,
                                      float(scale), _ptr(logits), _ptr(dlogits), ldc if ldc is not None else N,
                                      _ptr(rownll), _stream()), "clv_gemm_bce_f32")


def out_head_train_supported(H, D):
    return bool(_lib.lib().clv_out_head_train_supported(H, D))


def out_head_train(R, H, D, hs, Wo, bo, Y, scale, rownll, dhs, dWo, dbo, ws, logits=None, dlogits=None, ldy=None,
                   defer=None):
    """Output head forward + Bernoulli NLL + dhs + dWo/dbo in one launch (see clv_out_head_train).  Y: the target frames,
    float32 or the bytes themselves (uint8).
    defer: a ReduceQueue that takes the pending reduction of the weight-gradient slabs."""
    L = _lib.lib()
    need = L.clv_out_head_train_workspace_bytes(R)
    buf = defer.scratch(need) if defer is not None else ws.ensure(need)
    job = defer.next_job() if defer is not None else None
    check(L.clv_out_head_train(R, H, D, _ptr(hs), _ptr(Wo), _ptr(bo), _ptr(Y), _is_u8(Y), ldy if ldy is not None else D,
                               float(scale), _ptr(logits), _ptr(rownll), _ptr(dlogits), _ptr(dhs), _ptr(dWo), _ptr(dbo),
                               _ptr(buf), buf.numel(), job, _stream()), "clv_out_head_train")


def latent_head_supported(H, L):
    return bool(_lib.lib().clv_latent_head_supported(H, L))


def latent_head_fwd(R, H, L, hs, Wz, bz, eps, zargs, Z, ldz, rowkl=None, noise=None):
    """zargs = hs.Wz + bz, the reparametrised sample into Z (row stride ldz) and the rows' KL in one launch.
    noise (noise_draw(...)): eps is drawn inside the kernel and written to `eps`, else read from it."""
    check(_lib.lib().clv_latent_head_fwd(R, H, L, _ptr(hs), _ptr(Wz), _ptr(bz), _ptr(eps), _ptr(zargs), _ptr(Z), ldz,
                                         _ptr(rowkl), _noise_ref(noise), _stream()), "clv_latent_head_fwd")


def latent_head_bwd(R, H, L, hs, Wz, zargs, eps, dZ, lddz, kl_scale, dhs, dWz, dbz, ws, dzargs=None, defer=None):
    """The latent head's backward in one launch: dzargs (kept on chip unless a buffer is given), dhs = dzargs.Wz^T and the
    layer's kernel / bias gradient.  defer: a ReduceQueue that takes the pending reduction of the weight-gradient slabs."""
    Lb = _lib.lib()
    need = Lb.clv_latent_head_bwd_workspace_bytes(R, L)
    buf = defer.scratch(need) if defer is not None else ws.ensure(need)
    job = defer.next_job() if defer is not None else None
    check(Lb.clv_latent_head_bwd(R, H, L, _ptr(hs), _ptr(Wz), _ptr(zargs), _ptr(eps), _ptr(dZ), lddz, float(kl_scale),
                                 _ptr(dzargs), _ptr(dhs), _ptr(dWz), _ptr(dbz), _ptr(buf), buf.numel(), job, _stream()),
          "clv_latent_head_bwd")


def _prob_array(probs, N):
    arr = (_lib.GemmProb * len(probs))()
    for i, p in enumerate(probs):
        A = p.get('A')
        arr[i] = _lib.GemmProb(A.data_ptr() if A is not None else None, p.get('lda', p['M']), p['M'],
                               p['C'].data_ptr(), p.get('ldc', N), p.get('shift', 0), p.get('zero_period', 0),
                               int(p.get('ones', 0)))
    return arr


def gemm_grouped_tn_small2(probs0, B0, probs1, B1, N, K, ldb=None):
    """Two few-row grouped products (different B operands) in one launch; see clv_gemm_grouped_tn_small2."""
    a0, a1 = _prob_array(probs0, N), _prob_array(probs1, N)
    check(_lib.lib().clv_gemm_grouped_tn_small2(a0, len(probs0), _ptr(B0), a1, len(probs1), _ptr(B1), N, K,
                                                ldb if ldb is not None else N, _stream()), "clv_gemm_grouped_tn_small2")


def loss_sums(terms, out):
    """terms: five (tensor, n, stride); out[k] = mean of term k."""
    a = []
    for t, n, st in terms:
        a += [_ptr(t), n, st]
    check(_lib.lib().clv_loss_sums(*a, _ptr(out), _stream()), "clv_loss_sums")


def colsum(X, M, N, out, ws, ldx=None, beta=0.0):
    L = _lib.lib()
    buf = ws.ensure(L.clv_colsum_workspace_bytes(M, N))
    check(L.clv_colsum_f32(M, N, _ptr(X), ldx if ldx is not None else N, float(beta), _ptr(out), _ptr(buf),
                           buf.numel(), _stream()), "clv_colsum_f32")


def lstm_seq_fwd(B, T, xproj, rowbias, U, hs, cs, gates, h0=None, c0=None, hT=None, cT=None, gate_act=0, H=88):
    check(_lib.lib().clv_lstm_seq_fwd(B, T, H, gate_act, _ptr(xproj), _ptr(rowbias), _ptr(U), _ptr(h0), _ptr(c0),
                                      _ptr(hs), _ptr(cs), _ptr(gates), _ptr(hT), _ptr(cT), _stream()),
          "clv_lstm_seq_fwd")


def lstm_seq_bwd_z(B, T, U, dhs, cs, gates_inout, dzsum, Kz, nz, dZ, lddz, c0=None, gate_act=0, H=88):
    """clv_lstm_seq_bwd + dZ = dz . Kz^T in the same launch."""
    check(_lib.lib().clv_lstm_seq_bwd_z(B, T, H, gate_act, _ptr(U), _ptr(dhs), _ptr(cs), _ptr(c0), _ptr(gates_inout),
                                        _ptr(dzsum), _ptr(Kz), nz, _ptr(dZ), lddz, _stream()), "clv_lstm_seq_bwd_z")


def lstm_seq_bwd(B, T, U, dhs, cs, gates_inout, dzsum, c0=None, gate_act=0, H=88):
    check(_lib.lib().clv_lstm_seq_bwd(B, T, H, gate_act, _ptr(U), _ptr(dhs), _ptr(cs), _ptr(c0), _ptr(gates_inout),
                                      _ptr(dzsum), _stream()), "clv_lstm_seq_bwd")


def lstm_mx_supported(B, nx, nz, H=88):
    """the large-batch LSTM kernels on the bf16 matrix cores (csrc/lstm_mx.hip) take this shape"""
    return bool(_lib.lib().clv_lstm_mx_supported(B, H, nx, nz))


def lstm_mx_fwd(B, T, X, ldx, nx, Kx, Z, ldz, nz, Kz, rowbias, U, hs, coef, aux, gate_act=0, H=88):
    """LSTM training forward with the input products inside the kernel: frames X (sparse rows of Kx; float32 or the bytes
    themselves, ldx in elements) and latents Z."""
    check(_lib.lib().clv_lstm_mx_fwd(B, T, H, gate_act, _ptr(X), _is_u8(X) if X is not None else 0, ldx, nx, _ptr(Kx), _ptr(Z), ldz, nz, _ptr(Kz),
                                     _ptr(rowbias), _ptr(U), _ptr(hs), _ptr(coef), _ptr(aux), _stream()), "clv_lstm_mx_fwd")


def lstm_mx_bwd(B, T, U, dhs, aux, coef_inout, dzsum, Kz=None, nz=0, dZ=None, lddz=0, H=88):
    """BPTT of lstm_mx_fwd; coef_inout becomes dz; nz > 0: dZ = dz . Kz^T from the same launch."""
    check(_lib.lib().clv_lstm_mx_bwd(B, T, H, _ptr(U), _ptr(dhs), _ptr(aux), _ptr(coef_inout), _ptr(dzsum), _ptr(Kz), nz,
                                     _ptr(dZ), lddz, _stream()), "clv_lstm_mx_bwd")


def sparse_proj_supported(nx, N):
    return bool(_lib.lib().clv_sparse_proj_supported(nx, N))


def sparse_proj2(R, N, a, b, ldo=None):
    """Two projections over the same R frames in one launch; a, b = (nx, X, ldx, K, out); X float32, or both the bytes themselves."""
    if _is_u8(a[1]) != _is_u8(b[1]):
        raise TypeError("both projections read float frames or both read bytes")
    check(_lib.lib().clv_sparse_proj2(R, N, ldo if ldo is not None else N, _is_u8(a[1]), a[0], _ptr(a[1]), a[2], _ptr(a[3]), _ptr(a[4]),
                                      b[0], _ptr(b[1]), b[2], _ptr(b[3]), _ptr(b[4]), _stream()), "clv_sparse_proj2")


def sparse_proj(R, nx, N, X, ldx, K, out, ldo=None):
    """out[r,:N] = sum_k X[r,k] K[k,:] visiting only the nonzero inputs of a frame (K resident in LDS)."""
    check(_lib.lib().clv_sparse_proj(R, nx, N, _ptr(X), _is_u8(X), ldx, _ptr(K), _ptr(out), ldo if ldo is not None else N,
                                     _stream()), "clv_sparse_proj")


def sparse_dense_supported(N):
    return bool(_lib.lib().clv_sparse_dense_supported(N))


def sparse_dense(R, nx, N, X, ldx, K, bias, act, out, ldo=None):
    """out[r,:N] = act(sum_j X[r,j] K[j,:] + bias) over the nonzero inputs of each row (K read from HBM/L2)."""
    check(_lib.lib().clv_sparse_dense(R, nx, N, _ptr(X), ldx, _ptr(K), _ptr(bias), act, _ptr(out),
                                      ldo if ldo is not None else N, _stream()), "clv_sparse_dense")


def sparse_outer(Bn, nx, N, X, ldx, G, ldg, out, ldo=None, colsum=None, gdot=None):
    """out[j,:N] = sum_b X[b,j] G[b,:] (kernel gradient of a Dense layer with sparse inputs); colsum[N] = sum_b G[b,:].
    gdot = (Hact, ldh, hbias, out[N]): also sum_b (Hact - hbias)[b,c] G[b,c] = sum_j K[j,c] dK[j,c] (the `known` sums of clv_adam_wn_step)."""
    h, ldh, hb, go = gdot if gdot is not None else (None, 0, None, None)
    check(_lib.lib().clv_sparse_outer(Bn, nx, N, _ptr(X), ldx, _ptr(G), ldg, _ptr(out), ldo if ldo is not None else N,
                                         _ptr(colsum), _ptr(h), int(ldh), _ptr(hb), _ptr(go), _stream()), "clv_sparse_outer")


def dense_outer_bf16_supported(Bn, nx, N, ldx, ldg):
    return bool(_lib.lib().clv_dense_outer_bf16_supported(Bn, nx, N, ldx, ldg))


def dense_outer_bf16(Bn, nx, N, X, ldx, G, ldg, out, ldo=None, colsum=None, gdot=None):
    """sparse_outer's product for inputs that are exactly bf16 numbers (frames kept as bytes), dense on the bf16 matrix cores
    (csrc/outer_bf16.hip); same arguments.  X: float32, or the bytes themselves (uint8; ldx in elements either way)."""
    h, ldh, hb, go = gdot if gdot is not None else (None, 0, None, None)
    check(_lib.lib().clv_dense_outer_bf16(Bn, nx, N, _ptr(X), _is_u8(X), ldx, _ptr(G), ldg, _ptr(out), ldo if ldo is not None else N,
                                          _ptr(colsum), _ptr(h), int(ldh), _ptr(hb), _ptr(go), _stream()), "clv_dense_outer_bf16")


def vrnn_generate_supported(D, H, L, Cn):
    return bool(_lib.lib().clv_vrnn_generate_supported(D, H, L, Cn))


def vrnn_generate(N, S, nsteps, D, H, L, Cn, gate_act, z_prior, seed, x_seed, w, Kx_enc, Kw_enc, b_enc, U_enc, Wz, bz,
                  Kx_dec, Kz, Kw_dec, b_dec, U_dec, Wo, bo, Xs, xhat=None):
    check(_lib.lib().clv_vrnn_generate(N, S, nsteps, D, H, L, Cn, gate_act, int(bool(z_prior)), int(seed), _ptr(x_seed),
                                       _ptr(w), _ptr(Kx_enc), _ptr(Kw_enc), _ptr(b_enc), _ptr(U_enc), _ptr(Wz), _ptr(bz),
                                       _ptr(Kx_dec), _ptr(Kz), _ptr(Kw_dec), _ptr(b_dec), _ptr(U_dec), _ptr(Wo), _ptr(bo),
                                       _ptr(Xs), _ptr(xhat), _stream()), "clv_vrnn_generate")


def vae_generate_supported(D, H, L, Cn):
    return bool(_lib.lib().clv_vae_generate_supported(D, H, L, Cn))


def vae_generate(N, nsteps, D, H, L, Cn, use_x_prev, z_prior, seed, x_seed, w, Kh, bh, Kz, bz, Kd, bd, Ko, bo, Xs, xhat=None):
    """cl_vae frame loop for N sequences in one persistent launch (csrc/vae_generate.hip)."""
    check(_lib.lib().clv_vae_generate(N, nsteps, D, H, L, Cn, int(bool(use_x_prev)), int(bool(z_prior)), int(seed), _ptr(x_seed),
                                      _ptr(w), _ptr(Kh), _ptr(bh), _ptr(Kz), _ptr(bz), _ptr(Kd), _ptr(bd), _ptr(Ko), _ptr(bo),
                                      _ptr(Xs), _ptr(xhat), _stream()), "clv_vae_generate")


def lstm_pair_supported(L, H=88):
    return bool(_lib.lib().clv_lstm_pair_supported(H, L))


def lstm_pair_pack_floats():
    return int(_lib.lib().clv_lstm_pair_pack_floats())


def lstm_pair_pack(L, U_enc, U_dec, Kz, Wz, pack, H=88):
    """Recurrent kernels, Kz and Wz in the lane order of the pair kernels (once per weight update)."""
    check(_lib.lib().clv_lstm_pair_pack(H, L, _ptr(U_enc), _ptr(U_dec), _ptr(Kz), _ptr(Wz), _ptr(pack), _stream()),
          "clv_lstm_pair_pack")


def lstm_pair_fwd(B, T, L, gates_enc, rb_enc, gates_dec, dec_has_xproj, rb_dec, pack, bz, eps,
                  hs_enc, aux_enc, hs_dec, aux_dec, zargs, Z, ldz, klterm, gate_act=0, H=88, noise=None, notes=None):
    """aux_* [B*T, 2H]: (kcarry, kc) of the backward pass; noise: a noise_draw(): eps is drawn in the kernel;
    notes = (notes_enc, Kx_enc, notes_dec, Kx_dec): the input projections are gathered in the kernel from note lists
    (gather_rows_multi(notes=...)) and the kernels' frame rows; gates_* are then outputs only."""
    ne, ke, nd, kd = notes if notes is not None else (None, None, None, None)
    check(_lib.lib().clv_lstm_pair_fwd(B, T, H, L, gate_act, _ptr(gates_enc), _ptr(rb_enc), _ptr(gates_dec),
                                       int(bool(dec_has_xproj)), _ptr(rb_dec), _ptr(pack), _ptr(bz),
                                       _ptr(eps), _ptr(hs_enc), _ptr(aux_enc), _ptr(hs_dec), _ptr(aux_dec), _ptr(zargs),
                                       _ptr(Z), ldz, _ptr(klterm), _ptr(ne), _ptr(ke), _ptr(nd), _ptr(kd), _noise_ref(noise),
                                       _stream()), "clv_lstm_pair_fwd")


def lstm_pair_bwd(B, T, L, kl_scale, pack, Wz, dhs_dec, aux_dec, aux_enc, gates_dec, gates_enc, dzsum_dec,
                  dzsum_enc, zargs, eps, dzargs, gate_act=0, H=88, head_grad=None, ws=None, defer=None, label=None):
    """head_grad = (hs_enc, dWz, dbz): the latent head's kernel / bias gradient is accumulated inside the kernel (per-row
    slabs, summed by the pending reduction of `defer`, or at once) instead of by a GEMM over hs_enc.
    label = dict(D, C, Kenc_w, Kdec_w, wargs, eps, onehot, W, hW, Ka, prior, class_weight, w_kl_weight, inv_b, dwargs, dhW,
    layer_grad=(dKa, dba) | None): the label path's backward (vrnn_label_bwd) as the kernel's epilogue."""
    Lb = _lib.lib()
    hs, dWz, dbz, buf, nbytes, job = None, None, None, None, 0, None
    need_h = Lb.clv_lstm_pair_bwd_workspace_bytes(B, H, L) if head_grad is not None else 0
    lg = label.get('layer_grad') if label is not None else None
    need_l = Lb.clv_vrnn_label_bwd_workspace_bytes(B, label['D'], label['C']) if lg is not None else 0
    lbuf_ptr, ljob = None, None
    if head_grad is not None:
        hs, dWz, dbz = head_grad
        if defer is not None:
            buf = defer.scratch(need_h)
            job = defer.next_job()
        else:                              # one scratch buffer, two regions
            buf = ws.ensure((need_h + 255) // 256 * 256 + need_l)
        nbytes = need_h
    rider = None
    if label is not None:
        if lg is not None:
            if defer is not None:
                lbuf = defer.scratch(need_l)
                lbuf_ptr, ljob = lbuf.data_ptr(), defer.next_job_addr()
            else:
                base = buf if buf is not None else ws.ensure(need_l)
                lbuf_ptr = base.data_ptr() + ((need_h + 255) // 256 * 256 if buf is not None else 0)
        rider = _lib.LabelBwdRider(label['D'], label['C'], _ptr(label['Kenc_w']), _ptr(label['Kdec_w']), _ptr(label['wargs']),
                                   _ptr(label['eps']), _ptr(label['onehot']), _ptr(label['W']), _ptr(label['hW']),
                                   _ptr(label['Ka']), float(label['prior']), float(label['class_weight']),
                                   float(label['w_kl_weight']), float(label['inv_b']), _ptr(label['dwargs']),
                                   _ptr(label['dhW']), _ptr(lg[0]) if lg else None, _ptr(lg[1]) if lg else None,
                                   lbuf_ptr, need_l, ljob)
    check(Lb.clv_lstm_pair_bwd(B, T, H, L, gate_act, float(kl_scale), _ptr(pack), _ptr(Wz),
                                  _ptr(dhs_dec), _ptr(aux_dec), _ptr(aux_enc), _ptr(gates_dec), _ptr(gates_enc),
                                  _ptr(dzsum_dec), _ptr(dzsum_enc), _ptr(zargs), _ptr(eps), _ptr(dzargs),
                                  _ptr(hs), _ptr(dWz), _ptr(dbz), _ptr(buf), nbytes, job,
                                  C.byref(rider) if rider is not None else None, _stream()),
          "clv_lstm_pair_bwd")


def label_fwd(B, Cn, mean, logvar, ld_in, eps, onehot, prior, w, rowloss):
    check(_lib.lib().clv_label_fwd(B, Cn, _ptr(mean), _ptr(logvar), ld_in, _ptr(eps), _ptr(onehot), float(prior),
                                   _ptr(w), _ptr(rowloss), _stream()), "clv_label_fwd")


def label_bwd(B, Cn, mean, logvar, ld_in, eps, onehot, w, dw, prior, class_weight, w_kl_weight, inv_b,
              dmean, dlogvar, ld_out):
    check(_lib.lib().clv_label_bwd(B, Cn, _ptr(mean), _ptr(logvar), ld_in, _ptr(eps), _ptr(onehot), _ptr(w), _ptr(dw),
                                   float(prior), float(class_weight), float(w_kl_weight), float(inv_b),
                                   _ptr(dmean), _ptr(dlogvar), ld_out, _stream()), "clv_label_bwd")


def vrnn_label_fwd(B, D, Cn, G4, hW, Ka, ba, eps, onehot, prior, Kenc_w, benc, Kdec_w, bdec, wargs, W, rowloss, rb_enc,
                   rb_dec):
    check(_lib.lib().clv_vrnn_label_fwd(B, D, Cn, G4, _ptr(hW), _ptr(Ka), _ptr(ba), _ptr(eps), _ptr(onehot),
                                        float(prior), _ptr(Kenc_w), _ptr(benc), _ptr(Kdec_w), _ptr(bdec), _ptr(wargs),
                                        _ptr(W), _ptr(rowloss), _ptr(rb_enc), _ptr(rb_dec), _stream()),
          "clv_vrnn_label_fwd")


def noise_draw(seed, stream, first, step=0, step_dev=None):
    """clv_noise_draw: the kernel that gets it draws eps itself (the clv_philox_normal values at indices first + e) and
    writes them to its eps buffer.  The struct is returned by value: keep it alive across the call (ctypes copies the
    fields when the launch is made / captured)."""
    return _lib.NoiseDraw(int(seed), int(first), int(stream), int(step), _ptr(step_dev))


def _is_u8(t):
    """1 for a uint8 tensor (frames kept as bytes), 0 for float32: the x_u8 flag of the entry points that take either"""
    if t.dtype == torch.uint8:
        return 1
    if t.dtype != torch.float32:
        raise TypeError("frames must be float32 or uint8, got %s" % t.dtype)
    return 0


def _frames_mode(X, exact):
    """CLV_FRAMES_* of include/clvae.h: 2 = X holds the frames as bytes, 1 = float values that are exactly bf16 numbers, 0 = any"""
    return 2 if _is_u8(X) else int(bool(exact))


def _noise_ref(noise):
    return C.byref(noise) if noise is not None else None


def label_stage(cur, hist, idx, row0, cursor, X, Xh, hist_chunk, hist_ld, w_src, w_out, bytes_out=None):
    """clv_label_stage for vrnn_label_fwd_x(stage=...): cur / hist = (uint8 store, stride, offset, table or None) of the
    current / history frames (hist None: no history frames), cursor = (step_dev, step0, period, stride, offset) or None.
    bytes_out = (X8, Xh8): the rows are copied as bytes into these uint8 [B, nx] buffers instead of widened into X / Xh."""
    g = _lib.LabelStage()
    g.cur, g.cur_stride, g.cur_offset, g.cur_table = _ptr(cur[0]), int(cur[1]), int(cur[2]), _ptr(cur[3])
    if hist is not None:
        g.hist, g.hist_stride, g.hist_offset, g.hist_table = _ptr(hist[0]), int(hist[1]), int(hist[2]), _ptr(hist[3])
    g.idx, g.row0 = _ptr(idx), int(row0)
    if cursor is not None:
        g.cursor = _lib.BatchCursor(_ptr(cursor[0]), int(cursor[1]), int(cursor[2]), int(cursor[3]), int(cursor[4]))
    g.X, g.Xh, g.hist_chunk, g.hist_ld = _ptr(X), _ptr(Xh), int(hist_chunk), int(hist_ld)
    g.w_src, g.w_out = _ptr(w_src), _ptr(w_out)
    if bytes_out is not None:
        g.X8, g.Xh8 = _ptr(bytes_out[0]), _ptr(bytes_out[1])
    return g


def vrnn_label_fwd_x_proj_supported(B, D, nx, T, N):
    return bool(_lib.lib().clv_vrnn_label_fwd_x_proj_supported(B, D, nx, T, N))


def vrnn_label_fwd_x(B, D, Cn, G4, X, ldx, nx, Kh, bh, hW_out, Ka, ba, eps, onehot, prior, Kenc_w, benc, Kdec_w, bdec, wargs,
                     W, rowloss, rb_enc, rb_dec, noise=None, pack=None, parts=None, stage=None, proj=None):
    """hW = relu(X . Kh + bh) over the nonzero inputs of each row, then vrnn_label_fwd, one workgroup per row.
    noise: a noise_draw(): eps is drawn in the kernel (and written to `eps`) instead of read.
    pack = (L, U_enc, U_dec, Kz, Wz, out): the launch also writes the pair LSTM kernels' weight pack (lstm_pair_pack).
    parts = (buffer, splits): X . Kh was formed by dense_window_fwd_bf16 as split-K partial sums; X / Kh are not read.
    proj = (T, N, K_cur, out_cur, K_hist or None, out_hist or None): the launch also forms the LSTMs' frame projections of the
    staged rows (sparse_proj2's outputs bit for bit; needs a stage with bytes_out)."""
    pj = None
    if proj is not None:
        if stage is None or parts is not None:
            raise ValueError("proj rides on a stage (the byte stores it names), not on parts")
        T_, N_, kc, oc, kh, oh = proj
        pj = _lib.FrameProj(int(T_), int(N_), int(N_), _ptr(kc), _ptr(oc), _ptr(kh), _ptr(oh))
    ps = None
    if pack is not None:
        L_, ue, ud, kz, wz, out = pack
        ps = _lib.PairPackSrc(88, int(L_), _ptr(ue), _ptr(ud), _ptr(kz), _ptr(wz), _ptr(out))
    tail = (_ptr(bh), _ptr(hW_out), _ptr(Ka), _ptr(ba), _ptr(eps), _ptr(onehot), float(prior), _ptr(Kenc_w), _ptr(benc),
            _ptr(Kdec_w), _ptr(bdec), _ptr(wargs), _ptr(W), _ptr(rowloss), _ptr(rb_enc), _ptr(rb_dec), _noise_ref(noise),
            C.byref(ps) if ps is not None else None, _stream())
    if parts is not None and stage is None:
        check(_lib.lib().clv_vrnn_label_fwd_parts(B, D, Cn, G4, _ptr(parts[0]), int(parts[1]), *tail), "clv_vrnn_label_fwd_parts")
    else:      # stage = a label_stage(): the launch assembles the mini-batch rows itself (X, history frames, labels)
        check(_lib.lib().clv_vrnn_label_fwd_x(B, D, Cn, G4, _ptr(X), ldx, nx, _ptr(Kh), C.byref(stage) if stage is not None else None,
                                              *tail[:-1], C.byref(pj) if pj is not None else None, tail[-1]), "clv_vrnn_label_fwd_x")


def dense_window_fwd_bf16_supported(Bn, nx, N, ldx, ldk):
    return bool(_lib.lib().clv_dense_window_fwd_bf16_supported(Bn, nx, N, ldx, ldk))


def dense_window_fwd_bf16(Bn, nx, N, X, ldx, K, ldk, ws):
    """X . K for byte-valued X as split-K partial sums on the bf16 matrix cores (csrc/outer_bf16.hip); returns (buffer, splits)
    for vrnn_label_fwd_x(parts=...).  ws: a Workspace that keeps the buffer."""
    L = _lib.lib()
    buf = ws.ensure(L.clv_dense_window_fwd_bf16_workspace_bytes(Bn, nx, N))
    check(L.clv_dense_window_fwd_bf16(Bn, nx, N, _ptr(X), _is_u8(X), ldx, _ptr(K), ldk, _ptr(buf), buf.numel(), _stream()),
          "clv_dense_window_fwd_bf16")
    return buf, L.clv_dense_window_fwd_bf16_splits(Bn, nx)


def vrnn_label_bwd(B, D, Cn, G4, dzsum_enc, dzsum_dec, Kenc_w, Kdec_w, wargs, eps, onehot, W, hW, Ka, prior,
                   class_weight, w_kl_weight, inv_b, dwargs, dhW, layer_grad=None, ws=None, defer=None):
    """layer_grad = (dKa, dba): the Wargs layer's kernel / bias gradient as per-row slabs summed by `defer`'s pending
    reductions (or at once) instead of a GEMM over K = batch."""
    Lb = _lib.lib()
    dKa, dba, buf, nbytes, job = None, None, None, 0, None
    if layer_grad is not None:
        dKa, dba = layer_grad
        need = Lb.clv_vrnn_label_bwd_workspace_bytes(B, D, Cn)
        buf = defer.scratch(need) if defer is not None else ws.ensure(need)
        nbytes = buf.numel()
        job = defer.next_job() if defer is not None else None
    check(Lb.clv_vrnn_label_bwd(B, D, Cn, G4, _ptr(dzsum_enc), _ptr(dzsum_dec), _ptr(Kenc_w), _ptr(Kdec_w),
                                   _ptr(wargs), _ptr(eps), _ptr(onehot), _ptr(W), _ptr(hW), _ptr(Ka),
                                   float(prior), float(class_weight), float(w_kl_weight), float(inv_b),
                                   _ptr(dwargs), _ptr(dhW), _ptr(dKa), _ptr(dba), _ptr(buf), nbytes, job, _stream()),
          "clv_vrnn_label_bwd")


def gauss_fwd(R, Ld, zargs, eps, z, ldz, rowkl):
    check(_lib.lib().clv_gauss_fwd(R, Ld, _ptr(zargs), _ptr(eps), _ptr(z), ldz, _ptr(rowkl), _stream()),
          "clv_gauss_fwd")


def gauss_bwd(R, Ld, zargs, eps, dz, lddz, kl_scale, dzargs):
    check(_lib.lib().clv_gauss_bwd(R, Ld, _ptr(zargs), _ptr(eps), _ptr(dz), lddz, float(kl_scale), _ptr(dzargs),
                                   _stream()), "clv_gauss_bwd")


def bernoulli_nll(R, D, logits, y, ldy, scale, rownll, dlogits):
    check(_lib.lib().clv_bernoulli_nll(R, D, _ptr(logits), _ptr(y), ldy, float(scale), _ptr(rownll), _ptr(dlogits),
                                       _stream()), "clv_bernoulli_nll")


def sum_strided(n, x, stride, scale, out):
    check(_lib.lib().clv_sum_strided(n, _ptr(x), stride, float(scale), _ptr(out), _stream()), "clv_sum_strided")


def axpy(n, alpha, x, y):
    check(_lib.lib().clv_axpy(n, float(alpha), _ptr(x), _ptr(y), _stream()), "clv_axpy")


def gather_rows(rows, row_elems, src, idx, out, chunk=0, out_ld=0):
    check(_lib.lib().clv_gather_rows(rows, row_elems, _ptr(src), _ptr(idx), _ptr(out), chunk, out_ld, _stream()),
          "clv_gather_rows")


NOTE_ROW, NOTE_NONE = 96, 88       # CLV_NOTE_ROW / CLV_NOTE_NONE (include/clvae.h)


def gather_rows_multi(rows, idx, segs, row0=0, notes=None, cursor=None):
    """segs: up to 4 (src, out, row_elems, chunk, out_ld[, stride, offset, table]); one launch; idx None = rows
    row0..row0+rows-1.  A uint8 src (binary frames kept as bytes) is converted to float on the way -- or copied as bytes when
    `out` is a uint8 tensor too.  With (stride,
    offset, table) source row r starts at element table[idx[r]] * stride + offset: windows of a frame store.
    notes: per segment None or a uint8 tensor [rows * pieces, NOTE_ROW] that receives the frames' note lists (uint8
    sources of binary frames only): what lstm_pair_fwd gathers the input projections from.
    cursor = (step_dev, step0, period, stride, offset): batch j = (step - step0) mod period, rows j * stride + offset ..
    of the row list, chosen by the launch itself from the device step counter (a node of the step's graph)."""
    n = len(segs)
    P, I, U = C.c_void_p * n, C.c_int64 * n, C.c_int32 * n
    src = P(*[s_[0].data_ptr() for s_ in segs])
    if any(s_[1].dtype == torch.uint8 and s_[0].dtype != torch.uint8 for s_ in segs):
        raise TypeError("a uint8 output takes a uint8 source")
    # 2: bytes in, bytes out (a batch that stays uint8, out_ld in bytes); 1: bytes widened to float; 0: float
    u8 = U(*[(2 if s_[1].dtype == torch.uint8 else 1) if s_[0].dtype == torch.uint8 else 0 for s_ in segs])
    out = P(*[s_[1].data_ptr() for s_ in segs])
    re = I(*[int(s_[2]) for s_ in segs])
    ch = I(*[int(s_[3]) for s_ in segs])
    ld = I(*[int(s_[4]) for s_ in segs])
    ext = [tuple(s_[5:8]) + (0, 0, None)[len(s_[5:8]):] for s_ in segs]
    st = I(*[int(e[0]) for e in ext])
    of = I(*[int(e[1]) for e in ext])
    tb = P(*[(e[2].data_ptr() if e[2] is not None else None) for e in ext])
    nt = P(*[(t.data_ptr() if t is not None else None) for t in (notes or [None] * n)])
    cur = None      # (step_dev int32 tensor, step0, period, stride, offset): the batch is chosen on the device
    if cursor is not None:
        cur = C.byref(_lib.BatchCursor(cursor[0].data_ptr(), int(cursor[1]), int(cursor[2]), int(cursor[3]), int(cursor[4])))
    check(_lib.lib().clv_gather_rows_multi(rows, _ptr(idx), int(row0), n, src, u8, out, re, ch, ld, st, of, tb, nt, cur, _stream()),
          "clv_gather_rows_multi")


def philox_normal2(out0, n0, stream0, first0, out1, n1, stream1, first1, seed, step=0, step_dev=None):
    check(_lib.lib().clv_philox_normal2(_ptr(out0), n0, stream0, first0, _ptr(out1), n1, stream1, first1, seed, step,
                                        _ptr(step_dev), _stream()), "clv_philox_normal2")


def philox_normal(out, n, seed, step=0, stream_id=0, first_index=0, step_dev=None):
    check(_lib.lib().clv_philox_normal(_ptr(out), n, seed, step, _ptr(step_dev), stream_id, first_index, _stream()),
          "clv_philox_normal")


def philox_uniform(out, n, seed, step=0, stream_id=0, first_index=0, step_dev=None):
    check(_lib.lib().clv_philox_uniform(_ptr(out), n, seed, step, _ptr(step_dev), stream_id, first_index, _stream()),
          "clv_philox_uniform")


def act_grad(n, act, y, dy, dpre):
    check(_lib.lib().clv_act_grad(n, act, _ptr(y), _ptr(dy), _ptr(dpre), _stream()), "clv_act_grad")


def i32_add(counter, v=1):
    check(_lib.lib().clv_i32_add(_ptr(counter), int(v), _stream()), "clv_i32_add")


def dropout_rows(R, T, n, X, ldx, U, ldu, rate, out, ldo, beta=0.0):
    """out[r, :n] = beta * out + X[r, :n] * mask(U[r // T, :n]), mask(u) = (u >= rate) / (1 - rate) (clv_dropout_rows)."""
    check(_lib.lib().clv_dropout_rows(R, T, n, _ptr(X), ldx, _ptr(U), ldu, float(rate), float(beta), _ptr(out), ldo, _stream()),
          "clv_dropout_rows")


def bernoulli_sample(n, p, u, x):
    check(_lib.lib().clv_bernoulli_sample(n, _ptr(p), _ptr(u), _ptr(x), _stream()), "clv_bernoulli_sample")


class Graph:
    """Capture the kernels enqueued inside the ``with`` block on the current stream; replay with launch()."""

    def __init__(self):
        self.handle = C.c_void_p()
        self._stream = None

    def __enter__(self):
        self._stream = torch.cuda.Stream()
        self._stream.wait_stream(torch.cuda.current_stream())
        self._ctx = torch.cuda.stream(self._stream)
        self._ctx.__enter__()
        check(_lib.lib().clv_graph_begin_capture(_stream()), "graph begin")
        return self

    def __exit__(self, et, ev, tb):
        code = _lib.lib().clv_graph_end_capture(_stream(), C.byref(self.handle))
        self._ctx.__exit__(et, ev, tb)
        torch.cuda.current_stream().wait_stream(self._stream)
        if et is None:
            check(code, "graph end")
        return False

    def launch(self):
        check(_lib.lib().clv_graph_launch(self.handle, _stream()), "graph launch")

    def __del__(self):
        try:
            if self.handle:
                _lib.lib().clv_graph_destroy(self.handle)
        except Exception:
            pass


def prof_enable(on=True):
    check(_lib.lib().clv_prof_enable(int(on)))


def prof_empty_scope():
    """an empty bracket of the profiler's events on the current stream (record name "event_pair")"""
    check(_lib.lib().clv_prof_empty_scope(_stream()))


def prof_collect(cap=64):
    arr = (_lib.ProfRecord * cap)()
    n = _lib.lib().clv_prof_collect(arr, cap)
    if n < 0:
        check(n, "clv_prof_collect")
    return [(arr[i].name.decode(), arr[i].launches, arr[i].total_ms) for i in range(n)]
