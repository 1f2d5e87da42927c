"""ctypes binding of libclvae_hip.so (the C ABI declared in include/clvae.h).

The product has NO CPU fallback: ``lib()`` raises if the shared library is not
built, ``require_gpu()`` raises if no gfx950 device is visible.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CLV_LIB: another build of the same ABI (A/B measurements inside one GPU session; tools/build_variant.sh)
LIB_PATH = os.environ.get("CLV_LIB") or os.path.join(_HERE, "libclvae_hip.so")

ABI_VERSION = 600      # CLV_ABI_VERSION of include/clvae.h
ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_MASKPOS = 0, 1, 2, 3
GATE_HARD_SIGMOID, GATE_SIGMOID = 0, 1

_f = C.c_float
_i = C.c_int
_p = C.c_void_p
_sz = C.c_size_t
_i64 = C.c_int64
_u64 = C.c_uint64
_u32 = C.c_uint32


class VaeStepOpts(C.Structure):
    """clv_vae_step_opts (include/clvae.h)."""
    _fields_ = [("draw", _i), ("stream_w", _u32), ("stream_z", _u32), ("step", _u32),
                ("noise_seed", _u64), ("first_w", _u64), ("first_z", _u64),
                ("step_dev", _p), ("loss_means", _p), ("bump_iterations", _p), ("bf16", _i)]



class NoiseDraw(C.Structure):
    """clv_noise_draw (include/clvae.h): a kernel draws its own eps, the values of clv_philox_normal."""
    _fields_ = [("seed", _u64), ("first", _u64), ("stream", _u32), ("step", _u32), ("step_dev", _p)]


class PairPackSrc(C.Structure):
    """clv_pair_pack_src (include/clvae.h)."""
    _fields_ = [("H", _i), ("L", _i), ("U_enc", _p), ("U_dec", _p), ("Kz", _p), ("Wz", _p), ("pack", _p)]


class AdamKnownSums(C.Structure):
    """clv_adam_known_sums (include/clvae.h)."""
    _fields_ = [("tensor", _i), ("use", _i), ("gdot", _p), ("vnorm2", _p)]


class BatchCursor(C.Structure):
    """clv_batch_cursor (include/clvae.h)."""
    _fields_ = [("step_dev", _p), ("step0", C.c_int32), ("period", C.c_int32), ("stride", _i64), ("offset", _i64)]


class LabelStage(C.Structure):
    """clv_label_stage (include/clvae.h)."""
    _fields_ = [("cur", _p), ("hist", _p),
                ("cur_stride", _i64), ("cur_offset", _i64), ("hist_stride", _i64), ("hist_offset", _i64), ("row0", _i64),
                ("cur_table", _p), ("hist_table", _p), ("idx", _p),
                ("cursor", BatchCursor),
                ("X", _p), ("Xh", _p), ("hist_chunk", C.c_int32), ("hist_ld", _i64),
                ("w_src", _p), ("w_out", _p), ("X8", _p), ("Xh8", _p)]


class FrameProj(C.Structure):
    """clv_frame_proj (include/clvae.h)."""
    _fields_ = [("T", C.c_int32), ("N", C.c_int32), ("ldo", C.c_int32), ("K_cur", _p), ("out_cur", _p), ("K_hist", _p), ("out_hist", _p)]


class WgradProblem(C.Structure):
    """clv_wgrad_problem (include/clvae.h)."""
    _fields_ = [("K", C.c_int32), ("N", C.c_int32),
                ("X", _p), ("ldx", C.c_int32), ("nx", C.c_int32), ("x_exact_bf16", C.c_int32),
                ("H", _p), ("ldh", C.c_int32), ("nh", C.c_int32), ("h_shift", C.c_int32), ("h_zero_period", C.c_int32),
                ("Z", _p), ("ldz", C.c_int32), ("nz", C.c_int32),
                ("dz", _p), ("lddz", C.c_int32),
                ("dKx", _p), ("ld_kx", C.c_int32),
                ("dU", _p), ("ld_u", C.c_int32),
                ("dKz", _p), ("ld_kz", C.c_int32),
                ("beta", C.c_float),
                ("ws", _p), ("ws_bytes", C.c_size_t)]


class ParamDesc(C.Structure):
    _fields_ = [("offset", C.c_int64), ("rows", C.c_int32), ("cols", C.c_int32),
                ("col_offset", C.c_int64), ("is_matrix", C.c_int32), ("pad_", C.c_int32)]


class GemmProb(C.Structure):
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int32), ("M", C.c_int32), ("C", C.c_void_p), ("ldc", C.c_int32),
                ("a_shift", C.c_int32), ("a_zero_period", C.c_int32), ("ones", C.c_int32)]


class ReduceJob(C.Structure):
    """clv_reduce_job: an opaque pending split-K reduction (include/clvae.h)."""
    _fields_ = [("opaque", C.c_ubyte * 160)]


class SkinnyProduct(C.Structure):
    """clv_skinny_product: a few-row product riding in the reduction launch (include/clvae.h)."""
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int), ("rows", C.c_int), ("B", C.c_void_p), ("ldb", C.c_int),
                ("N", C.c_int), ("K", C.c_int), ("C", C.c_void_p), ("ldc", C.c_int), ("bias_row", C.c_void_p)]


class LabelBwdRider(C.Structure):
    """clv_label_bwd_rider: the label path's backward as the pair backward kernel's epilogue (include/clvae.h)."""
    _fields_ = [("D", C.c_int), ("C", C.c_int), ("Kenc_w", C.c_void_p), ("Kdec_w", C.c_void_p), ("wargs", C.c_void_p),
                ("eps", C.c_void_p), ("onehot", C.c_void_p), ("W", C.c_void_p), ("hW", C.c_void_p), ("Ka", C.c_void_p),
                ("prior_logvar", C.c_float), ("class_weight", C.c_float), ("w_kl_weight", C.c_float), ("inv_b", C.c_float),
                ("dwargs", C.c_void_p), ("dhW", C.c_void_p), ("dKa", C.c_void_p), ("dba", C.c_void_p),
                ("ws", C.c_void_p), ("ws_bytes", C.c_size_t), ("job", C.c_void_p)]


class ProfRecord(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int32), ("total_ms", C.c_float)]


# name -> (restype, argtypes); must list every function of include/clvae.h
SIGNATURES = {
    "clv_version": (_i, []),
    "clv_device_count": (_i, []),
    "clv_error_string": (C.c_char_p, [_i]),
    "clv_gemm_auto_split": (_i, [_i, _i, _i]),
    "clv_gemm_workspace_bytes": (_sz, [_i, _i, _i]),
    "clv_gemm_f32": (_i, [_i, _i, _i, _i, _i, _f, _p, _i, _p, _i, _f, _p, _i, _p, _i, _p, _i, _p, _sz, _p, _p]),
    "clv_gemm_grouped_tn": (_i, [_p, _i, _i, _i, _p, _i, _f, _i, _p, _sz, _p, _p]),
    "clv_splitk_reduce_multi": (_i, [_p, _i, _p, _p, _p, _i, _p, _p, _i, _p]),
    "clv_lstm_wgrad_supported": (_i, [_i, _i, _i, _i, _i]),
    "clv_lstm_wgrad_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "clv_lstm_wgrad_pair_supported": (_i, [_p, _p]),
    "clv_lstm_wgrad_pair_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "clv_lstm_wgrad_pair": (_i, [_p, _p, _i, _p, _p, _p]),
    "clv_lstm_wgrad": (_i, [_i, _i, _p, _i, _i, _i, _p, _i, _i, _i, _i, _p, _i, _i, _p, _i, _p, _i, _p, _i, _p, _i, _f,
                               _i, _p, _sz, _p, _p]),
    "clv_gemm_grouped_tn_small2": (_i, [_p, _i, _p, _p, _i, _p, _i, _i, _i, _p]),
    "clv_gemm_grouped_auto_split": (_i, [_p, _i, _i, _i]),
    "clv_gemm_grouped_workspace_bytes": (_sz, [_p, _i, _i, _i]),
    "clv_colsum_workspace_bytes": (_sz, [_i, _i]),
    "clv_colsum_f32": (_i, [_i, _i, _p, _i, _f, _p, _p, _sz, _p]),
    "clv_lstm_seq_fwd": (_i, [_i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "clv_lstm_seq_bwd_z": (_i, [_i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _i, _p, _i, _p]),
    "clv_lstm_seq_bwd": (_i, [_i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p]),
    "clv_lstm_mx_supported": (_i, [_i, _i, _i, _i]),
    "clv_lstm_mx_fwd": (_i, [_i, _i, _i, _i, _p, _i, _i, _i, _p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p]),
    "clv_lstm_mx_bwd": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _p, _i, _p, _i, _p]),
    "clv_lstm_pair_supported": (_i, [_i, _i]),
    "clv_lstm_pair_pack_floats": (_sz, []),
    "clv_lstm_pair_pack": (_i, [_i, _i, _p, _p, _p, _p, _p, _p]),
    "clv_lstm_pair_fwd": (_i, [_i, _i, _i, _i, _i, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p,
                               _p, _p]),
    "clv_lstm_pair_bwd": (_i, [_i, _i, _i, _i, _i, _f] + [_p] * 12 + [_p, _p, _p, _p, _sz, _p, _p, _p]),
    "clv_lstm_pair_bwd_workspace_bytes": (_sz, [_i, _i, _i]),
    "clv_sparse_proj_supported": (_i, [_i, _i]),
    "clv_sparse_proj_lds_bytes": (_sz, [_i, _i]),
    "clv_sparse_proj": (_i, [_i, _i, _i, _p, _i, _i, _p, _p, _i, _p]),
    "clv_sparse_proj2": (_i, [_i, _i, _i, _i, _i, _p, _i, _p, _p, _i, _p, _i, _p, _p, _p]),
    "clv_sparse_dense_supported": (_i, [_i]),
    "clv_sparse_dense": (_i, [_i, _i, _i, _p, _i, _p, _p, _i, _p, _i, _p]),
    "clv_sparse_outer": (_i, [_i, _i, _i, _p, _i, _p, _i, _p, _i, _p, _p, _i, _p, _p, _p]),
    "clv_dense_outer_bf16_supported": (_i, [_i, _i, _i, _i, _i]),
    "clv_dense_outer_bf16": (_i, [_i, _i, _i, _p, _i, _i, _p, _i, _p, _i, _p, _p, _i, _p, _p, _p]),
    "clv_gemm_bce_f32": (_i, [_i, _i, _i, _p, _i, _p, _i, _p, _p, _i, _f, _p, _p, _i, _p, _p]),
    "clv_out_head_train_supported": (_i, [_i, _i]),
    "clv_out_head_train_workspace_bytes": (_sz, [_i]),
    "clv_out_head_train": (_i, [_i, _i, _i, _p, _p, _p, _p, _i, _i, _f, _p, _p, _p, _p, _p, _p, _p, _sz, _p, _p]),
    "clv_latent_head_supported": (_i, [_i, _i]),
    "clv_latent_head_bwd_workspace_bytes": (_sz, [_i, _i]),
    "clv_latent_head_fwd": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p]),
    "clv_latent_head_bwd": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _i, _f, _p, _p, _p, _p, _p, _sz, _p, _p]),
    "clv_vrnn_generate_supported": (_i, [_i, _i, _i, _i]),
    "clv_vrnn_generate": (_i, [_i] * 9 + [_u64] + [_p] * 18),
    "clv_vae_generate_supported": (_i, [_i, _i, _i, _i]),
    "clv_vae_generate": (_i, [_i] * 8 + [_u64] + [_p] * 13),
    "clv_label_fwd": (_i, [_i, _i, _p, _p, _i, _p, _p, _f, _p, _p, _p]),
    "clv_label_bwd": (_i, [_i, _i, _p, _p, _i, _p, _p, _p, _p, _f, _f, _f, _f, _p, _p, _i, _p]),
    "clv_vae_fused_supported": (_i, [_i, _i, _i, _i, _i]),
    "clv_vae_fused_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "clv_vae_fused_step": (_i, [_i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, C.c_long, _f, _f, _f, _f, _i,
                                _p, _p, _sz, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "clv_vrnn_label_fwd": (_i, [_i, _i, _i, _i, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "clv_vrnn_label_fwd_x": (_i, [_i, _i, _i, _i, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                  _p, _p, _p, _p]),
    "clv_vrnn_label_fwd_x_proj_supported": (_i, [_i, _i, _i, _i, _i]),
    "clv_vrnn_label_fwd_parts": (_i, [_i, _i, _i, _i, _p, _i, _p, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                  _p, _p, _p]),
    "clv_dense_window_fwd_bf16_supported": (_i, [_i, _i, _i, _i, _i]),
    "clv_dense_window_fwd_bf16_splits": (_i, [_i, _i]),
    "clv_dense_window_fwd_bf16_workspace_bytes": (_sz, [_i, _i, _i]),
    "clv_dense_window_fwd_bf16": (_i, [_i, _i, _i, _p, _i, _i, _p, _i, _p, _sz, _p]),
    "clv_vrnn_label_bwd_workspace_bytes": (_sz, [_i, _i, _i]),
    "clv_vrnn_label_bwd": (_i, [_i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _f, _f, _f, _f, _p, _p, _p, _p, _p, _sz,
                                   _p, _p]),
    "clv_gauss_fwd": (_i, [_i, _i, _p, _p, _p, _i, _p, _p]),
    "clv_gauss_bwd": (_i, [_i, _i, _p, _p, _p, _i, _f, _p, _p]),
    "clv_bernoulli_nll": (_i, [_i, _i, _p, _p, _i, _f, _p, _p, _p]),
    "clv_loss_sums": (_i, [_p, _i, _i, _p, _i, _i, _p, _i, _i, _p, _i, _i, _p, _i, _i, _p, _p]),
    "clv_sum_strided": (_i, [_i, _p, _i, _f, _p, _p]),
    "clv_act_grad": (_i, [_i64, _i, _p, _p, _p, _p]),
    "clv_axpy": (_i, [_i64, _f, _p, _p, _p]),
    "clv_gather_rows": (_i, [_i64, _i64, _p, _p, _p, _i64, _i64, _p]),
    "clv_adam_wn_plan_bytes": (_sz, [_p, _i]),
    "clv_adam_wn_plan_build": (_i, [_p, _i, _p]),
    "clv_adam_wn_workspace_bytes": (_sz, [_p, _i]),
    "clv_adam_wn_step": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _f, _f, _f, _f, _i, _p, _p, _sz, _p]),
    "clv_philox_normal2": (_i, [_p, _i64, _u32, _u64, _p, _i64, _u32, _u64, _u64, _u32, _p, _p]),
    "clv_gather_rows_multi": (_i, [_i64, _p, _i64, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "clv_philox_normal": (_i, [_p, _i64, _u64, _u32, _p, _u32, _u64, _p]),
    "clv_philox_uniform": (_i, [_p, _i64, _u64, _u32, _p, _u32, _u64, _p]),
    "clv_i32_add": (_i, [_p, C.c_int32, _p]),
    "clv_bernoulli_sample": (_i, [_i64, _p, _p, _p, _p]),
    "clv_dropout_rows": (_i, [_i, _i, _i, _p, _i, _p, _i, _f, _f, _p, _i, _p]),
    "clv_graph_begin_capture": (_i, [_p]),
    "clv_graph_end_capture": (_i, [_p, C.POINTER(_p)]),
    "clv_graph_launch": (_i, [_p, _p]),
    "clv_graph_destroy": (_i, [_p]),
    "clv_prof_enable": (_i, [_i]),
    "clv_prof_collect": (_i, [C.POINTER(ProfRecord), _i]),
    "clv_prof_empty_scope": (_i, [_p]),
}

_lib = None


class ClvError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises if the library is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ClvError(
                "libclvae_hip.so is not built (%s). Run `python __graft_entry__.py` or "
                "`make -C classifying-vae-lstm_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        # torch ships its own ROCm runtime: it has to be the first HIP runtime mapped into the process, otherwise
        # torch.cuda later reports "No HIP GPUs are available"
        import torch  # noqa: F401
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)      # AttributeError if the ABI drifted
            fn.restype = res
            fn.argtypes = args
        if h.clv_version() != ABI_VERSION:      # a library of another round: same symbols, other argument lists
            raise ClvError("libclvae_hip.so has ABI version %d, this binding was written for %d (include/clvae.h: "
                           "CLV_ABI_VERSION); rebuild with `make -C classifying-vae-lstm_amd/csrc`"
                           % (h.clv_version(), ABI_VERSION))
        _lib = h
    return _lib


def require_gpu():
    n = lib().clv_device_count()
    if n <= 0:
        raise ClvError("no gfx950 (MI355X) device visible; the HIP path has no CPU fallback")
    return n


def check(code, what=""):
    if code != 0:
        msg = lib().clv_error_string(int(code))
        raise ClvError("%s failed: %s (%d)" % (what or "clv call", msg.decode() if msg else "?", code))
