"""ctypes binding of libnvf_hip.so (the C ABI declared in include/nvf_hip.h).

The library is the only compute path: there is no CPU fallback.  ``lib()`` raises if the
shared object is missing (run ``python -m nvfpcc_amd.build`` or ``__graft_entry__.build()``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NVF_LIB", os.path.join(_HERE, "libnvf_hip.so"))   # NVF_LIB: A/B a second build

P = C.c_void_p
I = C.c_int
F = C.c_float
L = C.c_int64
U = C.c_uint64
Z = C.c_size_t

# name -> (restype, argtypes); mirrors include/nvf_hip.h one to one
PROTOTYPES = {
    "nvf_version": (I, []),
    "nvf_step_ctx_bytes": (Z, []),
    "nvf_step_ctx_init": (I, [P]),
    "nvf_step_ctx_set_direct": (I, [P, I]),
    "nvf_step_ctx_set_wgrad_forms": (I, [P, I, I]),
    "nvf_pack_conv_weight": (I, [P, I, I, I, P, P, P]),
    "nvf_pack_convT_weight": (I, [P, I, I, I, P, P, P]),
    "nvf_effective_params": (I, [P, P, P, P, I, P, P, P, I, I, U, U, P]),
    "nvf_layer_desc_size": (Z, []),
    "nvf_prepare_weights": (I, [P, I, I, U, U, P, P]),
    "nvf_conv3d_gather": (I, [P, P, P, P, P, P] + [I] * 14 + [P]),
    "nvf_convT3d_k5s2_fwd": (I, [P, P, P, P] + [I] * 12 + [P]),
    "nvf_pack_mfma_k4_floats": (Z, [I, I]),
    "nvf_pack_mfma_k4": (I, [P, I, I, I, P, P]),
    "nvf_pack_mfma_k4_multi": (I, [P, P, P, P, I, P]),
    "nvf_conv3d_k4_mfma": (I, [P, P, P, P, P, P] + [I] * 13 + [P]),
    "nvf_conv3d_k4_mfma_bias": (I, [P, P, P, P, P, P] + [I] * 13 + [P, P, P]),
    "nvf_pack_wino_k4_floats": (Z, []),
    "nvf_conv3d_k4_wino_bwd": (I, [P, P, P, P, I, I, I, P, P, P]),
    "nvf_conv3d_k4_wino_fwd": (I, [P, P, P, P, I, I, I, P]),
    "nvf_wgrad_k4_wino": (I, [P, P, P, P, P, Z, I, I, P]),
    "nvf_pack_wino16_k4_floats": (Z, []),
    "nvf_conv3d_k4_wino16_bwd": (I, [P, P, P, P, I, I, I, P, P, P]),
    "nvf_conv3d_k4_wino16_fwd": (I, [P, P, P, P, I, I, I, P]),
    "nvf_wgrad16_k4_wino_partial": (I, [P, P, P, I, I, I, I, P, P]),
    "nvf_pack_convT_mfma_floats": (Z, [I]),
    "nvf_pack_convT_mfma": (I, [P, I, I, P, P]),
    "nvf_convT3d_k5s2_mfma": (I, [P, P, P, P, I, I, I, I, I, I, P]),
    "nvf_pack_s2k5_mfma_floats": (Z, [I, I]),
    "nvf_pack_s2k5_mfma": (I, [P, I, I, P, P]),
    "nvf_conv3d_s2k5_mfma": (I, [P, P, P, P, P, I, I, I, I, I, I, P]),
    "nvf_pack_convT16_mfma_floats": (Z, [I, I]),
    "nvf_pack_convT16_mfma": (I, [P, I, I, P, P]),
    "nvf_convT3d_k5s2_mfma16": (I, [P, P, P, P, I, I, I, I, I, I, I, P]),
    "nvf_pack_g16_mfma_floats": (Z, [I, I, I]),
    "nvf_pack_g16_mfma": (I, [P, I, I, I, P, P]),
    "nvf_conv3d_g16_mfma": (I, [P, P, P, P, P, P] + [I] * 14 + [P]),
    "nvf_pack_mfma_all": (I, [P, P, P, P, P, I, P]),
    "nvf_heads3_fwd": (I, [P, P, P, P, P, P, I, I, P]),
    "nvf_heads3_bwd_data": (I, [P, P, P, P, P, P, I, P]),
    "nvf_heads3_loss_bwd_data": (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, I, P, Z, P, P]),
    "nvf_heads3_loss_bwd_data_bias": (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, I, P, P, Z, P, P]),
    "nvf_heads3_fwd_loss_bwd_data": (I, [P, P, P, P, I, P, P, P, P, P, P, P, P, P, P, P, P, I, P, P, Z, P, P, P]),
    "nvf_heads3_wgrad_partial": (I, [P, P, P, P, P, I, I, P, P]),
    "nvf_stem_fwd": (I, [P] * 10 + [I, I, I, I, P]),
    "nvf_stem_latent_fwd": (I, [P] * 12 + [I, U, U, P] + [P] * 9 + [I, I, I, I, P]),
    "nvf_stem_bwd_workspace": (Z, [I, I]),
    "nvf_stem_bwd_workspace_for": (Z, [I, I, I, I]),
    "nvf_stem_bwd": (I, [P] * 13 + [Z, I, I, I, I, P]),
    "nvf_stem_bwd_partial": (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, Z, I, I, I, I, P, P, P, P]),
    "nvf_wgrad_workspace": (Z, [I] * 7),
    "nvf_wgrad": (I, [P, P, P, P, Z] + [I] * 15 + [P]),
    "nvf_wgrad_partial": (I, [P, P, P, P, Z] + [I] * 14 + [P, P]),
    "nvf_wgrad_reduce_multi": (I, [P, P, P, P, I, P]),
    "nvf_wgrad_mfma3_partial": (I, [P, P, P, I, P, P, P]),
    "nvf_wgrad_up1_conv0_partial": (I, [P, P, P, I, P, P]),
    "nvf_wgrad_trunk5_partial": (I, [P, P, P, I, P, P, P]),
    "nvf_wgrad_trunk5_partial_bias": (I, [P, P, P, P, I, P, P, P]),
    "nvf_wgrad_trunk5_heads_partial": (I, [P, P, P, P, P, P, P, I, I, P, P, P, P]),
    "nvf_wgrad_trunk5_heads_sums_partial": (I, [P, P, P, P, P, P, P, I, P, P, P, P, I, P, Z, P, P, I, P, P, P, P]),
    "nvf_wgrad_reduce_finals_tail": (I, [P, P, P, P, I, P, P, P, P, P, I, P]),
    "nvf_wgrad_reduce_finals": (I, [P, P, P, P, I, P, P, P]),
    "nvf_channel_sum_workspace": (Z, [I]),
    "nvf_channel_sum": (I, [P, P, P, Z, I, I, I, I, P]),
    "nvf_multi_channel_sum_workspace": (Z, [I]),
    "nvf_multi_channel_sum": (I, [P, P, P, P, I, I, P, Z, P, P]),
    "nvf_wgrad_reduce_multi_and_sums": (I, [P, P, P, P, I, P, P, P, P, I, I, P, Z, P, P]),
    "nvf_latent_tail_queue": (I, [P, P, P, P, P, P, P, P, P, P, F, I, U, U, P, P, P, P, P, P, P, P, P, P, I, I, I]),
    "nvf_stem_bwd_queue": (I, [P] * 16 + [Z, P, I, I, I, I, P]),
    "nvf_stem_bwd_pending": (I, [P]),
    "nvf_latent_tail_pending": (I, [P]),
    "nvf_latent_tail_cancel": (None, [P]),
    "nvf_finals_begin": (I, [P]),
    "nvf_finals_flush": (I, [P, P]),
    "nvf_finals_cancel": (None, [P]),
    "nvf_gdn_fwd": (I, [P, P, P, P, I, I, I, I, P]),
    "nvf_gdn_bwd_workspace": (Z, [I]),
    "nvf_gdn_bwd": (I, [P, P, P, P, P, P, P, P, Z, I, I, I, I, P]),
    "nvf_latent_fwd": (I, [P] * 12 + [I, I, I, I, U, U, P, P]),
    "nvf_latent_rate": (I, [P] * 12 + [F, I, I, I, I, U, U, P, P]),
    "nvf_weight_rate": (I, [P, I, P, P, P, P, P, P, P, F, I, P]),
    "nvf_weight_rate_batch_workspace": (Z, []),
    "nvf_weight_rate_batch": (I, [P, P, P, I, P, P, P, P, P, P, F, P, Z, P, P]),
    "nvf_reduce_workspace": (Z, []),
    "nvf_focal_loss": (I, [P, P, P, F, F, P, P, P, F, P, Z, L, I, I, P]),
    "nvf_focal_loss_multi": (I, [P, P, P, P, P, P, P, I, P, I, P, Z, P, P]),
    "nvf_metrics": (I, [P, P, P, F, F, P, P, Z, L, I, P, P]),
    "nvf_metrics_workspace": (Z, []),
    "nvf_metrics3": (I, [P, P, P, P, I, F, F, P, P, Z, P, P]),
    "nvf_sigmoid_bwd": (I, [P, P, P, L, P]),
    "nvf_relu_bwd": (I, [P, P, P, L, P]),
    "nvf_squared_error_map": (I, [P, P, F, P, I, I, P]),
    "nvf_maxpool2": (I, [P, P, I, I, I, I, P]),
    "nvf_adam_step": (I, [P, P, P, P, L, F, F, F, F, I, P]),
    "nvf_step_tail": (I, [P, P]),
    "nvf_adam_coefficients": (I, [F, F, F, I, P]),
    "nvf_adam_coefficients_n": (I, [F, F, F, I, I, P]),
    "nvf_gather_rows": (I, [P, P, P, I, I, P]),
    "nvf_scatter_add_rows": (I, [P, P, P, I, I, P]),
    "nvf_gather_rows_multi": (I, [P, P, P, I, P, I, P]),
    "nvf_step_head": (I, [P, I, I, U, U, P, P, P, P, P, P, P, I, P, P, P, I, P, I, P, P]),
    "nvf_step_head_stem": (I, [P, I, I, U, U, P, P, P, P, P, P, P, I, P, P, P, I, P, I, P, P, P]),
    "nvf_weight_rate_batch_final": (I, [P, P, P, P, P, P]),
    "nvf_wgrad_reduce_multi_and_sums_fused": (I, [P, P, P, P, I, P, P, P, P, P, P, I, I, P, Z, P, P]),
    "nvf_finals_flush_tail": (I, [P, P, P, I, P]),
    "nvf_uniform": (I, [P, L, U, U, P]),
    "nvf_nearest_dist2": (I, [P, P, P, P, P, P, I, P]),
    "nvf_threshold_count": (I, [P, F, P, I, I, P]),
    "nvf_threshold_compact": (I, [P, F, P, P, P, I, I, P]),
}



class NvfStepTail(C.Structure):
    """include/nvf_hip.h: typedef struct NvfStepTail (field for field)."""
    _fields_ = [("p", P), ("g", P), ("m", P), ("v", P), ("n", L), ("coef_dev", P),
                ("coef0_host", F), ("coef1_host", F), ("beta1", F), ("beta2", F), ("eps", F), ("nnb", C.c_int32),
                ("loss_terms", P), ("lbits", P), ("nbits", P), ("inv_npts_dev", P),
                ("inv_npts_host", F), ("nbits_scale", F), ("counts", P), ("acc", P),
                ("sched_buf", P), ("sched_rows", P), ("sched_cursor", P), ("done", P),
                ("sched_words", C.c_int32), ("reserved", C.c_int32)]


class NvfRateJob(C.Structure):
    """include/nvf_hip.h: typedef struct NvfRateJob."""
    _fields_ = [("kernel", P * 8), ("dk", P * 8), ("n", C.c_int32 * 8), ("nlayers", C.c_int32), ("reserved", C.c_int32),
                ("sigma", P), ("mu", P), ("part", P), ("g", F), ("reserved2", F)]


class NvfStemHead(C.Structure):
    """include/nvf_hip.h: typedef struct NvfStemHead."""
    _fields_ = [(n, P) for n in ("emb", "lat_beta_hat", "lat_gamma_hat", "sigma", "mu", "beta_hat", "gamma_hat", "h", "lat",
                                 "x_rounded", "bits", "a0", "h0", "y1")] + \
               [(n, C.c_int32) for n in ("lat_row", "up0_row", "conv0_row", "mode", "ch", "c0", "c1", "reserved")]


class NvfAdamFuse(C.Structure):
    """include/nvf_hip.h: typedef struct NvfAdamFuse."""
    _fields_ = [("g_base", P), ("p_base", P), ("m_base", P), ("v_base", P), ("n", L), ("coef_dev", P),
                ("coef0_host", F), ("coef1_host", F), ("beta1", F), ("beta2", F), ("eps", F), ("reserved", F),
                ("bad_count", P)]


_lib = None


def lib():
    """Load (once) and return the ctypes handle with every prototype declared."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(python -m nvfpcc_amd.build). There is no CPU fallback for the NVF hot path.")
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(h, name)   # AttributeError if the .so is stale
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


class NvfError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        kind = {-1: "NVF_EINVAL (rejected argument)", -2: "NVF_EWORKSPACE (workspace too small)"}.get(
            rc, f"hipError_t {rc}")
        raise NvfError(f"{what} failed: {kind}")
