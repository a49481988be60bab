"""ctypes binding of libspeechclip_hip.so (the C ABI declared in include/speechclip_hip.h).

There is NO fallback: if the library is missing or a symbol cannot be resolved this module raises, and
every op in ``speechclip_plus_amd.ops`` fails loudly.  Build with ``python -m speechclip_plus_amd.build``
(or ``__graft_entry__.build()``).
"""
import contextlib
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SC_LIB_PATH") or os.path.join(_HERE, "csrc", "libspeechclip_hip.so")   # override: same-box A/B of two builds

c_void_p, c_int, c_i64, c_float = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float


class GemmArgs(ctypes.Structure):
    """Mirror of ``sc_gemm_args`` (include/speechclip_hip.h)."""
    _fields_ = [
        ("A", c_void_p), ("lda", c_i64),
        ("W", c_void_p), ("ldw", c_i64),
        ("C", c_void_p), ("ldc", c_i64),
        ("M", c_int), ("N", c_int), ("K", c_int),
        ("bias", c_void_p),
        ("residual", c_void_p), ("ldr", c_i64),
        ("act", c_int), ("out_f32", c_int),
        ("Ct", c_void_p), ("n_split", c_int), ("R", c_int), ("dh", c_int),
        ("nb1", c_int), ("nb2", c_int),
        ("sA1", c_i64), ("sA2", c_i64), ("sW1", c_i64), ("sW2", c_i64), ("sC1", c_i64), ("sC2", c_i64),
        ("sBias1", c_i64), ("sBias2", c_i64), ("sR1", c_i64), ("sR2", c_i64),
        ("tile", c_int), ("reserved", c_int),
        ("drop_p", ctypes.c_float), ("drop_seed", ctypes.c_uint32),
        ("tap_c", c_int), ("ln_ns", c_int),
        ("ln_stats", c_void_p), ("ln_colsum", c_void_p), ("res_stats", c_void_p), ("res_gamma", c_void_p), ("res_beta", c_void_p),
        ("stats_out", c_void_p), ("res_ns", c_int), ("ln_eps", ctypes.c_float),
        ("tn", c_int), ("k_total", c_int), ("aux_mode", c_int), ("reserved3", c_int),
        ("seg_chunk", c_void_p),
    ]


class Segments(ctypes.Structure):
    """Mirror of ``sc_segments``: the ragged row layout (device tables row0 [B + 1], chunk [rows / 32][4])."""
    _fields_ = [("row0", c_void_p), ("chunk", c_void_p), ("B", c_int), ("rows", c_int), ("max_pitch", c_int), ("reserved", c_int)]


class RtGemmArgs(ctypes.Structure):
    """Mirror of ``sc_rt_gemm_args``."""
    _fields_ = [
        ("A", c_void_p), ("lda", c_i64), ("a_slice", c_i64), ("a_z", c_i64), ("a_kmajor", c_int), ("a_ns", c_int),
        ("a_bias", c_void_p), ("a_rowscale", c_void_p), ("a_group", c_int), ("a_nscale", c_int),
        ("B", c_void_p), ("ldb", c_i64), ("b_z", c_i64), ("b_kmajor", c_int), ("nbatch", c_int),
        ("C", c_void_p), ("ldc", c_i64), ("c_slice", c_i64), ("c_z", c_i64),
        ("U", c_void_p),
        ("M", c_int), ("N", c_int), ("K", c_int), ("S", c_int),
        ("alpha", ctypes.c_float), ("beta", ctypes.c_float),
        ("bias", c_void_p), ("bias_z", c_i64),
        ("act", c_int), ("drop_p", ctypes.c_float), ("drop_seed", ctypes.c_uint32), ("pad_", c_int),
        ("gb", c_void_p), ("gb_z", c_i64),
    ]


class RtLnArgs(ctypes.Structure):
    """Mirror of ``sc_rt_ln_args``."""
    _fields_ = [
        ("y", c_void_p), ("y_slice", c_i64), ("ns", c_int), ("rows", c_int),
        ("bias", c_void_p),
        ("drop_p", ctypes.c_float), ("drop_seed", ctypes.c_uint32),
        ("res", c_void_p), ("res_stride", c_i64),
        ("g1", c_void_p), ("b1", c_void_p), ("out1", c_void_p), ("xhat1", c_void_p), ("rstd1", c_void_p),
        ("g2", c_void_p), ("b2", c_void_p), ("out2", c_void_p), ("xhat2", c_void_p), ("rstd2", c_void_p),
        ("eps1", ctypes.c_float), ("eps2", ctypes.c_float), ("D", c_int), ("pad_", c_int),
    ]


class RtLnBwdArgs(ctypes.Structure):
    """Mirror of ``sc_rt_ln_bwd_args``."""
    _fields_ = [
        ("dy", c_void_p), ("dy_slice", c_i64), ("ns", c_int), ("rows", c_int),
        ("add", c_void_p),
        ("xhat", c_void_p), ("gamma", c_void_p), ("rstd", c_void_p),
        ("dx", c_void_p), ("dx_masked", c_void_p),
        ("drop_p", ctypes.c_float), ("drop_seed", ctypes.c_uint32),
        ("dgamma", c_void_p), ("dbeta", c_void_p),
        ("D", c_int), ("pad_", c_int),
    ]


class HubertLayerArgs(ctypes.Structure):
    """Mirror of ``sc_hubert_layer_args`` (include/speechclip_hip.h)."""
    _fields_ = [
        ("x", c_void_p), ("out", c_void_p), ("valid_len", c_void_p),
        ("B", c_int), ("R", c_int), ("T", c_int), ("D", c_int), ("F", c_int), ("H", c_int), ("pre_ln", c_int), ("reserved", c_int),
        ("qkv_w", c_void_p), ("o_w", c_void_p), ("fc1_w", c_void_p), ("fc2_w", c_void_p),
        ("qkv_b", c_void_p), ("o_b", c_void_p), ("fc1_b", c_void_p), ("fc2_b", c_void_p),
        ("ln1_g", c_void_p), ("ln1_b", c_void_p), ("ln2_g", c_void_p), ("ln2_b", c_void_p),
        ("eps", ctypes.c_float), ("p_attn", ctypes.c_float), ("p_res", ctypes.c_float),
        ("seed_attn", ctypes.c_uint32), ("seed_o", ctypes.c_uint32), ("seed_fc2", ctypes.c_uint32),
        ("qk", c_void_p), ("vt", c_void_p), ("ctx", c_void_p), ("pre", c_void_p), ("x1", c_void_p), ("ffn", c_void_p),
        ("fused_ln", c_int), ("x_ns", c_int), ("x_stats", c_void_p), ("x_ln_g", c_void_p), ("x_ln_b", c_void_p),
        ("qkv_colsum", c_void_p), ("fc1_colsum", c_void_p), ("stats1", c_void_p), ("out_stats", c_void_p),
        ("seg", ctypes.POINTER(Segments)), ("attn_work", c_void_p), ("n_attn_work", c_int), ("reserved2", c_int),
    ]


# name -> argtypes (all return int except sc_last_error); kept in one table so tests can check that the
# library exports every symbol the header declares.
SIGNATURES = {
    "sc_abi_version": [],
    "sc_is_diag_build": [],
    "sc_set_option": [c_int, c_int],
    "sc_hubert_layer_fwd": [ctypes.POINTER(HubertLayerArgs), c_void_p],
    "sc_gemm_bf16": [ctypes.POINTER(GemmArgs), c_void_p],
    "sc_attn_fwd_bf16": [c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int,
                         c_float, ctypes.c_uint32, c_void_p],
    "sc_attn_fwd_seg_bf16": [c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_i64, ctypes.POINTER(Segments), c_void_p, c_int, c_int, c_int,
                             c_float, c_void_p, c_int, c_float, ctypes.c_uint32, c_void_p],
    "sc_wav_prep_seg": [c_void_p, c_i64, c_void_p, c_void_p, ctypes.POINTER(Segments), c_int, c_int, c_int, c_void_p],
    "sc_conv0_stats_len": [c_void_p, c_i64, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p],
    "sc_wav_prep_crop": [c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_void_p],
    "sc_wav_prep_seg_crop": [c_void_p, c_i64, c_void_p, c_void_p, c_void_p, ctypes.POINTER(Segments), c_int, c_int, c_int, c_void_p],
    "sc_conv0_stats_len_crop": [c_void_p, c_i64, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p],
    "sc_conv0_gn_gelu_seg": [c_void_p, ctypes.POINTER(Segments), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "sc_conv0_ln_gelu_seg": [c_void_p, ctypes.POINTER(Segments), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_void_p],
    "sc_posconv_prep_seg": [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.POINTER(Segments), c_int, c_int, c_int, c_void_p],
    "sc_posconv_seg_bf16": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.POINTER(Segments), c_int, c_int, c_int, c_void_p],
    "sc_wsum_fwd_seg": [c_void_p, c_void_p, c_int, c_void_p, ctypes.POINTER(Segments), c_int, c_int, c_int, c_int, c_void_p],
    "sc_wsum_bwd_seg": [c_void_p, c_void_p, c_int, c_void_p, c_int, ctypes.POINTER(Segments), c_int, c_int, c_int, c_int, c_void_p],
    "sc_attn_bwd_bf16": [c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_void_p,
                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_int, c_int,
                         c_int, c_int, c_float, c_int, c_float, ctypes.c_uint32, c_void_p],
    "sc_attn_bwd_fused_bf16": [c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_int, c_int,
                               c_int, c_int, c_float, c_int, c_float, ctypes.c_uint32, c_void_p],
    "sc_head_transpose_bf16": [c_void_p, c_i64, c_void_p, c_int, c_int, c_int, c_void_p],
    "sc_len_mask_u8": [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p],
    "sc_attn32_fwd_bf16": [c_void_p, c_i64, c_void_p, c_i64, c_int, c_int, c_float, c_void_p],
    "sc_attn32_bwd_bf16": [c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_int, c_int, c_float, c_void_p],
    "sc_layernorm_bf16": [c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_int, c_float, c_int, c_void_p],
    "sc_wav_prep": [c_void_p, c_i64, c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_void_p],
    "sc_conv0_stats": [c_void_p, c_i64, c_int, c_int, c_int, c_void_p, c_void_p],
    "sc_conv0_finalize": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p],
    "sc_conv0_gn_gelu": [c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "sc_conv0_ln_gelu": [c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_int, c_void_p],
    "sc_conv0_gn_gelu_f32": [c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "sc_conv0_ln_gelu_f32": [c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_int, c_void_p],
    "sc_softmax_fwd_f32": [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_int, ctypes.c_float, c_void_p],
    "sc_conv0_ln_bwd": [c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p],
    "sc_posconv_prep": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "sc_dropout_mult_f32": [c_void_p, c_i64, c_float, ctypes.c_uint32, c_void_p],
    "sc_posconv_bf16": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "sc_wsum_fwd": [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "sc_wsum_bwd": [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "sc_wsum_lazy_fwd": [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float,
                         c_void_p],
    "sc_wsum_lazy_bwd": [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                         c_float, c_void_p],
    "sc_cls_scores": [c_void_p, c_void_p, c_i64, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "sc_cif_fwd_rows": [c_void_p, c_int, c_i64, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, ctypes.c_float, c_void_p],
    "sc_cif_bwd_rows": [c_void_p, c_int, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_i64, c_int, c_int, c_void_p, c_void_p, c_int, c_int,
                        c_int, c_int, ctypes.c_float, c_void_p],
    "sc_rows_zero_pad_bf16": [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "sc_cif_head_bwd_rows": [c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_i64, c_void_p, c_void_p, c_int, c_int, c_int, c_float,
                             ctypes.c_uint32, c_float, ctypes.c_uint32, c_void_p],
    "sc_cif_fwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, ctypes.c_float, c_void_p],
    "sc_cif_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, ctypes.c_float, c_void_p],
    "sc_cif_prepare": [c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_int, c_int, c_int, c_float, c_float, c_int, c_int, c_void_p, c_void_p,
                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "sc_cif_prepare_bwd": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_int,
                           c_void_p, c_void_p],
    "sc_cif_tail": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                    c_void_p],
    "sc_vq_prep_f32": [c_void_p, c_i64, c_int, c_int, c_float, c_void_p, c_i64, c_void_p, c_void_p],
    "sc_sgemm_mfma_f32": [c_void_p, c_i64, c_int, c_void_p, c_i64, c_int, c_void_p, c_i64, c_int, c_int, c_int, c_void_p, c_void_p],
    "sc_sgemm_mfma_f32_split": [c_void_p, c_i64, c_int, c_void_p, c_i64, c_int, c_void_p, c_i64, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p],
    "sc_sgemm_mfma_slices": [c_int, c_int, c_int],
    "sc_split3_bf16": [c_void_p, c_i64, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p],
    "sc_vq_rowstats": [c_void_p, c_i64, c_int, c_int, c_float, ctypes.POINTER(c_int), c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                       c_void_p],
    "sc_vq_perplexity": [c_void_p, c_i64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p],
    "sc_vq_gather_f32": [c_void_p, c_i64, c_void_p, c_void_p, c_i64, c_int, c_int, c_void_p],
    "sc_vq_onehot_f32": [c_void_p, c_void_p, c_i64, c_int, c_int, c_void_p],
    "sc_vq_soft_bwd": [c_void_p, c_i64, c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_float, c_void_p, c_i64, c_int, c_void_p],
    "sc_vq_norm_bwd_f32": [c_void_p, c_i64, c_void_p, c_void_p, c_i64, c_float, c_void_p, c_i64, c_int, c_int, c_void_p],
    "sc_bn_rows_fwd": [c_void_p, c_i64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_void_p, c_i64,
                       c_void_p, c_void_p, c_void_p],
    "sc_bn_rows_bwd": [c_void_p, c_i64, c_void_p, c_i64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_void_p,
                       c_void_p],
    "sc_softmax_fwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_int, ctypes.c_float, ctypes.c_float, ctypes.c_uint32, c_void_p],
    "sc_softmax_bwd": [c_void_p, c_void_p, c_void_p, c_i64, c_int, ctypes.c_float, ctypes.c_float, ctypes.c_uint32, c_void_p],
    "sc_gemm_stats_strips": [ctypes.POINTER(GemmArgs)],
    "sc_cls_pool_fwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "sc_cls_pool_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                        c_void_p, c_void_p],
    "sc_sgemm_f32": [c_void_p, c_i64, c_i64, c_void_p, c_i64, c_i64, c_void_p, c_i64, c_int, c_int, c_int, c_float, c_void_p, c_void_p],
    "sc_infonce_fwd": [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                       c_void_p, c_void_p, c_void_p],
    "sc_infonce_grad": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_void_p,
                        c_void_p, c_void_p],
    "sc_layernorm_bwd_drop_bf16": [c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_i64, c_int, c_float,
                                   c_void_p, c_void_p, c_int, c_void_p, c_i64, c_float, ctypes.c_uint32, c_void_p, c_void_p],
    "sc_wsum_share_bf16": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "sc_layernorm_bwd_bf16": [c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_i64, c_int, c_float,
                              c_void_p, c_void_p, c_int, c_void_p],
    "sc_conv0_gn_bwd": [c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                        c_float, c_void_p, c_int, c_void_p, c_void_p],
    "sc_conv_overlap_add_bf16": [c_void_p, c_void_p, c_i64, c_int, c_void_p],
    "sc_conv_overlap_add_act_bf16": [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_int, c_void_p],
    "sc_posconv_wgrad_bf16": [c_void_p, c_void_p, c_void_p, c_int, c_i64, c_int, c_int, c_int, c_void_p],
    "sc_transpose_bf16": [c_void_p, c_i64, c_void_p, c_i64, c_int, c_int, c_void_p, c_void_p],
    "sc_colsum_bf16": [c_void_p, c_i64, c_i64, c_int, c_void_p, c_int, c_void_p],
    "sc_transpose_batched_bf16": [c_void_p, c_i64, c_i64, c_void_p, c_i64, c_i64, c_int, c_int, c_int, c_void_p],
    "sc_cast_transpose_f32_bf16": [c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_int, c_int, c_void_p],
    "sc_dropout_bf16": [c_void_p, c_i64, c_void_p, c_i64, c_i64, c_int, c_float, ctypes.c_uint32, c_void_p],
    "sc_act_bf16": [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_void_p],
    "sc_sgemm_f32_ex": [c_void_p, c_i64, c_i64, c_i64, c_void_p, c_i64, c_i64, c_i64, c_void_p, c_i64, c_i64, c_int, c_int, c_int, c_int,
                        c_float, c_float, c_void_p, c_i64, c_void_p, c_i64, c_void_p],
    "sc_rowln_f32_fwd": [c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p],
    "sc_rowln_f32_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    "sc_gelu_f32": [c_void_p, c_void_p, c_void_p, c_i64, c_void_p],
    "sc_colsum_f32": [c_void_p, c_i64, c_int, c_int, c_void_p, c_float, c_float, c_void_p],
    "sc_headmask_f32": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "sc_prompt_assemble": [c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "sc_prompt_assemble_bwd": [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "sc_rows_gather_bf16": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    "sc_rows_scatter_bf16": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "sc_cif_head_fwd": [c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, ctypes.c_uint32, c_float, ctypes.c_uint32, c_void_p],
    "sc_cif_head_bwd": [c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_int, c_int, c_int, c_float,
                        ctypes.c_uint32, c_float, ctypes.c_uint32, c_void_p],
    "sc_rt_gemm": [ctypes.POINTER(RtGemmArgs), c_void_p],
    "sc_rt_gemm_slices": [c_int, c_int, c_int, c_int],
    "sc_rt_ln_fwd": [ctypes.POINTER(RtLnArgs), c_void_p],
    "sc_rt_ln_bwd": [ctypes.POINTER(RtLnBwdArgs), c_void_p],
    "sc_rt_l2norm_fwd": [c_void_p, c_i64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    "sc_rt_elem": [c_void_p, c_i64, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_float, ctypes.c_uint32,
                   c_void_p],
    "sc_rt_value_bias_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "sc_rt_softmax_bwd_reduce": [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "sc_rt_l2norm_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    "sc_sumsq_f32": [c_void_p, c_i64, c_void_p, c_int, c_void_p],
    "sc_adam_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_float, c_float, c_float, c_float, c_float, c_int, c_void_p, c_int, c_float, c_void_p],
}

DIAG_LIB_PATH = os.path.join(_HERE, "csrc", "libspeechclip_hip_diag.so")
GELU_EXACT_LIB_PATH = os.path.join(_HERE, "csrc", "libspeechclip_hip_gelu_exact.so")   # checker build (build.py), tests only
_LIB = None
_DIAG = None
_VARIANTS = {}


def lib() -> ctypes.CDLL:
    """The product library."""
    global _LIB
    if _LIB is None:
        _LIB = _load(LIB_PATH)
    return _LIB


@contextlib.contextmanager
def using_library(path: str):
    """Tests / A-B tools only: route every op through another build of the SAME ABI (e.g. GELU_EXACT_LIB_PATH) inside the block, in one
    process, so two builds see identical inputs.  The libraries hold no state the ops share (weight copies live in torch tensors)."""
    global _LIB
    if path not in _VARIANTS:
        _VARIANTS[path] = _load(path)
    prev, _LIB = lib(), _VARIANTS[path]
    try:
        yield _LIB
    finally:
        _LIB = prev


def diag_lib() -> ctypes.CDLL:
    """libspeechclip_hip_diag.so: the same ABI built with SC_DIAG_BUILD - additionally the timing-only / stamped kernels (GEMM tile ids
    32 / 34, the attention section stamps) and the opt-in LayerNorm-folded GEMMs.  Loaded by tools/, bench.py's in-kernel clock probe
    and the SC_FUSED_LN experiment; the product path (lib()) cannot reach those kernels."""
    global _DIAG
    if _DIAG is None:
        _DIAG = _load(DIAG_LIB_PATH)
        if _DIAG.sc_is_diag_build() != 1:
            raise RuntimeError(f"{DIAG_LIB_PATH} is not a diagnostics build")
    return _DIAG


def _load(path: str) -> ctypes.CDLL:
    if True:
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} not found: the HIP extension is not built (python -m speechclip_plus_amd.build). "
                "speechclip_plus_amd has no CPU / eager fallback.")
        cdll = ctypes.CDLL(path)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(cdll, name)            # AttributeError if the symbol is missing: loud
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
        cdll.sc_last_error.argtypes = []
        cdll.sc_last_error.restype = ctypes.c_char_p
        cdll.sc_workspace_bytes.argtypes = [c_int, c_i64, c_i64, c_i64]
        cdll.sc_workspace_bytes.restype = ctypes.c_int64
        cdll.sc_infonce_workspace_floats.argtypes = [c_int]
        cdll.sc_infonce_workspace_floats.restype = ctypes.c_int64
        cdll.sc_sizeof.argtypes = [c_int]
        cdll.sc_sizeof.restype = ctypes.c_int64
        for what, cls in enumerate((GemmArgs, HubertLayerArgs, RtGemmArgs, RtLnArgs, RtLnBwdArgs, Segments)):
            if cdll.sc_sizeof(what) != ctypes.sizeof(cls):
                raise RuntimeError(f"{cls.__name__}: ctypes mirror has {ctypes.sizeof(cls)} bytes, the library's struct {cdll.sc_sizeof(what)} "
                                   "(include/speechclip_hip.h and _lib.py are out of step, or a stale .so)")
        cdll.sc_hash32.argtypes = [ctypes.c_uint32]
        cdll.sc_hash32.restype = ctypes.c_uint32
        return cdll


def check(rc: int, what: str = "", library=None) -> None:
    if rc != 0:
        raise RuntimeError(f"libspeechclip_hip {what} failed ({rc}): {(library or lib()).sc_last_error().decode()}")
