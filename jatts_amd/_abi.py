"""ctypes binding of libjatts_hip.so (declared in include/jatts_hip.h).

There is NO CPU fallback: if the shared library is missing or a call fails this
module raises.  The library is built in-tree by ``jatts_amd.build.build()``
(``make -C jatts_amd/csrc``) into ``jatts_amd/lib/libjatts_hip.so``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("JATTS_HIP_LIB") or os.path.join(_HERE, "lib", "libjatts_hip.so")  # override: profiling builds only

F32, F16, F32S, F32E, F32E6 = 0, 1, 2, 3, 4      # F32S: f32 in HBM, split f16 hi/lo MFMA operands (jatts_hifigan_resunit only)
ABI_VERSION = 4      # JATTS_ABI_VERSION of include/jatts_hip.h (tests/test_abi_cpu.py compares the two)
ACT_NONE, ACT_RELU, ACT_TANH, ACT_SWISH, ACT_MISH, ACT_SNAKEBETA = 0, 1, 2, 3, 4, 5
PRE_NONE, PRE_LRELU = 0, 1
PAD_ZERO, PAD_REFLECT = 0, 1


class Ragged(C.Structure):
    _fields_ = [("cu_rows", C.c_void_p), ("n_seq", C.c_int32), ("max_len", C.c_int32),
                ("len_mul", C.c_int32), ("total_rows", C.c_int32), ("host_lens", C.c_void_p)]


class ConvDesc(C.Structure):
    _fields_ = [
        ("rg", Ragged), ("dtype", C.c_int32), ("n_in", C.c_int32), ("x", C.c_void_p * 3),
        ("ldx", C.c_int32), ("in_scale", C.c_float), ("pre_act", C.c_int32), ("pre_slope", C.c_float),
        ("w", C.c_void_p), ("c_in", C.c_int32), ("n_out", C.c_int32), ("k_w", C.c_int32),
        ("dil", C.c_int32), ("pad", C.c_int32), ("bias", C.c_void_p), ("act", C.c_int32),
        ("alpha", C.c_float), ("resid", C.c_void_p), ("ldr", C.c_int32), ("y", C.c_void_p),
        ("ldy", C.c_int32), ("y_is_f32", C.c_int32), ("y_transposed", C.c_int32), ("y_seq_col0", C.c_void_p),
        ("pad_mode", C.c_int32), ("variant", C.c_int32), ("w_inv", C.c_void_p), ("act_a", C.c_void_p), ("act_b", C.c_void_p),
        ("n_split", C.c_int32), ("ldy2", C.c_int32), ("y2", C.c_void_p), ("y2_seq_col0", C.c_void_p), ("w_layout", C.c_int32),
    ]


class ResUnitDesc(C.Structure):
    _fields_ = [
        ("rg", Ragged), ("dtype", C.c_int32), ("channels", C.c_int32), ("k_w", C.c_int32),
        ("dil", C.c_int32), ("slope", C.c_float), ("x", C.c_void_p), ("y", C.c_void_p),
        ("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p),
        ("add0", C.c_void_p), ("add1", C.c_void_p), ("out_scale", C.c_float),
        ("ws1", C.c_void_p), ("ws2", C.c_void_p), ("w_layout", C.c_int32),
    ]


class ResBlockDesc(C.Structure):
    _fields_ = [
        ("rg", Ragged), ("dtype", C.c_int32), ("channels", C.c_int32), ("k_w", C.c_int32), ("n_units", C.c_int32),
        ("dil", C.c_int32 * 3), ("slope", C.c_float), ("x", C.c_void_p), ("y", C.c_void_p),
        ("w1", C.c_void_p * 3), ("b1", C.c_void_p * 3), ("w2", C.c_void_p * 3), ("b2", C.c_void_p * 3),
        ("add0", C.c_void_p), ("add1", C.c_void_p), ("out_scale", C.c_float),
        ("ws1", C.c_void_p * 3), ("ws2", C.c_void_p * 3),
    ]


class RelAttnDesc(C.Structure):
    _fields_ = [
        ("rg", Ragged), ("dtype", C.c_int32), ("n_heads", C.c_int32), ("d_k", C.c_int32),
        ("q", C.c_void_p), ("ldq", C.c_int32), ("k", C.c_void_p), ("ldk", C.c_int32),
        ("vt", C.c_void_p), ("ldvt", C.c_int32), ("g", C.c_void_p), ("ldg", C.c_int32),
        ("ku", C.c_void_p), ("scale", C.c_float), ("out", C.c_void_p), ("ldo", C.c_int32),
        ("rel_mode", C.c_int32), ("rel_center", C.c_int32), ("vt_col0", C.c_void_p), ("kv_len", C.c_void_p),
    ]


# name -> (restype, argtypes); every symbol include/jatts_hip.h declares
PROTOTYPES = {
    "jatts_abi_version": (C.c_int, []),
    "jatts_last_error": (C.c_char_p, []),
    "jatts_device_info": (C.c_int, [C.c_char_p, C.c_int]),
    "jatts_conv1d": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    "jatts_conv_weight_index": (C.c_int64, [C.c_int32] * 5),
    "jatts_hifigan_resunit": (C.c_int, [C.POINTER(ResUnitDesc), C.c_void_p]),
    "jatts_unit_weight_index_k32": (C.c_int64, [C.c_int32] * 4),
    "jatts_hifigan_resblock": (C.c_int, [C.POINTER(ResBlockDesc), C.c_void_p]),
    "jatts_debug_trace": (C.c_int, [C.c_void_p, C.c_int64]),
    "jatts_mfma_probe": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "jatts_mfma_probe_flops": (C.c_double, [C.c_int32, C.c_int32, C.c_int32]),
    "jatts_set_workspace": (C.c_int, [C.c_void_p, C.c_int64]),
    "jatts_pack_conv_weight_split": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "jatts_bgemm": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                              C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                              C.c_int32, C.c_void_p]),
    "jatts_pcm16": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "jatts_alignment_logp": (C.c_int, [C.POINTER(Ragged), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                       C.c_void_p, C.c_int32, C.c_void_p]),
    "jatts_mas_viterbi": (C.c_int, [C.POINTER(Ragged), C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "jatts_hifigan_output": (C.c_int, [C.POINTER(Ragged), C.c_int32, C.POINTER(C.c_void_p), C.c_int32,
                                       C.c_float, C.c_float, C.c_int32, C.c_int32, C.c_void_p, C.c_float,
                                       C.c_void_p, C.c_void_p]),
    "jatts_relpos_attention": (C.c_int, [C.POINTER(RelAttnDesc), C.c_void_p]),
    "jatts_rowdot": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32,
                               C.c_void_p, C.c_void_p, C.c_void_p]),
    "jatts_embed_scale": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_float, C.c_void_p,
                                    C.c_int64, C.c_void_p, C.c_void_p]),
    "jatts_layernorm": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                                  C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]),
    "jatts_affine_cast": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int64,
                                    C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "jatts_affine_slice": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int64,
                                     C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "jatts_glu_dwconv_bn_swish": (C.c_int, [C.POINTER(Ragged), C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                            C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "jatts_predictor_head": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p,
                                       C.c_float, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]),
    "jatts_variance_embed_add": (C.c_int, [C.POINTER(Ragged), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_int32, C.c_void_p]),
    "jatts_gated_tanh_sigmoid": (C.c_int, [C.POINTER(Ragged), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_int32, C.c_void_p]),
    "jatts_groupnorm_mish": (C.c_int, [C.POINTER(Ragged), C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                                       C.c_int32, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "jatts_snakebeta": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                  C.c_void_p]),
    "jatts_l2_normalize": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int32,
                                     C.c_float, C.c_void_p]),
    "jatts_gaussian_sample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float,
                                        C.c_void_p]),
    "jatts_flip_channels": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "jatts_add_seq_vector": (C.c_int, [C.POINTER(Ragged), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "jatts_lr_durations": (C.c_int, [C.POINTER(Ragged), C.c_void_p, C.c_float, C.c_int32, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p]),
    "jatts_lr_gather": (C.c_int, [C.POINTER(Ragged), C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                  C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "jatts_frame_signal": (C.c_int, [C.POINTER(Ragged), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                     C.c_int32, C.c_void_p]),
    "jatts_power_spectrum": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p]),
    "jatts_fbank_post": (C.c_int, [C.POINTER(Ragged), C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_void_p,
                                   C.c_int32, C.c_void_p]),
    "jatts_seq_mean_std": (C.c_int, [C.POINTER(Ragged), C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                     C.c_void_p, C.c_int32, C.c_float, C.c_void_p]),
    "jatts_seq_affine_act": (C.c_int, [C.POINTER(Ragged), C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                       C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "jatts_se_scale_add": (C.c_int, [C.POINTER(Ragged), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                     C.c_void_p]),
    "jatts_masked_loss": (C.c_int, [C.POINTER(Ragged), C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                    C.c_int32, C.c_float, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "jatts_conv1d_wgrad": (C.c_int, [C.POINTER(Ragged), C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "jatts_pack_conv_weight": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "jatts_col_sum": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "jatts_layernorm_bwd": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "jatts_act_fwd": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "jatts_act_bwd": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "jatts_glu_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "jatts_glu_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "jatts_dwconv": (C.c_int, [C.POINTER(Ragged), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "jatts_dwconv_wgrad": (C.c_int, [C.POINTER(Ragged), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "jatts_col_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "jatts_bn_bwd_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "jatts_index_add_rows": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "jatts_lr_segment_sum": (C.c_int, [C.POINTER(Ragged), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "jatts_shift_softmax_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_float, C.c_int32, C.c_void_p, C.c_void_p]),
    "jatts_shift_softmax_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "jatts_gate_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "jatts_weight_norm_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "jatts_weight_norm_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "jatts_split_add": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "jatts_concat2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "jatts_outer_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "jatts_col_wsum": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "jatts_row_dot": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "jatts_masked_loss_bwd": (C.c_int, [C.POINTER(Ragged), C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "jatts_groupnorm_fwd": (C.c_int, [C.POINTER(Ragged), C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "jatts_groupnorm_bwd": (C.c_int, [C.POINTER(Ragged), C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "jatts_snakebeta_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "jatts_snakebeta_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "jatts_ctc_forward_sum": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]),
    "jatts_seq_sum": (C.c_int, [C.POINTER(Ragged), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "jatts_dropout": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_uint64, C.c_void_p, C.c_void_p]),
    "jatts_qkv_split": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "jatts_qkv_split_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "jatts_act_dropout": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_uint64, C.c_void_p, C.c_void_p]),
    "jatts_dropout_add": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_uint64, C.c_void_p, C.c_void_p]),
    "jatts_sumsq": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "jatts_gather_grads": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "jatts_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int64, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    "jatts_zero_pad_rows": (C.c_int, [C.POINTER(Ragged), C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "jatts_cfm_mix": (C.c_int, [C.POINTER(Ragged), C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int32, C.c_void_p,
                                C.c_void_p, C.c_void_p]),
    "jatts_sq_err_sum": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "jatts_gaussian_upsample": (C.c_int, [C.POINTER(Ragged), C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                          C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
}

_lib = None


class JattsHipError(RuntimeError):
    pass


def load():
    """Load libjatts_hip.so (raises if absent: there is no fallback path)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise JattsHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C jatts_amd/csrc`.  jatts_amd has no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the .so misses a declared symbol
        fn.restype, fn.argtypes = res, args
    if lib.jatts_abi_version() != ABI_VERSION:
        raise JattsHipError(f"libjatts_hip.so ABI version {lib.jatts_abi_version()} != {ABI_VERSION} (include/jatts_hip.h): rebuild with __graft_entry__.build()")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().jatts_last_error().decode(errors="replace")
        raise JattsHipError(f"{what} failed (rc={rc}): {msg}")
