"""ctypes binding of libtethys_mi.so (the C ABI of include/tethys_mi.h).

The library is the product: there is no CPU or PyTorch fallback.  ``lib()`` raises if the
shared object is missing or does not export every symbol the header declares.
"""
from __future__ import annotations

import ctypes as C
import os

TMI_F32, TMI_BF16 = 0, 1
_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtethys_mi.so")

c_i64, c_i32, c_f32, c_vp = C.c_int64, C.c_int32, C.c_float, C.c_void_p


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", c_vp), ("B", c_vp), ("C", c_vp),
        ("M", c_i64), ("N", c_i64), ("K", c_i64),
        ("a_sm", c_i64), ("a_sk", c_i64), ("b_sk", c_i64), ("b_sn", c_i64), ("ldc", c_i64),
        ("nbatch", c_i64), ("a_sb", c_i64), ("b_sb", c_i64), ("c_sb", c_i64),
        ("kbatch", c_i64), ("a_skb", c_i64), ("b_skb", c_i64),
        ("bias", c_vp), ("bias_sb", c_i64),
        ("scale_cols", c_i64), ("scale", c_f32),
        ("accumulate", c_i32),
        ("act", c_i32),
        ("aux_out", c_vp), ("aux_in", c_vp),
        ("resid", c_vp), ("r_ld", c_i64), ("r_sb", c_i64),
        ("splitk", c_i32),
        ("in_dtype", c_i32), ("out_dtype", c_i32),
        ("workspace", c_vp), ("workspace_bytes", c_i64),
        ("dropout_p", c_f32), ("dropout_seed", C.c_uint64),
        ("nbatch2", c_i64), ("a_sb2", c_i64), ("b_sb2", c_i64), ("c_sb2", c_i64),
    ]


class AttnDesc(C.Structure):
    _fields_ = [
        ("q", c_vp), ("k", c_vp), ("v", c_vp), ("o", c_vp),
        ("q_sb", c_i64), ("q_st", c_i64), ("k_sb", c_i64), ("k_st", c_i64),
        ("v_sb", c_i64), ("v_st", c_i64), ("o_sb", c_i64), ("o_st", c_i64),
        ("stats", c_vp),
        ("B", c_i64), ("H", c_i64), ("Tq", c_i64), ("Tk", c_i64),
        ("mask_mode", c_i32),
        ("d_o", c_vp), ("dq", c_vp), ("dk", c_vp), ("dv", c_vp),
        ("do_sb", c_i64), ("do_st", c_i64), ("dq_sb", c_i64), ("dq_st", c_i64),
        ("dk_sb", c_i64), ("dk_st", c_i64), ("dv_sb", c_i64), ("dv_st", c_i64),
        ("delta", c_vp),
        ("dq_scale", c_f32),
        ("score_scale", c_f32),
        ("dropout_p", c_f32),
        ("dropout_seed", C.c_uint64),
        ("drop_mask", c_vp), ("drop_mask_bytes", c_i64),
        ("workspace", c_vp), ("workspace_bytes", c_i64),
        ("bwd_passes", c_i32),
    ]


# name -> (restype, argtypes); every symbol include/tethys_mi.h declares
SIGNATURES = {
    "tmi_abi_version": (c_i32, []),
    "tmi_last_error": (C.c_char_p, []),
    "tmi_set_deterministic": (c_i32, [c_i32]),
    "tmi_plan_create": (c_i32, [C.POINTER(c_vp)]),
    "tmi_plan_destroy": (c_i32, [c_vp]),
    "tmi_plan_begin": (c_i32, [c_vp]),
    "tmi_plan_end": (c_i32, [c_vp]),
    "tmi_plan_replay": (c_i32, [c_vp, C.c_uint64, c_i64]),
    "tmi_plan_size": (c_i64, [c_vp, c_i32]),
    "tmi_plan_note_event_record": (c_i32, [c_vp, c_vp]),
    "tmi_plan_note_stream_wait": (c_i32, [c_vp, c_vp]),
    "tmi_plan_note_callback": (c_i32, [c_vp]),
    "tmi_memset_async": (c_i32, [c_vp, c_i32, c_i64, c_vp]),
    "tmi_memset2d_async": (c_i32, [c_vp, c_i64, c_i32, c_i64, c_i64, c_vp]),
    "tmi_memcpy_async": (c_i32, [c_vp, c_vp, c_i64, c_vp]),
    "tmi_gemm": (c_i32, [C.POINTER(GemmDesc), c_vp]),
    "tmi_layernorm_fwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_f32, c_i32, c_vp]),
    "tmi_layernorm_dropout_fwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_f32, c_f32, C.c_uint64, c_i32, c_vp]),
    "tmi_layernorm_dropout_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i32, c_f32, C.c_uint64, c_vp, c_i64, c_i32, c_vp]),
    "tmi_layernorm_bwd_workspace_bytes": (c_i64, [c_i64, c_i64, c_i32]),
    "tmi_layernorm_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i32, c_vp, c_i64, c_i32, c_vp]),
    "tmi_colsum": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i64, c_i32, c_vp]),
    "tmi_colsum_batched": (c_i32, [c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_i64, c_i64, c_i32, c_vp]),
    "tmi_gelu_bwd": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_vp]),
    "tmi_dropout": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_f32, C.c_uint64, c_i32, c_vp]),
    "tmi_gelu_bwd_batched": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_i32, c_vp]),
    "tmi_softmax_fwd": (c_i32, [c_vp, c_i64, c_i64, c_i64, c_i32, c_vp]),
    "tmi_softmax_bwd": (c_i32, [c_vp, c_vp, c_i64, c_i64, c_vp]),
    "tmi_attn_workspace_bytes": (c_i64, [c_i64, c_i64, c_i64]),
    "tmi_attn_dropmask_bytes": (c_i64, [c_i64, c_i64, c_i64, c_i64]),
    "tmi_attn_fwd": (c_i32, [C.POINTER(AttnDesc), c_vp]),
    "tmi_attn_bwd": (c_i32, [C.POINTER(AttnDesc), c_vp]),
    "tmi_embed_fwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i32, c_i32, c_vp]),
    "tmi_embed_bwd": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i32, c_i32, c_vp]),
    "tmi_xent_fwd_bwd": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_i64, c_f32, c_i32, c_vp]),
    "tmi_linear_xent": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_i64, c_f32, c_i32, c_vp]),
    "tmi_sum_scale": (c_i32, [c_vp, c_vp, c_i64, c_f32, c_vp]),
    "tmi_adam_step": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32, c_i32, c_i32, c_f32, c_f32, c_vp, c_i32, c_i32, c_vp]),
    "tmi_adam_scalars": (c_i32, [c_f32, c_f32, c_f32, c_i32, c_i32, c_f32, c_vp]),
    "tmi_adam_step_segments": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32, c_f32, c_i32,
                                       c_i32, c_f32, c_f32, c_vp, c_i32, c_i32, c_vp]),
    "tmi_adam_step_rows": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_f32, c_f32, c_f32, c_f32, c_i32, c_i32, c_f32, c_f32, c_vp, c_i32, c_vp]),
    "tmi_adam_step_dev": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_vp, c_i32, c_f32, c_vp, c_vp]),
    "tmi_cast_bf16": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_vp]),
    "tmi_transpose_cast_bf16": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_vp]),
    "tmi_feat_to_channels_last": (c_i32, [c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_i32, c_vp]),
    "tmi_sumsq": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_vp]),
    "tmi_logmel_from_spectrum": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i64, c_i32, c_i32, c_f32, c_i32, c_i64, c_vp]),
    "tmi_groupnorm_chunks": (c_i64, [c_i64]),
    "tmi_groupnorm_gelu_fwd": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_f32, c_i32, c_vp]),
    "tmi_groupnorm_gelu_bwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i32, c_vp]),
    "tmi_group_pack": (c_i32, [c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i32, c_vp]),
    "tmi_group_unpack": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i32, c_vp]),
    "tmi_posconv_pack_weights": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i32, c_vp]),
    "tmi_vq_nearest": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i32, c_vp]),
    "tmi_vq_assign": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i32, c_vp]),
    "tmi_vq_bwd": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i32, c_vp]),
    "tmi_fir_chunks": (c_i64, [c_i64]),
    "tmi_fir_gn_workspace_floats": (c_i64, [c_i64, c_i64, c_i64]),
    "tmi_fir_groupnorm_gelu_fwd": (c_i32, [c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp,
                                           c_i64, c_i64, c_i64, c_i64, c_f32, c_i32, c_vp]),
    "tmi_fir_groupnorm_gelu_bwd": (c_i32, [c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp,
                                           c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i32, c_vp]),
    "tmi_layernorm_bwd_emit": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i32, c_vp, c_vp, c_f32,
                                       C.c_uint64, c_vp, c_i64, c_i32, c_vp]),
    "tmi_grad_pack": (c_i32, [c_vp, c_vp, c_i64, c_f32, c_vp]),
    "tmi_grad_unpack": (c_i32, [c_vp, c_i32, c_i64, c_i64, c_vp, c_i64, c_f32, c_vp]),
    "tmi_contrastive_fwd_bwd": (c_i32, [c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_i64, c_f32, c_f32, c_vp]),
    "tmi_segment_sumsq": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_vp]),
    "tmi_segment_sumsq_chunks": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp]),
    "tmi_segment_clip": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_f32, c_vp]),
    "tmi_loss_combine": (c_i32, [c_vp, c_vp, c_f32, c_f32, c_vp, c_vp]),
    "tmi_debug_gemm_stamps": (c_i32, [c_vp]),
}

ABI_VERSION = 26
_lib = None


class TmiError(RuntimeError):
    pass


def lib():
    """Load libtethys_mi.so once; fail loudly if it is absent or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TmiError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # PyTorch's wheel bundles its own libamdhip64: it must be in the process BEFORE this library is
    # loaded, or the dynamic linker resolves our DT_NEEDED entry to the system copy and the process
    # ends up with two HIP runtimes (ours then reports "no ROCm-capable device" on its first launch).
    import torch  # noqa: F401
    h = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(h, name)
        except AttributeError as e:
            raise TmiError(f"libtethys_mi.so does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if h.tmi_abi_version() != ABI_VERSION:
        raise TmiError(f"libtethys_mi.so ABI {h.tmi_abi_version()} != binding {ABI_VERSION}")
    _lib = h
    return h


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().tmi_last_error()
        raise TmiError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")
