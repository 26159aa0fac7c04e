"""ctypes binding of libmart_hip.so (the C ABI declared in include/mart_hip.h).

The library is built in-tree by ``mkg_analogy_amd._build`` / ``__graft_entry__.build()``.  There is NO fallback:
if the shared object is missing or a call fails, the product path raises.
"""
from __future__ import annotations

import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MART_HIP_LIB") or os.path.join(_PKG, "lib", "libmart_hip.so")   # MART_HIP_LIB: A/B builds of the same ABI (tools/ab_lib.sh)

vp, i32, i64, f32, u64, u8 = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_uint64, C.c_uint8


class MartError(RuntimeError):
    pass


class GemmNT(C.Structure):
    _fields_ = [("A", vp), ("B", vp), ("A2", vp), ("B2", vp), ("lda", i32), ("ldb", i32),
                ("M", i32), ("N", i32), ("K", i32), ("K2", i32), ("a_rows", vp), ("b_rows", vp),
                ("batch", i32), ("stride_a", i64), ("stride_b", i64), ("stride_c", i64), ("stride_aux", i64),
                ("bias", vp), ("bias2", vp), ("bias_by_brow", i32), ("act", i32), ("preact", vp),
                ("mulz", vp), ("mul_act", i32), ("res_f32", vp), ("res_bf16", vp), ("ldres", i32),
                ("alpha", f32), ("C", vp), ("ldc", i32), ("c_f32", i32), ("C2", vp), ("ldc2", i32), ("tile_cfg", i32), ("preact_grad", i32), ("b_blocked", i32), ("a_src_rows", i32), ("b_src_rows", i32),
                ("in_f16", i32), ("c_f16", i32), ("c_split3", i32),
                ("row_stats", vp), ("ln_mean", vp), ("ln_rstd", vp), ("ln_colsum", vp)]


class GemmTN(C.Structure):
    _fields_ = [("X", vp), ("Y", vp), ("ldx", i32), ("ldy", i32), ("M", i32), ("NX", i32), ("NY", i32),
                ("out", vp), ("ldo", i32), ("out_rows", vp), ("colsum", vp), ("colsum_by_row", i32),
                ("batch", i32), ("stride_x", i64), ("stride_y", i64), ("stride_o", i64), ("splits", i32), ("alpha", f32),
                ("workspace", vp), ("workspace_bytes", i64)]


class LnFwd(C.Structure):
    _fields_ = [("x_f32", vp), ("y_bf16", vp), ("p_drop", f32), ("seed", u64), ("gamma", vp), ("beta", vp), ("eps", f32),
                ("M", i32), ("H", i32), ("s_out", vp), ("out_f32", vp), ("out_bf16", vp), ("mean", vp), ("rstd", vp), ("y_f32", vp),
                ("out_f16", vp), ("x_rows", vp), ("out_split3", vp)]


class LnBwd(C.Structure):
    _fields_ = [("dy_f32", vp), ("dy_bf16", vp), ("s", vp), ("mean", vp), ("rstd", vp), ("gamma", vp), ("add_f32", vp),
                ("M", i32), ("H", i32), ("ds_f32", vp), ("ds_bf16", vp), ("p_drop", f32), ("seed", u64),
                ("dgamma", vp), ("dbeta", vp), ("bf16_total", i32), ("ws", vp), ("ws_bytes", i64), ("add2_f32", vp), ("defer_reduce", i32), ("add_bf16", vp)]


class TextEmbed(C.Structure):
    _fields_ = [("ids", vp), ("tt", vp), ("word", vp), ("pos", vp), ("type", vp), ("gamma", vp), ("beta", vp), ("eps", f32),
                ("p_drop", f32), ("seed", u64), ("B", i32), ("L", i32), ("H", i32), ("s_out", vp), ("mean", vp), ("rstd", vp),
                ("out_f32", vp), ("out_bf16", vp), ("out_f16", vp)]


class AttnFwd(C.Structure):
    _fields_ = [("q", vp), ("k", vp), ("v", vp), ("ldq", i32), ("ldk", i32), ("ldv", i32),
                ("pk", vp), ("pv", vp), ("ldp", i32), ("Lp", i32),
                ("B", i32), ("nh", i32), ("Sq", i32), ("Sk", i32), ("scale", f32),
                ("attn_mask", vp), ("sep", vp), ("sep_stride", i32), ("w0", vp), ("w1", vp),
                ("p_drop", f32), ("seed", u64), ("ctx", vp), ("ldctx", i32), ("lse", vp), ("rw_skip_row0", i32), ("ctx_f16", vp)]


class AttnF32(C.Structure):
    _fields_ = [("q", vp), ("k", vp), ("v", vp), ("ldq", i64), ("ldk", i64), ("ldv", i64),
                ("pk", vp), ("pv", vp), ("ldp", i64), ("Lp", i32),
                ("B", i32), ("nh", i32), ("D", i32), ("Sq", i32), ("Sk", i32), ("scale", f32),
                ("attn_mask", vp), ("sep", vp), ("sep_stride", i32), ("w0", vp), ("w1", vp), ("rw_skip_row0", i32),
                ("ctx", vp), ("ldctx", i64), ("fast", i32), ("ctx_split3", vp), ("ldctx3", i64)]


class AttnBwdF32(C.Structure):
    _fields_ = [("f", AttnF32), ("dctx", vp), ("lddctx", i64), ("dq", vp), ("dk", vp), ("dv", vp), ("lddq", i64), ("lddk", i64), ("lddv", i64),
                ("dpk", vp), ("dpv", vp), ("lddp", i64), ("dw", vp)]


class AttnBwd(C.Structure):
    _fields_ = [("f", AttnFwd), ("dctx", vp), ("lddctx", i32), ("delta", vp),
                ("dq", vp), ("dk", vp), ("dv", vp), ("lddq", i32), ("lddk", i32), ("lddv", i32),
                ("dpk", vp), ("dpv", vp), ("lddp", i32), ("accum_dkv", i32), ("dw", vp), ("dw_ws", vp)]


class FusionFwd(C.Structure):
    _fields_ = [("q", vp), ("ldq", i32), ("v", vp), ("ldv", i32), ("out", vp), ("ldo", i32), ("probs", vp), ("ldp", i32),
                ("B", i32), ("Lq", i32), ("Nv", i32), ("H", i32), ("out_f16", vp)]


class FusionBwd(C.Structure):
    _fields_ = [("q", vp), ("ldq", i32), ("v", vp), ("ldv", i32), ("dout", vp), ("lddo", i32), ("probs", vp), ("ldp", i32),
                ("dq", vp), ("lddq", i32), ("dv_f32", vp), ("lddv", i32), ("dv_bf16", vp), ("lddvb", i32),
                ("B", i32), ("Lq", i32), ("Nv", i32), ("H", i32)]


class AdamW(C.Structure):
    _fields_ = [("master", vp), ("grad", vp), ("m", vp), ("v", vp), ("shadow_bf16", vp), ("chunks", vp), ("n_chunks", i32),
                ("lr", f32), ("beta1", f32), ("beta2", f32), ("eps", f32), ("weight_decay", f32), ("bc1", f32), ("bc2", f32),
                ("grad_scale", f32), ("shadow_f16", vp), ("zero_grad", i32)]


_SIGS = {
    "mart_last_error": (C.c_char_p, []),
    "mart_abi_version": (i32, []),
    "mart_check_device": (i32, []),
    "mart_gemm_nt": (i32, [C.POINTER(GemmNT), vp]),
    "mart_gemm_tn": (i32, [C.POINTER(GemmTN), vp]),
    "mart_gemm_tn_workspace_bytes": (i64, [i32, i32, i32, i32]),
    "mart_ln_fwd": (i32, [C.POINTER(LnFwd), vp]),
    "mart_ln_bwd": (i32, [C.POINTER(LnBwd), vp]),
    "mart_ln_bwd_partials": (i32, [i32]),
    "mart_ln_dgb_reduce": (i32, [vp, i32, i32, vp, vp, vp]),
    "mart_ln_fold_prep": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]),
    "mart_ln_stats_finalize": (i32, [vp, i32, i32, f32, vp, vp, vp]),
    "mart_patchify": (i32, [vp, vp, i32, i32, i32, vp]),
    "mart_patchify_gather": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "mart_gather_images": (i32, [vp, vp, vp, i32, i32, vp]),
    "mart_vision_assemble": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "mart_vision_assemble_bwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "mart_text_embed_fwd": (i32, [C.POINTER(TextEmbed), vp]),
    "mart_dropout_bwd_f32": (i32, [vp, vp, vp, i64, f32, u64, vp]),
    "mart_text_embed_scatter": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "mart_text_embed_scatter_det": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, i64, vp, vp]),
    "mart_vision_assemble_bwd_det": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp, i64, vp]),
    "mart_attn_fwd": (i32, [C.POINTER(AttnFwd), vp]),
    "mart_attn_bwd": (i32, [C.POINTER(AttnBwd), vp]),
    "mart_fusion_supported": (i32, [i32, i32, i32]),
    "mart_fusion_fwd": (i32, [C.POINTER(FusionFwd), vp]),
    "mart_fusion_bwd": (i32, [C.POINTER(FusionBwd), vp]),
    "mart_softmax_fwd": (i32, [vp, i32, vp, i32, i32, i32, vp]),
    "mart_softmax_bwd": (i32, [vp, i32, vp, i32, vp, i32, i32, i32, vp]),
    "mart_transpose_bf16": (i32, [vp, i32, i64, vp, i32, i64, i32, i32, i32, vp]),
    "mart_lsce_fwd": (i32, [vp, i32, vp, i64, f32, vp, vp, vp, i32, vp, i32, i32, vp]),
    "mart_lsce_bwd": (i32, [vp, i32, vp, i64, vp, f32, vp, i32, f32, vp, vp, i32, vp, i32, i32, vp]),
    "mart_rank": (i32, [vp, i32, vp, vp, i32, i32, vp]),
    "mart_simloss_fwd": (i32, [vp, vp, vp, vp, vp, i32, vp, i32, i32, i32, vp]),
    "mart_simloss_bwd": (i32, [vp, vp, vp, vp, vp, i32, vp, f32, vp, i32, i32, i32, vp]),
    "mart_needed_rows": (i32, [vp, i32, i32, i64, vp, vp, vp, vp, vp, vp, vp]),
    "mart_rows_lookup": (i32, [vp, i32, vp, i32, i32, vp, vp, vp]),
    "mart_rows_dense": (i32, [vp, vp, i32, i32, i32, i32, vp, f32, vp]),
    "mart_sum_splits_f32": (i32, [vp, i32, i64, vp, vp]),
    "mart_find_token": (i32, [vp, i32, i32, i64, vp, vp, vp, vp]),
    "mart_cast_f32_bf16": (i32, [vp, vp, i64, vp]),
    "mart_cast_bf16_f32": (i32, [vp, vp, i64, vp]),
    "mart_cast_f32_f16": (i32, [vp, vp, i64, vp]),
    "mart_cast_bf16_f16": (i32, [vp, vp, i64, vp]),
    "mart_gather_rows_first_f32": (i32, [vp, i32, vp, i32, vp, i32, i32, vp]),
    "mart_scatter_rows": (i32, [vp, vp, i32, vp, i32, i32, i32, i32, i32, vp]),
    "mart_cast_pad_f32_bf16": (i32, [vp, i32, vp, i32, i32, i32, vp]),
    "mart_gather_rows_bf16": (i32, [vp, i32, vp, vp, i32, i32, vp]),
    "mart_act_bwd": (i32, [vp, vp, i32, vp, i64, vp]),
    "mart_gather_rows_f32": (i32, [vp, i32, vp, vp, i32, i32, vp]),
    "mart_scatter_add_rows_f32": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "mart_add_f32_bf16": (i32, [vp, vp, vp, vp, i64, vp]),
    "mart_dropout_mask": (i32, [vp, i64, f32, u64, vp]),
    "mart_adamw": (i32, [C.POINTER(AdamW), vp]),
    "mart_transpose_table": (i32, [vp, vp, vp, i32, vp]),
    "mart_block_table": (i32, [vp, vp, vp, i32, vp]),
    "mart_split_bf16x3": (i32, [vp, i64, vp, i32, i32, i32, i32, vp]),
    "mart_split_bf16x3_rows": (i32, [vp, i64, vp, vp, i32, i32, i32, i32, vp]),
    "mart_split_bf16x3_stack": (i32, [vp, i64, vp, i32, i32, i32, i32, vp]),
    "mart_act_f32": (i32, [vp, vp, i32, i64, vp]),
    "mart_act_bwd_f32": (i32, [vp, vp, i32, vp, i64, vp]),
    "mart_colsum_f32": (i32, [vp, i64, vp, i32, i32, vp]),
    "mart_attn_bwd_f32": (i32, [C.POINTER(AttnBwdF32), vp]),
    "mart_patchify_f32": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "mart_vision_assemble_f32": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "mart_attn_fwd_f32": (i32, [vp, vp]),
}

EXPORTS = tuple(_SIGS)
EXPECTED_ABI = 10           # the layout the ctypes structures above were written for (mart_abi_version() of the library must match)
_lib = None


def lib():
    """Load (once) and return the shared library.  Raises MartError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MartError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(the MKGformer path has no non-HIP fallback)")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        abi = int(l.mart_abi_version())
        if abi != EXPECTED_ABI:
            raise MartError(f"{LIB_PATH} has ABI {abi}, this package binds ABI {EXPECTED_ABI}: rebuild it "
                            "(`python -c 'import __graft_entry__ as g; g.build()'`)")
        _lib = l
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().mart_last_error()
        raise MartError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
