"""ctypes binding of libhh.so (C ABI declared in include/hh.h).  There is NO fallback: if the library is
missing or a call fails, a RuntimeError is raised (the product path never routes through CPU code)."""
import ctypes
import os

import torch  # noqa: F401  -- must be imported BEFORE libhh.so so that both share torch's HIP runtime (libamdhip64)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HH_LIBHH_PATH") or os.path.join(HERE, "libhh.so")      # override: A/B runs of two builds in one session

c_i64, c_int, c_float, c_vp = ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_void_p


class GemmEpilogue(ctypes.Structure):
    _fields_ = [("bias", c_vp), ("resid", c_vp), ("ldr", c_i64), ("colscale", c_float), ("colscale_cols", c_int),
                ("act", c_int), ("c_dtype", c_int), ("remap_group", c_i64), ("remap_skip", c_i64),
                ("remap_offset", c_i64), ("splitk", c_int), ("split_stride", c_i64), ("c_block_stride", c_i64),
                ("ln_stats", c_vp), ("ln_colsum", c_vp), ("z_resid", c_vp), ("z_ldr", c_i64), ("z_out", c_vp), ("z_ldc", c_i64),
                ("z_stats", c_vp), ("z_partials", c_vp), ("z_eps", c_float), ("skip_c", c_int), ("z_update", c_int), ("z_resid_dtype", c_int), ("z_resid_lo", c_vp), ("walk_reverse", c_int)]


class QGemmOpts(ctypes.Structure):
    _fields_ = [("a_scale", c_float), ("a_drop_p", c_float), ("a_drop_seed", ctypes.c_uint32), ("a_drop_ld", ctypes.c_int32),
                ("bias", c_vp), ("scale", c_float), ("scale_ncols", ctypes.c_int32), ("relu", ctypes.c_int32),
                ("drop_p", c_float), ("drop_seed", ctypes.c_uint32), ("relu_mask", c_vp), ("ldmask", c_i64), ("mask_scale", c_float),
                ("resid", c_vp), ("ldr", c_i64), ("colsum", c_vp), ("splitk", ctypes.c_int32),
                ("batch", ctypes.c_int32), ("stride_a", c_i64), ("stride_b", c_i64), ("stride_c", c_i64), ("stride_bias", c_i64),
                ("stride_colsum", c_i64), ("rowscale", c_vp), ("ld_rowscale", c_i64), ("stride_rowscale", c_i64)]


class QGemmItem(ctypes.Structure):
    """include/hh.h: hh_qgemm_item -- one product of hh_qgemm_f32x3_group (the argument list of hh_qgemm_f32x3)."""
    _fields_ = [("A", c_vp), ("lda", c_i64), ("B", c_vp), ("ldb", c_i64), ("C", c_vp), ("ldc", c_i64),
                ("M", ctypes.c_int32), ("N", ctypes.c_int32), ("K", ctypes.c_int32), ("mode", ctypes.c_int32), ("opts", QGemmOpts)]


QGEMM_GROUP_MAX = 12          # include/hh.h: HH_QGEMM_GROUP_MAX


# name -> argtypes (restype is int unless listed in _RESTYPES); must list every symbol of include/hh.h
SIGNATURES = {
    "hh_version": [],
    "hh_abi_sizeof": [ctypes.c_char_p],
    "hh_last_error_string": [],
    "hh_set_tuning": [ctypes.c_char_p, c_int],
    "hh_debug_gemm_timeline": [c_vp, c_int],
    "hh_stream_set_cu_budget": [c_vp, c_int],
    "hh_stream_get_cu_budget": [c_vp, ctypes.POINTER(c_int)],
    "hh_call_count": [],
    "hh_debug_space_redo_count": [c_int],
    "hh_prof_enable": [c_int],
    "hh_prof_kernel_name": [c_int],
    "hh_prof_read": [c_int, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)],
    "hh_prof_set_role": [c_int],
    "hh_prof_read_role": [c_int, c_int, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)],
    "hh_workspace_bytes_gemm_splitk": [c_i64, c_int, c_int],
    "hh_workspace_bytes_gemm_tn": [c_int, c_int, c_int],
    "hh_workspace_bytes_gemm_zstats": [c_i64, c_int],
    "hh_ln_rowstats": [c_vp, c_i64, c_vp, c_i64, c_int, c_float, c_vp],
    "hh_workspace_bytes_xattn_bwd": [c_int, c_int, c_int, c_int],
    "hh_workspace_bytes_xattn_fwd": [c_int, c_int, c_int, c_int],
    "hh_workspace_bytes_attn_cls_partial": [c_int, c_int, c_int, c_int, c_int],
    "hh_workspace_bytes_mattn_fwd": [c_int, c_int, c_int],
    "hh_workspace_bytes_mattn_bwd": [c_int, c_int, c_int],
    "hh_mattn_slices": [c_int, c_int],
    "hh_mattn_fwd": [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_float, ctypes.c_uint32, c_int, c_vp],
    "hh_mattn_bwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int,
                     c_int, c_int, c_float, ctypes.c_uint32, c_int, c_vp],
    "hh_gemm_tn_bf16_batched2": [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_i64, c_int, c_int, c_i64, c_int, c_vp],
    "hh_layernorm_fwd": [c_vp, c_int, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_i64, c_int, c_float, c_vp],
    "hh_add_layernorm_fwd": [c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_int, c_i64, c_int, c_float, c_vp],
    "hh_layernorm_bwd": [c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_vp],
    "hh_gemm_bf16": [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, c_int, c_int, ctypes.POINTER(GemmEpilogue), c_vp],
    "hh_sum_partials": [c_vp, c_vp, c_int, c_i64, c_vp],
    "hh_gemm_tn_bf16": [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_int, c_int, c_i64, c_int, c_vp],
    "hh_cast_f32_to_bf16": [c_vp, c_vp, c_i64, c_vp],
    "hh_cast_bf16_to_f32": [c_vp, c_vp, c_i64, c_vp],
    "hh_transpose_to_bf16": [c_vp, c_int, c_i64, c_vp, c_i64, c_i64, c_i64, c_vp],
    "hh_patch_im2col": [c_vp, c_vp, c_i64, c_int, c_int, c_int, c_int, c_vp],
    "hh_patch_im2col_u8": [c_vp, c_vp, c_i64, c_int, c_int, c_int, c_int, c_int, ctypes.POINTER(c_float), ctypes.POINTER(c_float), c_vp],
    "hh_embed_ln_pre": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_float, c_vp, c_vp, c_float, c_vp, c_vp],
    "hh_layernorm_split_cls_fwd": [c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_i64, c_int, c_int, c_float, c_vp, c_vp],
    "hh_space_attn_fwd": [c_vp, c_int, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp],
    "hh_time_attn_fwd": [c_vp, c_int, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp],
    "hh_cls_combine": [c_vp, c_int, c_vp, c_int, c_int, c_int, c_vp],
    "hh_cls_attn_fwd": [c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_vp],
    "hh_text_attn_fwd": [c_vp, c_vp, c_int, c_int, c_int, c_vp],
    "hh_xattn_fwd": [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_int, c_int, c_int, c_int, c_float, ctypes.c_uint32, c_vp],
    "hh_xattn_fwd_split": [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_float, ctypes.c_uint32, c_vp],
    "hh_xattn_bwd": [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_i64, c_int, c_int, c_int, c_int, c_float,
                     ctypes.c_uint32, c_vp],
    "hh_qgemm_f32x3_group": [ctypes.POINTER(QGemmItem), c_int, c_vp],
    "hh_qgemm_f32x3": [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_int, c_int, c_int, c_int, ctypes.POINTER(QGemmOpts), c_vp],
    "hh_qself_attn_fwd": [c_vp, c_vp, c_int, c_int, c_int, c_float, ctypes.c_uint32, c_vp],
    "hh_qself_attn_bwd": [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_float, ctypes.c_uint32, c_vp],
    "hh_layernorm_pos_fwd": [c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_i64, c_int, c_float, c_vp],
    "hh_layernorm_bwd_add": [c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_vp],
    "hh_match_boxes": [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_int, c_float, c_float, c_float, c_vp, c_float, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp],
    "hh_lsap_rows": [c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_vp],
    "hh_box_loss_fwd": [c_vp, c_int, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp],
    "hh_box_loss_bwd": [c_vp, c_int, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp],
    "hh_box_tail_fwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_i64, c_i64, c_float, c_float, c_float, c_vp, c_vp, c_vp],
    "hh_box_loss_bwd_scaled": [c_vp, c_int, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp],
    "hh_text_flags": [c_vp, c_int, c_int, c_vp, c_vp, c_vp],
    "hh_rownorm_fwd": [c_vp, c_i64, c_vp, c_vp, c_int, c_int, c_float, c_vp],
    "hh_rownorm_bwd": [c_vp, c_vp, c_vp, c_i64, c_vp, c_int, c_int, c_float, c_vp],
    "hh_workspace_bytes_egonce": [c_int, c_int],
    "hh_egonce_fwd": [c_vp, c_i64, c_vp, c_vp, c_vp, c_int, c_int, c_float, c_float, c_vp, c_vp, c_vp, c_vp],
    "hh_masked_ce_fwd": [c_vp, c_i64, c_vp, c_vp, c_vp, c_int, c_int, c_float, c_float, c_vp, c_vp, c_vp],
    "hh_tv_accuracy": [c_vp, c_i64, c_vp, c_vp, c_vp, c_int, c_vp, c_vp],
    "hh_adamw_step": [c_vp, c_vp, c_vp, c_vp, c_i64, c_float, c_float, c_float, c_float, c_float, c_int, c_vp],
    "hh_adamw_arena_step": [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_float, c_float, c_float, c_float,
                            c_float, c_int, c_vp],
}
_RESTYPES = {"hh_last_error_string": ctypes.c_char_p, "hh_call_count": c_i64, "hh_debug_space_redo_count": c_i64, "hh_prof_kernel_name": ctypes.c_char_p, "hh_workspace_bytes_gemm_splitk": c_i64, "hh_workspace_bytes_gemm_tn": c_i64, "hh_workspace_bytes_gemm_zstats": c_i64,
             "hh_workspace_bytes_xattn_bwd": c_i64, "hh_workspace_bytes_xattn_fwd": c_i64, "hh_workspace_bytes_attn_cls_partial": c_i64, "hh_workspace_bytes_egonce": c_i64,
             "hh_workspace_bytes_mattn_fwd": c_i64, "hh_workspace_bytes_mattn_bwd": c_i64}

_lib = None


def lib():
    """Load libhh.so (raises if absent -- build it with `python -m helping_hand_for_egocentric_videos_amd.build`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libhh.so (HIP kernels) is not built: run `python -m helping_hand_for_egocentric_videos_amd.build`. "
                "There is no CPU fallback for the product path.")
        L = ctypes.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the symbol is missing
            fn.argtypes = args
            fn.restype = _RESTYPES.get(name, c_int)
        for cname, cls in ((b"hh_gemm_epilogue", GemmEpilogue), (b"hh_qgemm_opts", QGemmOpts), (b"hh_qgemm_item", QGemmItem)):
            if L.hh_abi_sizeof(cname) != ctypes.sizeof(cls):
                raise RuntimeError("libhh.so was built from another include/hh.h: sizeof(%s) is %d there, %d in this binding -- rebuild it "
                                   "(`python -m helping_hand_for_egocentric_videos_amd.build --force`)" % (cname.decode(), L.hh_abi_sizeof(cname), ctypes.sizeof(cls)))
        _lib = L
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().hh_last_error_string()
        raise RuntimeError("%s failed (status %d): %s" % (what, rc, msg.decode() if msg else "?"))
