"""ctypes binding of libclipenc_hip.so (C ABI: include/clipenc.h).

There is no CPU fallback: if the shared library is missing or fails to load, every product entry
point raises. Build it with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C clip_assisted_data_labeling_amd/csrc`.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_long, c_longlong, c_size_t, c_ulonglong, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CLIPENC_LIB_PATH", os.path.join(_HERE, "libclipenc_hip.so"))   # env: developer A/B builds

c_float_p = POINTER(c_float)
c_float_pp = POINTER(c_float_p)


class ClipencError(RuntimeError):
    pass


class clipenc_config(ctypes.Structure):
    _fields_ = [("image_size", c_int), ("patch", c_int), ("width", c_int), ("layers", c_int),
                ("heads", c_int), ("mlp_dim", c_int), ("embed_dim", c_int), ("act", c_int),
                ("ln_eps", c_float)]


class clipenc_weights(ctypes.Structure):
    _fields_ = [("conv1_weight", c_float_p), ("class_embedding", c_float_p), ("positional_embedding", c_float_p),
                ("ln_pre_w", c_float_p), ("ln_pre_b", c_float_p),
                ("ln_1_w", c_float_pp), ("ln_1_b", c_float_pp),
                ("in_proj_w", c_float_pp), ("in_proj_b", c_float_pp),
                ("out_proj_w", c_float_pp), ("out_proj_b", c_float_pp),
                ("ln_2_w", c_float_pp), ("ln_2_b", c_float_pp),
                ("c_fc_w", c_float_pp), ("c_fc_b", c_float_pp),
                ("c_proj_w", c_float_pp), ("c_proj_b", c_float_pp),
                ("ln_post_w", c_float_p), ("ln_post_b", c_float_p), ("proj", c_float_p)]


# name -> (restype, argtypes); mirrors include/clipenc.h one to one
SIGNATURES = {
    "clipenc_last_error": (c_char_p, []),
    "clipenc_device_count": (c_int, [POINTER(c_int)]),
    "clipenc_create": (c_int, [POINTER(clipenc_config), POINTER(clipenc_weights), c_int, POINTER(c_void_p)]),
    "clipenc_destroy": (c_int, [c_void_p]),
    "clipenc_set_chunk": (c_int, [c_void_p, c_int]),
    "clipenc_set_precision": (c_int, [c_void_p, c_int]),
    "clipenc_set_pixel_norm": (c_int, [c_void_p, c_float_p, c_float_p]),
    "clipenc_get_info": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_size_t)]),
    "clipenc_encode": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
    "fcreg_create": (c_int, [c_int, POINTER(c_int), c_float_pp, c_float_pp, c_float, c_int, POINTER(c_void_p)]),
    "fcreg_destroy": (c_int, [c_void_p]),
    "fcreg_forward": (c_int, [c_void_p, c_void_p, c_int, c_long, c_int, c_int, POINTER(c_int), c_void_p, c_void_p]),
    "clipenc_encode_score": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, POINTER(c_int), c_int,
                                     c_void_p, c_void_p, c_void_p]),
    "dedup_find_pairs": (c_int, [c_void_p, c_int, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p,
                                 c_ulonglong, c_void_p, c_void_p]),
    "dedup_screen_ws_bytes": (c_size_t, [c_int, c_int, ctypes.c_ulonglong]),
    "dedup_find_pairs_screened": (c_int, [c_void_p, c_int, c_int, c_float, c_int, c_void_p, c_void_p, c_size_t, ctypes.c_ulonglong, c_void_p,
                                          c_void_p, ctypes.c_ulonglong, c_void_p, c_void_p]),
    "dedup_tile_order": (c_int, [c_int, c_int, POINTER(ctypes.c_uint), c_long]),
    "preproc_create": (c_int, [c_int, POINTER(c_void_p)]),
    "preproc_destroy": (c_int, [c_void_p]),
    "preproc_crops_u8": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, POINTER(c_int), c_int, c_void_p, c_void_p]),
    "preproc_axis_tables": (c_int, [c_int, c_int, c_int, c_int, POINTER(c_int), POINTER(c_int), c_int, POINTER(c_int)]),
    "clipenc_profile_enable": (c_int, [c_void_p, c_int]),
    "clipenc_profile_kinds": (c_int, []),
    "clipenc_clock_probe": (c_int, [c_int, c_void_p, c_int, c_void_p]),
    "clipenc_mfma_stream_probe": (c_int, [c_int, c_int, c_void_p, c_void_p, ctypes.c_longlong, ctypes.POINTER(ctypes.c_double), c_void_p]),
    "clipenc_profile_read": (c_int, [c_void_p, c_int, POINTER(c_char_p), POINTER(ctypes.c_double), POINTER(c_longlong),
                                     POINTER(ctypes.c_double), c_int]),
    "clipenc_op_gemm_nt": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "clipenc_op_attention": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "fctrain_create": (c_int, [c_int, POINTER(c_int), c_float_pp, c_float_pp, c_float, c_int, POINTER(c_void_p)]),
    "fctrain_destroy": (c_int, [c_void_p]),
    "fctrain_epoch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_float, c_float, c_float, ctypes.c_uint,
                              c_void_p, c_void_p]),
    "fctrain_predict": (c_int, [c_void_p, c_void_p, c_long, c_void_p, c_void_p]),
    "fctrain_get_params": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "preproc_crops_u8_batch": (c_int, [c_void_p, c_int, POINTER(c_void_p), POINTER(c_int), POINTER(c_int), POINTER(c_int),
                                       POINTER(c_int), POINTER(c_int), c_int, c_void_p, c_void_p]),
    "simsearch_distances": (c_int, [c_void_p, c_int, c_long, c_int, c_long, c_void_p, c_int, c_void_p, c_void_p]),
    "simsearch_topn_workspace": (c_size_t, [c_long, c_int]),
    "simsearch_topn": (c_int, [c_void_p, c_long, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "diversity_workspace": (c_size_t, [c_long]),
    "diversity_order": (c_int, [c_void_p, c_long, c_int, c_long, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t,
                                c_void_p]),
    "clipenc_op_attention_q": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "clipenc_op_quant_rows_fp8": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "clipenc_op_gemm_fp8_q": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                      c_void_p, c_void_p, c_void_p]),
    "clipenc_op_gemm_fp8": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                    c_void_p, c_void_p, c_void_p]),
    "clipenc_op_quant_block_fp8": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "clipenc_op_row_norm_consts": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "clipenc_op_gemm_fp8_lnf": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "clipenc_op_gemm_fp8_resid_q": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_int, c_void_p]),
    "clipenc_forward_tokens": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "jpegdec_create": (c_int, [c_int, c_void_p]),
    "jpegdec_destroy": (c_int, [c_void_p]),
    "jpegdec_plan": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "jpegdec_run": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "jpegdec_reserve": (c_int, [c_void_p, ctypes.c_ulonglong, ctypes.c_ulonglong]),
    "jpegdec_reason": (ctypes.c_char_p, [c_int]),
    "jpegdec_probe": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
}

# include/clipenc_diag.h: only in libclipenc_hip_diag.so (`make diag`), bound when present (developer tools)
DIAG_SIGNATURES = {
    "clipenc_op_gemm_lnfold": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                       c_void_p, c_void_p, c_void_p]),
    "clipenc_op_gemm_nt_stamps": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "clipenc_op_gemm_nt_ld": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "clipenc_op_gemm_resid": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "clipenc_diag_fp8_stamps": (c_int, [c_void_p]),
}

_lib = None


def load() -> ctypes.CDLL:
    """Load the HIP library (once). Raises ClipencError — never falls back to a CPU path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ClipencError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run `make -C clip_assisted_data_labeling_amd/csrc`). There is no CPU fallback.")
    try:
        import torch  # noqa: F401  (loads torch's libamdhip64 first so both share one HIP runtime)
    except Exception:  # pragma: no cover
        pass
    try:
        lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    except OSError as e:
        raise ClipencError(f"could not load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library drift: fail loudly
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in DIAG_SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().clipenc_last_error()
        raise ClipencError(f"{what}: {msg.decode() if msg else 'unknown error'}")


def current_stream_ptr(device) -> int:
    import torch
    return int(torch.cuda.current_stream(device).cuda_stream)
