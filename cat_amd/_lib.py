"""ctypes loader for the in-tree libcat_amd.so (built by `make` / __graft_entry__.build()).

Fails loudly when the HIP library is missing: there is no CPU fallback in the product path.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libcat_amd.so")

u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)
vp = C.c_void_p

# name -> (restype, argtypes); mirrors include/cat_amd.h one to one
SIGNATURES = {
    "cat_version": (C.c_char_p, []),
    "cat_last_error": (C.c_char_p, []),
    "cat_ctx_create": (C.c_int, [C.c_uint32, u64p, C.c_uint32, C.c_int, C.c_uint64, C.c_int, C.POINTER(vp)]),
    "cat_ctx_destroy": (C.c_int, [vp]),
    "cat_ctx_reserve": (C.c_int, [vp, C.c_size_t]),
    "cat_ctx_workspace_need": (C.c_size_t, [vp, C.c_uint32]),
    "cat_ctx_workspace_bytes": (C.c_size_t, [vp]),
    "cat_ctx_workspace_grew": (C.c_uint64, [vp]),
    "cat_ctx_set_pool": (C.c_int, [vp, C.c_size_t]),
    "cat_ctx_set_workspace_strict": (C.c_int, [vp, C.c_int]),
    "cat_ctx_log_n": (C.c_uint32, [vp]),
    "cat_ctx_n_primes": (C.c_uint32, [vp]),
    "cat_ctx_root": (C.c_uint64, [vp, C.c_uint32]),
    "cat_ctx_n_moduli": (C.c_uint32, [vp]),
    "cat_ctx_modulus": (C.c_uint64, [vp, C.c_uint32]),
    "cat_malloc": (C.c_int, [vp, C.c_size_t, C.POINTER(vp)]),
    "cat_free": (C.c_int, [vp, vp]),
    "cat_upload": (C.c_int, [vp, vp, vp, C.c_size_t, vp]),
    "cat_download": (C.c_int, [vp, vp, vp, C.c_size_t, vp]),
    "cat_stream_sync": (C.c_int, [vp, vp]),
    "cat_copy": (C.c_int, [vp, vp, vp, C.c_size_t, vp]),
    "cat_ntt_forward": (C.c_int, [vp, vp, u32p, C.c_uint32, C.c_uint32, C.c_int, vp]),
    "cat_ntt_inverse": (C.c_int, [vp, vp, u32p, C.c_uint32, C.c_uint32, C.c_int, vp]),
    "cat_add": (C.c_int, [vp, vp, vp, vp, u32p, C.c_uint32, C.c_uint32, vp]),
    "cat_sub": (C.c_int, [vp, vp, vp, vp, u32p, C.c_uint32, C.c_uint32, vp]),
    "cat_negate": (C.c_int, [vp, vp, vp, u32p, C.c_uint32, C.c_uint32, vp]),
    "cat_dyadic": (C.c_int, [vp, vp, vp, vp, u32p, C.c_uint32, C.c_uint32, vp]),
    "cat_mul_scalar": (C.c_int, [vp, vp, u64p, vp, u32p, C.c_uint32, C.c_uint32, vp]),
    "cat_add_scalar": (C.c_int, [vp, vp, u64p, vp, u32p, C.c_uint32, C.c_uint32, vp]),
    "cat_sub_scalar": (C.c_int, [vp, vp, u64p, vp, u32p, C.c_uint32, C.c_uint32, vp]),
    "cat_modulo": (C.c_int, [vp, vp, vp, u32p, C.c_uint32, C.c_uint32, vp]),
    "cat_ckks_multiply": (C.c_int, [vp, vp, vp, vp, C.c_uint32, C.c_uint32, vp]),
    "cat_bgv_multiply": (C.c_int, [vp, vp, vp, vp, C.c_uint32, C.c_uint32, vp]),
    "cat_bgv_inv_q_last_mod_t": (C.c_uint64, [vp, C.c_uint32]),
    "cat_ckks_square": (C.c_int, [vp, vp, vp, C.c_uint32, C.c_uint32, vp]),
    "cat_multiply_plain_ntt": (C.c_int, [vp, vp, vp, C.c_size_t, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp]),
    "cat_add_plain": (C.c_int, [vp, vp, vp, C.c_size_t, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp]),
    "cat_sub_plain": (C.c_int, [vp, vp, vp, C.c_size_t, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp]),
    "cat_switch_key": (C.c_int, [vp, vp, vp, vp, C.c_uint32, C.c_uint32, vp]),
    "cat_key_broadcast": (C.c_int, [vp, vp, vp, C.c_size_t, C.c_int, C.c_size_t, vp]),
    "cat_result_gather": (C.c_int, [vp, vp, vp, C.c_size_t, vp, vp]),
    "cat_relinearize": (C.c_int, [vp, vp, vp, C.c_uint32, C.c_uint32, vp]),
    "cat_relinearize_rescale": (C.c_int, [vp, vp, vp, vp, C.c_uint32, C.c_uint32, vp]),
    "cat_rescale": (C.c_int, [vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp]),
    "cat_rescale_noalloc": (C.c_int, [vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp]),
    "cat_rescale_strided": (C.c_int, [vp, vp, C.c_size_t, vp, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, vp]),
    "cat_mod_drop": (C.c_int, [vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp]),
    "cat_galois_elt_from_step": (C.c_uint32, [C.c_uint32, C.c_int]),
    "cat_apply_galois_ntt": (C.c_int, [vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp]),
    "cat_apply_galois": (C.c_int, [vp, vp, vp, C.c_uint32, u32p, C.c_uint32, C.c_uint32, vp]),
    "cat_apply_galois_inplace": (C.c_int, [vp, vp, C.c_uint32, vp, C.c_uint32, C.c_uint32, vp]),
    "cat_apply_galois_rescale": (C.c_int, [vp, vp, C.c_uint32, vp, vp, C.c_uint32, C.c_uint32, vp]),
    "cat_rotate_hoisted": (C.c_int, [vp, vp, C.POINTER(C.c_uint32), C.POINTER(C.c_void_p), C.c_uint32,
                                     C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32, vp]),
    "cat_hoisted_key_prepare": (C.c_int, [vp, vp, C.c_uint32, vp, vp]),
    "cat_bfv_multiply": (C.c_int, [vp, vp, vp, vp, C.c_uint32, C.c_uint32, vp]),
    "cat_bfv_base_Bsk_size": (C.c_uint32, [vp, C.c_uint32]),
    "cat_bfv_base_Bsk": (C.c_int, [vp, C.c_uint32, u64p]),
    "cat_fastbconv_m_tilde": (C.c_int, [vp, vp, vp, C.c_uint32, C.c_uint32, vp]),
    "cat_sm_mrq": (C.c_int, [vp, vp, vp, C.c_uint32, C.c_uint32, vp]),
    "cat_fast_floor": (C.c_int, [vp, vp, vp, C.c_uint32, C.c_uint32, vp]),
    "cat_fastbconv_sk": (C.c_int, [vp, vp, vp, C.c_uint32, C.c_uint32, vp]),
}

_lib = None


def load() -> C.CDLL:
    """Loads libcat_amd.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build the HIP library first (make, or __graft_entry__.build()); "
                "cat_amd has no CPU fallback"
            )
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(lib, name)  # AttributeError = a declared symbol is not exported
            f.restype = res
            f.argtypes = args
        _lib = lib
    return _lib
