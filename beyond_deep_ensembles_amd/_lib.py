"""ctypes binding of libbde_hip.so (the C ABI declared in include/bde_hip.h).

The library is built in-tree (``beyond_deep_ensembles_amd/lib/libbde_hip.so``)
by ``__graft_entry__.build()`` / ``make -C beyond_deep_ensembles_amd/csrc``.
There is NO fallback: if the library is missing or a symbol cannot be bound,
loading raises, and every product code path that needs a kernel goes through
here.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_uint32, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libbde_hip.so")

_P = c_void_p  # every device pointer / stream travels as void*

# name -> (restype, argtypes); mirrors include/bde_hip.h one to one
SIGNATURES = {
    "bde_version": (c_int, []),
    "bde_arch": (c_char_p, []),
    "bde_init": (c_int, []),
    "bde_init_optional_failures": (c_int, []),
    "bde_svgd_ws_bytes": (c_size_t, [c_int]),
    "bde_svgd_kstat_floats": (c_size_t, [c_int]),
    "bde_svgd_gram": (c_int, [_P, c_int, c_int64, c_int64, _P, _P]),
    "bde_svgd_set_gram_keep_bytes": (c_int, [c_int64]),
    "bde_svgd_kstats": (c_int, [_P, c_int, c_float, c_float, c_float, c_float, c_float, c_int, _P, _P]),
    "bde_svgd_gram_finish": (c_int, [_P, c_int, _P, _P]),
    "bde_svgd_kstats_gmat": (c_int, [_P, c_int, c_int64, c_int, c_float, c_float, c_float, c_float, c_float, c_int, _P, _P]),
    "bde_svgd_combine": (c_int, [_P, _P, _P, c_int, c_int64, c_int64, c_int64, _P, _P]),
    "bde_svgd_step": (c_int, [_P, _P, _P, c_int, c_int64, c_int64, c_float, c_float, c_float, c_float, _P, _P, _P]),
    "bde_svgd_small_supported": (c_int, [c_int, c_int64]),
    "bde_svgd_step_small": (c_int, [_P, _P, _P, c_int, c_int64, c_int64, c_float, c_float, c_float, c_float, c_float,
                                    c_int, _P, _P, _P]),
    "bde_svgd_step_small_sgd": (c_int, [_P, _P, _P, c_int, c_int64, c_int64, c_float, c_float, c_float, c_double, c_double,
                                        c_double, c_double, c_int, c_int, _P, _P, _P]),
    "bde_svgd_step_small_adam": (c_int, [_P, _P, _P, _P, c_int, c_int64, c_int64, c_float, c_float, c_float, c_double,
                                         c_double, c_double, c_double, c_double, c_int64, _P, _P, _P]),
    "bde_svgd_apply_sgd": (c_int, [_P, _P, _P, c_int, c_int64, c_int64, c_double, c_double, c_double, c_double,
                                   c_int, c_int, _P]),
    "bde_svgd_apply_adam": (c_int, [_P, _P, _P, _P, c_int, c_int64, c_int64, c_double, c_double, c_double, c_double,
                                    c_double, c_int64, _P]),
    "bde_svgd_fused_gram_supported": (c_int, [c_int]),
    "bde_svgd_fused_sgd": (c_int, [_P, _P, _P, c_int, c_int64, c_int64, c_int64, _P, c_double, c_double, c_double, c_double,
                                   c_int, c_int, _P, _P]),
    "bde_svgd_fused_adam": (c_int, [_P, _P, _P, _P, c_int, c_int64, c_int64, c_int64, _P, c_double, c_double, c_double,
                                    c_double, c_double, c_int64, _P, _P]),
    "bde_svgd_combine_seg": (c_int, [_P, _P, _P, c_int64, _P, c_int, c_int64, c_int64, _P, _P]),
    "bde_svgd_fused_sgd_seg": (c_int, [_P, _P, _P, c_int64, _P, c_int, c_int64, c_int64, _P, c_double, c_double, c_double,
                                       c_double, c_int, c_int, _P, _P]),
    "bde_svgd_fused_adam_seg": (c_int, [_P, _P, _P, c_int64, _P, _P, c_int, c_int64, c_int64, _P, c_double, c_double,
                                        c_double, c_double, c_double, c_int64, _P, _P]),
    "bde_svgd_gather_seg": (c_int, [_P, _P, c_int64, _P, c_int, c_int, c_int, c_int64, _P]),
    "bde_sum_scalars": (c_int, [_P, c_int, _P, _P]),
    "bde_mean_scalars": (c_int, [_P, c_int, c_float, _P, _P]),
    "bde_swag_update": (c_int, [_P, _P, _P, _P, c_int64, c_int64, _P]),
    "bde_swag_sample": (c_int, [_P, _P, _P, c_int, c_int64, c_int, _P, _P, c_uint64, c_uint64, _P, c_int64, _P]),
    "bde_swag_sample_batched": (c_int, [_P, _P, _P, c_int, c_int64, c_int, _P, _P, c_int64, c_uint64, c_uint64, _P, c_int64,
                                        c_int, c_int64, _P]),
    "bde_swag_philox_rounds": (c_int, []),
    "bde_philox_normal": (c_int, [c_uint64, c_uint64, _P, c_int, _P, c_int64, c_int, _P]),
    "bde_philox_bits": (c_int, [c_uint64, c_uint64, c_uint32, c_uint64, _P, c_int64, c_int, _P]),
    "bde_gauss_draw_fwd": (c_int, [_P, _P, _P, c_uint64, c_uint64, _P, _P, c_int64, _P]),
    "bde_gauss_draw_bwd": (c_int, [_P, _P, _P, c_uint64, c_uint64, _P, _P, c_int, c_int64, _P]),
    "bde_reduce_ws_bytes": (c_size_t, []),
    "bde_gauss_kl": (c_int, [_P, _P, c_float, c_float, c_float, _P, _P, _P, c_int, _P, _P, c_int64, _P]),
    "bde_l2": (c_int, [_P, c_float, c_float, _P, _P, c_int, _P, _P, c_int64, _P]),
    "bde_mixture_nll": (c_int, [_P, c_float, c_float, c_float, c_float, _P, _P, c_int, _P, _P, c_int64, _P]),
    "bde_local_reparam_fwd": (c_int, [_P, _P, _P, c_uint64, c_uint64, _P, c_int64, _P]),
    "bde_local_reparam_bwd": (c_int, [_P, _P, _P, c_uint64, c_uint64, _P, c_int64, _P]),
    "bde_var_operand_fwd": (c_int, [_P, c_int, _P, c_int64, _P]),
    "bde_var_operand_bwd": (c_int, [_P, _P, c_int, _P, c_int64, _P]),
    "bde_conv_lrt_supported": (c_int, [c_int] * 11),
    "bde_conv_lrt_plan": (c_int, [c_int] * 12 + [_P]),
    "bde_conv_lrt_bwd_weight_plan": (c_int, [c_int] * 11 + [_P]),
    "bde_conv_lrt_pass_geos": (c_int, [c_int, _P, _P, c_int]),
    "bde_conv_lrt_candidates": (c_int, [_P, _P, c_int, _P]),
    "bde_conv_lrt_set_tiling": (c_int, [_P, c_int, c_int, c_int, c_int]),
    "bde_conv_lrt_wgrad_candidates": (c_int, [_P, _P, c_int, _P]),
    "bde_conv_lrt_wgrad_set_tiling": (c_int, [_P, c_int, c_int, c_int, c_int]),
    "bde_conv_lrt_prep_floats": (c_size_t, [c_int] * 4),
    "bde_conv_lrt_prep": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "bde_conv_lrt_prep_strided": (c_int, [_P, _P, _P] + [c_int] * 8 + [_P, _P]),
    "bde_conv_lrt_bwd_data_phases": (c_int, [_P, _P, _P, _P, _P] + [c_int] * 11 + [_P]),
    "bde_conv_lrt_fwd": (c_int, [_P, _P, _P, c_int, _P, c_uint64, c_uint64, _P, _P] + [c_int] * 11 + [_P]),
    "bde_conv_lrt_bwd_data": (c_int, [_P, _P, _P, _P, _P] + [c_int] * 11 + [_P]),
    "bde_conv_lrt_gvar_ws_bytes": (c_size_t, [c_int, c_int]),
    "bde_conv_lrt_gvar_bias": (c_int, [_P, _P, _P, c_uint64, c_uint64, _P, _P, _P, _P, _P, c_int, c_int, c_int64, _P]),
    "bde_conv_lrt_bwd_weight_ws_bytes": (c_size_t, [c_int] * 11),
    "bde_conv_lrt_bwd_weight": (c_int, [_P, _P, _P, _P, _P, c_size_t, _P, _P] + [c_int] * 11 + [_P]),
    "bde_lrt_linear_supported": (c_int, [c_int, c_int, c_int]),
    "bde_lrt_linear_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "bde_lrt_linear_fwd": (c_int, [_P, c_int64, _P, _P, _P, _P, _P, c_int, _P, c_uint64, c_uint64, _P, _P, c_int, c_int, c_int,
                                   _P, _P]),
    "bde_lrt_sigma_cache_wanted": (c_int, [c_int, c_int]),
    "bde_lrt_sigma_cache": (c_int, [_P, _P, _P, c_int64, _P]),
    "bde_lrt_linear_bwd_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "bde_lrt_linear_bwd": (c_int, [_P, c_int64, _P, _P, _P, _P, _P, c_int, _P, _P, _P, c_uint64, c_uint64, _P, _P, _P, _P, _P,
                                   c_int, c_int, c_int, _P, _P]),
    "bde_ivon_sample": (c_int, [_P, _P, _P, c_uint64, c_uint64, c_float, c_int, c_int, _P, _P, c_int64, _P]),
    "bde_ivon_update": (c_int, [_P, _P, _P, _P, _P] + [c_float] * 11 + [c_int64, _P]),
}

_lib = None
ABI_VERSION = 406        # csrc/version.hip, include/bde_hip.h


class BdeLibraryError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load (once) and return the bound library; raises BdeLibraryError if it
    is missing -- there is deliberately no CPU or PyTorch fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BdeLibraryError(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C beyond_deep_ensembles_amd/csrc)")
    # torch ships its own libamdhip64.so (same SONAME); import it first so the
    # HIP runtime our kernels register with is the one torch's streams live in.
    import torch  # noqa: F401
    try:
        lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    except OSError as e:  # pragma: no cover
        raise BdeLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise BdeLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    # the C signatures of existing exports have changed between ABI versions (symbol NAMES alone do not show a stale
    # build): the table above describes exactly one version
    have = int(lib.bde_version())
    if have != ABI_VERSION:
        raise BdeLibraryError(f"{LIB_PATH} reports ABI version {have}, these bindings are written for {ABI_VERSION}: "
                              "rebuild it (make -C beyond_deep_ensembles_amd/csrc)")
    _lib = lib
    return lib


def is_built() -> bool:
    return os.path.exists(LIB_PATH)
