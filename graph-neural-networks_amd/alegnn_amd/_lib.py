"""ctypes binding of libgfhip.so (C ABI declared in include/gfhip.h).

There is NO fallback: if the HIP library is missing or a call fails, the product path raises.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgfhip.so")
if os.environ.get("GFHIP_EXPERIMENTS") == "1" and os.environ.get("GFHIP_LIB"):   # A/B builds of the library (Makefile `variant`)
    LIB_PATH = os.environ["GFHIP_LIB"]

GF_OK, GF_ERR_SHAPE = 0, -1
GF_OP_FWD, GF_OP_BWD = 0, 1

_lib = None

_c = ctypes
_vp, _i32, _i64, _sz = _c.c_void_p, _c.c_int32, _c.c_int64, _c.c_size_t
_SIGNATURES = {
    "gf_version": (_c.c_int, []),
    "gf_last_error": (_c.c_char_p, []),
    "gf_plan_create": (_c.c_int, [_i32, _i64, _vp, _vp, _vp, _i32, _c.c_uint32, _c.POINTER(_vp)]),
    "gf_plan_destroy": (_c.c_int, [_vp]),
    "gf_plan_info": (_c.c_int, [_vp, _c.POINTER(_i32), _c.POINTER(_i64), _c.POINTER(_i64)]),
    "gf_layout_bgn_to_bng": (_c.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "gf_layout_bng_to_bgn": (_c.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "gf_spmm_hop": (_c.c_int, [_vp, _i32, _vp, _vp, _i32, _i32, _vp]),
    "gf_khop": (_c.c_int, [_c.POINTER(_vp), _i32, _i32, _vp, _i32, _i32, _i32, _vp]),
    "gf_contract": (_c.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "gf_grad_taps_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32, _i32, _i32]),
    "gf_grad_taps": (_c.c_int, [_vp, _vp, _vp, _vp, _vp, _sz, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "gf_lsigf_forward": (_c.c_int, [_c.POINTER(_vp), _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "gf_lsigf_backward": (_c.c_int, [_c.POINTER(_vp), _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz,
                                     _i32, _i32, _i32, _i32, _i32, _vp]),
    "gf_lsigf_pipeline": (_c.c_int, [_c.POINTER(_vp), _i32, _i32, _i32, _i32]),
    "gf_khop_panel": (_c.c_int, [_c.POINTER(_vp), _i32, _i32, _vp, _i32, _i32, _i32, _vp]),
    "gf_khop_panel_uses_chain": (_c.c_int, [_vp, _i32, _i32]),
    "gf_time_khop_panel": (_c.c_int, [_c.POINTER(_vp), _i32, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _c.POINTER(_c.c_float)]),
    "gf_pack_panels": (_c.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "gf_unpack_panels": (_c.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "gf_spmm_hop_panel": (_c.c_int, [_vp, _i32, _vp, _vp, _i32, _vp]),
    "gf_time_spmm_hop_panel": (_c.c_int, [_vp, _i32, _vp, _vp, _i32, _i32, _vp, _c.POINTER(_c.c_float)]),
    "gf_contract_panel": (_c.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "gf_grad_taps_panel": (_c.c_int, [_vp, _vp, _vp, _vp, _vp, _sz, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "gf_plan_panel_info": (_c.c_int, [_vp, _i32, _c.POINTER(_i32), _c.POINTER(_i32), _c.POINTER(_c.c_double), _c.POINTER(_c.c_double)]),
    "gf_ev_plan_create": (_c.c_int, [_i32, _i64, _vp, _vp, _c.POINTER(_vp)]),
    "gf_ev_plan_destroy": (_c.c_int, [_vp]),
    "gf_ev_plan_info": (_c.c_int, [_vp, _c.POINTER(_i32), _c.POINTER(_i64), _c.POINTER(_i64)]),
    "gf_evgf_scratch_floats": (_sz, [_i32, _i32, _i32, _i32, _i32]),
    "gf_evgf_forward": (_c.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "gf_evgf_backward": (_c.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "gf_lsigf_forward_relu": (_c.c_int, [_c.POINTER(_vp), _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "gf_lsigf_backward_relu": (_c.c_int, [_c.POINTER(_vp), _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz,
                                          _i32, _i32, _i32, _i32, _i32, _vp]),
    "gf_db_hop": (_c.c_int, [_vp, _i64, _i64, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "gf_db_grad_gso": (_c.c_int, [_vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "gf_stack_adjoint": (_c.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp]),
    "gf_lsigf_db_forward": (_c.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "gf_lsigf_db_backward": (_c.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz,
                                        _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "gf_lsigf_forward_ex": (_c.c_int, [_c.POINTER(_vp), _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "gf_lsigf_backward_ex": (_c.c_int, [_c.POINTER(_vp), _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz,
                                        _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "gf_maxpool_forward": (_c.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "gf_maxpool_backward": (_c.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "gf_nvgf_scratch_floats": (_sz, [_i32, _i32, _i32, _i32, _i32, _i32, _i32]),
    "gf_nvgf_forward": (_c.c_int, [_c.POINTER(_vp), _i32, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i32, _i32, _i32, _i32, _i32, _vp]),
    "gf_nvgf_backward": (_c.c_int, [_c.POINTER(_vp), _i32, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i32, _i32, _i32, _i32, _i32, _vp]),
    "gf_nvgf_fold_taps": (_c.c_int, [_vp, _vp, _vp, _vp, _c.c_int64, _i32, _i32, _vp]),
    "gf_debug_msweep_trace": (_c.c_int, [_vp]),
    "gf_spmm_hop_kernel": (_c.c_int, [_vp, _i32, _i32, _i32]),
    "gf_debug_msweep_info": (_c.c_int, [_vp, _i32, _c.POINTER(_i32)]),
    "gf_msweep_status": (_c.c_int, [_c.POINTER(_c.c_uint32), _c.POINTER(_i32)]),
    "gf_time_khop": (_c.c_int, [_c.POINTER(_vp), _i32, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _c.POINTER(_c.c_float)]),
    "gf_time_spmm_hop": (_c.c_int, [_vp, _i32, _vp, _vp, _i32, _i32, _i32, _vp, _c.POINTER(_c.c_float)]),
    "gf_tune": (_c.c_int, [_c.c_char_p, _i32]),
}


def lib():
    """Load libgfhip.so once.  Raises RuntimeError (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the gfx950 HIP library has not been built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C graph-neural-networks_amd`). "
                "alegnn_amd has no CPU / eager fallback by design.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError here == header / library mismatch
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc: int, what: str = "") -> None:
    """0 -> ok.  GF_ERR_SHAPE -> AssertionError (what the reference raises for the same violation,
    graphML.py:135-140, 2118-2122); anything else -> RuntimeError."""
    if rc == GF_OK:
        return
    msg = lib().gf_last_error().decode("utf-8", "replace")
    if rc == GF_ERR_SHAPE:
        raise AssertionError(f"{what}: {msg}" if what else msg)
    raise RuntimeError(f"{what} failed with status {rc}: {msg}" if what else f"libgfhip status {rc}: {msg}")


def exported_symbols():
    return sorted(_SIGNATURES)
