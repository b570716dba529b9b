"""ctypes binding of libpfa_hip.so (include/pfa_hip.h).

The library is mandatory for anything that touches a GPU tensor: `lib()` raises if it is missing
or fails to load - there is no eager-PyTorch or CPU fallback for the HIP path.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_uint64, c_void_p

PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(PKG_DIR, "lib", "libpfa_hip.so")

F32, BF16 = 0, 1
ACT_NONE, ACT_LRELU, ACT_RELU, ACT_TANH = 0, 1, 2, 3
CONV_S2, CONV_S1, TCONV_S2, CONV_S1_DGRAD = 0, 1, 2, 3


class GatherArgs(Structure):
    _fields_ = [
        ("struct_size", c_int32), ("mode", c_int32), ("dtype", c_int32),
        ("nb", c_int32), ("hq", c_int32), ("wq", c_int32), ("hs", c_int32), ("ws", c_int32),
        ("c1", c_int32), ("c2", c_int32), ("n", c_int32), ("n1", c_int32), ("ld1", c_int32), ("ld2", c_int32),
        ("act", c_int32), ("slope", c_float), ("splitk", c_int32), ("ntiles_stat", c_int32),
        ("src1", c_void_p), ("src2", c_void_p), ("w", c_void_p), ("bias", c_void_p),
        ("out1", c_void_p), ("out2", c_void_p), ("psum", c_void_p), ("psq", c_void_p), ("slab", c_void_p),
        ("fuse_out", c_int32), ("fuse_act1", c_int32), ("fuse_act2", c_int32), ("c_real", c_int32),
        ("fuse_raw", c_void_p), ("fuse_g2", c_void_p), ("fuse_scale", c_void_p), ("fuse_shift", c_void_p),
        ("fuse_mean", c_void_p), ("fuse_invstd", c_void_p),
    ]


class WgradArgs(Structure):
    _fields_ = [
        ("struct_size", c_int32), ("dtype", c_int32), ("stride", c_int32),
        ("nb", c_int32), ("hq", c_int32), ("wq", c_int32), ("hg", c_int32), ("wg", c_int32),
        ("d1p", c_int32), ("d1r", c_int32), ("d2p", c_int32), ("d2r", c_int32),
        ("g1p", c_int32), ("g1r", c_int32), ("g2p", c_int32), ("g2r", c_int32),
        ("msplit", c_int32), ("reserved", c_int32),
        ("dir1", c_void_p), ("dir2", c_void_p), ("gat1", c_void_p), ("gat2", c_void_p), ("dw", c_void_p),
        ("slab", c_void_p),
    ]


class ScatterArgs(Structure):
    _fields_ = [
        ("struct_size", c_int32), ("dtype", c_int32), ("nb", c_int32), ("hs", c_int32), ("ws", c_int32),
        ("c1", c_int32), ("c2", c_int32), ("n", c_int32), ("ld", c_int32), ("act", c_int32), ("slope", c_float),
        ("reserved", c_int32),
        ("src1", c_void_p), ("src2", c_void_p), ("w", c_void_p), ("bias", c_void_p), ("out", c_void_p),
    ]


class ShadowDesc(Structure):
    _fields_ = [("master", c_void_p), ("same", c_void_p), ("tr", c_void_p),
                ("rows", c_int32), ("c1r", c_int32), ("c1p", c_int32), ("c2r", c_int32), ("c2p", c_int32),
                ("rows_p", c_int32), ("same_blk0", c_int32), ("tr_blk0", c_int32)]


# name -> (restype, argtypes); every symbol include/pfa_hip.h declares
PROTOTYPES = {
    "pfa_version": (c_int, []),
    "pfa_last_error_string": (c_char_p, []),
    "pfa_launch_count": (c_uint64, []),
    "pfa_set_knob": (c_int, [c_int, c_int]),
    "pfa_set_debug_buffer": (c_int, [c_void_p, c_size_t]),
    "pfa_last_gather_kernel": (c_char_p, []),
    "pfa_gather_timing": (c_int, [c_int]),
    "pfa_gather_timing_count": (c_int, []),
    "pfa_gather_timing_get": (c_int, [c_int, POINTER(c_float), POINTER(c_char_p), POINTER(c_double), POINTER(c_double),
                                      POINTER(c_int)]),
    "pfa_gather_stat_tiles": (c_int, [POINTER(GatherArgs)]),
    "pfa_gather_slab_floats": (c_size_t, [POINTER(GatherArgs)]),
    "pfa_gather_gemm": (c_int, [POINTER(GatherArgs), c_void_p]),
    "pfa_splitk_finish": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_float, c_int, c_void_p,
                                  c_int, c_void_p, c_int, c_void_p, c_void_p, POINTER(c_int), c_void_p]),
    "pfa_wgrad": (c_int, [POINTER(WgradArgs), c_void_p]),
    "pfa_tconv_scatter_supported": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "pfa_tconv_scatter": (c_int, [POINTER(ScatterArgs), c_void_p]),
    "pfa_wgrad_slab_floats": (c_size_t, [POINTER(WgradArgs)]),
    "pfa_wgrad_slices": (c_int, [POINTER(WgradArgs)]),
    "pfa_logit_supported": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "pfa_logit_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "pfa_logit_dgrad_tiles": (c_int, [c_int, c_int, c_int]),
    "pfa_logit_dgrad": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float,
                                c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "pfa_logit_wgrad": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "pfa_bn_finalize": (c_int, [c_void_p, c_void_p, c_int, c_int, c_double, c_void_p, c_void_p, c_float, c_float,
                                c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pfa_bn_eval_coeffs": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_void_p, c_void_p, c_void_p]),
    "pfa_bn_act": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p]),
    "pfa_bn_bwd_reduce": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_int, c_void_p, c_int, c_float, c_void_p, c_void_p, POINTER(c_int), c_void_p]),
    "pfa_bn_bwd_tiles": (c_int, [c_int64, c_int, c_int]),
    "pfa_bn_bwd_finalize": (c_int, [c_void_p, c_int, c_int, c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p]),
    "pfa_bn_bwd_apply": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "pfa_bce_logits": (c_int, [c_void_p, c_int64, c_int, c_int, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    "pfa_l1_tanh_bwd": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_int, c_int, c_float,
                                c_void_p, c_void_p, c_void_p]),
    "pfa_colsum_ws": (c_size_t, [c_int64, c_int]),
    "pfa_colsum": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "pfa_adam": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_int,
                         c_void_p, c_float, c_void_p]),
    "pfa_adam_tick": (c_int, [c_void_p, c_void_p]),
    "pfa_weight_shadow": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "pfa_shadow_blocks": (c_int, [c_int, c_int, c_int, c_int, POINTER(c_int), POINTER(c_int)]),
    "pfa_weight_shadow_batch": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "pfa_pack_nhwc": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pfa_unpack_nchw": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pfa_unpack_u8": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_float, c_float, c_void_p, c_void_p]),
    "pfa_fill_zero": (c_int, [c_void_p, c_size_t, c_void_p]),
}

_LIB = None


class PfaError(RuntimeError):
    pass


def lib():
    """The loaded library.  Raises if libpfa_hip.so has not been built (run __graft_entry__.build())."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise PfaError(f"{LIB_PATH} is missing: the HIP extension is mandatory, build it with "
                           f"`make -C {PKG_DIR}` (or __graft_entry__.build()); there is no fallback path")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)   # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = handle
        for item in filter(None, os.environ.get("PFA_KNOBS", "").split(",")):   # e.g. PFA_KNOBS=0=5 (A/B runs)
            k, v = item.split("=")
            handle.pfa_set_knob(int(k), int(v))
    return _LIB


def check(rc, what=""):
    if rc != 0:
        msg = lib().pfa_last_error_string().decode(errors="replace")
        raise PfaError(f"{what or 'libpfa_hip'} failed (rc={rc}): {msg}")


def launch_count():
    return int(lib().pfa_launch_count())
