"""ctypes binding of libesahrnet.so (include/esahrnet.h).  No torch types cross this boundary:
device buffers are passed as raw pointers (tensor.data_ptr()) and the stream as a hipStream_t."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libesahrnet.so")
MAX_BRANCHES = 4
ABI_VERSION = 5


class Cfg(C.Structure):
    _fields_ = [("cin", C.c_int32), ("num_keypoints", C.c_int32), ("stem_width", C.c_int32),
                ("widths", C.c_int32 * MAX_BRANCHES), ("blocks", (C.c_int32 * MAX_BRANCHES) * 4),
                ("modules", C.c_int32 * 4), ("final_conv_kernel", C.c_int32), ("variant", C.c_int32),
                ("precision", C.c_int32)]


class AuxDesc(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("shape", C.c_int32 * 4)]


class ConvDesc(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("bn", C.c_char * 96), ("cin", C.c_int32),
                ("cout", C.c_int32), ("k", C.c_int32), ("stride", C.c_int32),
                ("has_bias", C.c_int32), ("relu", C.c_int32)]


class OpDesc(C.Structure):
    _fields_ = [("kernel", C.c_char * 64), ("label", C.c_char * 96), ("flops", C.c_double),
                ("bytes", C.c_double)]


class EsaHrnetError(RuntimeError):
    pass


_lib = None

_SIGS = {
    "esahrnet_last_error": (C.c_char_p, []),
    "esahrnet_abi_version": (C.c_int, []),
    "esahrnet_create": (C.c_int, [C.POINTER(Cfg), C.c_int, C.POINTER(C.c_void_p)]),
    "esahrnet_destroy": (C.c_int, [C.c_void_p]),
    "esahrnet_handle_device": (C.c_int, [C.c_void_p]),
    "esahrnet_debug_devstate": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "esahrnet_debug_set_launch_limit": (C.c_int, [C.c_longlong]),
    "esahrnet_debug_op_schedule": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "esahrnet_conv_count": (C.c_int, [C.c_void_p]),
    "esahrnet_conv_desc_get": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(ConvDesc)]),
    "esahrnet_set_conv": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "esahrnet_aux_count": (C.c_int, [C.c_void_p]),
    "esahrnet_aux_desc_get": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(AuxDesc)]),
    "esahrnet_set_aux": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "esahrnet_commit": (C.c_int, [C.c_void_p]),
    "esahrnet_workspace_bytes": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "esahrnet_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                   C.c_void_p, C.c_size_t, C.c_void_p]),
    "esahrnet_keypoints": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "esahrnet_keypoints_ex": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
    "esahrnet_partial_tiles": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "esahrnet_forward_partials": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_size_t, C.c_void_p]),
    "esahrnet_keypoints_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                            C.c_void_p, C.c_void_p]),
    "esahrnet_crops": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_float,
                                 C.c_void_p, C.c_void_p]),
    "esahrnet_pnp_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "esahrnet_flops_per_crop": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "esahrnet_launch_count": (C.c_int, [C.c_void_p]),
    "esahrnet_op_desc_get": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(OpDesc)]),
    "esahrnet_forward_timed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_float)]),
    "esahrnet_tap_count": (C.c_int, [C.c_void_p]),
    "esahrnet_tap_name": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t]),
    "esahrnet_set_debug_keep": (C.c_int, [C.c_void_p, C.c_int]),
    "esahrnet_tap_shape": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_int),
                                     C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "esahrnet_tap_read": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                    C.c_void_p, C.c_void_p]),
    "esahrnet_op_conv": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "esahrnet_op_fuse": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int,
                                   C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "esahrnet_op_conv_ex": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "esahrnet_op_fuse_ex": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
}


def exported_symbols():
    """Every entry point include/esahrnet.h declares."""
    return sorted(_SIGS)


def lib():
    """Load libesahrnet.so (built in-tree by build.py).  There is NO fallback: without the HIP
    library the product path does not exist."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EsaHrnetError(
            f"{LIB_PATH} is missing — build it with `python __graft_entry__.py` "
            "(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback for this path.")
    # torch's ROCm wheel bundles its own libamdhip64.so.7; the process must have ONE HIP runtime and
    # it has to be the one torch's allocator/streams live in, so torch is loaded first and
    # libesahrnet.so's NEEDED libamdhip64.so.7 then resolves to the already-loaded copy.
    import torch  # noqa: F401
    l = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(l, name)           # AttributeError = header/library mismatch: fail loudly
        fn.restype, fn.argtypes = res, args
    if l.esahrnet_abi_version() != ABI_VERSION:
        raise EsaHrnetError("libesahrnet.so ABI version mismatch — rebuild it")
    _lib = l
    return l


def check(rc: int):
    if rc != 0:
        raise EsaHrnetError(lib().esahrnet_last_error().decode(errors="replace"))
