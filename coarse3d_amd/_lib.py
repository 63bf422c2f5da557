"""ctypes binding of libcoarse3d_hip.so (C ABI in include/coarse3d_hip.h).

There is NO fallback: if the library is missing or a call is refused this raises.  The
library is built in-tree by ``__graft_entry__.build()`` / ``make -C coarse3d_amd/csrc``."""
import ctypes as C
import os

import torch  # noqa: F401  (loads the HIP runtime the library binds to)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcoarse3d_hip.so")


class Src(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p),
                ("C", C.c_int32), ("cstride", C.c_int32), ("coff", C.c_int32),
                ("lrelu", C.c_int32)]


class ConvDesc(C.Structure):
    _fields_ = [("src", Src * 3), ("nsrc", C.c_int32),
                ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cout", C.c_int32),
                ("ntaps", C.c_int32), ("tap_dy", C.c_int32 * 9), ("tap_dx", C.c_int32 * 9),
                ("wpack", C.c_void_p), ("bias", C.c_void_p), ("epi_lrelu", C.c_int32),
                ("out", C.c_void_p), ("out_cstride", C.c_int32), ("out_coff", C.c_int32),
                ("accumulate", C.c_int32), ("stat_partial", C.c_void_p)]


class WgradDesc(C.Structure):
    _fields_ = [("x", Src), ("dz", C.c_void_p), ("dz_cstride", C.c_int32),
                ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cout", C.c_int32),
                ("ntaps", C.c_int32), ("tap_dy", C.c_int32 * 9), ("tap_dx", C.c_int32 * 9),
                ("Cin_total", C.c_int32), ("cin_off", C.c_int32),
                ("dw", C.c_void_p), ("accumulate", C.c_int32), ("partial", C.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'` or "
                "`make -C coarse3d_amd/csrc`). There is no CPU fallback.")
        _lib = C.CDLL(LIB_PATH)
        _lib.c3d_last_error.restype = C.c_char_p
        _lib.c3d_wgrad_partial_floats.restype = C.c_int64
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed (rc={rc}): {lib().c3d_last_error().decode()}")


def exported_symbols():
    """Every entry point declared in include/coarse3d_hip.h (parsed from the header)."""
    import re
    hdr = os.path.join(os.path.dirname(_HERE), "include", "coarse3d_hip.h")
    txt = open(hdr).read()
    return sorted(set(re.findall(r"\b(c3d_[a-z0-9_]+)\s*\(", txt)))
