"""ctypes binding of libcoarse3d_hip.so (C ABI in include/coarse3d_hip.h).

There is NO fallback: if the library is missing or a call is refused this raises.  The
library is built in-tree by ``__graft_entry__.build()`` / ``make -C coarse3d_amd/csrc``."""
import ctypes as C
import os

import torch  # noqa: F401  (loads the HIP runtime the library binds to)

_HERE = os.path.dirname(os.path.abspath(__file__))
# C3D_LIB: another build of the same library (same-box A/B of two kernel versions, tools/ab_env.sh)
LIB_PATH = os.environ.get("C3D_LIB") or os.path.join(_HERE, "libcoarse3d_hip.so")


class Src(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p),
                ("C", C.c_int32), ("cstride", C.c_int32), ("coff", C.c_int32),
                ("lrelu", C.c_int32), ("bf16", C.c_int32), ("reserved", C.c_int32)]


class ConvDesc(C.Structure):
    _fields_ = [("src", Src * 3), ("nsrc", C.c_int32),
                ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cout", C.c_int32),
                ("ntaps", C.c_int32), ("tap_dy", C.c_int32 * 9), ("tap_dx", C.c_int32 * 9),
                ("wpack", C.c_void_p), ("bias", C.c_void_p), ("epi_lrelu", C.c_int32),
                ("out", C.c_void_p), ("out_cstride", C.c_int32), ("out_coff", C.c_int32),
                ("accumulate", C.c_int32), ("stat_partial", C.c_void_p), ("lrelu_slope", C.c_float),
                ("mfma_bf16", C.c_int32), ("out_bf16", C.c_int32), ("wpack_planes", C.c_int32),
                ("stat_mul", C.c_void_p), ("stat_mul_cstride", C.c_int32), ("variant", C.c_int32),
                ("acc_scale_dev", C.c_void_p), ("stat_mul_bf16", C.c_int32), ("reserved", C.c_int32)]


class WgradDesc(C.Structure):
    _fields_ = [("x", Src), ("dz", C.c_void_p), ("dz_cstride", C.c_int32),
                ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cout", C.c_int32),
                ("ntaps", C.c_int32), ("tap_dy", C.c_int32 * 9), ("tap_dx", C.c_int32 * 9),
                ("Cin_total", C.c_int32), ("cin_off", C.c_int32),
                ("dw", C.c_void_p), ("accumulate", C.c_int32), ("partial", C.c_void_p), ("mfma_bf16", C.c_int32),
                ("lrelu_slope", C.c_float), ("dz_bf16", C.c_int32), ("variant", C.c_int32),
                ("bias_partial", C.c_void_p), ("dbias", C.c_void_p), ("bias_n", C.c_int32), ("reserved2", C.c_int32),
                ("dz_scale", C.c_void_p), ("out_scale_dev", C.c_void_p),
                ("fuse_dy", C.c_void_p), ("fuse_act", C.c_void_p), ("fuse_k1", C.c_void_p), ("fuse_k2", C.c_void_p),
                ("fuse_k3", C.c_void_p), ("fuse_sum", C.c_void_p), ("fuse_pre_scale", C.c_void_p), ("fuse_pre_shift", C.c_void_p),
                ("fold_out", C.c_void_p)]


class WgradFold(C.Structure):
    _fields_ = [("partial", C.c_void_p), ("dw", C.c_void_p),
                ("strips", C.c_int32), ("T", C.c_int32), ("CI", C.c_int32), ("CO", C.c_int32), ("ci_slices", C.c_int32),
                ("co_slices", C.c_int32), ("Cin_src", C.c_int32), ("Cout", C.c_int32), ("Cin_total", C.c_int32),
                ("cin_off", C.c_int32), ("accumulate", C.c_int32), ("main_blocks", C.c_int32), ("nblocks", C.c_int32),
                ("block0", C.c_int32), ("bias_partial", C.c_void_p), ("dbias", C.c_void_p), ("bias_n", C.c_int32),
                ("out_scale", C.c_float), ("out_scale_dev", C.c_void_p)]


class PackEntry(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("Cout", C.c_int32), ("Cin", C.c_int32),
                ("T", C.c_int32), ("mode", C.c_int32), ("c_off", C.c_int32), ("c_cnt", C.c_int32),
                ("Kpad", C.c_int32), ("reserved", C.c_int32)]


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
        for name, (res, args) in prototypes().items():
            try:
                fn = getattr(_lib, name)          # AttributeError = header/library mismatch
            except AttributeError:
                if os.environ.get("C3D_LIB"):     # an OLDER build loaded for a same-box A/B (tools/ab_lib.sh): newer entry points are absent
                    continue
                raise
            fn.restype, fn.argtypes = res, args
    return _lib


_CTYPES = {"int": C.c_int, "int32_t": C.c_int32, "int64_t": C.c_int64, "uint64_t": C.c_uint64, "float": C.c_float,
           "double": C.c_double, "c3d_stream": C.c_void_p, "void": None}


def prototypes():
    """{symbol: (restype, argtypes)} parsed from include/coarse3d_hip.h -- the header is the
    single source of truth for the ABI."""
    import re
    hdr = os.path.join(os.path.dirname(_HERE), "include", "coarse3d_hip.h")
    txt = re.sub(r"/\*.*?\*/", "", open(hdr).read(), flags=re.S)
    out = {}
    for m in re.finditer(r"([A-Za-z_0-9 ]+?[ \*]+)(c3d_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", txt):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        res = C.c_char_p if "char" in ret else _CTYPES[ret.replace("const", "").strip()]
        at = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    at.append(C.c_void_p)
                else:
                    at.append(_CTYPES[a.replace("const", "").split()[0]])
        out[name] = (res, at)
    return out


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed (rc={rc}): {lib().c3d_last_error().decode()}")


def exported_symbols():
    """Every entry point declared in include/coarse3d_hip.h."""
    return sorted(prototypes())
