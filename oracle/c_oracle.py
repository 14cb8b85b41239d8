"""ctypes loader of oracle/liboracle.so (C restatement; test infrastructure only)."""
import ctypes as C
import pathlib
import subprocess

import numpy as np

_HERE = pathlib.Path(__file__).resolve().parent
_lib = None


def load(build=True):
    global _lib
    if _lib is None:
        so = _HERE / "liboracle.so"
        if not so.exists() and build:
            subprocess.run(["make", "-C", str(_HERE)], check=True, stdout=subprocess.DEVNULL)
        _lib = C.CDLL(str(so))
        _lib.oracle_normxcorr2.restype = C.c_int
        _lib.oracle_max_threads.restype = C.c_int
    return _lib


def max_threads():
    return int(load().oracle_max_threads())


def normxcorr2(signal, kernel, max_dist=None, sym_upper=False, full=False, miss_row=None,
               miss_col=None, missing_tol=0.75, kernel_conv=None, kernel_sq=None, n_threads=0):
    """float64 coefficient map and present-pixel counts; miss_row / miss_col are boolean
    vectors (None = no mask)."""
    lib = load()
    sig = np.ascontiguousarray(signal, dtype=np.float64)
    k = np.ascontiguousarray(kernel, dtype=np.float64)
    ms, ns = sig.shape
    km, kn = k.shape
    dp = C.POINTER(C.c_double)
    up = C.POINTER(C.c_uint8)
    masked = miss_row is not None
    mr = np.ascontiguousarray(miss_row, dtype=np.uint8) if masked else None
    mc = np.ascontiguousarray(miss_col, dtype=np.uint8) if masked else None
    kc = np.ascontiguousarray(kernel_conv, dtype=np.float64) if kernel_conv is not None else None
    k2 = np.ascontiguousarray(kernel_sq, dtype=np.float64) if kernel_sq is not None else None
    out = np.empty((ms, ns))
    nobs = np.empty((ms, ns))
    lib.oracle_normxcorr2(
        sig.ctypes.data_as(dp), C.c_int(ms), C.c_int(ns), k.ctypes.data_as(dp),
        kc.ctypes.data_as(dp) if kc is not None else None,
        k2.ctypes.data_as(dp) if k2 is not None else None,
        C.c_int(km), C.c_int(kn), C.c_int(int(full)), C.c_int(int(sym_upper)),
        C.c_int(-1 if max_dist is None else int(max_dist)), C.c_int(int(masked)),
        mr.ctypes.data_as(up) if masked else None, mc.ctypes.data_as(up) if masked else None,
        C.c_double(missing_tol), out.ctypes.data_as(dp), nobs.ctypes.data_as(dp), C.c_int(n_threads))
    return out, nobs


def _flags(miss_row, miss_col):
    up = C.POINTER(C.c_uint8)
    masked = miss_row is not None
    mr = np.ascontiguousarray(miss_row, dtype=np.uint8) if masked else None
    mc = np.ascontiguousarray(miss_col, dtype=np.uint8) if masked else None
    return masked, mr, mc, (mr.ctypes.data_as(up) if masked else None), (mc.ctypes.data_as(up) if masked else None)


def normxcorr2_rows(signal, kernel, r0, r1, max_dist=None, sym_upper=False, full=False, miss_row=None,
                    miss_col=None, missing_tol=0.75, n_threads=0):
    """Rows [r0, r1) of the coefficient map of a dense signal, plus the conditioning of every
    pixel (see oracle.c pixel()): full-size parity checks in bounded time."""
    lib = load()
    sig = np.ascontiguousarray(signal, dtype=np.float64)
    k = np.ascontiguousarray(kernel, dtype=np.float64)
    ms, ns = sig.shape
    dp = C.POINTER(C.c_double)
    masked, mr, mc, pr, pc = _flags(miss_row, miss_col)
    out = np.empty((r1 - r0, ns))
    cond = np.empty((r1 - r0, ns))
    lib.oracle_normxcorr2_rows(
        sig.ctypes.data_as(dp), C.c_int(ms), C.c_int(ns), k.ctypes.data_as(dp), C.c_int(k.shape[0]),
        C.c_int(k.shape[1]), C.c_int(int(full)), C.c_int(int(sym_upper)),
        C.c_int(-1 if max_dist is None else int(max_dist)), C.c_int(int(masked)), pr, pc, C.c_double(missing_tol),
        C.c_int(r0), C.c_int(r1), out.ctypes.data_as(dp), cond.ctypes.data_as(dp), C.c_int(n_threads))
    return out, cond


def normxcorr2_band(band, n, lo, width, kernel, r0, r1, out_lo, out_w, max_dist=None, sym_upper=True, full=True,
                    miss_row=None, miss_col=None, missing_tol=0.75, n_threads=0):
    """Rows [r0, r1) of the coefficient band (diagonals out_lo .. out_lo + out_w - 1) of an n x n map
    stored as a diagonal band (band[i, j - i - lo]); returns (corr, cond), both (r1 - r0, out_w)."""
    lib = load()
    b = np.ascontiguousarray(band, dtype=np.float64)
    assert b.ndim == 2 and b.shape[0] == n and b.shape[1] >= width
    k = np.ascontiguousarray(kernel, dtype=np.float64)
    dp = C.POINTER(C.c_double)
    masked, mr, mc, pr, pc = _flags(miss_row, miss_col)
    out = np.empty((r1 - r0, out_w))
    cond = np.empty((r1 - r0, out_w))
    lib.oracle_normxcorr2_band(
        b.ctypes.data_as(dp), C.c_int(n), C.c_longlong(b.shape[1]), C.c_int(lo), C.c_int(width),
        k.ctypes.data_as(dp), C.c_int(k.shape[0]), C.c_int(k.shape[1]), C.c_int(int(full)), C.c_int(int(sym_upper)),
        C.c_int(-1 if max_dist is None else int(max_dist)), C.c_int(int(masked)), pr, pc, C.c_double(missing_tol),
        C.c_int(r0), C.c_int(r1), C.c_int(out_lo), C.c_int(out_w), out.ctypes.data_as(dp), cond.ctypes.data_as(dp),
        C.c_int(n_threads))
    return out, cond
