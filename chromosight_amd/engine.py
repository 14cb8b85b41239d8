"""Device-side plumbing between the reference-shaped Python API (chromosight_amd.utils) and
the C ABI: buffer staging, band geometry, and the calls themselves.  Everything here runs on
the GPU through libchromosight_hip.so; nothing falls back to the host.
"""
import ctypes as C
import os

import numpy as np
import scipy.sparse as sp

from . import _lib
from ._lib import (CS_F32, CS_F64, FOCUS_DTYPE, LAYOUT_BAND, LAYOUT_DENSE, MASK_BINS, MASK_EXPLICIT, MASK_NONE,
                   CsCsr, CsFociParams, CsKernel, CsMatrix, CsNormxcorr2Params, band_matrix, dense_matrix)

XCORR_THRESHOLD = 1e-4   # reference detection.py:595 (default `threshold` of xcorr2)
DENOM_EPS = 1e-10        # reference detection.py:1010, 1088
CS_U8 = 2

_precision = os.environ.get("CHROMOSIGHT_HIP_PRECISION", "f32")


def set_precision(mode):
    """'f32' (default): float32 arithmetic on the bulk map, float64 re-scoring of the pixels
    that decide detection.  'f64': float64 arithmetic everywhere."""
    global _precision
    if mode not in ("f32", "f64"):
        raise ValueError("precision must be 'f32' or 'f64'")
    _precision = mode


def get_precision():
    return _precision


def compute_code(precision=None):
    return CS_F64 if (precision or _precision) == "f64" else CS_F32


def _one_call_per_context(fn):
    """The library's contexts own scratch (template weights, mask tables, candidate pools): one call in flight each.  The
    reference's functions can be called from several threads at once; calls that share a Device take turns here."""
    import functools

    @functools.wraps(fn)
    def locked(dev, *args, **kwargs):
        with dev.lock:
            return fn(dev, *args, **kwargs)
    return locked


class KernelSpec:
    """Host-side description of one pattern template (keeps the float64 arrays alive while the
    C struct points at them).  With tsvd, the correlated kernels are the truncated-SVD
    reconstructions (reference detection.py:618-619, 1037-1043, preprocessing.py:810-847)."""

    def __init__(self, kernel, tsvd=None):
        kernel = np.ascontiguousarray(np.asarray(kernel, dtype=np.float64))
        if kernel.ndim != 2:
            raise ValueError("kernel must be 2-dimensional")
        self.kernel = kernel
        self.km, self.kn = kernel.shape
        self.conv = None
        self.sq = None
        if tsvd is not None:
            from .utils.preprocessing import factorise_kernel
            u, v = factorise_kernel(kernel.copy(), prop_info=tsvd)
            self.conv = np.ascontiguousarray(u @ v)
            u2, v2 = factorise_kernel(kernel ** 2, prop_info=tsvd)
            self.sq = np.ascontiguousarray(u2 @ v2)
        dp = C.POINTER(C.c_double)
        self.struct = CsKernel(
            self.km, self.kn,
            self.kernel.ctypes.data_as(dp),
            self.conv.ctypes.data_as(dp) if self.conv is not None else None,
            self.sq.ctypes.data_as(dp) if self.sq is not None else None,
        )


def min_present(kernel_shape, missing_tol):
    """int((1 - missing_tol) * km * kn), reference detection.py:1069-1072."""
    return int((1 - missing_tol) * kernel_shape[0] * kernel_shape[1])


@_one_call_per_context
def run_normxcorr2(dev, sig, shape, kspec, out, *, full, sym_upper, max_dist, mask_mode=MASK_NONE,
                   miss_row=None, miss_col=None, mask=None, missing_tol=0.75, nobs=None,
                   precision=None, stream=None, row_window=None):
    """sig / out / nobs are CsMatrix; miss_row / miss_col / mask are DeviceBuffers (uint8).  row_window = (a, b): only the
    output rows a <= i < b are produced (geometry and masks stay those of the whole matrix; sig / out may be slabs that
    start at CsMatrix.row0)."""
    params = _corr_params(shape, kspec, full, sym_upper, max_dist, mask_mode, miss_row, miss_col, mask, missing_tol,
                          compute_code(precision), row_window)
    dev._check(dev.lib.cs_normxcorr2(dev.ctx, stream, C.byref(sig), C.byref(kspec.struct), C.byref(params),
                                     C.byref(out), C.byref(nobs) if nobs is not None else None))
    return params


@_one_call_per_context
def run_normxcorr2_host(dev, signal, kspec, *, full, sym_upper, max_dist, missing_tol=0.75, out_dtype=np.float64):
    """Dense float32 host map -> coefficient map on the host through cs_normxcorr2_host (upload, kernel and
    download of row slabs overlap; float64 widening on the library's host threads).  None when the map holds non-finite
    pixels or values the float32 kernels cannot square (the library finds out on the device, beside the kernels)."""
    ms, ns = signal.shape
    params = _corr_params((ms, ns), kspec, full, sym_upper, max_dist, MASK_NONE, None, None, None, missing_tol, CS_F32)
    # result pages from the device's pool of page-locked buffers: a fresh 128 MB numpy array would be
    # page-faulted in by the widening threads (measured: 7 ms of an 11 ms call)
    out = dev.pinned_result((ms, ns), out_dtype)
    rc = dev.lib.cs_normxcorr2_host(dev.ctx, signal.ctypes.data, _lib.np_dtype_code(signal.dtype), ns,
                                    C.byref(kspec.struct), C.byref(params), out.ctypes.data,
                                    _lib.np_dtype_code(out_dtype), ns)
    if rc == -5:
        return None                  # CS_ERR_RANGE: non-finite pixels or magnitudes beyond the float32 range (caller: staged path)
    dev._check(rc)
    return out


@_one_call_per_context
def run_rescore(dev, sig, shape, kspec, rows, cols, *, full, sym_upper, max_dist, mask_mode=MASK_NONE,
                miss_row=None, miss_col=None, mask=None, missing_tol=0.75, stream=None):
    """float64 coefficients (and present-pixel counts) at the given pixels."""
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    cols = np.ascontiguousarray(cols, dtype=np.int32)
    n = rows.size
    if n == 0:
        return np.zeros(0), np.zeros(0)
    params = CsNormxcorr2Params(
        int(shape[0]), int(shape[1]), int(bool(full)), int(bool(sym_upper)),
        -1 if max_dist is None else int(max_dist), int(mask_mode),
        miss_row.ptr if miss_row is not None else None,
        miss_col.ptr if miss_col is not None else None,
        mask.ptr if mask is not None else None,
        min_present((kspec.km, kspec.kn), missing_tol), CS_F64, XCORR_THRESHOLD, DENOM_EPS,
    )
    d_rows, d_cols = dev.to_device(rows), dev.to_device(cols)
    d_r, d_n = dev.empty(n, np.float64), dev.empty(n, np.float64)
    dev._check(dev.lib.cs_rescore_f64(dev.ctx, stream, C.byref(sig), C.byref(kspec.struct), C.byref(params),
                                      d_rows.ptr, d_cols.ptr, n, d_r.ptr, d_n.ptr))
    return d_r.download(stream), d_n.download(stream)


# Detect mode on float32 maps (csrc/cs_device.h cand_screen_*): a pixel is re-evaluated in float64 on the device when its
# float32 coefficient is within RESCORE_MARGIN of the threshold, OR when its window is too ill-conditioned for the float32
# error to stay below a quarter of that margin (variance of the window relative to its mean square, or of the template over
# the present pixels, below 8 n 2^-24 / margin = 2.8e-3 for a 17 x 17 template), OR when one of its sums sits next to one of
# the reference's zeroing thresholds; the exact values are then thresholded.  The candidate set is a superset of the pixels
# whose exact coefficient passes whatever the data look like (tests/test_gpu_margin.py).
RESCORE_MARGIN = 0.05




def map_pitch(width, itemsize=4, quantum=16):
    """Row pitch (elements) for a device-resident dense map of `width` columns: rounded up to `quantum`, and never a multiple
    of 2 KiB -- with a power-of-two pitch the same column of every row lies on the same HBM channels, and a tile's 64 rows x
    256 bytes of stores (or the 80 rows a tile stages) all land on those few channels: the dense matrix-core kernel's stores
    cost 15 us of a 105 us call at pitch 4096 floats and 5 us at 4160.  CsMatrix.ld carries the pitch."""
    ld = (int(width) + quantum - 1) // quantum * quantum
    if (ld * itemsize) % 2048 == 0:
        ld += 64
    return ld


def to_device_map(dev, array, stream=None):
    """Upload a dense 2-D host array with the pitch of map_pitch: (DeviceBuffer, ld)."""
    array = np.asarray(array)
    ld = map_pitch(array.shape[1], array.dtype.itemsize)
    if ld == array.shape[1]:
        return dev.to_device(array, stream=stream), ld
    padded = np.zeros((array.shape[0], ld), dtype=array.dtype)
    padded[:, :array.shape[1]] = array
    return dev.to_device(padded, stream=stream), ld
# below this threshold most pixels of a float32 map would be candidates: such maps are computed in float64
LOW_PEARSON_F64 = 0.1


def rescore_margin(pearson):
    return min(RESCORE_MARGIN, 0.25 * float(pearson))


def _corr_params(shape, kspec, full, sym_upper, max_dist, mask_mode, miss_row, miss_col, mask, missing_tol, code,
                 row_window=None):
    rb, re = (0, 0) if row_window is None else (int(row_window[0]), int(row_window[1]))
    return CsNormxcorr2Params(
        int(shape[0]), int(shape[1]), int(bool(full)), int(bool(sym_upper)),
        -1 if max_dist is None else int(max_dist), int(mask_mode),
        miss_row.ptr if miss_row is not None else None,
        miss_col.ptr if miss_col is not None else None,
        mask.ptr if mask is not None else None,
        min_present((kspec.km, kspec.kn), missing_tol), code, XCORR_THRESHOLD, DENOM_EPS, rb, re)


def _host_buffers(dev, cap, kk, want_windows):
    """Pinned result buffers of a device (grow-only, reused by every call): records always, the window buffer only for
    callers that fetch windows -- cap * kk float64, sized for the template at hand (an 81 x 81 template at cap = 4096 is
    215 MB of page-locked memory per context; hipHostMalloc of that costs more than the detection it would serve)."""
    have = getattr(dev, "_foci_host", None) or (None, None)
    rec, win = have
    if rec is None or rec.shape[0] < cap:
        cap = max(cap, 1024)
        rec = dev.pinned_empty(cap, FOCUS_DTYPE)
        win = None                                   # its capacity follows the records'
    if want_windows and (win is None or win.shape[0] < rec.shape[0] or win.shape[1] < kk):
        win = dev.pinned_empty((rec.shape[0], max(kk, win.shape[1] if win is not None else 0)), np.float64)
    dev._foci_host = (rec, win)
    return rec, win


@_one_call_per_context
def run_detect_foci(dev, sig, shape, kspec, *, pearson, lo_diag, hi_diag, inter, diag_only, full, sym_upper, max_dist,
                    mask_mode=MASK_NONE, miss_row=None, miss_col=None, missing_tol=0.75, precision=None,
                    want_windows=True, min_size=2, stream=None):
    """detect mode of one sub-matrix x one template on the device (cs_detect_foci): records of the
    foci in the reference's order (numpy structured array, _lib.FOCUS_DTYPE) and their windows."""
    if float(pearson) <= LOW_PEARSON_F64:
        precision = "f64"
    params = _corr_params(shape, kspec, full, sym_upper, max_dist, mask_mode, miss_row, miss_col, None, missing_tol,
                          compute_code(precision))
    fp = CsFociParams(float(pearson), rescore_margin(pearson), int(min_size), int(bool(diag_only)), int(lo_diag), int(hi_diag),
                      int(bool(inter)), int(bool(want_windows)))
    kk = kspec.km * kspec.kn
    cap = 4096
    while True:
        rec, win = _host_buffers(dev, cap, kk, want_windows)
        cap = rec.shape[0]
        n = C.c_int64(0)
        rc = dev.lib.cs_detect_foci(dev.ctx, stream, C.byref(sig), C.byref(kspec.struct), C.byref(params), C.byref(fp),
                                    rec.ctypes.data, cap, C.byref(n), win.ctypes.data if want_windows else None)
        if rc == -4 and n.value > cap:          # CS_ERR_OVERFLOW: more foci than records
            cap = int(n.value) + int(n.value) // 4
            continue
        dev._check(rc)
        break
    k = int(n.value)
    windows = None
    if want_windows:
        windows = win.reshape(-1)[:k * kk].reshape(k, kspec.km, kspec.kn).copy()
    return rec[:k].copy(), windows


def _block_arrays(dev, sigs, sigs32, shapes, kspec, max_dists, miss_rows, miss_cols, missing_tol, code, sym_upper):
    """The per-block argument arrays of a batched call (cs_matrix x n, cs_normxcorr2_params x n), cached on the device object:
    the templates of a configuration -- and every step of a run -- pass the same staged blocks, and building 23 ctypes
    structures three times per call cost a tenth of a template's wall time."""
    key = (tuple(s.d_ptr for s in sigs), tuple((s.band_w, s.ld, s.layout, s.dtype) for s in sigs), tuple(tuple(x) for x in shapes), tuple(int(m) if m is not None else -1 for m in max_dists),
           tuple(s.d_ptr if s is not None else 0 for s in sigs32) if sigs32 is not None else None, kspec.km, kspec.kn, float(missing_tol),
           int(code), bool(sym_upper), tuple(m.ptr for m in miss_rows), tuple(m.ptr for m in miss_cols))
    cache = dev.__dict__.setdefault("_block_array_cache", {})
    hit = cache.get(key)
    if hit is None:
        n_blocks = len(sigs)
        sig_arr = (_lib.CsMatrix * n_blocks)(*sigs)
        s32_arr = None
        if sigs32 is not None and any(s is not None for s in sigs32):
            s32_arr = (_lib.CsMatrix * n_blocks)(*[s if s is not None else _lib.CsMatrix(None, CS_F32, 0, 0, 0, 0, 0) for s in sigs32])
        par_arr = (CsNormxcorr2Params * n_blocks)(*[
            _corr_params(shapes[b], kspec, True, sym_upper, max_dists[b], MASK_BINS, miss_rows[b], miss_cols[b], None, missing_tol, code)
            for b in range(n_blocks)])
        if len(cache) > 64:
            cache.clear()
        hit = cache[key] = (sig_arr, s32_arr, par_arr)
    return hit


@_one_call_per_context
def run_detect_foci_batch(dev, sigs, shapes, kspec, *, pearson, hi_diags, inter, diag_only, max_dists, miss_rows, miss_cols,
                          missing_tol=0.75, want_windows=True, min_size=2, stream=None, flat=False, begin_only=False):
    """detect mode of a 1-D pattern (<= 4 scanned diagonals) on MANY banded sub-matrices with one native call
    (cs_detect_foci_batch).  Returns a list of (records, windows) per sub-matrix, or None when the library
    says a block does not qualify (the caller then goes block by block).  kspec may be a list of up to 4 templates of one
    size (cs_detect_foci_batch_templates; flat=True only): the records then come template by template, counts[t * n + b].
    begin_only=True (templates, flat): the chain is only ENQUEUED on `stream` (the library's asynchronous form) and a callable
    is returned that waits for it and yields the flat result -- the caller puts other work on the device in between; nothing
    else may use `dev` until the callable has run."""
    kspecs = list(kspec) if isinstance(kspec, (list, tuple)) else None
    if kspecs is not None:
        if not flat or not 1 <= len(kspecs) <= 4 or any((k.km, k.kn) != (kspecs[0].km, kspecs[0].kn) for k in kspecs):
            return None
        kspec = kspecs[0]
    n_templates = len(kspecs) if kspecs is not None else 1
    n_blocks = len(sigs)
    sig_arr, _, par_arr = _block_arrays(dev, sigs, None, shapes, kspec, max_dists, miss_rows, miss_cols, missing_tol, CS_F64, True)
    fp_arr = (CsFociParams * n_blocks)(*[
        CsFociParams(float(pearson), rescore_margin(pearson), int(min_size), int(bool(diag_only)), 0, int(hi_diags[b]), int(bool(inter)),
                     int(bool(want_windows))) for b in range(n_blocks)])
    counts = (C.c_int64 * (n_blocks * n_templates))()
    kk = kspec.km * kspec.kn
    cap = 4096 * n_templates
    k_arr = (_lib.CsKernel * n_templates)(*[k.struct for k in kspecs]) if kspecs is not None else None
    if begin_only:
        if kspecs is None or not flat:
            return None
        rec, win = _host_buffers(dev, max(cap, getattr(dev, "_batch_cap_hint", 0)), kk, want_windows)
        fp_arr[0].reserved = 1                                   # asynchronous: return once the chain is enqueued
        rc = dev.lib.cs_detect_foci_batch_templates(dev.ctx, stream, n_blocks, sig_arr, n_templates, k_arr, par_arr, fp_arr,
                                                    rec.ctypes.data, rec.shape[0], counts, win.ctypes.data if want_windows else None)
        fp_arr[0].reserved = 0
        if rc == -3:
            return None
        dev._check(rc)

        def finish():
            with dev.lock:
                rc2 = dev.lib.cs_detect_foci_batch_finish(dev.ctx, stream, counts)
            if rc2 == -4:                                        # more foci than the buffers hold: once more, waiting for it
                dev._batch_cap_hint = int(sum(counts)) + int(sum(counts)) // 4
                return run_detect_foci_batch(dev, sigs, shapes, kspecs, pearson=pearson, hi_diags=hi_diags, inter=inter,
                                             diag_only=diag_only, max_dists=max_dists, miss_rows=miss_rows, miss_cols=miss_cols,
                                             missing_tol=missing_tol, want_windows=want_windows, min_size=min_size, stream=stream,
                                             flat=True)
            dev._check(rc2)
            cnt = np.frombuffer(counts, dtype=np.int64).copy()
            total = int(cnt.sum())
            windows = win.reshape(-1)[:total * kk].reshape(total, kspec.km, kspec.kn).copy() if want_windows else None
            return rec[:total].copy(), windows, cnt

        return finish
    while True:
        rec, win = _host_buffers(dev, cap, kk, want_windows)
        cap = rec.shape[0]
        if kspecs is not None:
            rc = dev.lib.cs_detect_foci_batch_templates(dev.ctx, stream, n_blocks, sig_arr, n_templates, k_arr, par_arr, fp_arr,
                                                        rec.ctypes.data, cap, counts, win.ctypes.data if want_windows else None)
        else:
            rc = dev.lib.cs_detect_foci_batch(dev.ctx, stream, n_blocks, sig_arr, C.byref(kspec.struct), par_arr, fp_arr,
                                              rec.ctypes.data, cap, counts, win.ctypes.data if want_windows else None)
        if rc == -3:
            return None
        if rc == -4 and sum(counts) > cap:
            cap = int(sum(counts)) + int(sum(counts)) // 4
            continue
        dev._check(rc)
        break
    cnt = np.frombuffer(counts, dtype=np.int64).copy()
    total = int(cnt.sum())
    if flat:
        # all records (and windows) in block order + the per-block counts: the caller post-processes them in one go
        windows = win.reshape(-1)[:total * kk].reshape(total, kspec.km, kspec.kn).copy() if want_windows else None
        return rec[:total].copy(), windows, cnt
    out, o = [], 0
    flat_w = win.reshape(-1) if want_windows else None
    for b in range(n_blocks):
        k = int(cnt[b])
        windows = flat_w[o * kk:(o + k) * kk].reshape(k, kspec.km, kspec.kn).copy() if want_windows else None
        out.append((rec[o:o + k].copy(), windows))
        o += k
    return out


@_one_call_per_context
def run_detect_foci_blocks(dev, sigs, sigs32, shapes, kspec, *, pearson, lo_diags, hi_diags, inter, diag_only, max_dists, miss_rows,
                           miss_cols, missing_tol=0.75, want_windows=True, min_size=2, stream=None, precision=None, exclusive=False):
    """detect mode of one template on MANY sub-matrices with one native call (cs_detect_foci_blocks): 2-D scans on the masked
    matrix-core tile kernel in candidate mode, one candidate list, one foci chain; 1-D scans are passed on to the narrow batch.
    sigs32: the float32 twins of the float64 maps (entries may be None).  Returns (records, windows, counts) -- the records of
    all blocks one block after the other -- or None when the library says a block does not qualify (caller: block by block)."""
    if (precision or _precision) == "f64" or float(pearson) <= LOW_PEARSON_F64:
        return None
    n_blocks = len(sigs)
    sig_arr, s32_arr, par_arr = _block_arrays(dev, sigs, sigs32, shapes, kspec, max_dists, miss_rows, miss_cols, missing_tol, CS_F32,
                                              not inter)
    fp_arr = (CsFociParams * n_blocks)(*[
        CsFociParams(float(pearson), rescore_margin(pearson), int(min_size), int(bool(diag_only)), int(lo_diags[b]), int(hi_diags[b]),
                     int(bool(inter)), int(bool(want_windows)), int(bool(exclusive)), 0) for b in range(n_blocks)])
    counts = (C.c_int64 * n_blocks)()
    kk = kspec.km * kspec.kn
    cap = 4096
    while True:
        rec, win = _host_buffers(dev, cap, kk, want_windows)
        cap = rec.shape[0]
        rc = dev.lib.cs_detect_foci_blocks(dev.ctx, stream, n_blocks, sig_arr, s32_arr, C.byref(kspec.struct), par_arr, fp_arr,
                                           rec.ctypes.data, cap, counts, win.ctypes.data if want_windows else None)
        if rc == -3:
            return None
        if rc == -4 and sum(counts) > cap:
            cap = int(sum(counts)) + int(sum(counts)) // 4
            continue
        dev._check(rc)
        break
    cnt = np.frombuffer(counts, dtype=np.int64).copy()
    total = int(cnt.sum())
    windows = win.reshape(-1)[:total * kk].reshape(total, kspec.km, kspec.kn).copy() if want_windows else None
    return rec[:total].copy(), windows, cnt


@_one_call_per_context
def run_candidates(dev, sig, shape, kspec, row_window, *, pearson, lo_diag, hi_diag, inter, full, sym_upper, max_dist,
                   mask_mode=MASK_NONE, miss_row=None, miss_col=None, missing_tol=0.75, precision=None, stream=None,
                   **_unused):
    """First half of detect mode on a row window of a sub-matrix (cs_candidates): coordinates and
    float64 coefficients of the pixels of rows row_window[0] <= i < row_window[1] that pass the
    threshold, row-major.  `sig` holds those rows and the template's halo (CsMatrix.row0)."""
    if float(pearson) <= LOW_PEARSON_F64:
        precision = "f64"
    params = _corr_params(shape, kspec, full, sym_upper, max_dist, mask_mode, miss_row, miss_col, None, missing_tol,
                          compute_code(precision), row_window)
    fp = CsFociParams(float(pearson), rescore_margin(pearson), 1, 0, int(lo_diag), int(hi_diag), int(bool(inter)), 0)
    cap = 1 << 14
    while True:
        rows, cols = np.empty(cap, np.int32), np.empty(cap, np.int32)
        vals = np.empty(cap, np.float64)
        n = C.c_int64(0)
        rc = dev.lib.cs_candidates(dev.ctx, stream, C.byref(sig), C.byref(kspec.struct), C.byref(params), C.byref(fp),
                                   rows.ctypes.data, cols.ctypes.data, vals.ctypes.data, cap, C.byref(n))
        if rc == -4 and n.value > cap:
            cap = int(n.value) + int(n.value) // 4
            continue
        dev._check(rc)
        break
    k = int(n.value)
    return rows[:k], cols[:k], vals[:k]


@_one_call_per_context
def run_label_foci(dev, shape, rows, cols, vals, *, min_size=2, diag_only=False, stream=None):
    """Second half of detect mode (cs_label_foci): the 4-connected foci of a candidate list -- the
    coordinates of each focus at its maximum and its size, in the order of cs_detect_foci."""
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    cols = np.ascontiguousarray(cols, dtype=np.int32)
    vals = np.ascontiguousarray(vals, dtype=np.float64)
    k = rows.size
    cap = max(k // max(int(min_size), 1), 1)
    f_rows, f_cols, f_size = (np.empty(cap, np.int32) for _ in range(3))
    n = C.c_int64(0)
    dev._check(dev.lib.cs_label_foci(dev.ctx, stream, int(shape[0]), int(shape[1]), rows.ctypes.data, cols.ctypes.data,
                                     vals.ctypes.data, k, int(min_size), int(diag_only), f_rows.ctypes.data,
                                     f_cols.ctypes.data, f_size.ctypes.data, cap, C.byref(n)))
    m = int(n.value)
    return f_rows[:m], f_cols[:m], f_size[:m]


@_one_call_per_context
def run_quantify_pixels(dev, sig, shape, kspec, rows, cols, *, inter, full, sym_upper, max_dist, mask_mode=MASK_NONE,
                        miss_row=None, miss_col=None, missing_tol=0.75, want_windows=True, stream=None):
    """quantify mode (cs_quantify_pixels): one record (and window) per given pixel, in input order."""
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    cols = np.ascontiguousarray(cols, dtype=np.int32)
    k = rows.size
    params = _corr_params(shape, kspec, full, sym_upper, max_dist, mask_mode, miss_row, miss_col, None, missing_tol, CS_F64)
    fp = CsFociParams(0.0, 0.0, 1, 0, 0, 0, int(bool(inter)), int(bool(want_windows)))
    kk = kspec.km * kspec.kn
    rec, win = _host_buffers(dev, max(k, 1), kk, want_windows)
    dev._check(dev.lib.cs_quantify_pixels(dev.ctx, stream, C.byref(sig), C.byref(kspec.struct), C.byref(params),
                                          C.byref(fp), rows.ctypes.data, cols.ctypes.data, k, rec.ctypes.data,
                                          win.ctypes.data if want_windows else None))
    windows = win.reshape(-1)[:k * kk].reshape(k, kspec.km, kspec.kn).copy() if want_windows else None
    return rec[:k].copy(), windows


@_one_call_per_context
def run_quantify_blocks(dev, blocks, kspec, blk, rows, cols, *, missing_tol=0.75, want_windows=True, stream=None):
    """quantify mode over MANY staged sub-matrices with one native call (cs_quantify_blocks): position t is pixel
    (rows[t], cols[t]) of blocks[blk[t]] (pipeline.StagedBlock: intra blocks banded or dense, inter blocks dense -- `inter`
    and `max_dist` per block).  One record (and window) per position, in input order."""
    blk = np.ascontiguousarray(blk, dtype=np.int32)
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    cols = np.ascontiguousarray(cols, dtype=np.int32)
    k, n_blocks = rows.size, len(blocks)
    sig_arr = (_lib.CsMatrix * n_blocks)(*[b.sig for b in blocks])
    par_arr = (CsNormxcorr2Params * n_blocks)(*[
        _corr_params(b.shape, kspec, True, not b.inter, b.max_dist, MASK_BINS, b.miss_row, b.miss_col, None, missing_tol, CS_F64)
        for b in blocks])
    fp_arr = (CsFociParams * n_blocks)(*[CsFociParams(0.0, 0.0, 1, 0, 0, 0, int(bool(b.inter)), int(bool(want_windows))) for b in blocks])
    kk = kspec.km * kspec.kn
    rec, win = _host_buffers(dev, max(k, 1), kk, want_windows)
    dev._check(dev.lib.cs_quantify_blocks(dev.ctx, stream, n_blocks, sig_arr, C.byref(kspec.struct), par_arr, fp_arr, blk.ctypes.data,
                                          rows.ctypes.data, cols.ctypes.data, k, rec.ctypes.data,
                                          win.ctypes.data if want_windows else None))
    windows = win.reshape(-1)[:k * kk].reshape(k, kspec.km, kspec.kn).copy() if want_windows else None
    return rec[:k].copy(), windows


@_one_call_per_context
def run_compact(dev, corr, shape, threshold, lo_diag, hi_diag, stream=None, guess=1 << 16):
    """(rows, cols, vals) of the stored pixels of `corr` with value >= threshold inside the
    diagonal range; sorted row-major."""
    # The device arrays are kept with the device between calls (three allocations per call cost more than the compaction of a
    # rank's share of a split block), and only the `count` entries that exist cross the link (the whole capacity did: 1 MB for a
    # hundred candidates).  A context serves one call in flight (the callers hold Device.lock or own the device).
    cap = int(guess)
    while True:
        have = getattr(dev, "_compact_scratch", None)
        if have is None or have[0] < cap:
            have = dev._compact_scratch = (cap, dev.empty(cap, np.int32), dev.empty(cap, np.int32), dev.empty(cap, np.float64),
                                           dev.empty(1, np.int64))
        _, d_rows, d_cols, d_vals, d_count = have
        d_count.zero(stream)
        dev._check(dev.lib.cs_compact_ge(dev.ctx, stream, C.byref(corr), int(shape[0]), int(shape[1]),
                                         float(threshold), int(lo_diag), int(hi_diag), d_rows.ptr, d_cols.ptr,
                                         d_vals.ptr, have[0], d_count.ptr))
        count = int(d_count.download(stream)[0])
        if count <= have[0]:
            break
        cap = count
    rows, cols, vals = np.empty(count, np.int32), np.empty(count, np.int32), np.empty(count, np.float64)
    if count:
        for host, buf in ((rows, d_rows), (cols, d_cols), (vals, d_vals)):
            dev._check(dev.lib.cs_memcpy_d2h(dev.ctx, host.ctypes.data, buf.ptr, host.nbytes, stream))
    order = np.lexsort((cols, rows))
    return rows[order], cols[order], vals[order]


# ------------------------------------------------------------------------------------------------
# CSR staging
# ------------------------------------------------------------------------------------------------
class DeviceCsr:
    def __init__(self, dev, mat, dtype=None, stream=None):
        mat = sp.csr_matrix(mat)
        if not mat.has_canonical_format:
            mat = mat.copy()
            mat.sum_duplicates()
        if dtype is None:
            dtype = np.float32 if mat.dtype == np.float32 else np.float64
        self.shape = mat.shape
        self.nnz = int(mat.nnz)
        self.indptr = dev.to_device(mat.indptr, np.int64, stream)
        self.indices = dev.to_device(mat.indices, np.int32, stream)
        self.data = dev.to_device(mat.data, dtype, stream)
        self.struct = CsCsr(self.shape[0], self.shape[1], self.nnz, self.indptr.ptr, self.indices.ptr,
                            self.data.ptr, _lib.np_dtype_code(dtype))


def diag_range(mat):
    """(lo, hi) of col - row over the stored entries of a sparse matrix, or None if empty."""
    if mat.nnz == 0:
        return None
    if mat.format == "csr":
        counts = np.diff(mat.indptr)
        rows = np.flatnonzero(counts)
        if mat.has_sorted_indices:       # first / last stored column of every non-empty row
            first = mat.indices[mat.indptr[rows]].astype(np.int64)
            last = mat.indices[mat.indptr[rows + 1] - 1].astype(np.int64)
            return int((first - rows).min()), int((last - rows).max())
        d = mat.indices.astype(np.int64) - np.repeat(np.arange(mat.shape[0], dtype=np.int64), counts)
        return int(d.min()), int(d.max())
    coo = mat.tocoo()
    d = coo.col.astype(np.int64) - coo.row.astype(np.int64)
    return int(d.min()), int(d.max())


def csr_to_matrix(dev, dcsr, *, layout, dtype, band_lo=0, band_w=0, law=None, max_val=0.0, stream=None):
    """Scatter a device CSR into a zero-filled dense or band buffer (optionally detrending).
    Returns (DeviceBuffer, CsMatrix)."""
    n_rows, n_cols = dcsr.shape
    if layout == LAYOUT_BAND:
        ld = (int(band_w) + 63) // 64 * 64
        buf = dev.empty((n_rows, ld), dtype)
        code = CS_U8 if np.dtype(dtype) == np.uint8 else _lib.np_dtype_code(dtype)
        mat = CsMatrix(buf.ptr, code, LAYOUT_BAND, ld, int(band_lo), int(band_w))
    else:
        ld = (n_cols + 15) // 16 * 16
        buf = dev.empty((n_rows, ld), dtype)
        code = CS_U8 if np.dtype(dtype) == np.uint8 else _lib.np_dtype_code(dtype)
        mat = CsMatrix(buf.ptr, code, LAYOUT_DENSE, ld, 0, 0)
    d_law = None
    n_law = 0
    if law is not None:
        law = np.ascontiguousarray(law, dtype=np.float64)
        d_law = dev.to_device(law, np.float64, stream)
        n_law = law.size
    dev._check(dev.lib.cs_csr_to_band(dev.ctx, stream, C.byref(dcsr.struct), d_law.ptr if d_law else None,
                                      n_law, float(max_val or 0.0), C.byref(mat)))
    return buf, mat


def distance_law_sums(dev, dcsr, detectable_mask, n_diags, stream=None):
    """Per-diagonal (sum, count) of the positive pixels between detectable bins."""
    d_det = dev.to_device(np.asarray(detectable_mask, dtype=np.uint8), np.uint8, stream) \
        if detectable_mask is not None else None
    d_sum, d_cnt = dev.empty(max(n_diags, 1), np.float64), dev.empty(max(n_diags, 1), np.int64)
    dev._check(dev.lib.cs_distance_law_csr(dev.ctx, stream, C.byref(dcsr.struct),
                                           d_det.ptr if d_det else None, int(n_diags), d_sum.ptr, d_cnt.ptr))
    return d_sum.download(stream)[:n_diags], d_cnt.download(stream)[:n_diags]


def detrend_values(dev, dcsr, law, max_val, stream=None):
    """Detrended stored values of the CSR (same order as dcsr.data)."""
    law = np.ascontiguousarray(law, dtype=np.float64)
    d_law = dev.to_device(law, np.float64, stream)
    out = dev.empty(max(dcsr.nnz, 1), dcsr.data.dtype)
    dev._check(dev.lib.cs_detrend_csr(dev.ctx, stream, C.byref(dcsr.struct), d_law.ptr, law.size,
                                      float(max_val or 0.0), out.ptr))
    return out.download(stream)[:dcsr.nnz]


def band_to_coo(band, band_lo, band_w, n_cols):
    """Non-zero entries of a host band array as (rows, cols, vals)."""
    view = band[:, :band_w]
    rows, offs = np.nonzero(view)
    cols = rows + band_lo + offs
    ok = (cols >= 0) & (cols < n_cols)
    return rows[ok], cols[ok], view[rows[ok], offs[ok]]
