"""Detection half of the hot path with the reference's call surface
(chromosight/utils/detection.py).  `xcorr2`, `normxcorr2` and the correlation inside
`pattern_detector` run on the MI355X (libchromosight_hip.so); foci picking, window
validation and neighbour removal are the host-side callers that turn the coefficient map
into pattern coordinates with the reference's exact rules.

Numerics: the bulk coefficient map is computed in float32 (or float64 with
chromosight_amd.set_precision("f64")); every pixel that decides a detection -- all pixels at
or above the Pearson threshold, and all reported scores -- is re-evaluated in float64 on the
device (cs_rescore_f64), so candidate coordinates do not depend on float32 rounding.
"""
import ctypes as C
import os
import pathlib
import warnings

import numpy as np
import pandas as pd
import scipy.sparse as sp
from scipy.sparse import csgraph

from . import preprocessing as preproc
from . import stats as cus
from .. import engine
from .._lib import (CS_F32, CS_F64, LAYOUT_BAND, LAYOUT_BAND_LAZY, LAYOUT_BAND_PADDED, LAYOUT_DENSE, MASK_BINS, MASK_EXPLICIT, MASK_NONE,
                    FOCUS_DTYPE, CsMatrix, get_device, np_dtype_code)

RESCORE_MARGIN = engine.RESCORE_MARGIN


# ============================================================================================
# signal staging
# ============================================================================================
class _Staged:
    """A signal resident in HBM in dense or band layout, plus the geometry of the output map."""

    def __init__(self, dev, signal, kernel_shape, sym_upper, full, extra_range=None,
                 out_diag_range=None):
        km, kn = kernel_shape
        self.dev = dev
        self.shape = ms, ns = signal.shape
        self.sparse = sp.issparse(signal)
        self.keep = []  # device buffers that must outlive the calls
        reach = (km - 1) // 2 + (kn - 1) // 2
        if self.sparse:
            rng = engine.diag_range(signal)
            if extra_range is not None:
                rng = extra_range if rng is None else (min(rng[0], extra_range[0]), max(rng[1], extra_range[1]))
            if rng is None:
                rng = (0, 0)
            lo, hi = rng
            out_lo, out_hi = max(lo - reach, -(ms - 1)), min(hi + reach, ns - 1)
            if sym_upper:
                out_lo = max(out_lo, (km - kn) if full else 0)
            if out_diag_range is not None:
                out_lo, out_hi = max(out_lo, out_diag_range[0]), min(out_hi, out_diag_range[1])
            out_hi = max(out_hi, out_lo)
            in_w, out_w = hi - lo + 1, out_hi - out_lo + 1
            use_band = 2 * max(in_w, out_w) < ns
            in_dtype = np.float32 if signal.dtype == np.float32 else np.float64
            self.dcsr = engine.DeviceCsr(dev, signal, in_dtype)
            if use_band:
                self.layout = LAYOUT_BAND
                self.in_lo, self.in_w = lo, in_w
                self.out_lo, self.out_w = out_lo, out_w
                buf, self.sig = engine.csr_to_matrix(dev, self.dcsr, layout=LAYOUT_BAND, dtype=in_dtype,
                                                     band_lo=lo, band_w=in_w)
            else:
                self.layout = LAYOUT_DENSE
                buf, self.sig = engine.csr_to_matrix(dev, self.dcsr, layout=LAYOUT_DENSE, dtype=in_dtype)
            self.keep.append(buf)
            self.sig_buf = buf
        else:
            arr = np.asarray(signal)
            if arr.dtype != np.float32:
                arr = arr.astype(np.float64, copy=False)
            self.layout = LAYOUT_DENSE
            ld = (ns + 15) // 16 * 16
            if ld == ns:
                padded = np.ascontiguousarray(arr)      # already a legal leading dimension: no copy
            else:
                padded = np.zeros((ms, ld), dtype=arr.dtype)
                padded[:, :ns] = arr
            buf = dev.to_device(padded)
            self.sig_buf = buf
            self.keep.append(buf)
            self.sig = CsMatrix(buf.ptr, np_dtype_code(arr.dtype), LAYOUT_DENSE, ld, 0, 0)

    def stage_mask(self, mask):
        """Explicit missing mask (scipy sparse bool) -> uint8 buffer with the signal's geometry."""
        dmask = engine.DeviceCsr(self.dev, sp.csr_matrix(mask, dtype=np.float32), np.float32)
        if self.layout == LAYOUT_BAND:
            buf, _ = engine.csr_to_matrix(self.dev, dmask, layout=LAYOUT_BAND, dtype=np.uint8,
                                          band_lo=self.in_lo, band_w=self.in_w)
        else:
            buf, _ = engine.csr_to_matrix(self.dev, dmask, layout=LAYOUT_DENSE, dtype=np.uint8)
        self.keep.append(buf)
        return buf

    def alloc_out(self, dtype):
        ms, ns = self.shape
        if self.layout == LAYOUT_BAND:
            ld = (self.out_w + 63) // 64 * 64
            buf = self.dev.zeros((ms, ld), dtype)
            mat = CsMatrix(buf.ptr, np_dtype_code(dtype), LAYOUT_BAND, ld, self.out_lo, self.out_w)
        else:
            ld = (ns + 15) // 16 * 16
            buf = self.dev.empty((ms, ld), dtype)
            mat = CsMatrix(buf.ptr, np_dtype_code(dtype), LAYOUT_DENSE, ld, 0, 0)
        return buf, mat

    def to_host_coo(self, buf):
        """(rows, cols, vals) of the non-zero entries of an output buffer."""
        host = buf.download()
        ms, ns = self.shape
        if self.layout == LAYOUT_BAND:
            return engine.band_to_coo(host, self.out_lo, self.out_w, ns)
        view = host[:, :ns]
        rows, cols = np.nonzero(view)
        return rows, cols, view[rows, cols]

    def to_host_dense(self, buf):
        # large maps come back through pooled page-locked memory (2.5x the pageable D2H rate)
        host = buf.download(pinned=buf.nbytes >= (1 << 22))
        view = host[:, :self.shape[1]]
        return view if view.flags.c_contiguous else np.ascontiguousarray(view)


def _out_dtype():
    return np.float64 if engine.get_precision() == "f64" else np.float32


# largest |pixel| the float32 kernels take: a 31 x 31 window of squares must stay below FLT_MAX (3.4e38) with room for the
# per-tile scaling of the matrix-core kernels
_F32_SAFE_PEAK = 1e15


# ============================================================================================
# xcorr2 / normxcorr2
# ============================================================================================
def xcorr2(signal, kernel, threshold=1e-4, tsvd=None):
    """Cross-correlate (no flip) a dense or sparse 2-D signal with a dense kernel; the result
    is aligned on the window centre, zero on the (k-1)//2 margins, and values with
    |v| < threshold are zeroed (reference detection.py:595-624, 627-723, 726-804).
    Same container type as the input.  With `tsvd`, the kernel is replaced by its truncated-SVD
    reconstruction (reference preprocessing.py:810-847)."""
    if sp.issparse(kernel):
        raise ValueError("cannot handle kernel in sparse format")
    kernel = np.asarray(kernel, dtype=np.float64)
    if tsvd is not None:
        u, v = preproc.factorise_kernel(kernel, prop_info=tsvd)
        kernel = u @ v
    km, kn = kernel.shape
    sm, sn = signal.shape
    if sm < km or sn < kn:
        raise ValueError("cannot have kernel bigger than signal")
    dev = get_device()
    st = _Staged(dev, signal, (km, kn), sym_upper=False, full=False)
    out_buf, out_mat = st.alloc_out(_out_dtype())
    w = np.ascontiguousarray(kernel, dtype=np.float64)
    with dev.lock:                                   # one call in flight per context (engine._one_call_per_context)
        dev._check(dev.lib.cs_xcorr2(dev.ctx, None, C.byref(st.sig), sm, sn,
                                     w.ctypes.data_as(C.POINTER(C.c_double)), km, kn, float(threshold),
                                     engine.compute_code(), C.byref(out_mat)))
    # even kernel sizes give an output one row/column short, like the reference's re-padding
    osm = sm - km + 1 + 2 * ((km - 1) // 2)
    osn = sn - kn + 1 + 2 * ((kn - 1) // 2)
    if st.sparse:
        rows, cols, vals = st.to_host_coo(out_buf)
        ok = (rows < osm) & (cols < osn)
        return sp.csr_matrix((vals[ok].astype(np.float64), (rows[ok], cols[ok])), shape=(osm, osn))
    return st.to_host_dense(out_buf)[:osm, :osn].astype(np.float64)


def normxcorr2(signal, kernel, max_dist=None, sym_upper=False, full=False, missing_mask=None,
               missing_tol=0.75, tsvd=None, pval=False):
    """Pearson correlation of every kernel-sized window of `signal` with `kernel`
    (reference detection.py:807-914 -> :917-1131 sparse, :1134-1273 dense).

    Returns (corr, log10_pvals or None), containers of the same type and shape as `signal`.
    See SURVEY.md section 8(a2) for the per-pixel definition that the device kernel evaluates
    (zeroing thresholds, missing-pixel handling, the `missing_tol` cut, clipping).

    Differences with the reference, all in places where the reference raises by accident:
    a dense signal with a missing mask, and dense full=True with pval=True, are evaluated with
    the sparse formulas instead of crashing (reference detection.py:1246, :1263)."""
    if sp.issparse(kernel):
        raise ValueError("cannot handle kernel in sparse format")
    kernel = np.asarray(kernel, dtype=np.float64)
    if missing_mask is not None:
        if not sp.issparse(missing_mask):
            raise ValueError("Missing mask must be a sparse matrix.")
        if not signal.shape == missing_mask.shape:
            raise ValueError("Signal and missing mask do not have the same shape")
        if missing_mask.dtype != bool:
            raise ValueError(f"Missing mask dtype is {missing_mask.dtype}. Should be bool.")
        if min(kernel.shape) >= max(signal.shape):
            raise ValueError("cannot have kernel bigger than signal")
        preproc.check_missing_mask(signal, missing_mask)
    if not (kernel.std() > 0):
        raise ValueError("Cannot have flat kernel.")
    km, kn = kernel.shape
    ms, ns = signal.shape
    if not (km % 2 and kn % 2):
        raise ValueError("kernel dimensions must be odd")
    if not full and (ms < km or ns < kn):
        raise ValueError("cannot have kernel bigger than signal")
    dev = get_device()
    kspec = engine.KernelSpec(kernel, tsvd)
    if (missing_mask is None and isinstance(signal, np.ndarray) and signal.dtype in (np.float32, np.float64) and signal.ndim == 2
            and signal.flags.c_contiguous and engine.get_precision() == "f32" and signal.size >= (1 << 20)):
        # large host map in, host map out: one native call that pipelines the map over PCIe in row slabs
        # (cs_normxcorr2_host); the float64 array the reference returns is filled by the library's threads
        corr = engine.run_normxcorr2_host(dev, signal, kspec, full=full, sym_upper=sym_upper, max_dist=max_dist,
                                          missing_tol=missing_tol)
        if corr is not None:
            pvals = cus.corr_to_pval(corr.ravel(), km * kn).reshape(corr.shape) if pval else None
            return corr, pvals
        # (out of the float32 kernels' range, or non-finite pixels: the staged path below sorts that out)
    mask_range = engine.diag_range(missing_mask) if missing_mask is not None else None
    # Values the device arithmetic does not take as they are.  Non-finite pixels: in the reference a NaN or an infinite
    # pixel makes every window that holds it NaN and then 0 (detection.py:1088-1101) and leaves the others alone; the
    # device's running box sums would carry it into windows further down, so such pixels are staged as 0 and the windows
    # that hold them are zeroed afterwards.  Magnitudes: float32 arithmetic squares the pixels -- maps whose values leave
    # the range where a window's sum of squares fits are evaluated in float64 (the reference is float64 throughout).
    precision = None
    bad_pixels = None
    values = signal.data if sp.issparse(signal) else np.asarray(signal)
    peak = float(np.max(np.abs(values))) if values.size else 0.0
    if not np.isfinite(peak):
        if sp.issparse(signal):
            coo = signal.tocoo()
            nf = ~np.isfinite(coo.data)
            bad_pixels = (coo.row[nf].astype(np.int64), coo.col[nf].astype(np.int64))
            signal = sp.csr_matrix((np.where(nf, 0.0, coo.data), (coo.row, coo.col)), shape=signal.shape)
            finite = coo.data[~nf]
        else:
            nf = ~np.isfinite(values)
            bad_pixels = tuple(a.astype(np.int64) for a in np.nonzero(nf))
            signal = np.where(nf, 0.0, values)
            finite = values[~nf]
        peak = float(np.max(np.abs(finite))) if finite.size else 0.0
    if engine.get_precision() == "f32" and peak > _F32_SAFE_PEAK:
        precision = "f64"
    st = _Staged(dev, signal, (km, kn), sym_upper, full, extra_range=mask_range)
    mask_buf = None
    mask_mode = MASK_NONE
    mask_kw = {}
    if missing_mask is not None:
        if full:
            _check_framed_signal(signal, missing_mask, (km, kn), sym_upper, max_dist)
        bins = _mask_as_bins(missing_mask, sym_upper, max_dist)
        if bins is not None:
            # the mask is exactly make_missing_mask's: two flag vectors describe it, and the
            # device evaluates it analytically (factorised mask sums) instead of reading a map
            mask_mode = MASK_BINS
            mask_kw = dict(miss_row=dev.to_device(bins[0].astype(np.uint8)),
                           miss_col=dev.to_device(bins[1].astype(np.uint8)))
        else:
            mask_buf = st.stage_mask(missing_mask)
            mask_mode = MASK_EXPLICIT
    # dense inputs come back as float64 arrays (as the reference returns them): the kernel stores
    # float64 directly, which is cheaper than converting 8 bytes per pixel on the host
    out_dtype = (np.float64 if precision == "f64" else _out_dtype()) if st.sparse else np.float64
    out_buf, out_mat = st.alloc_out(out_dtype)
    want_nobs = pval and full and missing_mask is not None
    nobs_buf = nobs_mat = None
    if want_nobs:
        nobs_buf, nobs_mat = st.alloc_out(np.float32)
    engine.run_normxcorr2(dev, st.sig, (ms, ns), kspec, out_mat, full=full, sym_upper=sym_upper,
                          max_dist=max_dist, mask_mode=mask_mode, mask=mask_buf,
                          missing_tol=missing_tol, nobs=nobs_mat, precision=precision, **mask_kw)
    n = km * kn
    if st.sparse:
        rows, cols, vals = st.to_host_coo(out_buf)
        vals = vals.astype(np.float64)
        if bad_pixels is not None:
            # the centres whose window holds a bad pixel p: p - km // 2 .. p + (km - 1) // 2 (the window of centre i spans
            # i - (km - 1) // 2 .. i + km // 2: asymmetric for even sizes), as one set of dilated keys -- a map with a NaN row
            # has thousands of bad pixels, a pass over the result per pixel took minutes
            bp, bq = (np.asarray(x, dtype=np.int64) for x in bad_pixels)
            di = np.arange(-(km // 2), (km - 1) // 2 + 1, dtype=np.int64)
            dj = np.arange(-(kn // 2), (kn - 1) // 2 + 1, dtype=np.int64)
            ci = (bp[:, None, None] + di[None, :, None])
            cj = (bq[:, None, None] + dj[None, None, :])
            ok = (ci >= 0) & (ci < ms) & (cj >= 0) & (cj < ns)
            keys = np.unique((ci * ns + cj)[ok])
            hit = np.isin(rows.astype(np.int64) * ns + cols.astype(np.int64), keys)
            rows, cols, vals = rows[~hit], cols[~hit], vals[~hit]
        corr = sp.csr_matrix((vals, (rows, cols)), shape=(ms, ns))
        pvals = None
        if pval:
            if want_nobs:
                nobs_host = nobs_buf.download()
                if st.layout == LAYOUT_BAND:
                    n_obs = nobs_host[rows, cols - rows - st.out_lo].astype(np.float64)
                else:
                    n_obs = nobs_host[rows, cols].astype(np.float64)
                n_obs[n_obs == 0] = n
                logp = cus.corr_to_pval(vals, n_obs)
            else:
                logp = cus.corr_to_pval(vals, n)
            pvals = sp.csr_matrix((logp, (rows, cols)), shape=(ms, ns))
        return corr, pvals
    corr = st.to_host_dense(out_buf).astype(np.float64, copy=False)
    if bad_pixels is not None:
        for p, q in zip(*bad_pixels):                # centres p - km // 2 .. p + (km - 1) // 2 (see the sparse branch)
            corr[max(p - km // 2, 0):p + (km - 1) // 2 + 1, max(q - kn // 2, 0):q + (kn - 1) // 2 + 1] = 0.0
    pvals = None
    if pval:
        if want_nobs:
            n_obs = st.to_host_dense(nobs_buf).astype(np.float64)
            pvals = cus.corr_to_pval(corr, n_obs)
        else:
            pvals = cus.corr_to_pval(corr.ravel(), n).reshape(corr.shape)
    return corr, pvals


def _mask_as_bins(mask, sym_upper, max_dist):
    """(missing_rows, missing_cols) boolean vectors if `mask` is exactly the mask make_missing_mask
    builds from them (reference preprocessing.py:535-633), else None.  Candidate flags: the main
    diagonal of an upper-symmetric mask, the completely flagged rows / columns of a rectangular one."""
    ms, ns = mask.shape
    m = sp.csr_matrix(mask)
    if m.nnz == 0:
        return np.zeros(ms, dtype=bool), np.zeros(ns, dtype=bool)
    if sym_upper:
        if ms != ns:
            return None
        miss_r = np.asarray(m.diagonal()).astype(bool)
        miss_c = miss_r
    else:
        miss_r = np.diff(m.indptr) == ns
        miss_c = np.asarray(m.sum(axis=0)).ravel() == ms
    try:
        want = preproc.make_missing_mask((ms, ns), np.flatnonzero(~miss_r), np.flatnonzero(~miss_c),
                                         max_dist=max_dist, sym_upper=sym_upper)
    except ValueError:
        return None
    want = sp.csr_matrix(want)
    if want.nnz != m.nnz or (want != m).nnz != 0:
        return None
    return miss_r, miss_c


def _check_framed_signal(signal, mask, kernel_shape, sym_upper, max_dist):
    """Second safety check of the reference (detection.py:1022): in full mode the framed mask
    also flags the first max(mk, nk) sub-diagonals of upper-symmetric maps, which must then
    hold no signal."""
    if not sym_upper:
        return
    big_k = max(kernel_shape)
    mk, nk = kernel_shape
    coo = sp.coo_matrix(signal) if sp.issparse(signal) else sp.coo_matrix(np.asarray(signal))
    off = coo.col.astype(np.int64) - coo.row.astype(np.int64) + (nk - mk)
    bad = (off <= -1) & (off >= -big_k) & (np.abs(coo.data) > 0)
    n_bad = int(np.count_nonzero(bad))
    if n_bad:
        raise ValueError("There are", n_bad, "non-zero elements reported as missing.")


# ============================================================================================
# foci picking (host)
# ============================================================================================
def label_foci(matrix):
    """Label the 4-connected components of the non-zero pixels of a sparse 0/1 matrix
    (reference detection.py:459-554).  Returns (num_foci, coo matrix of labels starting at 1);
    labels are numbered by the row-major position of each focus' first pixel."""
    mat = sp.coo_matrix(sp.csr_matrix(matrix))
    n_rows, n_cols = mat.shape
    rows, cols = mat.row.astype(np.int64), mat.col.astype(np.int64)
    n_px = rows.size
    key = rows * n_cols + cols                       # sorted: csr -> coo is row-major
    if n_px and np.any(np.diff(key) <= 0):
        raise ValueError("matrix_sp is not properly double sorted")
    # right neighbour: next pixel in row-major order on the same row, next column
    right = np.flatnonzero((np.diff(rows) == 0) & (np.diff(cols) == 1))
    # lower neighbour: pixel with key + n_cols
    pos = np.searchsorted(key, key + n_cols)
    pos_ok = pos < n_px
    has_low = np.zeros(n_px, dtype=bool)
    has_low[pos_ok] = key[pos[pos_ok]] == key[pos_ok] + n_cols
    low_src = np.flatnonzero(has_low)
    src = np.concatenate([right, low_src])
    dst = np.concatenate([right + 1, pos[low_src]])
    adj = sp.coo_matrix((np.ones(src.size, dtype=np.int8), (src, dst)), shape=(n_px, n_px))
    num_foci, labels = csgraph.connected_components(adj, directed=False)
    foci_mat = sp.coo_matrix((labels + 1, (mat.row, mat.col)), shape=(n_rows, n_cols))
    return num_foci, foci_mat


def filter_foci(foci_mat, min_size=2):
    """Erase foci with fewer than min_size pixels (reference detection.py:557-592).  Like the
    reference, operates on (and modifies) the data array of the input matrix."""
    data = foci_mat.data
    ids, sizes = np.unique(data, return_counts=True)
    small = ids[sizes < min_size]
    data[np.isin(data, small)] = 0
    filtered = foci_mat.copy()
    filtered.data = data
    filtered.eliminate_zeros()
    return int(np.count_nonzero(sizes >= min_size)), filtered


def pick_foci(mat_conv, pearson, min_size=2):
    """Threshold the coefficient map at `pearson` (>= passes), label 4-connected foci, drop foci
    of fewer than min_size pixels and return, per focus, the first pixel (row-major) holding its
    maximum coefficient (reference detection.py:387-456).  Returns (coords, labelled matrix) or
    (None, None)."""
    cand = sp.coo_matrix(mat_conv).copy()
    cand.data = np.where(cand.data < pearson, 0.0, cand.data)
    cand.data[cand.data != 0] = 1
    cand.eliminate_zeros()
    if cand.nnz == 0:
        return None, None
    _, labelled = label_foci(cand)
    num_foci, labelled = filter_foci(labelled, min_size=min_size)
    if num_foci == 0:
        return None, None
    conv = sp.csr_matrix(mat_conv)
    scores = np.asarray(conv[labelled.row, labelled.col]).ravel()
    coords = _argmax_per_label(labelled.row, labelled.col, labelled.data, scores)
    return coords, labelled


def _argmax_per_label(rows, cols, labels, scores):
    """First row-major pixel with the maximal score of each label, labels in increasing order."""
    order = np.lexsort((-scores, labels))      # stable: ties keep the row-major order
    lab_sorted = labels[order]
    first = np.concatenate([[True], lab_sorted[1:] != lab_sorted[:-1]])
    best = order[first]
    out = np.zeros((best.size, 2), dtype=int)
    out[:, 0] = rows[best]
    out[:, 1] = cols[best]
    return out


# ============================================================================================
# window validation / neighbours / pileup (host)
# ============================================================================================
def _validate(coords, matrix, scores, missing_rows, missing_cols, kernel_shape, drop, zero_tol,
              missing_tol, pad=(0, 0), stripe_k=0):
    """Shared body of validate_patterns: `scores[i]` is the coefficient at coords[i].

    pad = (rows, cols) and stripe_k describe, without materialising it, the map pattern_detector
    validates on in full mode (reference detection.py:287-310): `matrix` framed by `pad` zero rows /
    columns on every side, with NaN on the stripe_k sub-diagonals -1..-stripe_k of the framed map;
    coords and the missing rows / columns are then indices into that framed map."""
    matrix = sp.csr_matrix(matrix)
    win_h, win_w = kernel_shape
    half_h, half_w = win_h // 2 + 1, win_w // 2 + 1
    n = coords.shape[0]
    out_scores = np.full(n, np.nan)
    windows = np.full((n, win_h, win_w), np.nan)
    pad_r, pad_c = pad
    inner_shape = matrix.shape
    framed_shape = (inner_shape[0] + 2 * pad_r, inner_shape[1] + 2 * pad_c)
    miss_r = np.zeros(framed_shape[0], dtype=bool)
    miss_c = np.zeros(framed_shape[1], dtype=bool)
    miss_r[missing_rows] = True
    miss_c[missing_cols] = True
    p1 = coords[:, 0].astype(np.int64)
    p2 = coords[:, 1].astype(np.int64)
    high, low = p1 - half_h + 1, p1 + half_h
    left, right = p2 - half_w + 1, p2 + half_w
    # strict upper bounds, as in the reference (detection.py:99-104)
    inside = (high >= 0) & (low < framed_shape[0]) & (left >= 0) & (right < framed_shape[1])
    failed = ~inside
    scores = np.asarray(scores, dtype=np.float64)
    # all windows of a chunk are read with one vectorised CSR lookup instead of one sparse slice
    # per pattern (the reference slices in a Python loop, detection.py:96-141)
    dr = np.arange(low[0] - high[0] if n else 0)[None, :, None]
    dc = np.arange(right[0] - left[0] if n else 0)[None, None, :]
    todo = np.flatnonzero(inside)
    for c0 in range(0, todo.size, 16384):
        sel = todo[c0:c0 + 16384]
        rr = high[sel, None, None] + dr
        cc = left[sel, None, None] + dc
        shape = (sel.size, dr.shape[1], dc.shape[2])
        flat_r = np.broadcast_to(rr, shape).ravel()
        flat_c = np.broadcast_to(cc, shape).ravel()
        if pad_r or pad_c:
            src_r, src_c = flat_r - pad_r, flat_c - pad_c
            stored = (src_r >= 0) & (src_r < inner_shape[0]) & (src_c >= 0) & (src_c < inner_shape[1])
            win = np.asarray(matrix[np.where(stored, src_r, 0), np.where(stored, src_c, 0)], dtype=np.float64).ravel()
            win = np.where(stored, win, 0.0).reshape(shape)
        else:
            win = np.asarray(matrix[flat_r, flat_c], dtype=np.float64).reshape(shape)
        if stripe_k:
            d = np.broadcast_to(cc - rr, shape)
            win[(d <= -1) & (d >= -stripe_k)] = np.nan
        win[np.broadcast_to(miss_r[rr] | miss_c[cc], shape)] = np.nan
        tot = win[0].size
        n_zero = np.count_nonzero(win == 0, axis=(1, 2))
        n_miss = np.count_nonzero(~np.isfinite(win), axis=(1, 2))
        with np.errstate(all="ignore"):
            prop_undetected = n_miss.astype(np.float64) / np.float64(tot)
            prop_zero = n_zero.astype(np.float64) / (tot - n_miss).astype(np.float64)   # 0/0 -> nan -> rejected
        ok = (prop_undetected < missing_tol) & (prop_zero < zero_tol)
        out_scores[sel[ok]] = scores[sel[ok]]
        windows[sel[ok]] = win[ok]
        failed[sel[~ok]] = True
    table = pd.DataFrame({"bin1": coords[:, 0], "bin2": coords[:, 1], "score": out_scores})
    if drop:
        return table.loc[~failed, :], windows[~failed]
    return table, windows


def validate_patterns(coords, matrix, conv_mat, detectable_bins, kernel_matrix, drop=True,
                      zero_tol=0.3, missing_tol=0.75):
    """Drop (or flag with NaN) the patterns whose window leaves the matrix or holds too many
    missing / zero pixels, and return the window of each pattern
    (reference detection.py:18-155)."""
    matrix = sp.csr_matrix(matrix)
    coords = np.asarray(coords)
    conv = sp.csr_matrix(conv_mat)
    rr = np.clip(coords[:, 0].astype(int), 0, conv.shape[0] - 1)
    cc = np.clip(coords[:, 1].astype(int), 0, conv.shape[1] - 1)
    scores = np.asarray(conv[rr, cc]).ravel() if coords.shape[0] else np.zeros(0)
    missing_rows = preproc.valid_to_missing(detectable_bins[0], matrix.shape[0])
    missing_cols = preproc.valid_to_missing(detectable_bins[1], matrix.shape[1])
    return _validate(coords, matrix, scores, missing_rows, missing_cols, np.shape(kernel_matrix),
                     drop, zero_tol, missing_tol)


def pileup_patterns(pattern_windows):
    """Pixel-wise mean of a stack of windows, ignoring NaN (reference detection.py:158-174)."""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return np.nanmean(pattern_windows, axis=0)


def remove_neighbours(patterns, win_size=8):
    """Greedy suppression by decreasing score: a kept pattern blacklists every other pattern
    closer than win_size bins on both axes (reference detection.py:348-384).  Returns a boolean
    keep-mask aligned on the input rows.  The visiting order is the reference's own sort; the
    scan is the library's grid-hashed cs_remove_neighbours (O(n) instead of O(n^2))."""
    return remove_neighbours_arrays(patterns.bin1.to_numpy(), patterns.bin2.to_numpy(), patterns.score.to_numpy(),
                                    win_size=win_size)


def remove_neighbours_arrays(bin1, bin2, score, win_size=8):
    """remove_neighbours on plain columns."""
    from .._lib import load_library
    n = len(score)
    if n == 0:
        return np.ones(0, dtype=bool)
    # the reference's visiting order: pandas' sort of the score column, descending (detection.py:368); a one-column
    # DataFrame sort and a Series sort go through the same nargsort, so ties come out in the same order
    order = pd.Series(np.asarray(score)).sort_values(ascending=False).index.to_numpy()
    order = np.ascontiguousarray(order, dtype=np.int64)
    b1 = np.ascontiguousarray(bin1, dtype=np.int64)
    b2 = np.ascontiguousarray(bin2, dtype=np.int64)
    keep = np.zeros(n, dtype=np.uint8)
    rc = load_library().cs_remove_neighbours(b1.ctypes.data, b2.ctypes.data, order.ctypes.data, n,
                                             max(int(win_size), 1), keep.ctypes.data)
    if rc != 0:
        raise ValueError("cs_remove_neighbours: bad arguments")
    return keep.astype(bool)


# ============================================================================================
# pattern_detector
# ============================================================================================
def pattern_detector(contact_map, kernel_config, kernel_matrix, coords=None, dump=None, full=False,
                     tsvd=None):
    """Detect patterns in one contact map (or, with `coords`, quantify the given positions)
    (reference detection.py:177-345).

    contact_map needs the attributes `matrix` (scipy sparse), `detectable_bins` (pair of index
    arrays -- boolean masks are accepted when full=False, as in the reference), `max_dist`,
    `inter` (and `name` when dumping).  Returns (DataFrame[bin1, bin2, score, pvalue],
    windows (n, km, kn)) or (None, None)."""
    kernel_matrix = np.asarray(kernel_matrix, dtype=np.float64)
    km, kn = kernel_matrix.shape
    kh, kw = (km - 1) // 2, (kn - 1) // 2
    run_mode = "detect" if coords is None else "quantify"
    matrix = contact_map.matrix
    ms, ns = matrix.shape
    if min(ms, ns) <= max(km, kn):
        return None, None
    inter = bool(contact_map.inter)
    max_dist = contact_map.max_dist
    missing_tol = kernel_config["max_perc_undetected"] / 100
    zero_tol = kernel_config["max_perc_zero"] / 100
    pearson = kernel_config["pearson"]
    sym_upper = not inter

    dev = get_device()
    kspec = engine.KernelSpec(kernel_matrix, tsvd)
    csr = sp.csr_matrix(matrix)
    if dump:
        return _pattern_detector_dump(contact_map, kernel_config, kernel_matrix, coords, dump, full, tsvd)

    out_range = None
    if not inter and max_dist is not None:
        out_range = (0, max_dist)       # diag_trim(mat_conv, max_dist), detection.py:269-270
    if full and sym_upper:
        # normxcorr2's second safety check (detection.py:1022): stored pixels on the sub-diagonals the framed mask
        # flags.  Upper-triangular maps can only trip it with a template taller than wide.
        rng = engine.diag_range(csr)
        if rng is not None and rng[0] + (kn - km) <= -1:
            _check_framed_signal(csr, None, (km, kn), sym_upper, max_dist)
    st = _Staged(dev, csr, (km, kn), sym_upper, full, out_diag_range=out_range)
    miss_r = miss_c = None
    if full:
        miss_r = dev.to_device(preproc.missing_flags(contact_map.detectable_bins[0], ms))
        miss_c = dev.to_device(preproc.missing_flags(contact_map.detectable_bins[1], ns))
    return detect_on_device(dev, st.sig, (ms, ns), kspec, kernel_config, miss_r, miss_c, inter=inter,
                            max_dist=max_dist, full=full, coords=coords)


def detect_on_device(dev, sig, shape, kspec, kernel_config, miss_row, miss_col, *, inter, max_dist, full,
                     coords=None, want_windows=True, raw=False, stream=None, defer=False):
    """The part of pattern_detector that follows the staging of the contact map in HBM
    (reference detection.py:240-345): correlation, foci, validation statistics -- one native call
    (cs_detect_foci / cs_quantify_pixels) -- then the acceptance rules on the few returned records.

    sig: CsMatrix of the staged (detrended) map; miss_row / miss_col: uint8 device flags of the
    undetectable bins (None when full is False).  Returns (table, windows) or (None, None); with
    raw=True the table is a (n, 4) float64 array (bin1, bin2, score, pvalue) instead of a DataFrame
    (the whole-genome drivers call this once per block and template: pandas would cost more than the
    GPU work of a small block)."""
    ms, ns = shape
    km, kn = kspec.km, kspec.kn
    kh, kw = (km - 1) // 2, (kn - 1) // 2
    missing_tol = kernel_config["max_perc_undetected"] / 100
    zero_tol = kernel_config["max_perc_zero"] / 100
    pearson = kernel_config["pearson"]
    sym_upper = not inter
    diag_only = (not inter) and kernel_config["max_dist"] == 0       # 1-D patterns live on the diagonal (:311-315)
    lo_diag, hi_diag = -(ms - 1), ns - 1
    if not inter:
        lo_diag = 0
        if max_dist is not None:
            hi_diag = max_dist
    mask_kw = dict(mask_mode=MASK_NONE)
    if full:
        mask_kw = dict(mask_mode=MASK_BINS, miss_row=miss_row, miss_col=miss_col)
    common = dict(inter=inter, full=full, sym_upper=sym_upper, max_dist=max_dist, missing_tol=missing_tol,
                  want_windows=want_windows, stream=stream, **mask_kw)
    if coords is None:
        run_mode = "detect"
        rec, windows = engine.run_detect_foci(dev, sig, shape, kspec, pearson=pearson, lo_diag=lo_diag, hi_diag=hi_diag,
                                              diag_only=diag_only, **common)
        if defer:
            # the caller applies the acceptance rules to the records of all its blocks at once (accept_many)
            return rec, windows
        if rec.shape[0] == 0:
            return None, None
    else:
        run_mode = "quantify"
        coords_in = coords
        pts = np.array(coords_in, dtype=int, copy=True)
        if diag_only:
            pts[:, 0] = pts[:, 1] + ((kw - kh) if full else 0)      # forced on the diagonal AFTER the (kh, kw) shift
        rr, cc = pts[:, 0].astype(np.int64), pts[:, 1].astype(np.int64)
        big = np.iinfo(np.int32).max // 2
        rec, windows = engine.run_quantify_pixels(dev, sig, shape, kspec, np.clip(rr, -big, big), np.clip(cc, -big, big),
                                                  **common)
        # the reference shifts the caller's array in place when coords are given (:297-298)
        if full:
            try:
                coords_in[:, 0] += kh
                coords_in[:, 1] += kw
                if diag_only:
                    coords_in[:, 0] = coords_in[:, 1]
            except (TypeError, ValueError, IndexError):
                pass
    if run_mode == "detect":
        rr, cc = rec["bin1"].astype(np.int64), rec["bin2"].astype(np.int64)
    conv = _offset_scores(dev, sig, shape, kspec, rr, cc, common) if (full and km != kn) else None
    return _accept_records(rec, windows, rr, cc, run_mode, shape, kspec, kernel_config, inter=inter, max_dist=max_dist,
                           full=full, raw=raw, conv=conv)


def quantify_many_on_device(dev, blocks, kspec, kernel_config, coords_list, *, want_windows=True, stream=None):
    """detect_on_device(coords=...) -- `quantify`, pattern_detector with given coordinates in full mode (reference
    detection.py:277, 297-298) -- for MANY staged sub-matrices (pipeline.StagedBlock; intra and inter mixed) with ONE native
    call per template (cs_quantify_blocks) instead of one per sub-matrix and template.  coords_list[b]: (n_b, 2) integer
    array of block-local (row, col) bins.  Returns (table (n, 4): bin1, bin2, score, pvalue -- the rows of the blocks one after
    the other, in input order --, windows or None), or None when the batch does not apply (non-square template: the
    caller goes block by block).  kspec may be a LIST of templates of one size: the coordinate lists and the per-record
    geometry are then prepared once (they cost as much as a template's native call) and the result is a list of
    (table, windows), one per template."""
    templates = list(kspec) if isinstance(kspec, (list, tuple)) else [kspec]
    km, kn = templates[0].km, templates[0].kn
    if km != kn or not blocks or any((t.km, t.kn) != (km, kn) for t in templates):
        return None
    kh, kw = (km - 1) // 2, (kn - 1) // 2
    live = [b for b in range(len(blocks)) if min(blocks[b].shape) > max(km, kn)]     # (:237-238: smaller blocks are skipped)
    counts = [len(c) for c in coords_list]
    total = int(sum(counts))

    def empty():
        return np.full((total, 4), np.nan), (np.full((total, km, kn), np.nan) if want_windows else None)
    if total == 0 or not live:
        out = [empty() for _ in templates]
        return out if isinstance(kspec, (list, tuple)) else out[0]
    starts = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    diag_only = kernel_config["max_dist"] == 0
    big = np.iinfo(np.int32).max // 2
    # the coordinate lists of the live blocks one after the other (every block is live in a normal run: then `where` is the
    # identity and the results are handed on as they come, without a 2.4 MB scatter per template)
    pts = np.concatenate([np.asarray(coords_list[b], dtype=np.int64).reshape(-1, 2) for b in live])
    n_live = np.array([counts[b] for b in live], dtype=np.int64)
    blk = np.repeat(np.arange(len(live), dtype=np.int32), n_live)
    used = [blocks[b] for b in live]
    rr, cc = pts[:, 0].copy(), pts[:, 1].copy()
    if diag_only:
        on_diag = ~np.array([bool(b.inter) for b in used])[blk]
        rr[on_diag] = cc[on_diag] + (kw - kh)                  # forced on the diagonal AFTER the (kh, kw) shift (:311-315)
    everything = len(live) == len(blocks)
    where = None if everything else np.concatenate([np.arange(starts[b], starts[b + 1]) for b in live])
    rr_c, cc_c = np.clip(rr, -big, big), np.clip(cc, -big, big)
    # per-record geometry of the acceptance rules (_accept_records takes arrays)
    ms = np.array([b.shape[0] for b in used], dtype=np.int64)[blk]
    ns = np.array([b.shape[1] for b in used], dtype=np.int64)[blk]
    inter = np.array([bool(b.inter) for b in used])[blk]
    md = np.array([(-1 if b.max_dist is None else b.max_dist) for b in used], dtype=np.int64)[blk]
    md = np.where(md < 0, np.iinfo(np.int64).max // 4, md)
    out = []
    for t in templates:
        rec, win = engine.run_quantify_blocks(dev, used, t, blk, rr_c, cc_c, missing_tol=kernel_config["max_perc_undetected"] / 100,
                                              want_windows=want_windows, stream=stream)
        rec4, win = _accept_records(rec, win, rr, cc, "quantify", (ms, ns), t, kernel_config, inter=inter, max_dist=md, full=True,
                                    raw=True)
        if everything:
            out.append((rec4, win if want_windows else None))
        else:
            table, windows = empty()
            table[where] = rec4
            if want_windows:
                windows[where] = win
            out.append((table, windows))
    return out if isinstance(kspec, (list, tuple)) else out[0]


def _offset_scores(dev, sig, shape, kspec, rr, cc, common):
    """Non-square templates in full mode: pattern_detector pads the coefficient map by (kw rows, kh columns)
    (zero_pad_sparse(mat_conv, kh, kw), preprocessing.py:636-676) yet shifts the coordinates by (kh, kw)
    (detection.py:287-298), so validate_patterns reads the score of pattern (r, c) at (r + kh - kw, c + kw - kh).
    Returns (rows, cols, float64 coefficients there)."""
    d = (kspec.km - 1) // 2 - (kspec.kn - 1) // 2
    big = np.iinfo(np.int32).max // 2
    sr, sc = np.clip(rr + d, -big, big), np.clip(cc - d, -big, big)
    keys = ("full", "sym_upper", "max_dist", "mask_mode", "miss_row", "miss_col", "missing_tol", "stream")
    vals, _ = engine.run_rescore(dev, sig, shape, kspec, sr, sc, **{k: common[k] for k in keys if k in common})
    return sr, sc, vals


def detect_many_on_device(dev, blocks, kspec, kernel_config, *, want_windows=True, raw=False, stream=None, defer=False, merged=False,
                          begin_only=False):
    """detect_on_device for a 1-D pattern (kernel_config["max_dist"] == 0: borders, hairpins) on many intra
    sub-matrices at once: `blocks` = objects with sig, shape, miss_row, miss_col, max_dist (pipeline.StagedBlock).
    One native call (cs_detect_foci_batch) instead of one launch chain and synchronisation per sub-matrix.
    Returns the list of (table, windows) detect_on_device would return, or None if the batch entry does not
    apply (caller falls back to one call per block).  kspec = a list of templates of one size (raw=True): all of them in
    the same launch chain (cs_detect_foci_batch_templates); the result is then a list over the templates."""
    templates = list(kspec) if isinstance(kspec, (list, tuple)) else None      # several templates of one size: ONE launch chain
    if templates is not None:
        if not raw or not templates:
            return None
        kspec = templates[0]
    if kernel_config["max_dist"] != 0 or not blocks or kspec.km != kspec.kn:
        return None
    if any(b.inter or b.max_dist is None or getattr(b, "row_window", None) is not None for b in blocks):
        return None
    missing_tol = kernel_config["max_perc_undetected"] / 100
    if templates is not None:
        res = engine.run_detect_foci_batch(
            dev, [b.sig for b in blocks], [b.shape for b in blocks], templates, pearson=kernel_config["pearson"],
            hi_diags=[b.max_dist for b in blocks], inter=False, diag_only=True, max_dists=[b.max_dist for b in blocks],
            miss_rows=[b.miss_row for b in blocks], miss_cols=[b.miss_col for b in blocks], missing_tol=missing_tol,
            want_windows=want_windows, stream=stream, flat=True, begin_only=begin_only)
        if res is None:
            return None
        n = len(blocks)

        def per_template():
            # (begin_only: the chain is enqueued; res() waits for it and ends the native call)
            rec, windows, counts = res() if begin_only else res
            # the acceptance rules of ALL templates in one native call (cs_accept_records threads beyond 2000 records: three
            # calls of ~ 60 us each on a rank's share of a genome were a sixth of its step), cut per template afterwards
            n_t = len(templates)
            cnt = np.ascontiguousarray(counts, dtype=np.int64)
            rec4, ok, kept = accept_native(rec, cnt, [b.shape for b in blocks] * n_t, [b.max_dist for b in blocks] * n_t, templates[0],
                                           kernel_config, inter=False, full=True, compact=True, pvals=True)
            win_ok = windows[ok] if windows is not None else None
            cut = np.concatenate([[0], np.cumsum(kept)])
            out = []
            for t in range(n_t):
                a, z = int(cut[t * n]), int(cut[(t + 1) * n])
                kept_t = kept[t * n:(t + 1) * n]
                win_t = win_ok[a:z] if win_ok is not None else None
                if merged:
                    out.append((rec4[a:z], kept_t, win_t))
                    continue
                cuts = np.cumsum(kept_t)[:-1]
                parts = np.split(rec4[a:z], cuts)
                wparts = np.split(win_t, cuts) if win_t is not None else [None] * n
                out.append([(None, None) if c == 0 else (tb, w) for c, tb, w in zip(cnt[t * n:(t + 1) * n], parts, wparts)])
            return out

        return per_template if (defer or begin_only) else per_template()
    res = engine.run_detect_foci_batch(
        dev, [b.sig for b in blocks], [b.shape for b in blocks], kspec, pearson=kernel_config["pearson"],
        hi_diags=[b.max_dist for b in blocks], inter=False, diag_only=True, max_dists=[b.max_dist for b in blocks],
        miss_rows=[b.miss_row for b in blocks], miss_cols=[b.miss_col for b in blocks], missing_tol=missing_tol,
        want_windows=want_windows, stream=stream, flat=raw)
    if res is None:
        return None
    if raw:
        rec, windows, counts = res
        if defer:           # the native call is done; the acceptance rules (numpy) when the caller asks for them
            return lambda: accept_many(blocks, rec, windows, counts, kspec, kernel_config, merged=merged, pvals=True)
        return accept_many(blocks, rec, windows, counts, kspec, kernel_config, merged=merged, pvals=True)
    out = []
    for b, (rec, windows) in zip(blocks, res):
        if rec.shape[0] == 0:
            out.append((None, None))
            continue
        rr, cc = rec["bin1"].astype(np.int64), rec["bin2"].astype(np.int64)
        out.append(_accept_records(rec, windows, rr, cc, "detect", b.shape, kspec, kernel_config, inter=False,
                                   max_dist=b.max_dist, full=True, raw=raw))
    return out


def detect_blocks_on_device(dev, blocks, kspec, kernel_config, *, want_windows=True, stream=None, defer=False, merged=False,
                            exclusive=False):
    """detect_on_device (raw tables) for a 2-D pattern on many banded intra sub-matrices with ONE native call
    (cs_detect_foci_blocks: tile kernels in candidate mode -> one candidate list -> one foci chain): `blocks` = objects with
    sig, sig32, shape, miss_row, miss_col, max_dist (pipeline.StagedBlock).  Returns the list of raw (table, windows) per
    block, or None when the batch entry does not apply (caller: one call per block).  exclusive: no other launch chain runs
    beside this call (cs_foci_params.exclusive: one persistent launch for the tiles of all blocks)."""
    if not blocks or kspec.km != kspec.kn or kernel_config["max_dist"] == 0:
        return None
    if any(b.inter or b.max_dist is None or getattr(b, "row_window", None) is not None or b.sig.layout not in (LAYOUT_BAND, LAYOUT_BAND_LAZY, LAYOUT_BAND_PADDED) for b in blocks):
        return None
    if any(b.sig.layout == LAYOUT_BAND_LAZY and getattr(b, "sig32", None) is None for b in blocks):
        return None
    res = engine.run_detect_foci_blocks(
        dev, [b.sig for b in blocks], [getattr(b, "sig32", None) for b in blocks], [b.shape for b in blocks], kspec,
        pearson=kernel_config["pearson"], lo_diags=[0] * len(blocks), hi_diags=[b.max_dist for b in blocks], inter=False,
        diag_only=False, max_dists=[b.max_dist for b in blocks], miss_rows=[b.miss_row for b in blocks],
        miss_cols=[b.miss_col for b in blocks], missing_tol=kernel_config["max_perc_undetected"] / 100,
        want_windows=want_windows, stream=stream, exclusive=exclusive)
    if res is None:
        return None
    rec, windows, counts = res
    if int(np.sum(counts)) == 0 and not merged:
        return (lambda: [(None, None)] * len(blocks)) if defer else [(None, None)] * len(blocks)
    if defer:
        return lambda: accept_many(blocks, rec, windows, counts, kspec, kernel_config, merged=merged, pvals=True)
    return accept_many(blocks, rec, windows, counts, kspec, kernel_config, merged=merged, pvals=True)


def accept_many(blocks, rec, windows, counts, kspec, kernel_config, merged=False, pvals=False):
    """The acceptance rules (detect mode, full maps) on the records of SEVERAL intra sub-matrices at once -- `rec` /
    `windows`: the records of `blocks` one block after the other, `counts` per block -- then cut at the block
    boundaries: the list of raw (table, windows) per block.  (23 numpy passes over a few hundred records each cost
    more than the native calls that produced them; the rules themselves run in the library: cs_accept_records.)
    merged=True: no cutting -- (table of all blocks, accepted records per block, windows)."""
    counts = np.ascontiguousarray(counts, dtype=np.int64)
    rec4, ok, kept = accept_native(rec, counts, [b.shape for b in blocks], [b.max_dist for b in blocks], kspec, kernel_config,
                                   inter=False, full=True, compact=True, pvals=pvals)
    if windows is not None:
        windows = windows[ok]
    if merged:
        return rec4, kept, windows
    cuts = np.cumsum(kept)[:-1]
    parts = np.split(rec4, cuts)
    wparts = np.split(windows, cuts) if windows is not None else [None] * len(blocks)
    return [(None, None) if n == 0 else (t, w) for n, t, w in zip(counts, parts, wparts)]


def device_pvalues():
    """False with CHROMOSIGHT_HIP_HOST_PVALUES=1: cs_accept_records computes the p-values of device records itself (the
    cross-check of cs_focus.pval; tests/test_gpu_device_pipeline.py)."""
    return not os.environ.get("CHROMOSIGHT_HIP_HOST_PVALUES")


def accept_native(rec, counts, shapes, max_dists, kspec, kernel_config, *, inter, full, compact, pvals=False):
    """cs_accept_records on the records of len(counts) sub-matrices: (table (k, 4), accepted mask over the records,
    accepted per sub-matrix); what _accept_records computes with numpy (and is tested against).  pvals: the records come from
    a device entry of the library and carry their p-values (cs_focus.pval)."""
    from .._lib import load_library
    rec = np.ascontiguousarray(rec)
    n = int(rec.shape[0])
    nb = len(counts)
    geo = np.empty((3, max(nb, 1)), dtype=np.int32)
    geo[0, :nb] = [s[0] for s in shapes]
    geo[1, :nb] = [s[1] for s in shapes]
    geo[2, :nb] = [-1 if m is None else m for m in max_dists]
    table = np.empty((n, 4), dtype=np.float64)
    ok = np.empty(n, dtype=np.uint8)
    kept = np.empty(max(nb, 1), dtype=np.int64)
    rc = load_library().cs_accept_records(
        rec.ctypes.data, nb, counts.ctypes.data, geo[0].ctypes.data, geo[1].ctypes.data, geo[2].ctypes.data, int(bool(inter)),
        int(kspec.km), int(kspec.kn), kernel_config["max_perc_undetected"] / 100, kernel_config["max_perc_zero"] / 100,
        int(bool(full)), int(bool(compact)) | (2 if pvals and device_pvalues() else 0), table.ctypes.data, ok.ctypes.data, kept.ctypes.data)
    if rc:
        raise ValueError(f"cs_accept_records: bad arguments ({rc})")
    kept = kept[:nb]
    return (table[:int(kept.sum())] if compact else table), ok.view(np.bool_), kept


def _accept_records(rec, windows, rr, cc, run_mode, shape, kspec, kernel_config, *, inter, max_dist, full, raw, return_ok=False,
                    conv=None):
    """The acceptance rules of pattern_detector / validate_patterns on the records the device returned
    (reference detection.py:121-141, 269-270, 332-336): (table, windows).  `shape` and `max_dist` may be
    per-record arrays (records of several sub-matrices in one call).  conv = (rows, cols, coefficients): where the
    reference reads the score when that is not (bin1, bin2) (_offset_scores)."""
    ms, ns = shape
    km, kn = kspec.km, kspec.kn
    missing_tol = kernel_config["max_perc_undetected"] / 100
    zero_tol = kernel_config["max_perc_zero"] / 100
    r64, nobs = rec["score"], rec["n_obs"]
    # coefficient on the trimmed map (:269-270), p-value on the untrimmed one (:332-336)
    sr, sc, sv = (rr, cc, r64) if conv is None else conv
    in_band = (sr >= 0) & (sr < ms) & (sc >= 0) & (sc < ns)
    if isinstance(inter, np.ndarray):                          # records of intra and inter sub-matrices in one call
        d = sc - sr
        in_band &= inter | ((d >= 0) & (d <= max_dist))
    elif not inter:
        d = sc - sr
        in_band &= (d >= 0) & ((d <= max_dist) if max_dist is not None else True)
    conv_at = np.where(in_band, sv, 0.0)
    n_obs = nobs if full else np.full(r64.shape, float(km * kn))
    with np.errstate(all="ignore"):
        logp = np.where(r64 != 0, cus.corr_to_pval(r64, np.where(n_obs == 0, km * kn, n_obs)), 0.0)
    # acceptance rules of validate_patterns (:121-141) on the window statistics
    tot = km * kn
    n_zero, n_miss = rec["n_zero"], rec["n_missing"]
    with np.errstate(all="ignore"):
        prop_undetected = n_miss.astype(np.float64) / np.float64(tot)
        prop_zero = n_zero.astype(np.float64) / (tot - n_miss).astype(np.float64)   # 0/0 -> nan -> rejected
    ok = (rec["inside"] != 0) & (prop_undetected < missing_tol) & (prop_zero < zero_tol)
    scores = np.where(ok, conv_at, np.nan)
    if windows is not None:
        windows[~ok] = np.nan
    if raw:
        with np.errstate(all="ignore"):
            rec4 = np.column_stack([rr.astype(np.float64), cc.astype(np.float64), scores, 10.0 ** logp])
        if run_mode == "detect":
            rec4 = rec4[ok]
            windows = windows[ok] if windows is not None else None
        return (rec4, windows, ok) if return_ok else (rec4, windows)
    table = pd.DataFrame({"bin1": rr, "bin2": cc, "score": scores})
    if run_mode == "detect":
        table = table.loc[ok, :]
        windows = windows[ok] if windows is not None else None
    kept = table.index.to_numpy()
    table["pvalue"] = 10 ** logp[kept] if len(kept) else None
    return table, windows


def detect_split_on_device(dev, sig, shape, row_window, kspec, kernel_config, miss_row, miss_col, *, max_dist, all_gather,
                           full=True, want_windows=True, raw=False, stream=None):
    """detect mode of ONE intra sub-matrix split over several GPUs by row windows.
    This rank holds the rows row_window = (a, b) of the staged map plus the template's halo (`sig`,
    CsMatrix.row0) and does the correlation, thresholding and float64 re-scoring of its rows
    (cs_candidates); the candidate pixels of all ranks are exchanged (`all_gather(array2d)` returns the
    rank-order concatenation on every rank -- a few thousand (row, col, value) triples) and every rank
    labels the same merged list (cs_label_foci), so foci that straddle a cut come out exactly as on one
    GPU; each rank then scores the foci whose final row it owns (cs_quantify_pixels) and the records are
    gathered.  Returns what detect_on_device returns for the whole sub-matrix, on every rank."""
    ms, ns = shape
    a, b = int(row_window[0]), int(row_window[1])
    km, kn = kspec.km, kspec.kn
    kk = km * kn
    missing_tol = kernel_config["max_perc_undetected"] / 100
    diag_only = kernel_config["max_dist"] == 0
    lo_diag, hi_diag = 0, (max_dist if max_dist is not None else ns - 1)
    mask_kw = dict(mask_mode=MASK_BINS, miss_row=miss_row, miss_col=miss_col) if full else dict(mask_mode=MASK_NONE)
    common = dict(inter=False, full=full, sym_upper=True, max_dist=max_dist, missing_tol=missing_tol, stream=stream,
                  **mask_kw)
    rows, cols, vals = engine.run_candidates(dev, sig, shape, kspec, (a, b), pearson=kernel_config["pearson"],
                                             lo_diag=lo_diag, hi_diag=hi_diag, **common)
    cand = all_gather(np.column_stack([rows.astype(np.float64), cols.astype(np.float64), vals]).reshape(-1, 3))
    if cand.shape[0] == 0:
        return None, None
    diag_code = (2 * ((kn - 1) // 2 - (km - 1) // 2) + 1 if full else 1) if diag_only else 0
    f_rows, f_cols, f_size = engine.run_label_foci(dev, shape, cand[:, 0].astype(np.int32), cand[:, 1].astype(np.int32),
                                                   cand[:, 2], min_size=2, diag_only=diag_code, stream=stream)
    order = np.arange(f_rows.size)
    mine = (f_rows >= a) & (f_rows < b)
    width = len(FOCUS_DTYPE.names) + 1 + (kk if want_windows else 0)
    local = np.zeros((int(mine.sum()), width))
    if mine.any():
        rec, win = engine.run_quantify_pixels(dev, sig, shape, kspec, f_rows[mine], f_cols[mine],
                                              want_windows=want_windows, **common)
        rec["focus_size"] = f_size[mine]
        local[:, 0] = order[mine]
        for k, name in enumerate(FOCUS_DTYPE.names):
            local[:, 1 + k] = rec[name]
        if want_windows:
            local[:, 1 + len(FOCUS_DTYPE.names):] = win.reshape(-1, kk)
    merged = all_gather(local)
    if merged.shape[0] == 0:
        return None, None
    merged = merged[np.argsort(merged[:, 0], kind="stable")]
    rec = np.zeros(merged.shape[0], dtype=FOCUS_DTYPE)
    for k, name in enumerate(FOCUS_DTYPE.names):
        rec[name] = merged[:, 1 + k]
    windows = merged[:, 1 + len(FOCUS_DTYPE.names):].reshape(-1, km, kn).copy() if want_windows else None
    rr, cc = rec["bin1"].astype(np.int64), rec["bin2"].astype(np.int64)
    if full and km != kn:
        raise ValueError("a sub-matrix split by rows needs a square template in full mode")
    return _accept_records(rec, windows, rr, cc, "detect", shape, kspec, kernel_config, inter=False, max_dist=max_dist,
                           full=full, raw=raw)


def _pattern_detector_dump(contact_map, kernel_config, kernel_matrix, coords, dump, full, tsvd):
    """pattern_detector with the reference's intermediate dumps (03_normxcorr2, 04_diag_trim,
    05_foci; reference detection.py:227-231, 265, 272, 285): materialises the whole coefficient
    map on the host, so it is the slow, inspectable variant."""
    class _NoDump:
        pass
    shadow = _NoDump()
    for attr in ("matrix", "detectable_bins", "max_dist", "inter"):
        setattr(shadow, attr, getattr(contact_map, attr))
    km, kn = kernel_matrix.shape

    def save(base, mat):
        sp.save_npz(pathlib.Path(dump) / f"{contact_map.name}_{base}", mat)

    mask = None
    if full:
        mask = preproc.make_missing_mask(
            contact_map.matrix.shape, valid_rows=contact_map.detectable_bins[0],
            valid_cols=contact_map.detectable_bins[1], max_dist=contact_map.max_dist,
            sym_upper=not contact_map.inter)
    conv, _ = normxcorr2(sp.csr_matrix(contact_map.matrix), kernel_matrix, max_dist=contact_map.max_dist,
                         sym_upper=not contact_map.inter, full=full, missing_mask=mask, tsvd=tsvd,
                         pval=False, missing_tol=kernel_config["max_perc_undetected"] / 100)
    save("03_normxcorr2", conv)
    conv.data[np.isnan(conv.data)] = 0
    if not contact_map.inter:
        conv = preproc.diag_trim(conv.tocsr(), contact_map.max_dist)
        save("04_diag_trim", conv)
    if coords is None:
        conv = conv.tocoo()
        conv.eliminate_zeros()
        _, foci = pick_foci(conv, kernel_config["pearson"])
        if foci is not None:
            save("05_foci", foci)
    return pattern_detector(shadow, kernel_config, kernel_matrix, coords=coords, dump=None, full=full, tsvd=tsvd)
