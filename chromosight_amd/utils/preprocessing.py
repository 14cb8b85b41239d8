"""Preprocessing half of the hot path, with the reference's call surface
(chromosight/utils/preprocessing.py): `distance_law` and `detrend` run on the MI355X through
libchromosight_hip.so; the mask / trim / pad / template helpers are small host-side index
manipulations that the detection path and API users need alongside them.

Each function cites the reference lines whose behaviour it reproduces; the parity tests in
tests/ compare against vectors generated from those lines (tests/golden/make_golden.py).
"""
import sys

import numpy as np
import numpy.linalg as la
import scipy.ndimage as ndi
import scipy.sparse as sp

from .. import engine
from .._lib import LAYOUT_BAND, get_device


# --------------------------------------------------------------------------------------------
# index helpers
# --------------------------------------------------------------------------------------------
def valid_to_missing(valid, size):
    """Indices of the bins that are NOT in `valid` (reference preprocessing.py:850-875).
    `valid` may be an index array or a boolean mask, as in the reference."""
    flag = np.ones(size, dtype=bool)
    try:
        flag[valid] = False
    except IndexError:  # no valid index at all
        pass
    return np.flatnonzero(flag)


def missing_flags(valid, size):
    """uint8 vector, 1 where the bin is missing (device-side form of valid_to_missing)."""
    flag = np.ones(size, dtype=np.uint8)
    try:
        flag[valid] = 0
    except IndexError:
        pass
    return flag


def set_mat_diag(mat, diag=0, val=0):
    """Set one diagonal of a square dense array in place (reference preprocessing.py:71-90)."""
    m = mat.shape[0]
    mat.flat[diag:m ** 2 - diag * m:m + 1] = val


def diag_trim(mat, n):
    """Keep diagonals 0..n of the upper triangle of a CSR matrix; for a dense array, zero the
    upper diagonals n, n+1, ... (reference preprocessing.py:93-126, including the reference's
    off-by-one between the two container types)."""
    if sp.issparse(mat):
        if mat.format != "csr":
            raise ValueError("input type must be scipy.sparse.csr_matrix")
        coo = mat.tocoo()
        off = coo.col.astype(np.int64) - coo.row.astype(np.int64)
        keep = (off >= 0) & (off <= n)
        out = sp.coo_matrix((coo.data[keep], (coo.row[keep], coo.col[keep])), shape=mat.shape)
        return out.tocsr()
    trimmed = mat.copy()
    for d in range(n, trimmed.shape[0]):
        set_mat_diag(trimmed, d, 0)
    return trimmed


def zero_pad_sparse(mat, margin_h, margin_v, fmt="coo"):
    """Add margin_h empty columns left and right and margin_v empty rows above and below
    (reference preprocessing.py:636-676)."""
    coo = sp.coo_matrix(mat)
    sm, sn = coo.shape
    out = sp.coo_matrix(
        (coo.data, (coo.row + margin_v, coo.col + margin_h)),
        shape=(sm + 2 * margin_v, sn + 2 * margin_h), dtype=coo.dtype,
    )
    return out.asformat(fmt)


# --------------------------------------------------------------------------------------------
# missing-pixel masks (host containers; the device evaluates the same predicate analytically,
# chromosight_amd/csrc/cs_device.h missing_pred)
# --------------------------------------------------------------------------------------------
def make_missing_mask(shape, valid_rows, valid_cols, max_dist=None, sym_upper=False):
    """CSR bool mask, True on pixels of missing bins (reference preprocessing.py:535-633).
    Upper-symmetric matrices are only flagged on diagonals 0..max_dist of the upper triangle."""
    sm, sn = shape
    if sym_upper and (sm != sn or len(valid_rows) != len(valid_cols)):
        raise ValueError("Rectangular matrices cannot be upper symmetric")
    miss_r = valid_to_missing(valid_rows, sm)
    miss_c = miss_r if sym_upper else valid_to_missing(valid_cols, sn)
    if not sym_upper:
        flag_r = np.zeros(sm, dtype=bool)
        flag_c = np.zeros(sn, dtype=bool)
        flag_r[miss_r] = True
        flag_c[miss_c] = True
        rows_a = np.repeat(miss_r, sn)
        cols_a = np.tile(np.arange(sn), miss_r.size)
        ok_rows = np.flatnonzero(~flag_r)
        rows_b = np.tile(ok_rows, miss_c.size)
        cols_b = np.repeat(miss_c, ok_rows.size)
        rows = np.concatenate([rows_a, rows_b])
        cols = np.concatenate([cols_a, cols_b])
        return sp.csr_matrix((np.ones(rows.size, dtype=bool), (rows, cols)), shape=shape, dtype=bool)
    md = min(shape) if max_dist is None else max_dist
    shift = np.arange(md + 1)
    # pixels above each missing bin (same column) and to its right (same row)
    up_r = (miss_r[:, None] - shift[None, :]).ravel()
    up_c = np.repeat(miss_r, md + 1)
    rt_r = np.repeat(miss_r, md + 1)
    rt_c = (miss_r[:, None] + shift[None, :]).ravel()
    rows = np.concatenate([up_r, rt_r])
    cols = np.concatenate([up_c, rt_c])
    ok = (rows >= 0) & (cols < sm)
    mask = sp.coo_matrix((np.ones(int(ok.sum()), dtype=bool), (rows[ok], cols[ok])), shape=shape, dtype=bool)
    return mask.tocsr()


def frame_missing_mask(mask, kernel_shape, sym_upper=False, max_dist=None):
    """Surround a missing mask with the (mk-1, nk-1) frame used by 'full' mode
    (reference preprocessing.py:404-498): all four margins flagged, except for upper-symmetric
    maps with a max_dist, where only the parts of the top and right margins a band window can
    reach are flagged; upper-symmetric maps also get the first max(mk, nk) sub-diagonals."""
    if mask.dtype != bool:
        raise ValueError("Mask must contain boolean values")
    if not sp.issparse(mask):
        raise ValueError("Mask must be a sparse matrix")
    ms, ns = mask.shape
    mk, nk = kernel_shape
    H, W = ms + 2 * (mk - 1), ns + 2 * (nk - 1)
    banded = sym_upper and (max_dist is not None)
    inner = mask.tocsr()
    if banded:
        inner = diag_trim(inner, max_dist + max(nk, mk))
    inner = inner.tocoo()
    rows = [inner.row[inner.data != 0] + (mk - 1)]
    cols = [inner.col[inner.data != 0] + (nk - 1)]

    def add_block(r0, r1, c0, c1):
        if r1 > r0 and c1 > c0:
            rr, cc = np.meshgrid(np.arange(r0, r1), np.arange(c0, c1), indexing="ij")
            rows.append(rr.ravel())
            cols.append(cc.ravel())

    if banded:
        add_block(0, mk - 1, nk - 1, nk - 1 + min(max_dist + nk, ns))       # top, up to scan distance
        add_block(max(H - (max_dist + mk + 1), 0), H, nk - 1 + ns, W)       # right, last rows
        add_block(0, mk - 1, 0, nk - 1)                                     # top-left corner
    else:
        add_block(0, mk - 1, 0, W)
        add_block(mk - 1 + ms, H, 0, W)
        add_block(0, H, 0, nk - 1)
        add_block(0, H, nk - 1 + ns, W)
    if sym_upper:
        for off in range(1, max(mk, nk) + 1):
            r = np.arange(off, min(H, W + off))
            rows.append(r)
            cols.append(r - off)
    rows = np.concatenate(rows)
    cols = np.concatenate(cols)
    framed = sp.coo_matrix((np.ones(rows.size, dtype=bool), (rows, cols)), shape=(H, W), dtype=bool)
    return framed.tocsr()


def check_missing_mask(signal, mask):
    """Raise ValueError when a pixel flagged missing holds a non-zero signal
    (reference preprocessing.py:501-532)."""
    if sp.issparse(mask):
        mrow, mcol = mask.nonzero()
        if mrow.size == 0:
            return
        vals = np.asarray(sp.csr_matrix(signal)[mrow, mcol]).ravel() if sp.issparse(signal) \
            else np.asarray(signal)[mrow, mcol]
        n_bad = int(np.count_nonzero(np.abs(vals) > 0))
        if n_bad > 0:
            raise ValueError("There are", n_bad, "non-zero elements reported as missing.")
    else:
        total = np.sum(np.abs(np.asarray(signal)[np.asarray(mask) > 0]))
        if total > 1e-10:
            raise ValueError("There are", str(total), "non-zero elements reported as missing.")


# --------------------------------------------------------------------------------------------
# distance law and detrend -- device
# --------------------------------------------------------------------------------------------
def _isotonic_non_increasing(y):
    """Least-squares non-increasing fit by pool-adjacent-violators (what
    sklearn.isotonic.IsotonicRegression(increasing=False) computes on unit weights,
    reference preprocessing.py:192-195)."""
    vals, wts, cnts = [], [], []
    for v in np.asarray(y, dtype=np.float64):
        vals.append(v)
        wts.append(1.0)
        cnts.append(1)
        while len(vals) > 1 and vals[-2] < vals[-1]:
            w = wts[-2] + wts[-1]
            v_new = (vals[-2] * wts[-2] + vals[-1] * wts[-1]) / w
            c = cnts[-2] + cnts[-1]
            vals.pop(); wts.pop(); cnts.pop()
            vals[-1], wts[-1], cnts[-1] = v_new, w, c
    return np.repeat(vals, cnts)


def distance_law(matrix, detectable_bins=None, max_dist=None, smooth=True, fun=np.nanmean):
    """Genomic distance law: mean of the strictly positive pixels of each of the first
    min(N, max_dist+1) upper diagonals, restricted to detectable bins; NaN for diagonals
    without such a pixel, 0 beyond (reference preprocessing.py:129-197).

    The per-diagonal sums and counts are reduced on the GPU (cs_distance_law_csr).  Any other
    reducer (the reference accepts e.g. np.nanmedian) needs the pixels of a diagonal together and is
    applied on the host, diagonal by diagonal, as the reference does."""
    matrix = sp.csr_matrix(matrix)
    if fun not in (np.nanmean, np.mean):
        return _distance_law_host(matrix, detectable_bins, max_dist, smooth, fun)
    mat_n = matrix.shape[0]
    if max_dist is None:
        max_dist = mat_n
    n_diags = min(mat_n, max_dist + 1)
    det = None
    if detectable_bins is not None:
        det = np.zeros(mat_n, dtype=np.uint8)
        det[detectable_bins] = 1
    dev = get_device()
    dcsr = engine.DeviceCsr(dev, matrix)
    sums, counts = engine.distance_law_sums(dev, dcsr, det, n_diags)
    dist = np.zeros(mat_n)
    with np.errstate(invalid="ignore", divide="ignore"):
        dist[:n_diags] = np.where(counts > 0, sums / np.maximum(counts, 1), np.nan)
    if smooth and mat_n > 2:
        dist[~np.isfinite(dist)] = 0
        dist = _isotonic_non_increasing(dist)
    return dist


def _distance_law_host(matrix, detectable_bins, max_dist, smooth, fun):
    """distance_law for an arbitrary reducer (reference preprocessing.py:173-188)."""
    mat_n = matrix.shape[0]
    if max_dist is None:
        max_dist = mat_n
    n_diags = min(mat_n, max_dist + 1)
    flag = np.ones(mat_n, dtype=bool)
    if detectable_bins is not None:
        flag[:] = False
        flag[detectable_bins] = True
    dist = np.zeros(mat_n)
    with np.errstate(all="ignore"):
        for d in range(n_diags):
            diag = matrix.diagonal(d)[flag[:mat_n - d] & flag[d:]]
            dist[d] = fun(diag[diag > 0])
    if smooth and mat_n > 2:
        dist[~np.isfinite(dist)] = 0
        dist = _isotonic_non_increasing(dist)
    return dist


def detrend(matrix, detectable_bins=None, max_dist=None, smooth=False, fun=np.nanmean, max_val=10):
    """Divide every stored pixel by the distance law of its diagonal, then set values
    >= max_val to 1 (reference preprocessing.py:256-310).  Returns CSR; the input is untouched."""
    matrix = sp.csr_matrix(matrix)
    y = distance_law(matrix, detectable_bins=detectable_bins, max_dist=max_dist, smooth=smooth, fun=fun)
    y[np.isnan(y)] = 0.0
    if not matrix.has_canonical_format:
        matrix = matrix.copy()
        matrix.sum_duplicates()
    dev = get_device()
    dcsr = engine.DeviceCsr(dev, matrix)
    # the device cap is "max_val > 0"; None = no cap; an explicit non-positive cap (every value >= it
    # becomes 1, reference :308-309) is applied here
    data = engine.detrend_values(dev, dcsr, y, max_val if (max_val is not None and max_val > 0) else 0.0)
    if max_val is not None and max_val <= 0:
        data = np.where(data >= max_val, 1.0, data)
    return sp.csr_matrix((data, matrix.indices.copy(), matrix.indptr.copy()), shape=matrix.shape)


# --------------------------------------------------------------------------------------------
# template editing (host, tiny)
# --------------------------------------------------------------------------------------------
def factorise_kernel(kernel, prop_info=0.999):
    """Truncated SVD keeping the fewest singular vectors whose cumulated squared singular
    values exceed prop_info of the total; each vector scaled by sqrt(sigma)
    (reference preprocessing.py:810-847).  Returns (U[:, :k], V[:k, :])."""
    u, sigma, v = la.svd(np.asarray(kernel, dtype=np.float64))
    energy = np.cumsum(sigma ** 2)
    keep = int(np.flatnonzero(energy > prop_info * energy[-1])[0]) + 1
    if keep > np.floor(min(np.shape(kernel)) / 2):
        sys.stderr.write(
            f"Warning: Kernel factorisation required {keep} singular,"
            "vectors this may result in slow operations.\n"
        )
    root = np.sqrt(sigma[:keep])
    return u[:, :keep] * root[None, :], v[:keep, :] * root[:, None]


def crop_kernel(kernel, target_size):
    """Centre crop to at most target_size, even targets bumped to the next odd number
    (reference preprocessing.py:679-728)."""
    target = [t if t % 2 else t + 1 for t in target_size]
    if list(target) != list(target_size):
        sys.stderr.write(
            f"WARNING: Cropped kernel size adjusted to {target[0]}x{target[1]} to keep odd dimensions.\n"
        )
    sm, sn = kernel.shape
    mr = (sm - target[0]) // 2 if sm > target[0] else 0
    mc = (sn - target[1]) // 2 if sn > target[1] else 0
    return kernel[mr:sm - mr, mc:sn - mc]


def resize_kernel(kernel, kernel_res=None, signal_res=None, factor=None, min_size=7, quiet=False):
    """Rescale a square odd template by kernel_res / signal_res (or `factor`) with order-1
    spline interpolation, never below min_size, re-zooming once to land on an odd size
    (reference preprocessing.py:731-807)."""
    km, kn = kernel.shape
    if km != kn:
        raise ValueError("kernel must be square.")
    if not (km % 2) or not (kn % 2):
        raise ValueError("kernel size must be odd.")
    if factor is not None:
        if kernel_res is not None or signal_res is not None:
            raise ValueError(
                "factor is mutually exclusive with resolution parameters (kernel_res and signal_res)."
            )
        zoom = factor
    else:
        if kernel_res is None or signal_res is None:
            raise ValueError("You must provide either a resize factor or the signal and kernel resolutions.")
        zoom = kernel_res / signal_res
    if km * zoom < min_size:
        zoom = min_size / km
    resized = ndi.zoom(kernel, zoom, order=1)
    if not resized.shape[0] % 2:
        adjusted = (resized.shape[0] - 1) / km
        if not quiet:
            sys.stderr.write(f"Adjusting resize factor from {zoom} to {adjusted}.\n")
        resized = ndi.zoom(kernel, adjusted, order=1)
    return resized
