"""CPU ORACLE (test infrastructure, not product code) for the callers of the correlation:
focus picking and window validation of chromosight's pattern_detector, restated independently of
chromosight_amd (dense / band numpy arrays, scipy.ndimage labelling, one plain loop per pattern).

Reference lines restated here (all in /root/reference/chromosight/utils/detection.py):
  pick_foci        :387-456   threshold (>= pearson passes), 4-connected foci, foci of fewer than
                              min_size pixels dropped, per focus the first row-major pixel holding its
                              maximum; foci numbered by the row-major position of their first pixel
  label_foci       :459-554   4-way adjacency (right / lower neighbour)
  filter_foci      :557-592
  pattern_detector :287-345   zero padding by (kh, kw) in full mode, NaN on the big_k first
                              sub-diagonals of intra maps, 1-D patterns forced on the diagonal
  validate_patterns:18-155    strict window bounds, missing bins -> NaN, zero / missing proportions

Pinned by tests/test_oracle_golden.py against the reference's own outputs (tests/golden/nms.npz,
example_blocks.npz) before it is trusted as the checker of the device path.
"""
import numpy as np
from scipy import ndimage as ndi

FOUR = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]])
# the same 4-neighbourhood in band coordinates (row i, x = j - i): (i, j+1) -> (i, x+1),
# (i+1, j) -> (i+1, x-1)
FOUR_BAND = np.array([[0, 0, 1], [1, 1, 1], [1, 0, 0]])


def pick_foci_dense(corr, pearson, min_size=2, structure=FOUR):
    """corr: dense 2-D array (zeros = nothing).  Returns an (n, 2) int array of (row, col), one per
    focus in label order, or an empty (0, 2) array."""
    cand = (corr >= pearson) & (corr != 0)
    if not cand.any():
        return np.zeros((0, 2), dtype=int)
    labels, n_lab = ndi.label(cand, structure=structure)   # numbered in raster order of first pixel
    sizes = np.bincount(labels.ravel(), minlength=n_lab + 1)
    out = []
    rows, cols = np.nonzero(labels)                         # row-major
    lab = labels[rows, cols]
    vals = corr[rows, cols]
    order = np.argsort(lab, kind="stable")
    rows, cols, lab, vals = rows[order], cols[order], lab[order], vals[order]
    starts = np.flatnonzero(np.concatenate([[True], lab[1:] != lab[:-1]]))
    ends = np.concatenate([starts[1:], [lab.size]])
    for s, e in zip(starts, ends):
        if sizes[lab[s]] < min_size:
            continue
        k = s + int(np.argmax(vals[s:e]))                   # first maximum in row-major order
        out.append((rows[k], cols[k]))
    return np.array(out, dtype=int).reshape(-1, 2)


def pick_foci_band(corr_band, lo, pearson, min_size=2):
    """corr_band[i, x] = coefficient of pixel (i, i + lo + x).  Same rules, band coordinates."""
    foci = pick_foci_dense(corr_band, pearson, min_size, structure=FOUR_BAND)
    if foci.shape[0]:
        foci[:, 1] = foci[:, 0] + lo + foci[:, 1]
    return foci


def validate(coords, value_at, shape, miss_row, miss_col, kernel_shape, zero_tol, missing_tol, inter, full=True):
    """Window statistics and validity of every pattern, as pattern_detector + validate_patterns do
    them.  value_at(p, q) -> contact value of matrix pixel (p, q) (vectorised, 0 when not stored).
    coords: (n, 2) matrix coordinates (unpadded).  Returns (valid bool[n], windows[n, km, kn])."""
    km, kn = kernel_shape
    kh, kw = (km - 1) // 2, (kn - 1) // 2
    half_h, half_w = km // 2 + 1, kn // 2 + 1
    ms, ns = shape
    pad_r, pad_c = (kw, kh) if full else (0, 0)     # zero_pad_sparse(mat, kh, kw): kh columns, kw rows
    sh_r, sh_c = (kh, kw) if full else (0, 0)       # coords[:, 0] += kh; coords[:, 1] += kw
    H, W = ms + 2 * pad_r, ns + 2 * pad_c
    # framed bins that are NOT detectable (det + kh / det + kw are the detectable ones)
    fr_miss = np.ones(H, dtype=bool)
    fc_miss = np.ones(W, dtype=bool)
    ok_r = np.flatnonzero(~np.asarray(miss_row, dtype=bool)) + sh_r
    ok_c = np.flatnonzero(~np.asarray(miss_col, dtype=bool)) + sh_c
    fr_miss[ok_r[ok_r < H]] = False
    fc_miss[ok_c[ok_c < W]] = False
    big_k = max(km, kn)
    n = coords.shape[0]
    valid = np.zeros(n, dtype=bool)
    windows = np.full((n, km, kn), np.nan)
    for t in range(n):
        p1, p2 = int(coords[t, 0]) + sh_r, int(coords[t, 1]) + sh_c
        high, low = p1 - half_h + 1, p1 + half_h
        left, right = p2 - half_w + 1, p2 + half_w
        if not (high >= 0 and low < H and left >= 0 and right < W):
            continue
        rr, cc = np.meshgrid(np.arange(high, low), np.arange(left, right), indexing="ij")
        src_r, src_c = rr - pad_r, cc - pad_c
        inside = (src_r >= 0) & (src_r < ms) & (src_c >= 0) & (src_c < ns)
        win = np.where(inside, value_at(np.where(inside, src_r, 0), np.where(inside, src_c, 0)), 0.0).astype(np.float64)
        if not inter:
            d = cc - rr
            win[(d <= -1) & (d >= -big_k)] = np.nan
        win[fr_miss[rr] | fc_miss[cc]] = np.nan
        tot = win.size
        n_zero = int(np.sum(win == 0))
        n_miss = int(np.sum(~np.isfinite(win)))
        with np.errstate(all="ignore"):
            prop_undetected = n_miss / tot
            prop_zero = np.float64(n_zero) / np.float64(tot - n_miss)
        if prop_undetected < missing_tol and prop_zero < zero_tol:
            valid[t] = True
            windows[t] = win
    return valid, windows


def detect_table(matrix, corr_trimmed, miss_row, miss_col, kernel_shape, pearson, zero_tol, missing_tol,
                 inter=False, diag_only=False, full=True, return_windows=False):
    """Dense test-sized maps: (bin1, bin2, score) of the validated foci in the reference's order.
    corr_trimmed: coefficient map already restricted to the scanned diagonals (diag_trim).

    Non-square templates in full mode (detection.py:287-298, preprocessing.py:636-676): the maps are padded by
    (kw rows, kh columns) but the coordinates shifted by (kh, kw), so the score of pattern (r, c) is read at
    (r + kh - kw, c + kw - kh) (0 outside the map) and a 1-D pattern, forced on the diagonal after the shift, ends
    up at (c + kw - kh, c); validate() restates the same offsets for the windows."""
    matrix = np.asarray(matrix, dtype=np.float64)
    corr_trimmed = np.asarray(corr_trimmed, dtype=np.float64)
    km, kn = kernel_shape
    shift = ((km - 1) // 2 - (kn - 1) // 2) if full else 0
    foci = pick_foci_dense(corr_trimmed, pearson)
    if foci.shape[0] == 0:
        return (np.zeros((0, 3)), np.zeros((0, km, kn))) if return_windows else np.zeros((0, 3))
    if diag_only and not inter:
        foci[:, 0] = foci[:, 1] - shift
    valid, wins = validate(foci, lambda p, q: matrix[p, q], matrix.shape, miss_row, miss_col, kernel_shape,
                           zero_tol, missing_tol, inter, full=full)
    keep = foci[valid]
    sr, sc = keep[:, 0] + shift, keep[:, 1] - shift
    inside = (sr >= 0) & (sr < corr_trimmed.shape[0]) & (sc >= 0) & (sc < corr_trimmed.shape[1])
    scores = np.where(inside, corr_trimmed[np.where(inside, sr, 0), np.where(inside, sc, 0)], 0.0)
    table = np.column_stack([keep[:, 0], keep[:, 1], scores]).astype(np.float64)
    return (table, wins[valid]) if return_windows else table


def detect_table_band(band, band_lo, corr_band, out_lo, n, miss, kernel_shape, pearson, zero_tol, missing_tol,
                      diag_only=False):
    """Intra maps in band storage (band[i, j - i - band_lo], corr_band[i, j - i - out_lo], columns of
    corr_band = the scanned diagonals only): the same table for maps whose dense form does not fit."""
    band = np.asarray(band)
    bw = band.shape[1]
    foci = pick_foci_band(np.asarray(corr_band, dtype=np.float64), out_lo, pearson)
    if foci.shape[0] == 0:
        return np.zeros((0, 3))
    scores = corr_band[foci[:, 0], foci[:, 1] - foci[:, 0] - out_lo]
    if diag_only:
        foci[:, 0] = foci[:, 1]
        x = foci[:, 1] - foci[:, 0] - out_lo
        scores = np.where((x >= 0) & (x < corr_band.shape[1]), corr_band[foci[:, 0], np.clip(x, 0, corr_band.shape[1] - 1)], 0.0)

    def value_at(p, q):
        x = q - p - band_lo
        ok = (x >= 0) & (x < bw)
        return np.where(ok, band[p, np.where(ok, x, 0)], 0.0)

    valid, _ = validate(foci, value_at, (n, n), miss, miss, kernel_shape, zero_tol, missing_tol, False)
    keep = foci[valid]
    return np.column_stack([keep[:, 0], keep[:, 1], scores[valid]]).astype(np.float64)
