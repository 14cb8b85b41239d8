"""CPU ORACLE (test infrastructure, not product code).

A plain numpy float64 restatement of chromosight's sliding-window Pearson
correlation path, written per output pixel from the validated specification in
SURVEY.md section 8(a2).  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import this module; the product path
(`chromosight_amd/`) never does.

Parity status: PINNED.  `tests/golden/make_golden.py` imports the reference
(`/root/reference/chromosight/utils/{detection,preprocessing,stats}.py`) in the
authoring container and stores its outputs under `tests/golden/*.npz`;
`tests/test_oracle_golden.py` checks every function here against those vectors
(float64, <= 1e-12).

Reference lines restated (all relative to /root/reference/chromosight/utils/):
  xcorr2_oracle            detection.py:595-624, 627-723 (sparse), 726-804 (dense)
  framed_missing_predicate preprocessing.py:535-633 (make_missing_mask) +
                           preprocessing.py:404-498 (frame_missing_mask)
  normxcorr2_oracle        detection.py:807-914, 917-1131 (sparse), 1134-1273 (dense)
  corr_to_pval_oracle      stats.py:43-81
  distance_law_oracle      preprocessing.py:129-197
  detrend_oracle           preprocessing.py:256-310
"""
import math

import numpy as np
from numpy.lib.stride_tricks import sliding_window_view

XCORR_THRESHOLD = 1e-4  # default `threshold` of xcorr2, detection.py:595
DENOM_EPS = 1e-10       # detection.py:1010, 1088, 1216, 1247


# --------------------------------------------------------------------------- #
# plain cross-correlation
# --------------------------------------------------------------------------- #
def window_sums(arr, weights):
    """valid-mode sum_{a,b} arr[i+a, j+b] * weights[a, b] in float64."""
    arr = np.asarray(arr, dtype=np.float64)
    weights = np.asarray(weights, dtype=np.float64)
    mk, nk = weights.shape
    win = sliding_window_view(arr, (mk, nk))
    return np.einsum("ijab,ab->ij", win, weights, optimize=False)


def xcorr2_oracle(signal, kernel, threshold=XCORR_THRESHOLD):
    """Cross-correlation (no flip), output aligned on the window centre and
    zero on the (k-1)//2 margins, tiny values zeroed (detection.py:716-722,
    797-803).  Odd kernel dimensions only (the reference's contract,
    preprocessing.py:774-775)."""
    signal = np.asarray(signal, dtype=np.float64)
    kernel = np.asarray(kernel, dtype=np.float64)
    sm, sn = signal.shape
    km, kn = kernel.shape
    kh, kw = (km - 1) // 2, (kn - 1) // 2
    valid = window_sums(signal, kernel)
    valid[np.abs(valid) < threshold] = 0.0
    out = np.zeros((sm, sn))
    out[kh:kh + valid.shape[0], kw:kw + valid.shape[1]] = valid
    return out


# --------------------------------------------------------------------------- #
# missing-pixel predicate in the framed coordinate system
# --------------------------------------------------------------------------- #
def framed_missing_predicate(shape, kernel_shape, miss_row, miss_col,
                             sym_upper, max_dist):
    """Boolean array of shape (ms + 2(mk-1), ns + 2(nk-1)): True where the
    reference's make_missing_mask + frame_missing_mask flag a pixel missing.

    miss_row / miss_col are boolean vectors (True = bin not detectable)."""
    ms, ns = shape
    mk, nk = kernel_shape
    miss_row = np.asarray(miss_row, dtype=bool)
    miss_col = np.asarray(miss_col, dtype=bool)
    H, W = ms + 2 * (mk - 1), ns + 2 * (nk - 1)
    out = np.zeros((H, W), dtype=bool)
    ii = np.arange(ms)[:, None]
    jj = np.arange(ns)[None, :]
    inner = miss_row[:, None] | miss_col[None, :]
    if sym_upper:
        # make_missing_mask fills, for each missing bin, max_dist+1 pixels
        # upwards and to the right (preprocessing.py:588-627)
        md = min(ms, ns) if max_dist is None else max_dist
        inner = inner & (jj - ii >= 0) & (jj - ii <= md)
    out[mk - 1:mk - 1 + ms, nk - 1:nk - 1 + ns] = inner
    if sym_upper and max_dist is not None:
        max_m = max_dist + mk
        max_n = max_dist + nk
        # top margin: columns [0, max_n) of the *unframed* width (:461-463)
        out[:mk - 1, nk - 1:nk - 1 + min(max_n, ns)] = True
        # right margin: last max_m+1 framed rows (:474-475)
        out[max(H - (max_m + 1), 0):, nk - 1 + ns:] = True
        # left margin: only the top-left corner (:476-477)
        out[:mk - 1, :nk - 1] = True
    else:
        out[:mk - 1, :] = True
        out[mk - 1 + ms:, :] = True
        out[:, :nk - 1] = True
        out[:, nk - 1 + ns:] = True
    if sym_upper:
        big_k = max(mk, nk)
        fi = np.arange(H)[:, None]
        fj = np.arange(W)[None, :]
        off = fj - fi
        out |= (off <= -1) & (off >= -big_k)
    return out


# --------------------------------------------------------------------------- #
# normalised cross-correlation
# --------------------------------------------------------------------------- #
def _thr(x, threshold=XCORR_THRESHOLD):
    x = x.copy()
    x[np.abs(x) < threshold] = 0.0
    return x


def normxcorr2_oracle(signal, kernel, max_dist=None, sym_upper=False,
                      full=False, missing=None, missing_tol=0.75,
                      kernel_conv=None, sparse_semantics=True):
    """Per-pixel Pearson coefficient map, float64.

    signal : 2-D array (dense view of the reference's CSR or ndarray input).
    missing : None, or a boolean array with the shape of the (framed, when
        full=True) signal: the framed missing mask.
    kernel_conv : optional kernel actually used for the sum(S*K) correlation
        (the tsvd-reconstructed kernel, detection.py:1016,1082); statistics of
        `kernel` itself are still used for the mean/std terms.
    sparse_semantics : follow _normxcorr2_sparse (True) or _normxcorr2_dense.

    Returns (corr, n_obs) with the input's shape; n_obs = number of present
    pixels where the window touches a missing pixel, else mk*nk.
    """
    S = np.asarray(signal, dtype=np.float64)
    K = np.asarray(kernel, dtype=np.float64)
    Kc = K if kernel_conv is None else np.asarray(kernel_conv, dtype=np.float64)
    ms, ns = S.shape
    mk, nk = K.shape
    n = mk * nk
    kh, kw = (mk - 1) // 2, (nk - 1) // 2
    ones = np.ones((mk, nk))
    if full:
        F = np.zeros((ms + 2 * (mk - 1), ns + 2 * (nk - 1)))
        F[mk - 1:mk - 1 + ms, nk - 1:nk - 1 + ns] = S
    else:
        F = S
    H, W = F.shape

    def xc(arr, ker):
        v = _thr(window_sums(arr, ker))
        o = np.zeros((H, W))
        o[kh:kh + v.shape[0], kw:kw + v.shape[1]] = v
        return o

    with np.errstate(all="ignore"):
        if missing is None:
            kmean = float(K.mean())
            kstd = float(K.std())
            m1 = xc(F, ones / n)
            den = xc(F ** 2, ones / n) - m1 ** 2
            den = np.sqrt(den) * kstd
            num = xc(F, Kc / n) - m1 * kmean
            r = np.where(np.abs(den) < DENOM_EPS, 0.0, num / den)
            n_obs = np.full((H, W), float(n))
        else:
            M = np.asarray(missing, dtype=bool)
            if M.shape != F.shape:
                raise ValueError("missing mask shape mismatch")
            ksum = np.sum(K)
            kmean = ksum / n
            k2sum = np.sum(K ** 2)
            k2mean = k2sum / n
            Mf = M.astype(np.float64)
            nm = xc(Mf, ones)
            touched = nm != 0
            npres = np.where(touched, n - nm, float(n))
            kmw = (ksum - xc(Mf, Kc)) / npres
            k2mw = (k2sum - xc(Mf, Kc ** 2)) / npres
            m1 = xc(F, ones / n)
            m1 = np.where(touched, m1 * n / npres, m1)
            m2 = xc(F ** 2, ones / n)
            m2 = np.where(touched, m2 * n / npres, m2)
            kvar = k2mean - kmean ** 2
            den = (m2 - m1 ** 2) * kvar
            den = np.where(touched, den / kvar * (k2mw - kmw ** 2), den)
            den = np.sqrt(den)
            cut = int((1 - missing_tol) * n)
            den = np.where(touched & (npres < cut), 0.0, den)
            out = m1 * kmean
            out = np.where(touched, out * kmw * npres / (kmean * n), out)
            num = xc(F, Kc / n) - out
            num = np.where(touched, num * n / npres, num)
            r = np.where(np.abs(den) < DENOM_EPS, 0.0, num / den)
            n_obs = npres
        if sym_upper:
            r = np.triu(r)
        r[~np.isfinite(r)] = 0.0
        r[r < -1] = -1.0
        r[r > 1] = 1.0
    if full:
        r = r[mk - 1:mk - 1 + ms, nk - 1:nk - 1 + ns]
        n_obs = n_obs[mk - 1:mk - 1 + ms, nk - 1:nk - 1 + ns]
    return r, n_obs


def corr_to_pval_oracle(corr, n_obs):
    """log10 two-sided p-value through Fisher's z (stats.py:43-81):
    log10(2 * Phi(-|atanh(r)| * sqrt(n - 3)))."""
    corr = np.asarray(corr, dtype=np.float64)
    n_obs = np.broadcast_to(np.asarray(n_obs, dtype=np.float64), corr.shape)
    out = np.empty(corr.shape)
    flat_r, flat_n, flat_o = corr.ravel(), n_obs.ravel(), out.ravel()
    for idx in range(flat_r.size):
        r, nn = flat_r[idx], flat_n[idx]
        with np.errstate(all="ignore"):
            z = math.atanh(r) if abs(r) < 1 else math.copysign(math.inf, r)
            s = math.sqrt(nn - 3) if nn >= 3 else math.nan
            x = -abs(z * s)
        if math.isnan(x):
            flat_o[idx] = math.nan
            continue
        p = math.erfc(-x / math.sqrt(2.0))  # 2 * Phi(x), x <= 0
        flat_o[idx] = math.log10(p) if p > 0 else -math.inf
    return out


# --------------------------------------------------------------------------- #
# distance law / detrend
# --------------------------------------------------------------------------- #
def distance_law_oracle(dense, detectable, max_dist=None):
    """Mean of the strictly positive pixels of each diagonal whose two bins are
    detectable (preprocessing.py:173-188); no smoothing.  NaN where a diagonal
    has no such pixel; 0 beyond min(N, max_dist+1) diagonals."""
    A = np.asarray(dense, dtype=np.float64)
    n = A.shape[0]
    det = np.zeros(n, dtype=bool)
    det[np.asarray(detectable)] = True
    if max_dist is None:
        max_dist = n
    n_diags = min(n, max_dist + 1)
    law = np.zeros(n)
    for d in range(n_diags):
        vals = np.diagonal(A, d)
        ok = det[:n - d] & det[d:]
        vals = vals[ok]
        vals = vals[vals > 0]
        law[d] = vals.mean() if vals.size else np.nan
    return law


def detrend_oracle(dense, stored, detectable, max_dist=None, max_val=10):
    """dense: symmetric float64 array (NaN allowed); stored: boolean array of
    the pixels explicitly stored in the reference's sparse matrix (only those
    are divided, preprocessing.py:300-309).  Returns (detrended dense with 0
    at non-stored pixels, law with NaN->0)."""
    A = np.asarray(dense, dtype=np.float64)
    law = distance_law_oracle(np.where(stored, A, 0.0), detectable, max_dist)
    y = law.copy()
    y[np.isnan(y)] = 0.0
    n = A.shape[0]
    ii, jj = np.indices((n, n))
    with np.errstate(all="ignore"):
        out = np.where(stored, A / y[np.abs(ii - jj)], 0.0)
        if max_val is not None:
            out = np.where(stored & (out >= max_val), 1.0, out)
    return out, y
