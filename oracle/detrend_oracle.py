"""CPU ORACLE (test infrastructure, not product code) for the block preparation that precedes the
correlation: balancing of a cooler pixel table, distance law, detrend and trimming of one
intra-chromosomal block, restated on diagonal-band numpy arrays (band[i, d] = pixel (i, i + d)).

Reference lines restated (relative to /root/reference/chromosight/):
  utils/contacts_map.py:527-548  create_mat: balanced block (count * w[bin1] * w[bin2], what cooler's
                                 matrix(balance=True) yields), detrend, remove_diags, NaN -> 0
  utils/contacts_map.py:603-638  detrend over keep_distance = min(max_dist, N) + largest_kernel
                                 diagonals, max_val 10; remove_diags = upper band 0..keep_distance
  utils/preprocessing.py:129-197 distance_law: per diagonal, nanmean of the strictly positive pixels
                                 whose row and column bins are both detectable
  utils/preprocessing.py:256-310 detrend: divide by the law (NaN -> 0 first), values >= max_val -> 1

Pinned by tests/test_oracle_golden.py against the reference's own laws and prepared blocks
(tests/golden/example_blocks.npz)."""
import numpy as np


def balanced_band(cool, chrom_idx, keep):
    """(band float64 [n, keep + 1], detectable bool [n]) of one chromosome of a decoded .cool;
    pixels of unweighted bins are NaN, unstored pixels 0."""
    off = cool["chrom_offset"]
    s, e = int(off[chrom_idx]), int(off[chrom_idx + 1])
    n = e - s
    b1, b2, w = np.asarray(cool["bin1_id"]), np.asarray(cool["bin2_id"]), np.asarray(cool["weight"], dtype=np.float64)
    cnt = np.asarray(cool["count"])
    if b1.size and np.all(b1[1:] >= b1[:-1]):            # cooler order: the chromosome is one run of pixels
        lo, hi = np.searchsorted(b1, [s, e])
        b1, b2, cnt = b1[lo:hi], b2[lo:hi], cnt[lo:hi]
    sel = (b1 >= s) & (b1 < e) & (b2 >= s) & (b2 < e) & (b2 - b1 >= 0) & (b2 - b1 <= keep)
    r, c = b1[sel] - s, b2[sel] - s
    vals = cnt[sel] * w[b1[sel]] * w[b2[sel]]
    band = np.zeros((n, keep + 1))
    np.add.at(band, (r, c - r), vals)
    return band, np.isfinite(w[s:e])


def distance_law_band(band, detectable):
    """law[d] for the stored diagonals: mean of the > 0 pixels between detectable bins (NaN if none)."""
    n, w = band.shape
    law = np.full(w, np.nan)
    for d in range(min(w, n)):
        diag = band[:n - d, d]
        ok = detectable[:n - d] & detectable[d:]
        v = diag[ok]
        v = v[v > 0]
        if v.size:
            law[d] = v.mean()
    return law


def prepare_band(band, detectable, max_val=10.0):
    """Detrended, capped, NaN-free band (the matrix pattern_detector receives) and the law."""
    n, w = band.shape
    law = distance_law_band(band, detectable)
    y = np.where(np.isnan(law), 0.0, law)
    with np.errstate(all="ignore"):
        out = band / y[None, :]
        out[out >= max_val] = 1.0
    out[np.isnan(out)] = 0.0
    cols = np.arange(n)[:, None] + np.arange(w)[None, :]
    out[cols >= n] = 0.0
    return out, law
