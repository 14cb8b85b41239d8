"""`detect` end to end on one genome: the callers either side of the hot path (SURVEY.md 8(f),
next-2 / next-3), so that "detect wall-clock on the test .cool" can be measured and the
reference's committed outputs reproduced.

Input is a *decoded* .cool (tools/dump_cool.py -> npz: pixels, ICE weights, chromosome offsets,
bin size); neither cooler nor h5py exists in this image.  The steps mirror the reference:
  block assembly        ContactMap.create_mat        contacts_map.py:527-548
  detrend + trim        ContactMap.detrend / remove_diags   contacts_map.py:603-638 (device)
  per kernel, per block pattern_detector(full=True)  cli/chromosight.py:601-614, 730-791 (device)
  post-processing       remove_neighbours, min_dist, NaN p-values, BH q-values
                        cli/chromosight.py:806-871
"""
import numpy as np
import pandas as pd
import scipy.sparse as sp

from .utils import detection as cid
from .utils import preprocessing as preproc
from .utils.stats import fdr_correction

OUTPUT_COLUMNS = ["chrom1", "start1", "end1", "chrom2", "start2", "end2", "bin1", "bin2", "kernel_id",
                  "iteration", "score", "pvalue", "qvalue"]


class ContactBlock:
    """One intra-chromosomal sub-matrix, ready for pattern_detector (the four attributes it
    reads; reference contacts_map.py:453-526)."""

    def __init__(self, name, matrix, detectable_bins, max_dist, inter=False):
        self.name = name
        self.matrix = matrix
        self.detectable_bins = detectable_bins
        self.max_dist = max_dist
        self.inter = inter

    @property
    def shape(self):
        return self.matrix.shape


def balanced_intra_block(cool, chrom_idx):
    """count * w[bin1] * w[bin2] of one chromosome as a symmetric COO matrix, NaN where a bin
    has no weight -- what cooler's matrix(balance=True, sparse=True) returns for the block."""
    off = cool["chrom_offset"]
    s, e = int(off[chrom_idx]), int(off[chrom_idx + 1])
    b1, b2 = cool["bin1_id"], cool["bin2_id"]
    sel = (b1 >= s) & (b1 < e) & (b2 >= s) & (b2 < e)
    w = cool["weight"]
    r = (b1[sel] - s).astype(np.int64)
    c = (b2[sel] - s).astype(np.int64)
    v = cool["count"][sel] * w[b1[sel]] * w[b2[sel]]
    offd = r != c
    rows = np.concatenate([r, c[offd]])
    cols = np.concatenate([c, r[offd]])
    vals = np.concatenate([v, v[offd]])
    return sp.coo_matrix((vals, (rows, cols)), shape=(e - s, e - s))


def balanced_upper_band(cool, chrom_idx, keep):
    """Upper band (diagonals 0..keep) of the balanced intra block as CSR, built straight from the
    pixel table: cooler stores the upper triangle sorted by (bin1, bin2), so a chromosome is one
    contiguous run of pixels and the CSR row pointer is a searchsorted.  Everything the path uses
    downstream (distance law over diagonals >= 0, detrend, trim to 0..keep) only looks at this
    band, so the symmetric block of balanced_intra_block never needs to exist."""
    off = cool["chrom_offset"]
    s, e = int(off[chrom_idx]), int(off[chrom_idx + 1])
    n = e - s
    b1, b2, w = cool["bin1_id"], cool["bin2_id"], cool["weight"]
    if "_bin1_sorted" not in cool:
        cool["_bin1_sorted"] = bool(np.all(b1[1:] >= b1[:-1]))
    if cool["_bin1_sorted"]:
        lo, hi = np.searchsorted(b1, [s, e])
        r1, r2, cnt = b1[lo:hi], b2[lo:hi], cool["count"][lo:hi]
    else:
        sel = (b1 >= s) & (b1 < e)
        r1, r2, cnt = b1[sel], b2[sel], cool["count"][sel]
    d = r2 - r1
    sel = (r2 < e) & (d >= 0) & (d <= keep)
    r1, r2 = r1[sel], r2[sel]
    v = cnt[sel] * w[r1] * w[r2]
    r = (r1 - s).astype(np.int64)
    c = (r2 - s).astype(np.int32)
    if r.size == 0 or (np.all(r[1:] >= r[:-1]) and np.all((r[1:] > r[:-1]) | (c[1:] > c[:-1]))):
        indptr = np.searchsorted(r, np.arange(n + 1)).astype(np.int32 if r.size < 2 ** 31 else np.int64)
        return sp.csr_matrix((v, c, indptr), shape=(n, n))
    m = sp.coo_matrix((v, (r, c)), shape=(n, n)).tocsr()     # unsorted / duplicated pixel table
    return m


def prepare_intra_block(cool, chrom_idx, max_dist, largest_kernel, name=None):
    """ContactMap.create_mat for a balanced intra block: detrend by the distance law on the
    first keep_distance diagonals, set >= 10 to 1, keep the upper band, NaN -> 0."""
    off = cool["chrom_offset"]
    s, e = int(off[chrom_idx]), int(off[chrom_idx + 1])
    det_all = cool.get("_detectable")
    if det_all is None:
        det_all = cool["_detectable"] = np.flatnonzero(np.isfinite(cool["weight"]))
    lo, hi = np.searchsorted(det_all, [s, e])
    det = det_all[lo:hi] - s
    keep = min(max_dist, e - s) + largest_kernel
    band = balanced_upper_band(cool, chrom_idx, keep)
    mat = preproc.detrend(band, max_dist=keep, smooth=False, detectable_bins=det, max_val=10)
    mat.data[np.isnan(mat.data)] = 0
    mat.eliminate_zeros()
    return ContactBlock(name or f"chrom{chrom_idx}", mat, (det, det.copy()), max_dist)


def prepare_inter_block(cool, ca, cb, name=None):
    """ContactMap.create_mat for a balanced inter-chromosomal block (contacts_map.py:598-601,
    539-548): NaN -> 0, divide by the median of the stored values, drop zeros."""
    off = cool["chrom_offset"]
    s1, e1, s2, e2 = int(off[ca]), int(off[ca + 1]), int(off[cb]), int(off[cb + 1])
    b1, b2, w = cool["bin1_id"], cool["bin2_id"], cool["weight"]
    sel = (b1 >= s1) & (b1 < e1) & (b2 >= s2) & (b2 < e2)
    vals = cool["count"][sel] * w[b1[sel]] * w[b2[sel]]
    vals = np.where(np.isnan(vals), 0.0, vals)
    with np.errstate(all="ignore"):
        vals = vals / np.nanmedian(vals) if vals.size else vals
    vals = np.where(np.isnan(vals), 0.0, vals)
    mat = sp.coo_matrix((vals, (b1[sel] - s1, b2[sel] - s2)), shape=(e1 - s1, e2 - s2))
    mat.eliminate_zeros()
    det_all = np.flatnonzero(np.isfinite(w))
    det_r = det_all[(det_all >= s1) & (det_all < e1)] - s1
    det_c = det_all[(det_all >= s2) & (det_all < e2)] - s2
    return ContactBlock(name or f"chrom{ca}-chrom{cb}", mat.tocsr(), (det_r, det_c), None, inter=True)


def quantify_block(cool, ca, cb, coords, kernel_config, kernel, max_dist, largest_kernel, tsvd=None):
    """One task of `chromosight quantify` (cli/chromosight.py:229-260): scores of the given
    (bin1, bin2) block coordinates; rows that fail validation carry NaN scores."""
    if np.asarray(coords).shape[0] == 0:
        return None, None
    if ca == cb:
        block = prepare_intra_block(cool, ca, max_dist, largest_kernel)
    else:
        block = prepare_inter_block(cool, ca, cb)
    return cid.pattern_detector(block, kernel_config, kernel, coords=np.array(coords, dtype=int), full=True,
                                tsvd=tsvd)


def detect(cool, kernel_config, tsvd=None):
    """`chromosight detect` (intra-chromosomal, balanced, default options) on a decoded cool.
    Returns the output table (same columns and row order as the reference's <prefix>.tsv)."""
    binsize = int(cool["binsize"])
    off = cool["chrom_offset"]
    names = [str(n) for n in cool["chrom_names"]]
    n_chrom = len(names)
    max_dist = max(kernel_config["max_dist"] // binsize, 1)
    largest = max(k.shape[0] for k in kernel_config["kernels"])
    blocks = [prepare_intra_block(cool, ci, max_dist, largest, names[ci]) for ci in range(n_chrom)]
    all_coords = []
    for kernel_id, kernel in enumerate(kernel_config["kernels"]):
        for it in range(kernel_config["max_iterations"]):
            tables, windows = [], []
            for ci, block in enumerate(blocks):
                tab, win = cid.pattern_detector(block, kernel_config, kernel, full=True, tsvd=tsvd)
                if tab is None:
                    continue
                tab = tab.copy()
                tab["bin1"] += int(off[ci])
                tab["bin2"] += int(off[ci])
                tables.append(tab)
                windows.append(win)
            if not tables:
                break
            coords = pd.concat(tables, axis=0).reset_index(drop=True)
            coords["kernel_id"] = kernel_id
            coords["iteration"] = it
            all_coords.append(coords)
            kernel = cid.pileup_patterns(np.concatenate(windows, axis=0))
    if not all_coords:
        return pd.DataFrame(columns=OUTPUT_COLUMNS)
    coords = pd.concat(all_coords, axis=0).reset_index(drop=True)
    separation = max(int(kernel_config["min_separation"] // binsize), 1)
    coords = coords.loc[cid.remove_neighbours(coords, win_size=separation), :]
    # bins -> genomic coordinates
    bin_chrom = np.repeat(np.arange(n_chrom), np.diff(off))
    for tag in ("1", "2"):
        b = coords[f"bin{tag}"].to_numpy(dtype=np.int64)
        coords[f"chrom{tag}"] = [names[c] for c in bin_chrom[b]]
        coords[f"start{tag}"] = cool["bin_start"][b]
        coords[f"end{tag}"] = cool["bin_end"][b]
    coords = coords.reset_index(drop=True)
    too_close = (coords.chrom1 == coords.chrom2) & (np.abs(coords.start2 - coords.start1) < kernel_config["min_dist"])
    coords = coords.loc[~too_close, :]
    coords = coords.loc[~coords.pvalue.isnull(), :]
    coords["qvalue"] = fdr_correction(coords["pvalue"])
    return coords.loc[:, OUTPUT_COLUMNS].reset_index(drop=True)
