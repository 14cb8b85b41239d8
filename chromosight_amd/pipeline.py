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
import ctypes as C

import numpy as np
import pandas as pd
import scipy.sparse as sp

from . import engine
from ._lib import CS_F32, CS_F64, LAYOUT_BAND, LAYOUT_DENSE, CsCsr, CsMatrix, get_device, np_dtype_code
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


# ================================================================================================
# device-resident genome: the .cool pixel table as ONE CSR in HBM, sub-matrices as views
# ================================================================================================
class _Ptr:
    """Anything the engine addresses through `.ptr` (a slice of a DeviceBuffer)."""

    def __init__(self, ptr):
        self.ptr = ptr


class _Scratch:
    """Grow-only device buffer reused by the blocks of one genome (they are staged one after the
    other on one stream, so reuse needs no synchronisation)."""

    def __init__(self, dev):
        self.dev, self.buf, self.nbytes = dev, None, 0

    def get(self, nbytes):
        if nbytes > self.nbytes:
            self.dev.sync()
            self.buf = self.dev.empty(int(nbytes * 1.25) + 256, np.uint8)
            self.nbytes = self.buf.nbytes
        return self.buf.ptr


class StagedBlock:
    """One sub-matrix staged in HBM (detrended band or dense map + the flags of its undetectable
    bins): what pattern_detector works on after ContactMap.create_mat (contacts_map.py:527-548)."""

    def __init__(self, name, sig, shape, miss_row, miss_col, max_dist, inter, keep):
        self.name, self.sig, self.shape = name, sig, shape
        self.miss_row, self.miss_col = miss_row, miss_col
        self.max_dist, self.inter, self.keep = max_dist, inter, keep


class DeviceCool:
    """A decoded .cool resident in HBM.  cooler stores the upper triangle of the whole genome
    sorted by (bin1, bin2): that table IS a CSR matrix (row pointer = searchsorted of bin1), so it
    is uploaded once -- counts, column bins, ICE weights -- and every sub-matrix is a view on it:
    balancing (count * w[bin1] * w[bin2], what cooler's matrix(balance=True) returns,
    contacts_map.py:531), the slicing of the block, diag_trim, the distance law, detrend and the
    band tiler all read the same arrays (cs_csr views, include/chromosight_hip.h)."""

    def __init__(self, cool, dev=None):
        self.dev = dev = dev or get_device()
        off = np.asarray(cool["chrom_offset"], dtype=np.int64)
        self.offsets = off
        self.n_bins = n_bins = int(off[-1])
        self.names = [str(n) for n in cool["chrom_names"]]
        self.binsize = int(cool["binsize"])
        self.bin_start, self.bin_end = cool.get("bin_start"), cool.get("bin_end")
        b1 = np.asarray(cool["bin1_id"])
        b2 = np.asarray(cool["bin2_id"])
        cnt = np.asarray(cool["count"])
        if b1.size and not (np.all(b1[1:] >= b1[:-1]) and np.all((b1[1:] > b1[:-1]) | (b2[1:] > b2[:-1]))):
            order = np.lexsort((b2, b1))                      # unsorted / duplicated pixel table
            b1, b2, cnt = b1[order], b2[order], cnt[order]
        weight = np.asarray(cool["weight"], dtype=np.float64)
        indptr = np.searchsorted(b1, np.arange(n_bins + 1)).astype(np.int64)
        # integer counts below 2^24 are exact in float32 (half the bytes of every pass)
        small = cnt.size == 0 or (np.issubdtype(cnt.dtype, np.integer) and cnt.max() < (1 << 24)) or \
            (cnt.max() < (1 << 24) and np.all(cnt == np.rint(cnt)))
        self.val_dtype = np.float32 if small else np.float64
        self.nnz = int(b1.size)
        self.indptr = dev.to_device(indptr, np.int64)
        self.indices = dev.to_device(b2, np.int32)
        self.data = dev.to_device(cnt, self.val_dtype)
        self.weight = dev.to_device(weight, np.float64)
        miss = ~np.isfinite(weight)
        self.miss_host = miss
        self.miss = dev.to_device(miss.astype(np.uint8))
        self.det = dev.to_device((~miss).astype(np.uint8))
        self.upload_bytes = self.indptr.nbytes + self.indices.nbytes + self.data.nbytes + self.weight.nbytes
        self._band = _Scratch(dev)
        self._ext = _Scratch(dev)

    @property
    def n_chrom(self):
        return len(self.offsets) - 1

    def chrom_size(self, ci):
        return int(self.offsets[ci + 1] - self.offsets[ci])

    def _view(self, s, e, cs, ce, begin=None, end=None):
        return CsCsr(e - s, ce - cs, max(self.nnz, 1),
                     begin if begin is not None else self.indptr.ptr + 8 * s, self.indices.ptr, self.data.ptr,
                     np_dtype_code(self.val_dtype), cs, end, self.weight.ptr + 8 * s, self.weight.ptr + 8 * cs)

    def stage_intra(self, ci, max_dist, largest_kernel, smooth=False, band_dtype=np.float64, name=None, stream=None,
                    resident=False):
        """ContactMap.create_mat of one balanced intra block, on the device: distance law over the
        first keep_distance diagonals of the detectable bins, detrend, >= 10 -> 1, NaN -> 0, upper
        band only (contacts_map.py:527-548, 603-638; preprocessing.py:129-197, 256-310)."""
        dev, lib = self.dev, self.dev.lib
        s, e = int(self.offsets[ci]), int(self.offsets[ci + 1])
        n = e - s
        keep = min(max_dist, n) + largest_kernel
        ext = self._ext.get(16 * n + 16 * (keep + 2) + 512)
        d_begin, d_end = ext, ext + 8 * n
        d_sum = d_end + 8 * n
        n_diags = min(n, keep + 1)
        d_cnt, d_law = d_sum + 8 * (keep + 2), None
        raw = self._view(s, e, s, e)
        dev._check(lib.cs_csr_band_extent(dev.ctx, stream, C.byref(raw), 0, keep, d_begin, d_end))
        view = self._view(s, e, s, e, d_begin, d_end)
        dev._check(lib.cs_distance_law_csr(dev.ctx, stream, C.byref(view), self.det.ptr + s, n_diags, d_sum, d_cnt))
        if smooth and n > 2:
            sums = np.empty(n_diags)
            cnts = np.empty(n_diags, dtype=np.int64)
            dev._check(lib.cs_memcpy_d2h(dev.ctx, sums.ctypes.data, d_sum, 8 * n_diags, stream))
            dev._check(lib.cs_memcpy_d2h(dev.ctx, cnts.ctypes.data, d_cnt, 8 * n_diags, stream))
            law = np.zeros(n)
            with np.errstate(invalid="ignore", divide="ignore"):
                law[:n_diags] = np.where(cnts > 0, sums / np.maximum(cnts, 1), 0.0)
            law = np.ascontiguousarray(preproc._isotonic_non_increasing(law)[:n_diags])
            law[np.isnan(law)] = 0.0
            dev._check(lib.cs_memcpy_h2d(dev.ctx, d_sum, law.ctypes.data, 8 * n_diags, stream))
            d_law = d_sum
        else:
            d_law = d_sum                                     # finished in place
            dev._check(lib.cs_distance_law_finish(dev.ctx, stream, d_sum, d_cnt, n_diags, d_law))
        # layout rule of the host path (_Staged): band when it is less than half of the dense map
        in_w = min(keep, n - 1) + 1
        out_w = min(max_dist, n - 1) + 1
        esz = np.dtype(band_dtype).itemsize
        band = 2 * max(in_w, out_w) < n
        ld = (in_w + 63) // 64 * 64 if band else (n + 15) // 16 * 16
        # resident: the block owns its buffer (288 GB of HBM hold every block of a genome at once, so
        # blocks are staged once and reused by all templates / iterations); else a shared scratch
        own = dev.empty(n * ld * esz, np.uint8) if resident else None
        ptr = own.ptr if resident else self._band.get(n * ld * esz)
        sig = CsMatrix(ptr, np_dtype_code(band_dtype), LAYOUT_BAND if band else LAYOUT_DENSE, ld, 0, in_w if band else 0)
        dev._check(lib.cs_csr_to_band(dev.ctx, stream, C.byref(view), d_law, n_diags, 10.0, C.byref(sig)))
        flags = _Ptr(self.miss.ptr + s)
        block = StagedBlock(name or self.names[ci], sig, (n, n), flags, flags, max_dist, False, keep)
        block.buffer = own
        return block

    def block_bins(self, ci):
        s, e = int(self.offsets[ci]), int(self.offsets[ci + 1])
        return np.flatnonzero(~self.miss_host[s:e])


def detect_block(dcool, block, kernel_config, kernel, tsvd=None, coords=None, want_windows=True):
    """pattern_detector(full=True) on a staged block (cli/chromosight.py:601-614)."""
    kernel = np.asarray(kernel, dtype=np.float64)
    if min(block.shape) <= max(kernel.shape):
        return None, None
    kspec = engine.KernelSpec(kernel, tsvd)
    return cid.detect_on_device(dcool.dev, block.sig, block.shape, kspec, kernel_config, block.miss_row,
                                block.miss_col, inter=block.inter, max_dist=block.max_dist, full=True, coords=coords,
                                want_windows=want_windows)


def detect(cool, kernel_config, tsvd=None, smooth=False, band_dtype=np.float64):
    """`chromosight detect` (intra-chromosomal, balanced, default options) on a decoded cool (dict)
    or a DeviceCool.  Every block is staged once in HBM (distance law, detrend, band) and stays
    resident across templates and iterations; each (block, template) is one native call.
    Returns the output table (same columns and row order as the reference's <prefix>.tsv)."""
    dcool = cool if isinstance(cool, DeviceCool) else DeviceCool(cool)
    binsize = dcool.binsize
    off = dcool.offsets
    names = dcool.names
    n_chrom = dcool.n_chrom
    max_dist = max(kernel_config["max_dist"] // binsize, 1)
    largest = max(np.shape(k)[0] for k in kernel_config["kernels"])
    blocks = [dcool.stage_intra(ci, max_dist, largest, smooth=smooth, band_dtype=band_dtype, resident=True)
              for ci in range(n_chrom)]
    all_coords = []
    for kernel_id, kernel in enumerate(kernel_config["kernels"]):
        for it in range(kernel_config["max_iterations"]):
            tables, windows = [], []
            for ci, block in enumerate(blocks):
                tab, win = detect_block(dcool, block, kernel_config, kernel, tsvd=tsvd)
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
    return postprocess(coords, kernel_config, binsize, off, names, dcool.bin_start, dcool.bin_end)


def postprocess(coords, kernel_config, binsize, off, names, bin_start, bin_end):
    """cmd_detect after the per-block loop (cli/chromosight.py:806-871): neighbour removal, bins ->
    genomic coordinates, min_dist and NaN-p filters, Benjamini-Hochberg q-values, column order."""
    n_chrom = len(names)
    separation = max(int(kernel_config["min_separation"] // binsize), 1)
    coords = coords.loc[cid.remove_neighbours(coords, win_size=separation), :]
    # bins -> genomic coordinates
    bin_chrom = np.repeat(np.arange(n_chrom), np.diff(off))
    name_arr = np.asarray(names, dtype=object)
    for tag in ("1", "2"):
        b = coords[f"bin{tag}"].to_numpy(dtype=np.int64)
        coords[f"chrom{tag}"] = name_arr[bin_chrom[b]]
        coords[f"start{tag}"] = np.asarray(bin_start)[b]
        coords[f"end{tag}"] = np.asarray(bin_end)[b]
    coords = coords.reset_index(drop=True)
    too_close = (coords.chrom1 == coords.chrom2) & (np.abs(coords.start2 - coords.start1) < kernel_config["min_dist"])
    coords = coords.loc[~too_close, :]
    coords = coords.loc[~coords.pvalue.isnull(), :]
    coords["qvalue"] = fdr_correction(coords["pvalue"])
    return coords.loc[:, OUTPUT_COLUMNS].reset_index(drop=True)
