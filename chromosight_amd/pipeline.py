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
import os

import threading

import numpy as np
import pandas as pd
import scipy.sparse as sp

from . import engine
from ._lib import (COUNTS_HEADER_BYTES, CS_F32, CS_F64, LAYOUT_BAND, LAYOUT_BAND_COUNTS, LAYOUT_BAND_COUNTS_VIEW, LAYOUT_BAND_LAZY, LAYOUT_BAND_PADDED,
                   LAYOUT_DENSE, LAZY_BAND_BYTES, CsCsr, CsMatrix, CsStageBlock,
                   get_device, np_dtype_code)
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


def _pitch(width, quantum):
    """Row pitch (elements) of a staged map: `width` rounded up to `quantum`, and never a multiple of 4 KiB in float64 --
    with a power-of-two pitch the same slot of every row lies on the same memory channels, and the waves of the staging and
    tile kernels, which sweep their rows at the same pace, all hit those few channels at once (measured: the tiler of the
    23-block genome 2.9 -> see DESIGN.md)."""
    ld = (int(width) + quantum - 1) // quantum * quantum
    if (ld * 8) % 4096 == 0:
        ld += quantum
    return ld


class _FreeList:
    """Device buffers of released blocks, best fit first; every access under one lock (blocks are released by the
    garbage collector on whatever thread drops the last reference, taken by the staging threads)."""

    def __init__(self):
        self._lock = threading.Lock()
        self._bufs = []

    def release(self, buf):
        with self._lock:
            self._bufs.append(buf)

    def take(self, nbytes):
        with self._lock:
            best = None
            for k, buf in enumerate(self._bufs):
                if buf.nbytes >= nbytes and (best is None or buf.nbytes < self._bufs[best].nbytes):
                    best = k
            return self._bufs.pop(best) if best is not None else None


class _Shared:
    """HBM shared by the blocks of one staging call (laws and descriptors of lazily evaluated bands): back to the genome's
    free list when the last block that refers to it is gone."""

    def __init__(self, pool, buffer):
        self.pool, self.buffer = pool, buffer

    def __del__(self):
        try:
            self.pool.release(self.buffer)
        except Exception:
            pass


_FULL_LOCK = threading.RLock()


class StagedBlock:
    """One sub-matrix staged in HBM (detrended band or dense map + the flags of its undetectable
    bins): what pattern_detector works on after ContactMap.create_mat (contacts_map.py:453-526)."""

    def __init__(self, name, sig, shape, miss_row, miss_col, max_dist, inter, keep):
        self.name, self.sig, self.shape = name, sig, shape
        self.miss_row, self.miss_col = miss_row, miss_col
        self.max_dist, self.inter, self.keep = max_dist, inter, keep
        self.buffer, self.pool = None, None
        self.row_window = None          # (a, b): the block holds only these rows (+ halo) of the sub-matrix
        self.sig32 = None               # the same map in float32 (same layout), written by the same staging pass
        self.restage = None             # lazily evaluated float64 band (sig.layout == LAYOUT_BAND_LAZY): stages the block
        self._full = None               # again with the band stored, for the paths that read it themselves (full())

    def full(self):
        """This block with its float64 band in memory.  The batched native entries read a lazily evaluated band through
        its descriptor (include/chromosight_hip.h cs_stage_block); every other path gets the block staged once more, the
        plain way (same values: the staging pass and the descriptor compute the same expression)."""
        if self.sig.layout != LAYOUT_BAND_LAZY:
            return self
        with _FULL_LOCK:                    # (the worker threads of a block-by-block pass ask for the same block's band at once)
            if self._full is None:
                self._full = self.restage()
        return self._full

    def __del__(self):
        # a resident block hands its HBM back to the genome's free list (hipFree synchronises and costs
        # ~0.2 ms; a genome is restaged for every pattern configuration).  The list is shared with the worker
        # threads (_resident pops under the same lock).  Reuse needs no event: every path that drops a block has
        # synchronised the streams that read it (detect / quantify return host tables), and the next staging of the
        # buffer is ordered after that synchronisation.
        try:
            if self.buffer is not None and self.pool is not None:
                self.pool.release(self.buffer)
                if getattr(self, "buffer32", None) is not None:
                    self.pool.release(self.buffer32)
        except Exception:
            pass


class _Workers:
    """A few host threads, each with its own context (scratch, weights, pinned result buffers) and
    stream on the same GPU.  One (block, template) call is a chain of ~25 small launches and three
    host synchronisations; calls for different blocks are independent, so running several at once lets
    the GPU overlap their latency-bound kernels (ctypes releases the GIL during the native call)."""

    def __init__(self, index, n):
        import concurrent.futures
        import threading
        from ._lib import Device
        self.local = threading.local()
        self.index = index
        self.device_cls = Device
        self.pool = concurrent.futures.ThreadPoolExecutor(max_workers=n)

    def device(self):
        dev = getattr(self.local, "dev", None)
        if dev is None:
            dev = self.local.dev = self.device_cls(self.index)
            # worker streams carry the short launch chains that run beside another pattern's tile kernels: served first
            self.local.stream = dev.new_stream(high_priority=True)
            self.local.scratch = _Scratch(dev)
        return dev, self.local.stream

    def scratch(self):
        self.device()
        return self.local.scratch

    def map(self, fn, items):
        return list(self.pool.map(fn, items))


_WORKER_POOLS = {}
_STAGE_STREAMS = {}            # per device object: (stream, extent scratch) pairs of DeviceCool.stage_blocks


class DeviceCool:
    """A decoded .cool resident in HBM.  cooler stores the upper triangle of the whole genome
    sorted by (bin1, bin2): that table IS a CSR matrix (row pointer = searchsorted of bin1), so it
    is uploaded once -- counts, column bins, ICE weights -- and every sub-matrix is a view on it:
    balancing (count * w[bin1] * w[bin2], what cooler's matrix(balance=True) returns,
    contacts_map.py:531), the slicing of the block, diag_trim, the distance law, detrend and the
    band tiler all read the same arrays (cs_csr views, include/chromosight_hip.h)."""

    def __init__(self, cool, dev=None):
        if isinstance(cool, (str, bytes)) or hasattr(cool, "__fspath__"):
            from . import io as cio
            cool = cio.load_cool(cool)                     # a .cool path: decoded here (chromosight_amd/io.py)
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
        # ... and, when none is negative, can be staged as a band of raw counts that its readers detrend (CS_LAYOUT_BAND_COUNTS)
        # (and no weight: the readers' NaN -> 0 is a max with 0)
        self.counts_ok = bool(small and (cnt.size == 0 or cnt.min() >= 0) and not np.any(weight < 0))
        self.nnz = int(b1.size)
        self.indptr = dev.to_device(indptr, np.int64)
        self.indices = dev.to_device(b2, np.int32)
        self.data = dev.to_device(cnt, self.val_dtype)
        self.weight = dev.to_device(weight, np.float64)
        miss = ~np.isfinite(weight)
        self.miss_host = miss
        self.miss = dev.to_device(miss.astype(np.uint8))
        self.det = dev.to_device((~miss).astype(np.uint8))
        self.host = {"binsize": self.binsize, "chrom_offset": off, "chrom_names": np.asarray(self.names), "bin1_id": b1,
                     "bin2_id": b2, "count": cnt, "weight": weight, "bin_start": self.bin_start, "bin_end": self.bin_end}
        self.upload_bytes = self.indptr.nbytes + self.indices.nbytes + self.data.nbytes + self.weight.nbytes
        # what cs_stage_blocks needs: every stored pixel on or above the diagonal (a .cool's symmetric-upper storage)
        self.upper = bool(b1.size == 0 or np.all(b2 >= b1))
        self._band = _Scratch(dev)
        self._ext = _Scratch(dev)
        self._stage_lock = threading.RLock()
        self._free = _FreeList()        # HBM of released resident blocks, reused by the next staging
        self._workers = None

    def _resident(self, nbytes, dev=None):
        buf = self._free.take(nbytes)
        return buf if buf is not None else (dev or self.dev).empty(nbytes, np.uint8)

    def stage_blocks(self, chroms, max_dist, largest_kernel, workers=1, **options):
        """stage_intra(resident=True) of several chromosomes.  Default: one host thread deals the blocks to a few streams
        (below).  workers > 1: host threads with their own contexts stage several blocks at a time (see _Workers) --
        measured slower than ONE stream on the 23-block genome (the hand-over costs two synchronisations per block).
        Every block's staging is complete on return."""
        chroms = list(chroms)
        lazy64 = options.pop("lazy64", False)                # (only the one-call staging knows lazily evaluated bands)
        counts = options.pop("counts", None)                 # (... and bands of raw counts)
        # One staging call at a time per pixel table: the calls share this context's staging scratch, its page-locked
        # table slots and the law scratch, and they are ordered on one stream.  (Patterns scanned side by side by several
        # host threads each stage the blocks they cannot take as views -- short chromosomes staged dense for a wider
        # pattern: two such calls at once handed each other's block tables to the kernels.)
        with self._stage_lock:
            fast = self._stage_fast(chroms, max_dist, largest_kernel, lazy64=lazy64, counts=counts, **options)
        if fast is not None:
            return fast
        if workers <= 1 or len(chroms) <= 1:
            # one host thread, the blocks dealt to a few streams of this context (each with its own extent scratch):
            # a block's chain is 7 short launches that leave most of the chip idle, the chains of different blocks
            # are independent, and issuing them costs the host less than running them costs the GPU (23-block genome,
            # per step: 1 stream 12.8 ms, 2: 12.5, 4: 11.4-11.7, 6-12: 11.2)
            n_streams = min(6, len(chroms))
            if n_streams <= 1 or options.get("reduce") is not None or options.get("smooth"):
                return [self.stage_intra(ci, max_dist, largest_kernel, resident=True, **options) for ci in chroms]
            # process-wide like the worker pools: creating streams per DeviceCool cost 8 ms per `detect` on a small genome
            pairs = _STAGE_STREAMS.setdefault(id(self.dev), (self.dev, []))[1]      # (the entry keeps the device alive)
            if len(pairs) < n_streams:
                pairs += [(self.dev.new_stream(), _Scratch(self.dev)) for _ in range(n_streams - len(pairs))]
            self.dev.sync()
            blocks = []
            for k, ci in enumerate(chroms):
                stream, ext = pairs[k % n_streams]
                blocks.append(self.stage_intra(ci, max_dist, largest_kernel, resident=True, stream=stream, ext=ext, **options))
            for stream, _ in pairs[:n_streams]:
                self.dev.sync(stream)
            return blocks
        self.dev.sync()                         # uploads / a previous pass on the default stream
        pool = self.workers(workers)

        def one(ci):
            dev, stream = pool.device()
            block = self.stage_intra(ci, max_dist, largest_kernel, resident=True, stream=stream, dev=dev, ext=pool.scratch(),
                                     **options)
            dev.sync(stream)
            return block

        return pool.map(one, chroms)

    def _stage_fast(self, chroms, max_dist, largest_kernel, smooth=False, band_dtype=np.float64, reduce=None, rows=None,
                    stream=None, lazy64=False, counts=None, **unused):
        """All the given chromosomes with ONE native call (cs_stage_blocks: three launches for the whole genome -- a
        pass over the pixel table for the distance laws, the laws' finish, and the detrend / tiler that writes every
        block once in float64 (exact re-scoring, windows) and float32 (what the matrix-core kernel stages)); None when
        an option needs the block-by-block path (isotonic smoothing, a block split over ranks, float32-only bands)."""
        if (smooth or reduce is not None or rows is not None or unused or not self.upper or not chroms
                or np.dtype(band_dtype) not in (np.dtype(np.float64), np.dtype(np.float32))):
            return None
        only32 = np.dtype(band_dtype) == np.float32       # float32 maps only (map-level callers: no exact re-scoring)
        dev, lib = self.dev, self.dev.lib
        geo = []
        for ci in chroms:
            s, e = int(self.offsets[ci]), int(self.offsets[ci + 1])
            n = e - s
            keep = min(max_dist, n) + largest_kernel
            n_diags = min(n, keep + 1)
            if n_diags > 4096:
                return None
            in_w = min(keep, n - 1) + 1
            out_w = min(max_dist, n - 1) + 1
            band = 2 * max(in_w, out_w) < n
            ld = _pitch(in_w, 64) if band else _pitch(n, 16)
            geo.append([ci, s, n, keep, n_diags, in_w, band, ld])
        # lazy64: the float64 band of a banded block is NOT stored beyond its first `near` diagonals (where the runs of a
        # 1-D pattern live); the float64 kernels behind the batched entries recompute the few pixels they read from the
        # pixel table (include/chromosight_hip.h cs_stage_block: a third of the staging pass was writing that band).  The
        # laws and the descriptors then live as long as the blocks do: one shared buffer each, not the rewritten scratch.
        # (lazy64 = "all": every diagonal stored AND a descriptor -- what tests/test_gpu_device_pipeline.py checks the recomputed
        # pixels against)
        near = 64
        lazy = [bool(lazy64 and not only32 and g[6] and (lazy64 == "all" or g[5] > 2 * near)) for g in geo]
        # counts (default wherever the float64 band is lazily evaluated -- the detection path --; float32-only, map-level callers
        # ask for it): the float32 band of a banded block holds the RAW COUNTS, written by the pass that reduces the distance
        # law -- no detrend / tiler pass over the pixel table, no float64 near band; the tile kernel detrends the tiles it
        # fetches (float32) and the float64 kernels every pixel they read (the staging pass's own expression;
        # include/chromosight_hip.h CS_LAYOUT_BAND_COUNTS).  Needs counts that are exact in float32 and not negative (what a
        # .cool holds).  Measured (profiles/r05_counts_band.txt): the C3 step 0.222 -> 0.212 ms and a quarter less HBM
        # traffic; a rank's share of 8 of the C4 genome 0.520 -> 0.507 ms (rank 5) / 0.470 -> 0.445 (rank 0), the whole genome
        # 2.82 -> 2.81 ms.  CHROMOSIGHT_HIP_NO_COUNTS_BAND=1: the detrended bands of the tiler pass.
        if counts is None:
            counts = bool(lazy64) and lazy64 != "all"
        counts = bool(counts and self.counts_ok and not os.environ.get("CHROMOSIGHT_HIP_NO_COUNTS_BAND"))
        cnt = [bool(counts and g[6] and (only32 or lazy[k])) for k, g in enumerate(geo)]
        for k, g in enumerate(geo):
            if cnt[k]:
                g[7] = _pitch(g[5] + 4, 64)                     # (>= 4 zero slots behind every row's diagonals)
        # The laws (and the descriptors of lazily evaluated bands) live as long as the blocks do: one buffer per staging call,
        # held by every block of the call -- never the rewritten law scratch.  A recorded staging (plan.StepPlan replays the
        # cs_stage_blocks table verbatim, d_law included) must not point into a grow-only scratch that a later staging with a
        # longer law re-allocates (silent corruption on the next replay).
        # (counts: CS_COUNTS_LAW_BYTES -- the law, its reciprocals, float32 copies of the reciprocals and of the block's weights)
        law_len = [(2 * g[4] + 2 + (g[4] + 3) // 2 + (g[2] + 1) // 2) if cnt[k] else g[4] for k, g in enumerate(geo)]
        shared = _Shared(self._free, self._resident(8 * sum(law_len) + 256 + LAZY_BAND_BYTES * len(geo)))
        laws = shared.buffer.ptr + LAZY_BAND_BYTES * len(geo)
        table = (CsStageBlock * len(geo))()
        blocks, off = [], 0
        for k, (ci, s, n, keep, n_diags, in_w, band, ld) in enumerate(geo):
            near_k = in_w if lazy64 == "all" else near
            ld64 = (ld if lazy64 == "all" else _pitch(near, 2)) if lazy[k] else ld
            b64 = None if (only32 or cnt[k]) else self._resident(n * ld64 * 8)
            head = COUNTS_HEADER_BYTES if cnt[k] else 0
            b32 = self._resident(n * ld * 4 + head)
            p32 = b32.ptr + head
            desc = shared.buffer.ptr + LAZY_BAND_BYTES * k if lazy[k] else None
            if cnt[k]:
                table[k] = CsStageBlock(s, n, keep, LAYOUT_BAND, in_w, ld, None, p32, laws + off, 0, 0, 1, desc)
            else:
                table[k] = CsStageBlock(s, n, keep, LAYOUT_BAND if band else LAYOUT_DENSE, in_w if band else 0, ld,
                                        b64.ptr if b64 is not None else None, p32, laws + off,
                                        ld64 if lazy[k] else 0, near_k if lazy[k] else 0, 0, desc)
            off += 8 * law_len[k]
            layout = LAYOUT_BAND if band else LAYOUT_DENSE
            flags = _Ptr(self.miss.ptr + s)
            # (the staging pass assembles every row in zeroed pieces of ld slots: the float32 band is zero behind its stored
            # diagonals -- the tile kernel fetches its rim tiles like the inner ones, include/chromosight_hip.h)
            sig32 = CsMatrix(p32, CS_F32, LAYOUT_BAND_COUNTS if cnt[k] else LAYOUT_BAND_PADDED if band and ld >= in_w + 4 else layout, ld, 0,
                             in_w if band else 0, 0)
            if only32:
                sig64 = sig32
            elif lazy[k]:
                sig64 = CsMatrix(desc, CS_F64, LAYOUT_BAND_LAZY, ld, 0, in_w, 0)
            else:
                sig64 = CsMatrix(b64.ptr, CS_F64, layout, ld, 0, in_w if band else 0, 0)
            blk = StagedBlock(self.names[ci], sig64, (n, n), flags, flags, max_dist, False, keep)
            blk.sig32 = sig32
            blk.buffer, blk.pool = (b32 if (only32 or b64 is None) else b64), self._free
            blk.buffer32 = None if (only32 or b64 is None) else b32
            blk.shared = shared                                 # (the law buffer of the call)
            if lazy[k]:
                blk.genome = self                               # (the descriptor points into the pixel table)
                blk.restage = (lambda ci=ci: self._restage_synced(ci, max_dist, largest_kernel, band_dtype=band_dtype))
            blocks.append(blk)
        genome = CsCsr(self.n_bins, self.n_bins, max(self.nnz, 1), self.indptr.ptr, self.indices.ptr, self.data.ptr,
                       np_dtype_code(self.val_dtype), 0, None, self.weight.ptr, self.weight.ptr)
        # asynchronous on `stream`: whoever reads the blocks on another stream or context synchronises first (the callers
        # that hand blocks to worker threads do; the law scratch is rewritten in stream order)
        with dev.lock:
            dev._check(lib.cs_stage_blocks(dev.ctx, stream, C.byref(genome), table, len(geo), 10.0))
        return blocks

    def _restage_synced(self, ci, max_dist, largest_kernel, **options):
        """One block staged again with its float64 band stored (StagedBlock.full), COMPLETE on return: the staging call is
        asynchronous on this genome's stream, and who asks for the full band is a per-block path on a worker thread with a
        context and stream of its own -- it read a band that was still being written (a 2-D pattern with pearson < 0.1 scanned
        beside a 1-D pattern lost a third of its records, differently every run: found by
        tests/test_gpu_device_pipeline.py::test_2d_chain_retry_and_fallback_paths)."""
        with self._stage_lock:
            block = self.stage_blocks([ci], max_dist, largest_kernel, **options)[0]
            self.dev.sync()
        return block

    def view_for(self, block, max_dist, largest_kernel):
        """The staged block of the same chromosome for a pattern with a shorter scanning distance, WITHOUT staging it again:
        ContactMap.create_mat with keep' = min(max_dist, n) + largest_kernel <= keep is the band of the first keep' + 1
        diagonals of the block staged with `keep` -- the distance law of a diagonal does not depend on how many others are
        kept (preprocessing.py:173-188, without --smooth-trend), so the detrended values are the same numbers, and a band
        view with fewer stored diagonals reads every pixel beyond them as 0 (diag_trim).  None when a view cannot stand in
        (dense-staged short chromosomes, row windows, a longer distance than the block holds)."""
        n = block.shape[0]
        keep = min(max_dist, n) + largest_kernel
        if (block.inter or block.sig.layout not in (LAYOUT_BAND, LAYOUT_BAND_LAZY, LAYOUT_BAND_PADDED, LAYOUT_BAND_COUNTS, LAYOUT_BAND_COUNTS_VIEW)
                or getattr(block, "row_window", None) is not None
                or block.keep is None or keep > block.keep or getattr(block, "smooth", False)):
            return None
        in_w = min(keep, n - 1) + 1
        out_w = min(max_dist, n - 1) + 1
        if not 2 * max(in_w, out_w) < n:
            return None                              # this pattern would have been staged dense: keep the layouts identical
        s = block.sig
        # (a narrower view of a zero-padded band is a plain band: the slots behind ITS diagonals hold the block's further ones)
        narrower = {LAYOUT_BAND_PADDED: LAYOUT_BAND, LAYOUT_BAND_COUNTS: LAYOUT_BAND_COUNTS_VIEW}
        view = StagedBlock(block.name, CsMatrix(s.d_ptr, s.dtype, narrower.get(s.layout, s.layout), s.ld, 0, in_w, s.row0), block.shape, block.miss_row,
                           block.miss_col, max_dist, False, keep)
        if s.layout == LAYOUT_BAND_LAZY:                 # (a view shares the descriptor; read by itself it is staged for itself)
            ci = self.names.index(block.name)
            view.restage = lambda: self._restage_synced(ci, max_dist, largest_kernel)
        if block.sig32 is not None:
            t = block.sig32
            view.sig32 = CsMatrix(t.d_ptr, t.dtype, LAYOUT_BAND_COUNTS_VIEW if t.layout in (LAYOUT_BAND_COUNTS, LAYOUT_BAND_COUNTS_VIEW) else LAYOUT_BAND,
                                  t.ld, 0, in_w, t.row0)
        view.parent = block                          # keeps the buffers alive; the view owns none
        return view

    def workers(self, n=4):
        # process-wide: the workers' contexts, streams and scratch outlive this DeviceCool (creating them costs
        # several milliseconds, more than a whole detect on a small genome)
        key = (self.dev.index, n)
        if key not in _WORKER_POOLS:
            _WORKER_POOLS[key] = _Workers(self.dev.index, n)
        return _WORKER_POOLS[key]

    @property
    def n_chrom(self):
        return len(self.offsets) - 1

    def chrom_size(self, ci):
        return int(self.offsets[ci + 1] - self.offsets[ci])

    def _view(self, s, e, cs, ce, begin=None, end=None):
        return CsCsr(e - s, ce - cs, max(self.nnz, 1),
                     begin if begin is not None else self.indptr.ptr + 8 * s, self.indices.ptr, self.data.ptr,
                     np_dtype_code(self.val_dtype), cs, end, self.weight.ptr + 8 * s, self.weight.ptr + 8 * cs)

    def stage_intra(self, *args, dev=None, **options):
        """_stage_intra holding the context it uses (one call in flight per context: engine._one_call_per_context)."""
        with (dev or self.dev).lock:
            return self._stage_intra(*args, dev=dev, **options)

    def stage_inter(self, *args, **options):
        with self.dev.lock:
            return self._stage_inter(*args, **options)

    def _stage_intra(self, ci, max_dist, largest_kernel, smooth=False, band_dtype=np.float64, name=None, stream=None,
                     resident=False, rows=None, reduce=None, dev=None, ext=None):
        """ContactMap.create_mat of one balanced intra block, on the device: distance law over the
        first keep_distance diagonals of the detectable bins, detrend, >= 10 -> 1, NaN -> 0, upper
        band only (contacts_map.py:527-548, 603-638; preprocessing.py:129-197, 256-310).

        rows = (a, b): stage only the rows a <= i < b of the block plus the template's halo of
        (largest_kernel - 1) // 2 rows on either side (one block split over several GPUs, SURVEY.md
        8(e)); the distance law is then the sum of every part's per-diagonal (sum, count): `reduce`
        maps this part's float64 array (2, n_diags) to the total (an all-reduce).  The block's
        `row_window` is (a, b) and its matrix carries row0."""
        # dev / ext: a worker's own context and extent scratch (stage_blocks); the buffers live in the device's one
        # address space whichever context allocated them
        dev = dev or self.dev
        lib = dev.lib
        s, e = int(self.offsets[ci]), int(self.offsets[ci + 1])
        n = e - s
        keep = min(max_dist, n) + largest_kernel
        n_diags = min(n, keep + 1)
        in_w = min(keep, n - 1) + 1
        out_w = min(max_dist, n - 1) + 1
        # layout rule of the host path (_Staged): band when it is less than half of the dense map
        band = 2 * max(in_w, out_w) < n
        a, b = (0, n) if rows is None else (int(rows[0]), int(rows[1]))
        if not (0 <= a < b <= n):
            raise ValueError("row window outside the block")
        split = (a, b) != (0, n)
        if split and not band:
            raise ValueError("only blocks staged as a band can be split by rows")
        halo = (largest_kernel - 1) // 2 if split else 0
        ra, rb = max(0, a - halo), min(n, b + halo)
        m = rb - ra
        ext = (ext or self._ext).get(16 * m + 16 * (keep + 2) + 512)
        d_begin, d_end = ext, ext + 8 * m
        d_sum = d_end + 8 * m
        d_cnt, d_law = d_sum + 8 * (keep + 2), None
        # a view of the rows ra .. rb-1 whose column origin moves with it: diagonals are unchanged
        raw = self._view(s + ra, s + rb, s + ra, e)
        dev._check(lib.cs_csr_band_extent(dev.ctx, stream, C.byref(raw), 0, keep, d_begin, d_end))
        view = self._view(s + ra, s + rb, s + ra, e, d_begin, d_end)
        own = self._view(s + a, s + b, s + a, e, d_begin + 8 * (a - ra), d_end + 8 * (a - ra)) if split else view
        dev._check(lib.cs_distance_law_csr(dev.ctx, stream, C.byref(own), self.det.ptr + s + a, n_diags, d_sum, d_cnt))
        if (smooth and n > 2) or reduce is not None:
            sums = np.empty(n_diags)
            cnts = np.empty(n_diags, dtype=np.int64)
            dev._check(lib.cs_memcpy_d2h(dev.ctx, sums.ctypes.data, d_sum, 8 * n_diags, stream))
            dev._check(lib.cs_memcpy_d2h(dev.ctx, cnts.ctypes.data, d_cnt, 8 * n_diags, stream))
            if reduce is not None:
                total = np.asarray(reduce(np.stack([sums, cnts.astype(np.float64)])), dtype=np.float64)
                sums, cnts = np.ascontiguousarray(total[0]), np.rint(total[1]).astype(np.int64)
            with np.errstate(invalid="ignore", divide="ignore"):
                law = np.where(cnts > 0, sums / np.maximum(cnts, 1), 0.0)      # cs_distance_law_finish
            if smooth and n > 2:
                full_law = np.zeros(n)
                full_law[:n_diags] = law
                law = preproc._isotonic_non_increasing(full_law)[:n_diags]
            law = np.ascontiguousarray(law, dtype=np.float64)
            law[np.isnan(law)] = 0.0
            dev._check(lib.cs_memcpy_h2d(dev.ctx, d_sum, law.ctypes.data, 8 * n_diags, stream))
            d_law = d_sum
        else:
            d_law = d_sum                                     # finished in place
            dev._check(lib.cs_distance_law_finish(dev.ctx, stream, d_sum, d_cnt, n_diags, d_law))
        esz = np.dtype(band_dtype).itemsize
        ld = _pitch(in_w, 64) if band else _pitch(n, 16)
        # resident: the block owns its buffer (288 GB of HBM hold every block of a genome at once, so
        # blocks are staged once and reused by all templates / iterations); else a shared scratch
        buf = self._resident(m * ld * esz, dev) if resident else None
        ptr = buf.ptr if resident else self._band.get(m * ld * esz)
        sig = CsMatrix(ptr, np_dtype_code(band_dtype), LAYOUT_BAND if band else LAYOUT_DENSE, ld, 0, in_w if band else 0, ra)
        dev._check(lib.cs_csr_to_band(dev.ctx, stream, C.byref(view), d_law, n_diags, 10.0, C.byref(sig)))
        flags = _Ptr(self.miss.ptr + s)
        block = StagedBlock(name or self.names[ci], sig, (n, n), flags, flags, max_dist, False, keep)
        block.buffer, block.pool = buf, self._free
        block.row_window = (a, b) if split else None
        block.smooth = bool(smooth)
        return block

    def _stage_inter(self, ca, cb, name=None, stream=None, resident=False, dtype=np.float64):
        """ContactMap.create_mat of a balanced inter-chromosomal block (ca < cb), on the device:
        NaN -> 0, divided by the median of its stored values (contacts_map.py:598-601), dense layout."""
        dev, lib = self.dev, self.dev.lib
        s1, e1 = int(self.offsets[ca]), int(self.offsets[ca + 1])
        s2, e2 = int(self.offsets[cb]), int(self.offsets[cb + 1])
        n_r, n_c = e1 - s1, e2 - s2
        ext = self._ext.get(16 * n_r + 8 * max(n_r, n_c) + 512)
        d_begin, d_end, d_law = ext, ext + 8 * n_r, ext + 16 * n_r
        raw = self._view(s1, e1, s2, e2)
        dev._check(lib.cs_csr_band_extent(dev.ctx, stream, C.byref(raw), -n_r, n_c, d_begin, d_end))
        view = self._view(s1, e1, s2, e2, d_begin, d_end)
        med = C.c_double(0.0)
        dev._check(lib.cs_csr_median(dev.ctx, stream, C.byref(view), C.byref(med)))
        scale = np.full(max(n_r, n_c), med.value)
        dev._check(lib.cs_memcpy_h2d(dev.ctx, d_law, scale.ctypes.data, scale.nbytes, stream))
        esz = np.dtype(dtype).itemsize
        ld = (n_c + 15) // 16 * 16
        own = self._resident(n_r * ld * esz) if resident else None
        ptr = own.ptr if resident else self._band.get(n_r * ld * esz)
        sig = CsMatrix(ptr, np_dtype_code(dtype), LAYOUT_DENSE, ld, 0, 0)
        # "law" = the median on every diagonal, no cap: value / median, NaN -> 0
        dev._check(lib.cs_csr_to_band(dev.ctx, stream, C.byref(view), d_law, scale.size, 0.0, C.byref(sig)))
        block = StagedBlock(name or f"{self.names[ca]}-{self.names[cb]}", sig, (n_r, n_c), _Ptr(self.miss.ptr + s1),
                            _Ptr(self.miss.ptr + s2), None, True, None)
        block.buffer, block.pool = own, self._free
        return block

    def stage_inter_many(self, pairs, dtype=np.float64, stream=None):
        """stage_inter(resident=True) for SEVERAL inter-chromosomal blocks: the extents of all of them, then their medians with
        one native call (cs_csr_median_many: two synchronisations in all instead of two per block), then the dense maps."""
        pairs = list(pairs)
        if not pairs:
            return []
        with self.dev.lock:
            dev, lib = self.dev, self.dev.lib
            geo, total = [], 0
            for ca, cb in pairs:
                s1, e1 = int(self.offsets[ca]), int(self.offsets[ca + 1])
                s2, e2 = int(self.offsets[cb]), int(self.offsets[cb + 1])
                n_r, n_c = e1 - s1, e2 - s2
                geo.append((ca, cb, s1, e1, s2, e2, n_r, n_c, total))
                total += (16 * n_r + 511) // 256 * 256
            # (the blocks' "laws" -- their median on every diagonal -- side by side behind the row extents: ONE upload)
            law_off, laws = [], 0
            for _, _, _, _, _, _, n_r, n_c, _ in geo:
                law_off.append(laws)
                laws += (max(n_r, n_c) + 31) // 32 * 32
            ext = self._ext.get(total + 8 * laws + 512)
            views = (type(self._view(0, 1, 0, 1)) * len(pairs))()
            for i, (ca, cb, s1, e1, s2, e2, n_r, n_c, off) in enumerate(geo):
                d_begin, d_end = ext + off, ext + off + 8 * n_r
                raw = self._view(s1, e1, s2, e2)
                dev._check(lib.cs_csr_band_extent(dev.ctx, stream, C.byref(raw), -n_r, n_c, d_begin, d_end))
                views[i] = self._view(s1, e1, s2, e2, d_begin, d_end)
            med = (C.c_double * len(pairs))()
            dev._check(lib.cs_csr_median_many(dev.ctx, stream, views, len(pairs), med))
            out = []
            esz = np.dtype(dtype).itemsize
            scale = np.empty(laws)
            for i, (_, _, _, _, _, _, n_r, n_c, _) in enumerate(geo):
                scale[law_off[i]:law_off[i] + max(n_r, n_c)] = med[i]
            dev._check(lib.cs_memcpy_h2d(dev.ctx, ext + total, scale.ctypes.data, scale.nbytes, stream))
            for i, (ca, cb, s1, e1, s2, e2, n_r, n_c, off) in enumerate(geo):
                d_law = ext + total + 8 * law_off[i]
                ld = (n_c + 15) // 16 * 16
                own = self._resident(n_r * ld * esz)
                sig = CsMatrix(own.ptr, np_dtype_code(dtype), LAYOUT_DENSE, ld, 0, 0)
                # "law" = the median on every diagonal, no cap: value / median, NaN -> 0
                dev._check(lib.cs_csr_to_band(dev.ctx, stream, C.byref(views[i]), d_law, max(n_r, n_c), 0.0, C.byref(sig)))
                block = StagedBlock(f"{self.names[ca]}-{self.names[cb]}", sig, (n_r, n_c), _Ptr(self.miss.ptr + s1),
                                    _Ptr(self.miss.ptr + s2), None, True, None)
                block.buffer, block.pool = own, self._free
                out.append(block)
            return out

    def subsampled(self, sample, seed=0, inter=False):
        """A DeviceCool whose counts are a random subsample of this one's, drawn per sub-matrix without
        replacement like the reference's --subsample (contacts_map.py:552-596, preprocessing.py:359-401):
        `sample` in (0, 1] is the proportion of the contacts of every sub-matrix to keep.  Intra blocks
        are sampled as the reference sees them (symmetric matrix, both triangles drawn independently;
        the path then only reads the upper one).  Unlike the reference's unseeded np.random.choice the
        draw is reproducible: numpy Generator(seed), multivariate hypergeometric."""
        sample = float(sample)
        if sample < 0:
            raise ValueError("Subsample must be strictly positive.")
        if sample > 1:
            raise ValueError("Subsample cannot be above 1")
        rng = np.random.default_rng(seed)
        h = self.host
        b1, b2, cnt = h["bin1_id"], h["bin2_id"], np.asarray(h["count"]).astype(np.int64)
        off = self.offsets
        chrom_of = np.repeat(np.arange(self.n_chrom), np.diff(off))
        c1, c2 = chrom_of[b1], chrom_of[b2]
        new = np.zeros_like(cnt)
        for ca in range(self.n_chrom):
            for cb in range(ca, self.n_chrom if inter else ca + 1):
                sel = np.flatnonzero((c1 == ca) & (c2 == cb))
                if sel.size == 0:
                    continue
                x = cnt[sel]
                if ca == cb:
                    offd = b1[sel] != b2[sel]
                    pool = np.concatenate([x, x[offd]])            # upper triangle + diagonal, then the mirror
                    keep = int(sample * pool.sum())
                    drawn = rng.multivariate_hypergeometric(pool, keep, method="marginals") if keep < pool.sum() else pool
                    new[sel] = drawn[:x.size]
                else:
                    keep = int(sample * x.sum())
                    new[sel] = rng.multivariate_hypergeometric(x, keep, method="marginals") if keep < x.sum() else x
        cool = dict(h)
        nz = new > 0
        cool["bin1_id"], cool["bin2_id"], cool["count"] = b1[nz], b2[nz], new[nz]
        return DeviceCool(cool, self.dev)

    def bins_of(self, chroms, positions):
        """Whole-genome bin of (chromosome name, base pair) pairs; -1 outside the genome
        (HicGenome.coords_to_bins, contacts_map.py:404-450, for fixed-size bins)."""
        ci = pd.Categorical(pd.Series(np.asarray(chroms, dtype=object)).astype(str), categories=list(self.names)).codes.astype(np.int64)
        local = np.asarray(positions, dtype=np.int64) // self.binsize
        sizes = np.diff(self.offsets)
        ok = (ci >= 0) & (local >= 0) & (local < sizes[np.maximum(ci, 0)])
        return np.where(ok, self.offsets[np.maximum(ci, 0)] + local, -1)

    def block_bins(self, ci):
        s, e = int(self.offsets[ci]), int(self.offsets[ci + 1])
        return np.flatnonzero(~self.miss_host[s:e])


def detect_block(dcool, block, kernel_config, kernel, tsvd=None, coords=None, want_windows=True, raw=False, dev=None,
                 stream=None, all_gather=None, defer=False):
    """pattern_detector(full=True) on a staged block (cli/chromosight.py:601-614).  A block staged as a
    row window (stage_intra(rows=...)) is this rank's part of a sub-matrix split over several GPUs:
    `all_gather` exchanges the candidate pixels and the records (detect_split_on_device)."""
    kernel = np.asarray(kernel, dtype=np.float64)
    if min(block.shape) <= max(kernel.shape):
        return None, None
    _check_template(kernel)
    kspec = engine.KernelSpec(kernel, tsvd)
    block = block.full() if hasattr(block, "full") else block     # (the per-block entries read the float64 band themselves)
    if getattr(block, "row_window", None) is not None:
        if coords is not None or all_gather is None:
            raise ValueError("a row window of a block needs detect mode and an all_gather")
        return cid.detect_split_on_device(dev or dcool.dev, block.sig, block.shape, block.row_window, kspec,
                                          kernel_config, block.miss_row, block.miss_col, max_dist=block.max_dist,
                                          all_gather=all_gather, want_windows=want_windows, raw=raw, stream=stream)
    return cid.detect_on_device(dev or dcool.dev, block.sig, block.shape, kspec, kernel_config, block.miss_row,
                                block.miss_col, inter=block.inter, max_dist=block.max_dist, full=True, coords=coords,
                                want_windows=want_windows, raw=raw, stream=stream, defer=defer)


def _check_template(kernel):
    """normxcorr2's template check (detection.py:888-889).  It is what stops the reference when a 1-D pattern is
    iterated: the windows of intra maps carry NaN on the first sub-diagonals (detection.py:300-310), so does their
    pileup, and `kernel.std() > 0` is False for a template with NaN (tests/golden/iterations.npz: borders_error)."""
    if not (np.std(kernel) > 0):
        raise ValueError("Cannot have flat kernel.")


def b_is_band(block):
    return block.sig.layout in (LAYOUT_BAND, LAYOUT_BAND_LAZY, LAYOUT_BAND_PADDED) and not block.inter and getattr(block, "row_window", None) is None


def detect_blocks(dcool, blocks, kernel_config, kernel, tsvd=None, raw=True, workers=4, batch=True, want_windows=True, defer=False,
                  dev=None, stream=None, merged=False, exclusive=True):
    """detect_block for every staged block of `blocks`; results in the order of `blocks`.  Banded intra blocks go to the
    device in ONE native call per template (1-D patterns: cs_detect_foci_batch; 2-D patterns: cs_detect_foci_blocks);
    what the library cannot batch (dense-staged short chromosomes, inter blocks, odd templates) goes block by block,
    several at a time (see _Workers).  defer=True: returns a callable that yields the results -- the native calls are
    done when detect_blocks returns, the acceptance rules (numpy on the returned records) run in the callable, so a
    caller can overlap them with the next template's device work (parallel.detect_genome).  dev / stream: another
    context and stream of the same GPU for the batched calls (templates scanned concurrently by several host threads).
    merged=True: when ONE native call covered every block, the result is the tuple (table of all blocks (k, 4), accepted
    records per block, windows or None) instead of the per-block list (the genome drivers concatenate the tables anyway).
    exclusive=False: other templates / patterns are being scanned on this GPU at the same time (parallel.detect_patterns)."""
    kernel = np.asarray(kernel, dtype=np.float64)
    bdev = dev or dcool.dev
    done = (lambda res: (lambda: res)) if defer else (lambda res: res)
    if any(min(b.shape) > max(kernel.shape) for b in blocks):
        _check_template(kernel)
    square = kernel.shape[0] == kernel.shape[1]
    if len(blocks) > 1 and raw and batch and square and tsvd is None:
        live = [k for k, b in enumerate(blocks) if min(b.shape) > max(kernel.shape)]
        banded = [k for k in live if b_is_band(blocks[k])]
        kspec = engine.KernelSpec(kernel, tsvd)
        fin = None
        if len(banded) > 1:
            many = cid.detect_many_on_device if kernel_config["max_dist"] == 0 else cid.detect_blocks_on_device
            extra = dict(raw=True) if kernel_config["max_dist"] == 0 else dict(exclusive=exclusive)
            whole = merged and len(banded) == len(blocks)
            fin = many(bdev, [blocks[k] for k in banded], kspec, kernel_config, want_windows=want_windows, defer=True, stream=stream,
                       merged=whole, **extra)
            if fin is not None and whole:
                return fin if defer else fin()
        if fin is not None:
            rest = {k: detect_block(dcool, blocks[k], kernel_config, kernel, tsvd=tsvd, raw=raw, want_windows=want_windows, dev=dev,
                                    stream=stream)
                    for k in live if k not in banded}

            def finish():
                out = [(None, None)] * len(blocks)
                for k, r in zip(banded, fin()):
                    out[k] = r
                for k, r in rest.items():
                    out[k] = r
                return out

            return finish if defer else finish()
    elif len(blocks) > 1 and kernel_config["max_dist"] == 0 and batch and square:
        # 1-D patterns, tables as DataFrames: the same batch, accepted block by block
        live = [k for k, b in enumerate(blocks) if min(b.shape) > max(kernel.shape)]
        banded = [k for k in live if b_is_band(blocks[k])]
        # (on the caller's context and stream: several host threads may be scanning templates side by side)
        res = cid.detect_many_on_device(bdev, [blocks[k] for k in banded], engine.KernelSpec(kernel, tsvd), kernel_config,
                                        raw=raw, want_windows=want_windows, stream=stream) if len(banded) > 1 else None
        if res is not None:
            out = [(None, None)] * len(blocks)
            for k, r in zip(banded, res):
                out[k] = r
            for k in live:
                if k not in banded:
                    out[k] = detect_block(dcool, blocks[k], kernel_config, kernel, tsvd=tsvd, raw=raw, want_windows=want_windows,
                                          dev=dev, stream=stream)
            return done(out)
    if workers <= 1 or len(blocks) <= 1 or dev is not None:
        # (a caller that brought its own context and stream -- a template scanned beside others -- keeps every call on them:
        # that stream is the one that waited for the staging of the blocks)
        return done([detect_block(dcool, b, kernel_config, kernel, tsvd=tsvd, raw=raw, want_windows=want_windows, dev=dev, stream=stream)
                     for b in blocks])
    dcool.dev.sync()
    pool = dcool.workers(workers)

    # raw tables of banded intra blocks: the workers only run the native call; the acceptance rules are applied to all
    # their records in one go (numpy under the interpreter lock was a good part of a worker's time on a small block)
    deferred = raw and square and all(b_is_band(b) and min(b.shape) > max(kernel.shape) for b in blocks)

    def one(block):
        dev, stream = pool.device()
        return detect_block(dcool, block, kernel_config, kernel, tsvd=tsvd, raw=raw, dev=dev, stream=stream,
                            want_windows=want_windows, defer=deferred)

    results = pool.map(one, blocks)
    if not deferred:
        return done(results)
    counts = [len(r[0]) for r in results]
    if sum(counts) == 0:
        return done([(None, None)] * len(blocks))
    rec = np.concatenate([r[0] for r in results])
    windows = np.concatenate([r[1] for r in results]) if want_windows else None
    return done(cid.accept_many(blocks, rec, windows, counts, engine.KernelSpec(kernel, tsvd), kernel_config, pvals=True))


def detect_blocks_templates(dcool, blocks, kernel_config, kernels, want_windows=False, dev=None, stream=None, begin_only=False):
    """A 1-D pattern's templates (one size: the three borders templates, the one hairpins template) on banded intra blocks with ONE native call
    (cs_detect_foci_batch_templates) instead of one launch chain per template.  Returns a callable that yields, per template,
    the merged result of detect_blocks (table of all blocks, accepted records per block, windows) -- the native call is done
    when this returns, the acceptance rules run in the callable -- or None when the entry does not apply.
    begin_only=True: the chain is only enqueued on (dev, stream) when this returns; the callable waits for it first."""
    kernels = [np.asarray(k, dtype=np.float64) for k in kernels]
    if kernel_config["max_dist"] != 0 or len(kernels) < 1 or len(kernels) > 4 or len(blocks) < 1:
        return None
    if any(k.shape != kernels[0].shape or k.shape[0] != k.shape[1] for k in kernels):
        return None
    if not all(b_is_band(b) and min(b.shape) > max(kernels[0].shape) for b in blocks):
        return None
    for k in kernels:
        _check_template(k)
    return cid.detect_many_on_device(dev or dcool.dev, blocks, [engine.KernelSpec(k) for k in kernels], kernel_config,
                                     want_windows=want_windows, raw=True, stream=stream, defer=True, merged=True, begin_only=begin_only)


def with_win_size(kernel_config, win_size):
    """--win-size of the reference's CLI (cli/chromosight.py:365-370, 689-695): every template of the
    config resized to win_size x win_size (odd), "auto" / None keeps the config's own sizes."""
    if win_size is None or win_size == "auto":
        return kernel_config
    win_size = int(win_size)
    if not win_size % 2:
        raise ValueError("--win-size must be odd")
    cfg = dict(kernel_config)
    cfg["kernels"] = [preproc.resize_kernel(np.asarray(k, dtype=np.float64), factor=win_size / np.shape(k)[0], quiet=True)
                      for k in kernel_config["kernels"]]
    return cfg


def sub_matrices(dcool, inter):
    """(chrom a, chrom b) of every sub-matrix in the reference's order (contacts_map.py:274-312)."""
    return [(a, b) for a in range(dcool.n_chrom) for b in range(dcool.n_chrom) if a == b or (a < b and inter)]


def detect(cool, kernel_config, tsvd=None, smooth=False, band_dtype=np.float64, inter=False, subsample=None, seed=0,
           return_windows=False, win_size=None):
    """`chromosight detect` (balanced matrix) on a decoded cool (dict) or a DeviceCool; options
    --inter, --smooth-trend, --tsvd, --subsample (seeded), --iterations through the config.  Every
    block is staged once in HBM (distance law, detrend, band / median scaling) and stays resident
    across templates and iterations; each (block, template) is one native call.
    Returns the output table (same columns and row order as the reference's <prefix>.tsv); with
    return_windows also the windows of its rows (what the reference saves as <prefix>.json / .npy)."""
    kernel_config = with_win_size(kernel_config, win_size)
    dcool = cool if isinstance(cool, DeviceCool) else DeviceCool(cool)
    if subsample is not None:
        dcool = dcool.subsampled(subsample, seed=seed, inter=inter)
    binsize = dcool.binsize
    off = dcool.offsets
    names = dcool.names
    n_chrom = dcool.n_chrom
    max_dist = max(kernel_config["max_dist"] // binsize, 1)
    largest = max(np.shape(k)[0] for k in kernel_config["kernels"])
    if not inter and not return_windows and not smooth and np.dtype(band_dtype) == np.float64 and dcool.upper and hasattr(dcool, "view_for"):
        # ONE orchestration: the intra-chromosomal blocks of a plain `detect` go through the genome drivers bench.py times
        # (parallel.genome_step: blocks staged by one native call, batched chains, and -- for the configurations a StepPlan covers,
        # chromosight_amd/plan.py -- every call after the first on this DeviceCool as one native call list); the per-block
        # machinery below keeps --inter, --smooth-trend and the runs that return windows.
        from . import parallel
        # (local: `detect` is this process's call -- under an initialised process group it neither shards nor gathers; the sharded
        # driver is parallel.detect_genome / genome_step)
        rec = parallel.genome_step(dcool, [kernel_config], tsvd=tsvd, local=True)[0]
        if rec.shape[0] == 0:
            return pd.DataFrame(columns=OUTPUT_COLUMNS)
        first = np.asarray(off, dtype=np.int64)[rec[:, 0].astype(np.int64)]
        coords = {"bin1": rec[:, 1].astype(np.int64) + first, "bin2": rec[:, 2].astype(np.int64) + first, "score": rec[:, 3],
                  "pvalue": rec[:, 4], "kernel_id": rec[:, 5].astype(np.int64), "iteration": rec[:, 6].astype(np.int64)}
        return postprocess(coords, kernel_config, binsize, off, names, dcool.bin_start, dcool.bin_end)
    pairs = sub_matrices(dcool, inter)
    intra = dict(zip([a for a, b in pairs if a == b],
                     dcool.stage_blocks([a for a, b in pairs if a == b], max_dist, largest, smooth=smooth, band_dtype=band_dtype)))
    blocks = [intra[a] if a == b else dcool.stage_inter(a, b, resident=True) for a, b in pairs]
    all_coords, all_windows = [], []
    for kernel_id, kernel in enumerate(kernel_config["kernels"]):
        for it in range(kernel_config["max_iterations"]):
            tables, windows = [], []
            # windows feed the next iteration's template and the optional output; otherwise they stay on the device
            need_windows = return_windows or it + 1 < kernel_config["max_iterations"]
            results = detect_blocks(dcool, blocks, kernel_config, kernel, tsvd=tsvd, raw=True, want_windows=need_windows)
            for (ca, cb), (tab, win) in zip(pairs, results):
                if tab is None or len(tab) == 0:
                    continue
                tab[:, 0] += int(off[ca])
                tab[:, 1] += int(off[cb])
                tables.append(tab)
                windows.append(win)
            if not tables:
                break
            rec = np.concatenate(tables, axis=0)
            all_coords.append({"bin1": rec[:, 0].astype(np.int64), "bin2": rec[:, 1].astype(np.int64), "score": rec[:, 2],
                               "pvalue": rec[:, 3], "kernel_id": np.full(len(rec), kernel_id, dtype=np.int64),
                               "iteration": np.full(len(rec), it, dtype=np.int64)})
            if need_windows:
                kernel_windows = np.concatenate(windows, axis=0)
                all_windows.append(kernel_windows)
                kernel = cid.pileup_patterns(kernel_windows)
    if not all_coords:
        empty = pd.DataFrame(columns=OUTPUT_COLUMNS)
        return (empty, np.zeros((0,) + np.shape(kernel_config["kernels"][0]))) if return_windows else empty
    coords = {k: np.concatenate([c[k] for c in all_coords]) for k in all_coords[0]}
    # windows of different templates may differ in size only across configs, never inside one
    windows = np.concatenate(all_windows, axis=0) if return_windows else None
    return postprocess(coords, kernel_config, binsize, off, names, dcool.bin_start, dcool.bin_end, windows=windows)


def detect_to_files(cool, kernel_config, prefix, win_fmt="json", dec=10, **options):
    """detect + the two files `chromosight detect` leaves behind (cli/chromosight.py:873-881):
    <prefix>.tsv and <prefix>.json | .npy.  Returns the table."""
    from . import io as cio
    cio.check_prefix_dir(prefix)
    table, windows = detect(cool, kernel_config, return_windows=True, **options)
    cio.write_patterns(table, prefix, dec=dec)
    cio.save_windows(windows, prefix, fmt=win_fmt)
    return table


def postprocess(coords, kernel_config, binsize, off, names, bin_start, bin_end, windows=None):
    """cmd_detect after the per-block loop (cli/chromosight.py:806-871): neighbour removal, bins ->
    genomic coordinates, min_dist and NaN-p filters, Benjamini-Hochberg q-values, column order.
    `coords`: DataFrame or dict of columns bin1, bin2, score, pvalue, kernel_id, iteration.
    `windows` (one per row of coords) are filtered alongside and returned as a second value.
    Column arithmetic is plain numpy and the table is built once (pandas' per-column inserts cost more
    than the detection itself on a small genome)."""
    col = {k: np.asarray(coords[k]) for k in ("bin1", "bin2", "score", "pvalue", "kernel_id", "iteration")}
    b1 = col["bin1"].astype(np.int64, copy=False)
    b2 = col["bin2"].astype(np.int64, copy=False)
    separation = max(int(kernel_config["min_separation"] // binsize), 1)
    sel = np.flatnonzero(cid.remove_neighbours_arrays(b1, b2, col["score"], win_size=separation))
    # bins -> genomic coordinates
    bin_chrom = np.repeat(np.arange(len(names)), np.diff(off))
    b1, b2 = b1[sel], b2[sel]
    c1, c2 = bin_chrom[b1], bin_chrom[b2]
    bin_start, bin_end = np.asarray(bin_start), np.asarray(bin_end)
    s1, s2 = bin_start[b1], bin_start[b2]
    pval = col["pvalue"][sel].astype(np.float64, copy=False)
    too_close = (c1 == c2) & (np.abs(s2 - s1) < kernel_config["min_dist"])
    keep = ~too_close & ~np.isnan(pval)
    sel, b1, b2, c1, c2, pval = sel[keep], b1[keep], b2[keep], c1[keep], c2[keep], pval[keep]
    name_arr = np.asarray(names, dtype=object)
    table = pd.DataFrame({
        "chrom1": name_arr[c1], "start1": bin_start[b1], "end1": bin_end[b1],
        "chrom2": name_arr[c2], "start2": bin_start[b2], "end2": bin_end[b2],
        "bin1": b1, "bin2": b2, "kernel_id": col["kernel_id"][sel], "iteration": col["iteration"][sel],
        "score": col["score"][sel], "pvalue": pval, "qvalue": fdr_correction(pval)}, columns=OUTPUT_COLUMNS)
    if windows is not None:
        return table, windows[sel]
    return table


def quantify(cool, positions, kernel_config, inter=False, tsvd=None, subsample=None, seed=0, smooth=False,
             max_dist_bp=None, win_size=None, shard=None):
    """`chromosight quantify` (cli/chromosight.py:264-470): score the given 2-D positions with every
    template of the config and keep, per position, the row the reference keeps (sorted by score,
    last of each (chrom1, start1, chrom2, start2) group).  `positions`: DataFrame with chrom1, start1,
    end1, chrom2, start2, end2.  Returns (table in the reference's output order and columns, windows).
    shard: parallel.QuantifyShard -- the sub-matrices are dealt to the ranks (parallel.quantify_genome)."""
    dcool = cool if isinstance(cool, DeviceCool) else DeviceCool(cool)
    if subsample is not None:
        dcool = dcool.subsampled(subsample, seed=seed, inter=inter)
    cfg = dict(with_win_size(kernel_config, win_size))
    bed2d = positions.loc[:, ["chrom1", "start1", "end1", "chrom2", "start2", "end2"]].reset_index(drop=True).copy()
    furthest = np.max(bed2d.start2 - bed2d.start1)
    cfg["max_dist"] = min(furthest, dcool.n_bins * dcool.binsize)       # scan up to the furthest pattern
    if max_dist_bp is not None:
        cfg["max_dist"] = int(max_dist_bp)                              # (tests: pin the scanning distance)
    cfg["min_dist"] = 0
    kernels = [np.asarray(k, dtype=np.float64) for k in cfg["kernels"]]
    km, kn = kernels[0].shape
    max_dist = max(cfg["max_dist"] // dcool.binsize, 1)
    largest = max(k.shape[0] for k in kernels)
    # chromosome of every position, once (the four bin lookups below share it); HicGenome.coords_to_bins
    # (contacts_map.py:404-450): -1 where the genome has no such bin
    c1 = pd.Categorical(bed2d.chrom1.astype(str), categories=list(dcool.names)).codes.astype(np.int64)
    c2 = pd.Categorical(bed2d.chrom2.astype(str), categories=list(dcool.names)).codes.astype(np.int64)
    sizes = np.diff(dcool.offsets)

    def bins(codes, bp):
        local = np.asarray(bp, dtype=np.int64) // dcool.binsize
        ok = (codes >= 0) & (local >= 0) & (local < sizes[np.maximum(codes, 0)])
        return np.where(ok, dcool.offsets[np.maximum(codes, 0)] + local, -1)

    s1, e1, s2, e2 = (bed2d[c].to_numpy() for c in ("start1", "end1", "start2", "end2"))
    g1 = bins(c1, (s1 + e1) // 2)
    g2 = bins(c2, (s2 + e2) // 2)
    n_pos = len(bed2d)
    # plain arrays while the blocks are scanned (one pandas assignment per block and template cost more than the
    # native calls); the tables are assembled once below
    score_out = [np.full(n_pos, np.nan) for _ in kernels]
    pval_out = [np.full(n_pos, np.nan) for _ in kernels]
    # windows: in position order where a path needs them there (block-by-block fall-back, sharded runs); the batched path hands
    # its windows on in block order with the positions they belong to (three 2.4 MB scatters and as many NaN fills less)
    win_out = [None for _ in kernels]
    win_src = [None for _ in kernels]

    def win_array(k):
        if win_out[k] is None:
            win_out[k] = np.full((n_pos, km, kn), np.nan)
        return win_out[k]
    located = (g1 >= 0) & (g2 >= 0)
    # positions of every sub-matrix, in input order: one stable sort by (chrom1, chrom2); only the sub-matrices that hold a
    # position are visited (no pattern on a sub-matrix: it is not scanned, :240), in the order of sub_matrices()
    pair_key = np.where(located, c1 * (dcool.n_chrom + 1) + c2, -1)
    order = np.argsort(pair_key, kind="stable")
    keys_sorted = pair_key[order]
    present, first = np.unique(keys_sorted, return_index=True)
    ends = np.append(first[1:], keys_sorted.size)
    todo = []
    for key, lo, hi in zip(present.tolist(), first.tolist(), ends.tolist()):
        ca, cb = divmod(key, dcool.n_chrom + 1)
        if key < 0 or ca > cb or (ca != cb and not inter):
            continue                                        # (sub_matrices(): upper blocks, inter ones only with --inter)
        sel = order[lo:hi]
        todo.append((ca, cb, sel, np.column_stack([g1[sel] - dcool.offsets[ca], g2[sel] - dcool.offsets[cb]]).astype(int)))
    # Sharded run (parallel.quantify_genome): this rank scores the positions of its own sub-matrices, the scores of all ranks
    # are exchanged once below -- the reference's pool over sub-matrices (cli/chromosight.py:396-410)
    mine = todo if shard is None else shard.select(todo, dcool, max_dist)
    # every sub-matrix that holds a position is staged once: the intra blocks with ONE native call (cs_stage_blocks)
    staged = {}
    intra = sorted({ca for ca, cb, _, _ in mine if ca == cb})
    if intra:
        for ca, blk in zip(intra, dcool.stage_blocks(intra, max_dist, largest, smooth=smooth)):
            staged[(ca, ca)] = blk
    inter_pairs = [(ca, cb) for ca, cb, _, _ in mine if ca != cb]
    if len(inter_pairs) > 1:
        for pair, blk in zip(inter_pairs, dcool.stage_inter_many(inter_pairs)):
            staged[pair] = blk
    else:
        for ca, cb in inter_pairs:
            staged[(ca, cb)] = dcool.stage_inter(ca, cb, resident=True)
    blocks = [staged[(ca, cb)] for ca, cb, _, _ in mine]
    # one native call per template for the positions of all sub-matrices (cs_quantify_blocks), the coordinate lists prepared
    # once for all templates ...
    res_all = None
    if mine and tsvd is None and not os.environ.get("CHROMOSIGHT_HIP_NO_QUANTIFY_BATCH"):
        for kernel in kernels:
            _check_template(kernel)
        res_all = cid.quantify_many_on_device(dcool.dev, blocks, [engine.KernelSpec(k) for k in kernels], cfg, [c for _, _, _, c in mine])
    if res_all is not None:
        where = np.concatenate([sel for _, _, sel, _ in mine])
        for kernel_id, (table, wins) in enumerate(res_all):
            score_out[kernel_id][where] = table[:, 2]
            pval_out[kernel_id][where] = table[:, 3]
            if shard is None:
                win_src[kernel_id] = (wins, where)
            else:
                win_array(kernel_id)[where] = wins
    for kernel_id, kernel in enumerate(kernels):
        if res_all is not None:
            continue
        # ... or, where the batch does not apply (truncated SVD, non-square templates), one per sub-matrix and template
        for (ca, cb, sel, coords), block in zip(mine, blocks):
            rec, wins = detect_block(dcool, block, cfg, kernel, tsvd=tsvd, coords=coords.copy(), raw=True)
            if rec is None:
                continue
            score_out[kernel_id][sel] = rec[:, 2]
            pval_out[kernel_id][sel] = rec[:, 3]
            win_array(kernel_id)[sel] = wins
    if shard is not None:
        win_out = [win_array(k) for k in range(len(kernels))]
        score_out, pval_out, win_out = shard.merge(score_out, pval_out, win_out, [sel for _, _, sel, _ in mine])
    # best score of every coordinate among the templates, as the reference selects it (:432-441): the tables of the templates
    # one below the other, sort_values("score"), last row of every (chrom1, start1, chrom2, start2) group.  In numpy -- the
    # same argsort pandas runs (quicksort on the finite scores, NaN last), so ties fall as they do there -- because the
    # pandas version (three assigns, a concat of 3 n rows, a sort and a groupby) cost more than scoring the positions.
    n_k = len(kernels)
    score_all, pval_all = np.concatenate(score_out), np.concatenate(pval_out)
    idx = np.arange(score_all.size)
    nan = np.isnan(score_all)
    in_sorted = np.concatenate([idx[~nan][np.argsort(score_all[~nan], kind="quicksort")], idx[nan]])
    # (group key: the four columns factorised into one integer -- a row-wise np.unique costs ten times as much)
    gid = np.zeros(n_pos, dtype=np.int64)
    for col in (bed2d.chrom1, s1, bed2d.chrom2, s2):
        codes, uniques = pd.factorize(col)
        gid = gid * (len(uniques) + 1) + (codes + 1)
    groups_sorted = np.tile(gid, n_k)[in_sorted]
    _, first_from_end = np.unique(groups_sorted[::-1], return_index=True)
    pick = in_sorted[np.sort(groups_sorted.size - 1 - first_from_end)]          # groupby(sort=False).tail(1): in sorted-frame order
    src = pick % n_pos
    score, pvalue = score_all[pick], pval_all[pick]
    windows = np.empty((pick.size, km, kn))
    for k in range(n_k):
        m = pick // n_pos == k
        if win_src[k] is not None:
            wins, where = win_src[k]
            inv = np.full(n_pos, -1, dtype=np.int64)
            inv[where] = np.arange(where.size)
            at = inv[src[m]]
            if (at >= 0).all():
                windows[m] = wins[at]
            else:                                       # (positions no sub-matrix holds: NaN windows, as the position-order arrays had)
                w = np.full((at.size, km, kn), np.nan)
                w[at >= 0] = wins[at[at >= 0]]
                windows[m] = w
        else:
            windows[m] = win_array(k)[src[m]]
    # the bin columns come from the interval STARTS (coords_to_bins of start1 / start2, :446-455), the scores from
    # the interval midpoints: they differ for intervals wider than one bin
    out1, out2 = bins(c1[src], s1[src]), bins(c2[src], s2[src])
    if (out1 < 0).any() or (out2 < 0).any():            # no such bin: NaN, as the reference's merge leaves it
        out1, out2 = np.where(out1 < 0, np.nan, out1), np.where(out2 < 0, np.nan, out2)
    qvalue = np.asarray(fdr_correction(pvalue), dtype=np.float64)
    bad = np.isnan(score)
    pvalue = np.where(bad, np.nan, pvalue)
    qvalue = np.where(bad, np.nan, qvalue)
    final = np.lexsort((out2, out1))                    # sort_values(["bin1", "bin2"]): stable, NaN last
    cols = {c: bed2d[c].to_numpy()[src][final] for c in ("chrom1", "start1", "end1", "chrom2", "start2", "end2")}
    cols.update(bin1=out1[final], bin2=out2[final], score=score[final], pvalue=pvalue[final], qvalue=qvalue[final])
    return pd.DataFrame(cols), windows                   # (the windows keep the selection order, as before the final sort of the table: cli/chromosight.py:441-470)
