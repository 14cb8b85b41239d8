"""Bands of raw counts (CS_LAYOUT_BAND_COUNTS, include/chromosight_hip.h): the law pass of cs_stage_blocks writes a block's
counts once and the readers balance and detrend what they fetch -- the masked tile kernel a landed tile, the float64
kernels a pixel (reference: contacts_map.py:527-548 create_mat -> detrend, preprocessing.py:256-310).  Checked against the
pixel table itself (the staged counts), against the detrended band of the tiler pass (same staging call, counts off), against
the C oracle on the band the detrend oracle prepares, and at the discrete cap (>= max_val -> 1) where the float32 product the
tile kernel forms and the float64 product of the reference may fall on different sides."""
import numpy as np
import pytest

import chromosight_amd
from chromosight_amd import engine, pipeline
from chromosight_amd._lib import (LAYOUT_BAND, LAYOUT_BAND_COUNTS, LAYOUT_BAND_PADDED, MASK_BINS, CsMatrix, HipLibraryError, get_device,
                                  np_dtype_code)
from oracle import c_oracle, detrend_oracle
from tools.synthetic_genome import make_cool

pytestmark = pytest.mark.gpu

from parity_util import assert_parity

KERNEL_MFMA_REG = 5


def loops():
    return np.asarray(chromosight_amd.kernels.loops["kernels"][0], dtype=np.float64)


def staged_map(dev, dcool, max_dist, counts, kernel=None):
    """The coefficient map of chromosome 0 (band of max_dist + 1 diagonals) from a float32-only staging, and the staged band."""
    kernel = loops() if kernel is None else kernel
    n = int(dcool.offsets[1] - dcool.offsets[0])
    block = dcool.stage_blocks([0], max_dist, kernel.shape[0], band_dtype=np.float32, counts=counts)[0]
    out_w = max_dist + 1
    ld_out = (out_w + 63) // 64 * 64
    d_out = dev.zeros((n, ld_out), np.float32)
    engine.run_normxcorr2(dev, block.sig, (n, n), engine.KernelSpec(kernel),
                          CsMatrix(d_out.ptr, np_dtype_code(np.float32), LAYOUT_BAND, ld_out, 0, out_w), precision="f32",
                          miss_row=block.miss_row, miss_col=block.miss_col, full=True, sym_upper=True, max_dist=max_dist,
                          mask_mode=MASK_BINS, missing_tol=0.5)
    sig = block.sig
    staged = np.empty((n, sig.ld), dtype=np.float32)
    dev._check(dev.lib.cs_memcpy_d2h(dev.ctx, staged.ctypes.data, sig.d_ptr, staged.nbytes, None))
    return d_out.download()[:, :out_w], staged, sig, dev.lib.cs_last_kernel(dev.ctx)


@pytest.mark.parametrize("seed,n,max_dist", [(3, 1500, 100), (4, 2600, 333), (5, 900, 61)])
def test_counts_band_against_pixel_table_detrended_band_and_oracle(seed, n, max_dist):
    cool, _ = make_cool(n, max_dist, 2000, seed=seed, loops_per_10k=0, chrom_sizes=[n])
    dev = get_device()
    dcool = pipeline.DeviceCool(cool, dev)
    assert dcool.counts_ok
    got_c, staged_c, sig_c, kern_c = staged_map(dev, dcool, max_dist, True)
    got_d, staged_d, sig_d, kern_d = staged_map(dev, dcool, max_dist, False)
    assert kern_c == kern_d == KERNEL_MFMA_REG
    assert sig_c.layout == LAYOUT_BAND_COUNTS and sig_d.layout in (LAYOUT_BAND_PADDED, LAYOUT_BAND)
    w = sig_c.band_w
    assert w == min(max_dist + 17, n - 1) + 1 and sig_c.ld >= w + 4 and not staged_c[:, w:].any()
    want_counts = np.zeros((n, w), dtype=np.float32)
    b1, b2, c = np.asarray(cool["bin1_id"]), np.asarray(cool["bin2_id"]), np.asarray(cool["count"])
    keep = (b2 - b1) < w
    want_counts[b1[keep], (b2 - b1)[keep]] = c[keep]
    assert np.array_equal(staged_c[:, :w], want_counts)
    band, det = detrend_oracle.balanced_band(cool, 0, max_dist + 17)
    prepared, _ = detrend_oracle.prepare_band(band, det)
    miss = (~det).astype(np.uint8)
    want, cond = c_oracle.normxcorr2_band(prepared, n, 0, prepared.shape[1], loops(), 0, n, 0, max_dist + 1, max_dist=max_dist,
                                          miss_row=miss, miss_col=miss, missing_tol=0.5)
    assert_parity(got_c, want, cond, "f32", f"band of counts, seed {seed}", max_ill_frac=1e-3)
    assert_parity(got_d, want, cond, "f32", f"detrended band, seed {seed}", max_ill_frac=1e-3)
    # float32 arithmetic on a landed tile against one rounding of the float64 expression: a few units in the last place per pixel
    assert np.quantile(np.abs(got_c - got_d), 0.999) < 5e-6


def planted_cool(n, w, plant):
    """Poisson counts on the diagonals 0 .. w - 1 with unit weights, and diagonal `d` replaced by `plant` (list of counts laid
    on its first pixels, nothing else stored on it)."""
    rng = np.random.default_rng(11)
    rows = np.repeat(np.arange(n), w)
    diag = np.tile(np.arange(w), n)
    cnt = rng.poisson(40.0 / (diag + 1.0)).astype(np.int32)
    d, values = plant
    cnt[diag == d] = 0
    first = np.flatnonzero(diag == d)[:len(values)]
    cnt[first] = values
    ok = (rows + diag < n) & (cnt > 0)
    return {"binsize": 1000, "chrom_offset": np.array([0, n]), "chrom_names": np.array(["chr1"]), "bin1_id": rows[ok].astype(np.int64),
            "bin2_id": (rows + diag)[ok].astype(np.int64), "count": cnt[ok], "weight": np.ones(n), "bin_start": None, "bin_end": None}


def test_counts_band_at_the_cap():
    """A diagonal whose law is exactly 2 with one pixel of 20 (20 / 2 = 10 = max_val: capped to 1 -- the reference's
    `>= max_val`), one of 19 (9.5: kept) and one of 21 (10.5: capped): the tile kernel's float32 product lands ON max_val for
    the first, which is where it hands the tile to the float64 function of the staging pass."""
    n, max_dist, d = 700, 40, 33
    values = [20, 19, 21] + [1] * 54                  # 114 over 57 positive pixels: the diagonal's law is exactly 2
    assert sum(values) == 2 * len(values)
    cool = planted_cool(n, max_dist + 18, (d, values))
    dev = get_device()
    dcool = pipeline.DeviceCool(cool, dev)
    got_c, staged_c, sig_c, _ = staged_map(dev, dcool, max_dist, True)
    got_d, staged_d, _, _ = staged_map(dev, dcool, max_dist, False)
    assert staged_c[0, d] == 20 and staged_c[1, d] == 19 and staged_c[2, d] == 21
    assert staged_d[0, d] == 1.0 and staged_d[1, d] == np.float32(9.5) and staged_d[2, d] == 1.0       # the reference's cap
    band, det = detrend_oracle.balanced_band(cool, 0, max_dist + 17)
    prepared, _ = detrend_oracle.prepare_band(band, det)
    assert prepared[0, d] == 1.0 and prepared[1, d] == 9.5 and prepared[2, d] == 1.0
    miss = (~det).astype(np.uint8)
    want, cond = c_oracle.normxcorr2_band(prepared, n, 0, prepared.shape[1], loops(), 0, n, 0, max_dist + 1, max_dist=max_dist,
                                          miss_row=miss, miss_col=miss, missing_tol=0.5)
    assert_parity(got_c, want, cond, "f32", "band of counts at the cap", max_ill_frac=1e-3)
    # the windows that hold the three pixels: a pixel taken for 10 instead of 1 moves a coefficient by 1e-2 and more
    rows = slice(0, 12)
    assert np.abs(got_c[rows] - got_d[rows]).max() < 5e-5


def test_counts_band_is_refused_by_other_kernels():
    """Only the masked float32 tile kernel detrends what it reads: a template it does not serve, plain cross-correlation and
    float64 arithmetic fail loudly instead of correlating raw counts."""
    n, max_dist = 1500, 100
    cool, _ = make_cool(n, max_dist, 2000, seed=3, loops_per_10k=0, chrom_sizes=[n])
    dev = get_device()
    dcool = pipeline.DeviceCool(cool, dev)
    block = dcool.stage_blocks([0], max_dist, 21, band_dtype=np.float32, counts=True)[0]
    assert block.sig.layout == LAYOUT_BAND_COUNTS
    out_w = max_dist + 1
    ld_out = (out_w + 63) // 64 * 64
    d_out = dev.zeros((n, ld_out), np.float32)
    out = CsMatrix(d_out.ptr, np_dtype_code(np.float32), LAYOUT_BAND, ld_out, 0, out_w)
    big = np.random.default_rng(0).normal(size=(21, 21))
    with pytest.raises((NotImplementedError, ValueError, HipLibraryError)):
        engine.run_normxcorr2(dev, block.sig, (n, n), engine.KernelSpec(big), out, precision="f32", miss_row=block.miss_row,
                              miss_col=block.miss_col, full=True, sym_upper=True, max_dist=max_dist, mask_mode=MASK_BINS, missing_tol=0.5)
    with pytest.raises((NotImplementedError, ValueError, HipLibraryError)):
        engine.run_normxcorr2(dev, block.sig, (n, n), engine.KernelSpec(loops()), out, precision="f64", miss_row=block.miss_row,
                              miss_col=block.miss_col, full=True, sym_upper=True, max_dist=max_dist, mask_mode=MASK_BINS, missing_tol=0.5)
