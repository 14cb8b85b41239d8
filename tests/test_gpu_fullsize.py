"""Full-size parity of the benched launches (C2, C3, C4') against the C
oracle, through the C ABI exactly as bench.py issues them (same generators, same resident
float32 buffers, same launch shapes), at the north-star tolerance: float32 coefficient within
1e-5 of the float64 oracle on every well-defined pixel.

A pixel is *ill-defined* when one factor of its denominator is an (almost) exactly degenerate
quantity -- the variance of the window, or the variance of the template over the present pixels,
below COND_EPS of its scale (oracle.c: `cond`): there the float64 reference value itself is
cancellation noise (the reference's dense and sparse paths disagree by 5e-8 on such windows,
tests/test_oracle_golden.py).  Such pixels are counted, reported and bounded, never silently
dropped."""
import numpy as np
import pytest

import chromosight_amd
from chromosight_amd import engine
from chromosight_amd._lib import (LAYOUT_BAND, LAYOUT_DENSE, MASK_BINS, MASK_NONE, CsMatrix, get_device,
                                  np_dtype_code)
from oracle import c_oracle
from tools.synthetic_genome import band_workload

pytestmark = pytest.mark.gpu

from parity_util import assert_parity


def loops():
    return np.asarray(chromosight_amd.kernels.loops["kernels"][0], dtype=np.float64)


def check(got, want, cond, what):
    return assert_parity(got, want, cond, "f32", what, max_ill_frac=1e-4)


@pytest.mark.parametrize("pitch", ["engine", "tight"])
def test_c2_dense_4096_full_map(pitch):
    """C2 as benched: dense 4096^2 float32 gamma(4, 0.25) seed 0, loops 17x17, full=False, no mask -- device maps with the
    engine's row pitch (4160 elements: what bench.py times) and with rows packed tight (4096)."""
    dev = get_device()
    n = 4096
    sig = np.random.default_rng(0).gamma(4.0, 0.25, size=(n, n)).astype(np.float32)
    if pitch == "engine":
        d_sig, ld_in = engine.to_device_map(dev, sig)
        ld_out = engine.map_pitch(n, 4)
        assert ld_in == ld_out == 4160
    else:
        d_sig, ld_in, ld_out = dev.to_device(sig), n, n
    d_out = dev.empty((n, ld_out), np.float32)
    engine.run_normxcorr2(dev, CsMatrix(d_sig.ptr, np_dtype_code(np.float32), LAYOUT_DENSE, ld_in, 0, 0), (n, n),
                          engine.KernelSpec(loops()),
                          CsMatrix(d_out.ptr, np_dtype_code(np.float32), LAYOUT_DENSE, ld_out, 0, 0),
                          full=False, sym_upper=False, max_dist=None, mask_mode=MASK_NONE, precision="f32")
    got = d_out.download()[:, :n]
    assert last_kernel() == KERNEL_MFMA_DENSE            # the kernel bench.py times on this workload
    want, cond = c_oracle.normxcorr2_rows(sig.astype(np.float64), loops(), 0, n, full=False)
    check(got, want, cond, "C2 4096x4096")
    assert np.all(got[:8] == 0) and np.all(got[:, -8:] == 0)      # valid-mode margins


KERNEL_STREAM, KERNEL_MFMA_DENSE, KERNEL_MFMA_REG = 2, 4, 5


def last_kernel():
    dev = get_device()
    return dev.lib.cs_last_kernel(dev.ctx)


def run_band(name, n=None, expect=KERNEL_MFMA_REG, padded=False):
    """padded: the band handed over as CS_LAYOUT_BAND_PADDED (band_workload's rows are zero behind their stored diagonals and
    beyond the matrix) -- the tile kernel then fetches the rim tiles like the inner ones; same map either way."""
    from chromosight_amd._lib import LAYOUT_BAND_PADDED
    dev = get_device()
    band, band_w, miss, n, max_dist = band_workload(name, n=n)
    out_w = max_dist + 1
    ld_out = (out_w + 63) // 64 * 64
    d_sig, d_out = dev.to_device(band), dev.zeros((n, ld_out), np.float32)
    d_miss = dev.to_device(miss)
    assert band.shape[1] >= band_w + 4
    engine.run_normxcorr2(dev, CsMatrix(d_sig.ptr, np_dtype_code(np.float32), LAYOUT_BAND_PADDED if padded else LAYOUT_BAND, band.shape[1], 0, band_w),
                          (n, n), engine.KernelSpec(loops()),
                          CsMatrix(d_out.ptr, np_dtype_code(np.float32), LAYOUT_BAND, ld_out, 0, out_w),
                          full=True, sym_upper=True, max_dist=max_dist, mask_mode=MASK_BINS, miss_row=d_miss,
                          miss_col=d_miss, missing_tol=0.5, precision="f32")
    assert last_kernel() == expect
    return d_out.download()[:, :out_w], band, band_w, miss, n, max_dist


@pytest.mark.parametrize("kernel", ["tile", "tile_padded", "stream"])
def test_c3_band_50000_full_map(kernel, monkeypatch):
    """C3 as benched: N = 50 000, diagonals 0..250, 2 % missing bins, max_dist 233, full, sym_upper,
    missing_tol 0.5 -- every pixel of the 11.7 M-pixel band, on the masked matrix-core tile kernel (the default,
    what bench.py times) and on the packed-FMA streaming kernel (CHROMOSIGHT_HIP_MFMA_REG=0)."""
    if kernel == "stream":
        monkeypatch.setenv("CHROMOSIGHT_HIP_MFMA_REG", "0")
    got, band, band_w, miss, n, max_dist = run_band("c3", expect=KERNEL_STREAM if kernel == "stream" else KERNEL_MFMA_REG,
                                                    padded=kernel == "tile_padded")
    want, cond = c_oracle.normxcorr2_band(band.astype(np.float64), n, 0, band_w, loops(), 0, n, 0, max_dist + 1,
                                          max_dist=max_dist, miss_row=miss, miss_col=miss, missing_tol=0.5)
    check(got, want, cond, "C3 50000 x 234")


@pytest.mark.parametrize("kernel", ["tile", "tile_padded", "stream"])
def test_c4p_band_200000_row_windows(kernel, monkeypatch):
    """C4' as benched: N = 200 000 single block, max_dist 1000 -- seven windows of 2000 rows (both
    matrix ends, the middle, strip-height boundaries), 14 M pixels against the oracle; both kernels as above."""
    if kernel == "stream":
        monkeypatch.setenv("CHROMOSIGHT_HIP_MFMA_REG", "0")
    got, band, band_w, miss, n, max_dist = run_band("c4p", expect=KERNEL_STREAM if kernel == "stream" else KERNEL_MFMA_REG,
                                                    padded=kernel == "tile_padded")
    band64 = band.astype(np.float64)
    del band
    for r0 in (0, 1990, 49_000, 99_137, 150_000, 187_654, n - 2000):
        want, cond = c_oracle.normxcorr2_band(band64, n, 0, band_w, loops(), r0, r0 + 2000, 0, max_dist + 1,
                                              max_dist=max_dist, miss_row=miss, miss_col=miss, missing_tol=0.5)
        check(got[r0:r0 + 2000], want, cond, f"C4' rows {r0}..{r0 + 2000}")


@pytest.mark.parametrize("env", ["CHROMOSIGHT_HIP_FORCE_GENERIC", "CHROMOSIGHT_HIP_NO_SYMMETRY"])
def test_c2_alternative_kernels(env, monkeypatch):
    """The generic LDS-tiled kernel and the unfolded streaming kernel on a 1024-row slab of C2: the
    in-library cross-checks of the benched kernel, against the same oracle."""
    monkeypatch.setenv(env, "1")
    dev = get_device()
    n, rows = 4096, 1024
    sig = np.random.default_rng(0).gamma(4.0, 0.25, size=(n, n)).astype(np.float32)[:rows]
    d_sig, d_out = dev.to_device(sig), dev.empty((rows, n), np.float32)
    engine.run_normxcorr2(dev, CsMatrix(d_sig.ptr, np_dtype_code(np.float32), LAYOUT_DENSE, n, 0, 0), (rows, n),
                          engine.KernelSpec(loops()),
                          CsMatrix(d_out.ptr, np_dtype_code(np.float32), LAYOUT_DENSE, n, 0, 0),
                          full=False, sym_upper=False, max_dist=None, mask_mode=MASK_NONE, precision="f32")
    want, cond = c_oracle.normxcorr2_rows(sig.astype(np.float64), loops(), 0, rows, full=False)
    check(d_out.download(), want, cond, f"C2 slab, {env}")


def test_c3_from_csr_50000_full_map():
    """C3 as BASELINE.md defines it and as bench.py's `c3_from_csr` leg times it: the N = 50 000 pixel table (CSR) in,
    DeviceCool.stage_blocks on the device and the masked tile kernel on the staged band -- every pixel of the 11.7 M-pixel map
    against the C oracle run on the band the detrend ORACLE prepared from the same pixel table (oracle/detrend_oracle.py).
    Both stagings: the band of raw counts written by the law pass and detrended by the tile kernel as it splits a landed tile
    (CS_LAYOUT_BAND_COUNTS: what bench.py times) -- the band itself against the pixel table's counts -- and the detrended band of
    the tiler pass -- against the oracle's band; both maps against the oracle."""
    from chromosight_amd import pipeline
    from chromosight_amd._lib import LAYOUT_BAND_COUNTS, LAYOUT_BAND_PADDED
    from oracle import detrend_oracle
    from tools.synthetic_genome import make_cool
    n, max_dist = 50_000, 233
    cool, _ = make_cool(n, max_dist, 2000, seed=1, loops_per_10k=0, chrom_sizes=[n])        # bench.py Workload("c3"), rank 0
    dev = get_device()
    dcool = pipeline.DeviceCool(cool, dev)
    assert dcool.counts_ok
    out_w = max_dist + 1
    ld_out = (out_w + 63) // 64 * 64
    band, det = detrend_oracle.balanced_band(cool, 0, max_dist + 17)
    prepared, _ = detrend_oracle.prepare_band(band, det)
    maps = {}
    for counts in (True, False):
        block = dcool.stage_blocks([0], max_dist, 17, band_dtype=np.float32, counts=counts)[0]
        d_out = dev.zeros((n, ld_out), np.float32)
        engine.run_normxcorr2(dev, block.sig, (n, n), engine.KernelSpec(loops()),
                              CsMatrix(d_out.ptr, np_dtype_code(np.float32), LAYOUT_BAND, ld_out, 0, out_w), precision="f32",
                              miss_row=block.miss_row, miss_col=block.miss_col, full=True, sym_upper=True, max_dist=max_dist,
                              mask_mode=MASK_BINS, missing_tol=0.5)
        assert last_kernel() == KERNEL_MFMA_REG
        maps[counts] = d_out.download()[:, :out_w]
        sig = block.sig
        assert sig.layout == (LAYOUT_BAND_COUNTS if counts else LAYOUT_BAND_PADDED) and sig.band_lo == 0 and sig.band_w == max_dist + 18
        staged = np.empty((n, sig.ld), dtype=np.float32)
        dev._check(dev.lib.cs_memcpy_d2h(dev.ctx, staged.ctypes.data, sig.d_ptr, staged.nbytes, None))
        assert not staged[:, sig.band_w:].any()                      # zero-padded rows: the staging pass's own
        if counts:
            # the raw counts of the pixel table, slot = diagonal
            want_counts = np.zeros((n, sig.band_w), dtype=np.float32)
            b1, b2, c = np.asarray(cool["bin1_id"]), np.asarray(cool["bin2_id"]), np.asarray(cool["count"])
            keep = (b2 - b1) < sig.band_w
            want_counts[b1[keep], (b2 - b1)[keep]] = c[keep]
            assert np.array_equal(staged[:, :sig.band_w], want_counts)
        else:
            # the staged float32 band: the oracle's band rounded once
            assert np.abs(staged[:, :sig.band_w] - prepared).max() <= 1e-6 * max(1.0, np.abs(prepared).max())
        del block, d_out, staged
    miss = (~det).astype(np.uint8)
    want, cond = c_oracle.normxcorr2_band(prepared, n, 0, prepared.shape[1], loops(), 0, n, 0, out_w, max_dist=max_dist,
                                          miss_row=miss, miss_col=miss, missing_tol=0.5)
    check(maps[True], want, cond, "C3 from CSR 50000 x 234, band of counts")
    check(maps[False], want, cond, "C3 from CSR 50000 x 234, detrended band")
    # (the tile kernel detrends a landed tile in float32: a few units in the last place of the float32 band per pixel)
    assert np.abs(maps[True] - maps[False]).max() < 2e-4 and np.quantile(np.abs(maps[True] - maps[False]), 0.999) < 2e-6
