"""The two-pass matrix-core kernel (chromosight_amd/csrc/cs_corr_wide.hip) -- templates with a side of 18 .. 33, what
`--win-size` makes (reference cli/chromosight.py:365-370, 689-695 -> preprocessing.py:731-807) and what API users pass --
against the C oracle at the 1e-5 bar, in every container the runtime-size kernel served before: dense and banded maps,
float32 and float64, per-bin and explicit masks (general plane form AND the factorised form of inner tiles), n_obs,
row windows, plain cross-correlations; and against the runtime-size kernel (CHROMOSIGHT_HIP_NO_WIDE=1)."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

import chromosight_amd
from chromosight_amd import engine
from chromosight_amd._lib import LAYOUT_BAND, LAYOUT_DENSE, MASK_BINS, MASK_NONE, CsMatrix, get_device, np_dtype_code
from chromosight_amd.utils import detection as cud
from chromosight_amd.utils import preprocessing as cup
from oracle import c_oracle
from parity_util import assert_parity

ILL_MASKED = 0.1
KERNEL_GENERIC, KERNEL_SEPARABLE, KERNEL_MFMA_WIDE = 1, 6, 7

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def f32_precision():
    old = chromosight_amd.get_precision()
    chromosight_amd.set_precision("f32")
    yield
    chromosight_amd.set_precision(old)


def last_kernel():
    dev = get_device()
    return dev.lib.cs_last_kernel(dev.ctx)


def template(shape, seed=0):
    """A full-rank template with structure (a blob on a gradient plus noise): what a zoomed loops template looks like."""
    rng = np.random.default_rng(1000 * shape[0] + shape[1] + seed)
    i, j = np.indices(shape)
    ci, cj = (shape[0] - 1) / 2, (shape[1] - 1) / 2
    blob = np.exp(-((i - ci) ** 2 + (j - cj) ** 2) / (0.08 * shape[0] * shape[1] + 1))
    return 0.4 + blob + 0.02 * (i - j) + 0.15 * rng.normal(size=shape)


def _signal(rng, shape, kind):
    if kind == "gamma":
        return rng.gamma(2.0, 1.0, size=shape)
    if kind == "hic":
        return np.minimum(rng.gamma(20, 0.05, size=shape), 10.0)
    if kind == "tiny":
        return rng.gamma(2.0, 1.0, size=shape) * 3e-7
    if kind == "huge":
        return rng.gamma(2.0, 1.0, size=shape) * 7e8
    if kind == "signed":
        return rng.normal(size=shape)
    raise ValueError(kind)


@pytest.mark.parametrize("shape,kshape,kind", [
    ((300, 300), (19, 19), "gamma"), ((513, 777), (21, 21), "hic"), ((130, 70), (33, 33), "signed"),
    ((64, 70), (23, 23), "gamma"), ((200, 333), (25, 19), "gamma"), ((150, 150), (5, 31), "hic"),
    ((97, 201), (33, 7), "gamma"), ((256, 256), (27, 27), "tiny"), ((256, 256), (29, 29), "huge"),
    ((190, 190), (19, 33), "gamma"), ((222, 111), (31, 21), "hic"), ((400, 400), (31, 31), "gamma"),
])
@pytest.mark.parametrize("full", [True, False, "slow"])
def test_dense_maps_match_oracle(shape, kshape, kind, full, monkeypatch):
    """Unmasked dense float32 maps: partial tiles, frames narrower than a tile, valid-mode margins, rectangular
    templates (one and two Toeplitz passes), data far from unit scale."""
    if full == "slow":                       # the pixel-by-pixel staging of every tile (CHROMOSIGHT_HIP_WIDE_SLOW=1)
        monkeypatch.setenv("CHROMOSIGHT_HIP_WIDE_SLOW", "1")
        full = True
    rng = np.random.default_rng(shape[0] * 1000 + shape[1] + kshape[0])
    sig = _signal(rng, shape, kind).astype(np.float32)
    kern = template(kshape)
    got, _ = cud.normxcorr2(sig, kern, full=full)
    assert last_kernel() == KERNEL_MFMA_WIDE
    want, cond = c_oracle.normxcorr2_rows(sig.astype(np.float64), kern, 0, shape[0], full=full)
    assert_parity(got, want, cond, "f32", f"wide dense {shape} {kshape} {kind} full={full}")


def test_equals_runtime_size_kernel(monkeypatch):
    """The two float32 kernels on one map: same coefficients to a few 1e-6, and they are different kernels."""
    rng = np.random.default_rng(5)
    sig = rng.gamma(2.0, 1.0, size=(700, 900)).astype(np.float32)
    kern = template((21, 21))
    a, _ = cud.normxcorr2(sig, kern, full=True)
    assert last_kernel() == KERNEL_MFMA_WIDE
    monkeypatch.setenv("CHROMOSIGHT_HIP_NO_WIDE", "1")
    b, _ = cud.normxcorr2(sig, kern, full=True)
    assert last_kernel() == KERNEL_GENERIC
    assert np.abs(a - b).max() < 5e-6
    assert not np.array_equal(a, b)


def test_sym_upper_dense_and_f64_container():
    rng = np.random.default_rng(9)
    sig = np.triu(rng.gamma(2.0, 1.0, size=(260, 260)))          # float64 container
    kern = template((21, 21))
    for full in (True, False):
        got, _ = cud.normxcorr2(sig, kern, sym_upper=True, full=full)
        assert last_kernel() == KERNEL_MFMA_WIDE
        want, cond = c_oracle.normxcorr2_rows(sig, kern, 0, 260, sym_upper=True, full=full)
        assert_parity(got, want, cond, "f32", f"wide dense sym_upper full={full}")
        assert np.all(np.tril(got, -1) == 0)


def _masked_band(n, md, keep, seed, frac=0.04):
    rng = np.random.default_rng(seed)
    ii, jj = np.indices((n, n))
    sig = np.triu(np.minimum(rng.gamma(20, 0.05, size=(n, n)), 10.0))
    sig[jj - ii > md + keep] = 0
    valid = np.flatnonzero(rng.random(n) > frac)
    miss = np.ones(n, bool)
    miss[valid] = False
    sig[miss, :] = 0
    sig[:, miss] = 0
    return sig.astype(np.float32), valid, miss, (jj - ii >= 0) & (jj - ii <= md)


@pytest.mark.parametrize("n,md,ksize,tol", [(900, 120, 21, 0.75), (1500, 400, 21, 0.5), (1300, 500, 33, 0.75),
                                             (700, 60, 19, 0.25), (400, 399, 25, 0.75), (1100, 300, 27, 0.5)])
@pytest.mark.parametrize("plane", [False, True, "slow", "one", "two"])
def test_banded_maps_with_bin_masks(n, md, ksize, tol, plane, monkeypatch):
    """The detect configuration: CSR in, band in / band out on the device, per-bin masks, full, sym_upper, coefficients
    and n_obs (through the p-values).  plane=False: inner tiles take the factorised form (1-D tables minus the cross
    plane), the rim the general plane; plane=True (CHROMOSIGHT_HIP_WIDE_PLANE=1): the general plane everywhere -- both
    against the oracle and against each other; "slow" (CHROMOSIGHT_HIP_WIDE_SLOW=1): every tile staged pixel by pixel
    with the general predicate (what the tiles on the frame of the matrix and explicit masks take)."""
    if plane == "one":                       # inner tiles and the rest in ONE launch
        monkeypatch.setenv("CHROMOSIGHT_HIP_WIDE_ONE_LAUNCH", "1")
    elif plane == "two":                     # ... in two, the inner tiles with the small LDS image (the default on wide bands only)
        monkeypatch.setenv("CHROMOSIGHT_HIP_WIDE_TWO_LAUNCHES", "1")
    elif plane == "slow":
        monkeypatch.setenv("CHROMOSIGHT_HIP_WIDE_SLOW", "1")
    elif plane:
        monkeypatch.setenv("CHROMOSIGHT_HIP_WIDE_PLANE", "1")
    sig, valid, miss, band = _masked_band(n, md, ksize, n + md)
    kern = template((ksize, ksize))
    mask = cup.make_missing_mask((n, n), valid, valid, max_dist=md, sym_upper=True)
    c, p = cud.normxcorr2(sp.csr_matrix(sig), kern, max_dist=md, sym_upper=True, full=True, missing_mask=mask,
                          missing_tol=tol, pval=True)
    assert last_kernel() == KERNEL_MFMA_WIDE
    want, cond = c_oracle.normxcorr2_rows(sig.astype(np.float64), kern, 0, n, max_dist=md, sym_upper=True, full=True,
                                          miss_row=miss, miss_col=miss, missing_tol=tol)
    assert_parity(c.toarray()[band], want[band], cond[band], "f32", f"wide band n={n} md={md} k={ksize} plane={plane}",
                  max_ill_frac=ILL_MASKED)
    # the runtime-size kernel on the same call: coefficients and p-values (n_obs) agree
    monkeypatch.setenv("CHROMOSIGHT_HIP_NO_WIDE", "1")
    c2, p2 = cud.normxcorr2(sp.csr_matrix(sig), kern, max_dist=md, sym_upper=True, full=True, missing_mask=mask,
                            missing_tol=tol, pval=True)
    assert last_kernel() == KERNEL_GENERIC
    well = band & (cond >= 1e-3)
    assert np.abs(c.toarray() - c2.toarray())[well].max() < 5e-6
    assert np.abs(p.toarray() - p2.toarray())[well].max() < 5e-3


def test_inter_block_and_explicit_mask():
    """An inter-chromosomal block (dense, masks on both axes, no diagonal limits: its middle tiles are inner tiles) and an
    explicit mask in valid mode."""
    rng = np.random.default_rng(21)
    shape = (500, 620)
    kern = template((21, 21))
    inter = rng.gamma(2.0, 1.0, size=shape)
    vr, vc = np.flatnonzero(rng.random(shape[0]) > 0.05), np.flatnonzero(rng.random(shape[1]) > 0.05)
    mr, mc = np.ones(shape[0], bool), np.ones(shape[1], bool)
    mr[vr] = False
    mc[vc] = False
    inter[mr, :] = 0
    inter[:, mc] = 0
    mask = cup.make_missing_mask(shape, vr, vc, max_dist=None, sym_upper=False)
    for full in (True, False):
        c, _ = cud.normxcorr2(sp.csr_matrix(inter), kern, max_dist=None, sym_upper=False, full=full, missing_mask=mask,
                              missing_tol=0.75)
        assert last_kernel() == KERNEL_MFMA_WIDE
        want, cond = c_oracle.normxcorr2_rows(inter, kern, 0, shape[0], full=full, miss_row=mr, miss_col=mc)
        assert_parity(c.toarray(), want, cond, "f32", f"wide inter block full={full}", max_ill_frac=ILL_MASKED)
    # an explicit mask that is NOT a union of rows and columns: dense signal, dense mask
    sig = rng.gamma(2.0, 1.0, size=(300, 340))
    m = rng.random(sig.shape) < 0.03
    sig[m] = 0
    m = sp.csr_matrix(m)
    got, _ = cud.normxcorr2(sig, kern, full=False, missing_mask=m, missing_tol=0.75)
    assert last_kernel() == KERNEL_MFMA_WIDE
    import os
    os.environ["CHROMOSIGHT_HIP_NO_WIDE"] = "1"
    try:
        ref, _ = cud.normxcorr2(sig, kern, full=False, missing_mask=m, missing_tol=0.75)
        assert last_kernel() == KERNEL_GENERIC
    finally:
        del os.environ["CHROMOSIGHT_HIP_NO_WIDE"]
    assert np.abs(got - ref).max() < 5e-6


def test_narrow_maps_and_f64_bands():
    """Maps narrower than a 16-byte piece (general staging), a float64 band container and a band whose rows start at
    every alignment (odd row pitch)."""
    rng = np.random.default_rng(77)
    kern = template((19, 19))
    for shape in ((60, 3), (3, 60), (40, 5)):
        sig = rng.gamma(2.0, 1.0, size=shape).astype(np.float32)
        got, _ = cud.normxcorr2(sig, kern, full=True)
        assert last_kernel() == KERNEL_MFMA_WIDE
        want, cond = c_oracle.normxcorr2_rows(sig.astype(np.float64), kern, 0, shape[0], full=True)
        assert_parity(got, want, cond, "f32", f"wide narrow {shape}", max_ill_frac=1.0)
    dev = get_device()
    n, md, bw = 700, 150, 150 + 19 + 1
    band = np.zeros((n, bw + 3))                                    # float64, odd pitch
    d = np.arange(bw)
    band[:, :bw] = np.minimum(rng.gamma(20, 0.05, size=(n, bw)), 10.0)
    for i in range(n):
        band[i, :bw][i + d >= n] = 0
    miss = (rng.random(n) < 0.03).astype(np.uint8)
    band[miss.astype(bool)] = 0
    for i in range(n):
        cols = i + d
        band[i, :bw][miss[np.minimum(cols, n - 1)].astype(bool)] = 0
    out_w = md + 1
    d_sig, d_out, d_miss = dev.to_device(band), dev.zeros((n, out_w), np.float32), dev.to_device(miss)
    engine.run_normxcorr2(dev, CsMatrix(d_sig.ptr, np_dtype_code(np.float64), LAYOUT_BAND, band.shape[1], 0, bw), (n, n),
                          engine.KernelSpec(kern), CsMatrix(d_out.ptr, np_dtype_code(np.float32), LAYOUT_BAND, out_w, 0, out_w),
                          full=True, sym_upper=True, max_dist=md, mask_mode=MASK_BINS, miss_row=d_miss, miss_col=d_miss,
                          missing_tol=0.5, precision="f32")
    assert last_kernel() == KERNEL_MFMA_WIDE
    want, cond = c_oracle.normxcorr2_band(band, n, 0, bw, kern, 0, n, 0, out_w, max_dist=md, miss_row=miss, miss_col=miss,
                                          missing_tol=0.5)
    assert_parity(d_out.download(), want, cond, "f32", "wide f64 band, odd pitch", max_ill_frac=ILL_MASKED)


def test_xcorr2_wide_template():
    """Plain cross-correlation (reference detection.py:595-804) with a 25 x 21 template."""
    rng = np.random.default_rng(33)
    sig = rng.gamma(2.0, 1.0, size=(210, 190))
    k = rng.normal(size=(25, 21))
    got = cud.xcorr2(sig, k, threshold=1e-4)
    assert last_kernel() == KERNEL_MFMA_WIDE
    from oracle import pearson_oracle as orc
    want = orc.xcorr2_oracle(sig, k, threshold=0)
    near = np.abs(np.abs(want) - 1e-4) < 1e-6
    ref = np.where(np.abs(want) < 1e-4, 0.0, want)
    assert np.abs(got - ref)[~near].max() < 3e-6 * np.abs(want).max()


def test_row_windows():
    """Row windows (slab inputs): equal to the rows of the whole map."""
    dev = get_device()
    rng = np.random.default_rng(4)
    n, cols = 500, 600
    sig_h = rng.gamma(2.0, 1.0, size=(n, cols)).astype(np.float32)
    kspec = engine.KernelSpec(template((23, 23)), None)
    code = np_dtype_code(np.float32)

    def run(sig, out, window=None):
        params = engine._corr_params((n, cols), kspec, True, False, None, MASK_NONE, None, None, None, 0.75,
                                     engine.compute_code("f32"), window)
        dev._check(dev.lib.cs_normxcorr2(dev.ctx, None, C.byref(sig), C.byref(kspec.struct), C.byref(params), C.byref(out), None))

    d_sig = dev.to_device(sig_h)
    d_full = dev.zeros(n * cols, np.float32)
    run(CsMatrix(d_sig.ptr, code, LAYOUT_DENSE, cols, 0, 0), CsMatrix(d_full.ptr, code, LAYOUT_DENSE, cols, 0, 0))
    full = d_full.download().reshape(n, cols)
    assert last_kernel() == KERNEL_MFMA_WIDE
    for a, b in [(0, 37), (37, 200), (200, 201), (201, 495), (495, 500)]:
        ra, rb = max(0, a - 11), min(n, b + 11)
        d_slab = dev.to_device(np.ascontiguousarray(sig_h[ra:rb]))
        d_out = dev.zeros((b - a) * cols, np.float32)
        run(CsMatrix(d_slab.ptr, code, LAYOUT_DENSE, cols, 0, 0, ra), CsMatrix(d_out.ptr, code, LAYOUT_DENSE, cols, 0, 0, a), (a, b))
        got = d_out.download().reshape(b - a, cols)
        assert np.abs(got - full[a:b]).max() <= 2e-6, (a, b)


@pytest.mark.parametrize("ksize", [19, 21, 33])
def test_c4p_band_200000_row_windows(ksize):
    """C4' (N = 200 000 single block, max_dist 1000, 2 % missing bins) under templates of 19, 21 and 33: seven windows of
    2000 rows (both matrix ends, the middle) against the oracle -- the full-size parity of the kernel
    profiles/r06_template_kernels.txt times."""
    from tools.synthetic_genome import band_workload
    dev = get_device()
    band, band_w, miss, n, max_dist = band_workload("c4p")
    out_w = max_dist + 1
    ld_out = (out_w + 63) // 64 * 64
    d_sig, d_out = dev.to_device(band), dev.zeros((n, ld_out), np.float32)
    d_miss = dev.to_device(miss)
    kern = template((ksize, ksize))
    engine.run_normxcorr2(dev, CsMatrix(d_sig.ptr, np_dtype_code(np.float32), LAYOUT_BAND, band.shape[1], 0, band_w),
                          (n, n), engine.KernelSpec(kern),
                          CsMatrix(d_out.ptr, np_dtype_code(np.float32), LAYOUT_BAND, ld_out, 0, out_w),
                          full=True, sym_upper=True, max_dist=max_dist, mask_mode=MASK_BINS, miss_row=d_miss,
                          miss_col=d_miss, missing_tol=0.5, precision="f32")
    assert last_kernel() == KERNEL_MFMA_WIDE
    got = d_out.download()[:, :out_w]
    band64 = band.astype(np.float64)
    del band
    for r0 in (0, 1990, 49_000, 99_137, 150_000, 187_654, n - 2000):
        rows = 2000
        want, cond = c_oracle.normxcorr2_band(band64, n, 0, band_w, kern, r0, r0 + rows, 0, max_dist + 1,
                                              max_dist=max_dist, miss_row=miss, miss_col=miss, missing_tol=0.5)
        assert_parity(got[r0:r0 + rows], want, cond, "f32", f"C4' k={ksize} rows {r0}..{r0 + rows}", max_ill_frac=1e-4)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("CS_SWEEP_FROM", "0")), int(__import__("os").environ.get("CS_SWEEP_TO", "32"))))
def test_random_wide_configuration(seed, monkeypatch):
    """Seeded random sweep (the shape of tests/test_gpu_random_sweep.py) over what only this kernel serves: template sides of 19 .. 33,
    square and rectangular, piecewise-constant and mirrored templates; dense / banded / inter maps; full / valid mode; max_dist below,
    around and above the 167 + diagonals an inner tile needs; missing-bin clusters at the matrix ends; every third seed with the two
    launches forced, every fifth with the plane everywhere.  CS_SWEEP_FROM / CS_SWEEP_TO widen it."""
    rng = np.random.default_rng(5000 + seed)
    if seed % 3 == 0:
        monkeypatch.setenv("CHROMOSIGHT_HIP_WIDE_TWO_LAUNCHES", "1")
    elif seed % 5 == 0:
        monkeypatch.setenv("CHROMOSIGHT_HIP_WIDE_PLANE", "1")
    ksz = [19, 21, 23, 25, 27, 29, 31, 33][seed % 8]
    km, kn = (ksz, ksz) if seed % 4 else (ksz, min(33, ksz + 2 * int(rng.integers(1, 4))))
    kern = rng.random((km, kn)) + 0.5
    if rng.random() < 0.3:
        kern = np.where(rng.random((km, kn)) < 0.5, 0.5, 1.5)
        kern[0, 0], kern[-1, -1] = 0.5, 1.5
    if seed % 7 in (1, 2, 5):
        kern = (kern + kern[::-1, :]) / 2
    mode = seed % 4
    if mode == 0:
        shape = (int(rng.integers(70, 500)), int(rng.integers(70, 500)))
        sig = (rng.gamma(3, 0.4, size=shape) * (rng.random(shape) > 0.3)).astype(np.float32)
        full = bool(seed & 4)
        got, _ = cud.normxcorr2(sig, kern, full=full)
        assert last_kernel() == KERNEL_MFMA_WIDE
        want, cond = c_oracle.normxcorr2_rows(sig.astype(np.float64), kern, 0, shape[0], full=full)
        assert_parity(got, want, cond, "f32", f"wide sweep {seed} dense {shape} {km}x{kn}", max_ill_frac=0.01)
    elif mode in (1, 2):
        n = int(rng.integers(120, 1400))
        max_dist = int([rng.integers(2, 40), rng.integers(150, 260), rng.integers(min(260, n), n + 50)][seed % 3])
        keep = min(max_dist, n) + max(km, kn)
        ii, jj = np.indices((n, n))
        a = rng.gamma(3, 0.4, size=(n, n)) * (rng.random((n, n)) > 0.25)
        a[(jj - ii < 0) | (jj - ii > keep)] = 0
        miss = rng.random(n) < 0.05
        miss[:3] = True
        miss[-2:] = True
        a[miss, :] = 0
        a[:, miss] = 0
        a = a.astype(np.float32)
        tol = float(rng.choice([0.25, 0.5, 0.75]))
        valid = np.flatnonzero(~miss)
        mask = cup.make_missing_mask((n, n), valid, valid, max_dist=max_dist, sym_upper=True)
        got, _ = cud.normxcorr2(sp.csr_matrix(a), kern, max_dist=max_dist, sym_upper=True, full=True, missing_mask=mask, missing_tol=tol)
        assert last_kernel() == KERNEL_MFMA_WIDE
        want, cond = c_oracle.normxcorr2_rows(a.astype(np.float64), kern, 0, n, max_dist=max_dist, sym_upper=True, full=True,
                                              miss_row=miss, miss_col=miss, missing_tol=tol)
        assert_parity(got.toarray(), want, cond, "f32", f"wide sweep {seed} band n={n} max_dist={max_dist} {km}x{kn}", max_ill_frac=0.05)
    else:
        shape = (int(rng.integers(60, 500)), int(rng.integers(60, 500)))
        a = rng.gamma(3, 0.4, size=shape) * (rng.random(shape) > 0.4)
        mr, mc = rng.random(shape[0]) < 0.05, rng.random(shape[1]) < 0.05
        a[mr, :] = 0
        a[:, mc] = 0
        mask = cup.make_missing_mask(shape, np.flatnonzero(~mr), np.flatnonzero(~mc), sym_upper=False)
        got, _ = cud.normxcorr2(sp.csr_matrix(a), kern, sym_upper=False, full=True, missing_mask=mask)
        assert last_kernel() == KERNEL_MFMA_WIDE
        want, cond = c_oracle.normxcorr2_rows(a, kern, 0, shape[0], sym_upper=False, full=True, miss_row=mr, miss_col=mc)
        assert_parity(got.toarray(), want, cond, "f32", f"wide sweep {seed} inter {shape} {km}x{kn}", max_ill_frac=0.05)


def test_candidate_sink_equals_the_map_path(monkeypatch):
    """cs_candidates (first half of detect mode) under a 21 x 21 template: the two-pass kernel appends the candidate pixels itself
    (CorrArgs::cand_keys: no coefficient map, no compaction pass) -- the same pixels and float64 scores as the map + compaction path
    on the runtime-size kernel, on a banded map with missing bins (whole map and a row window) and on an inter-like dense map."""
    dev = get_device()
    rng = np.random.default_rng(77)
    n, md = 1800, 420
    kern = template((21, 21))
    bw = md + 21 + 1
    ld = (bw + 4 + 63) // 64 * 64
    band = np.zeros((n, ld), dtype=np.float32)
    d = np.arange(bw)
    band[:, :bw] = np.minimum(rng.gamma(20, 0.05, size=(n, bw)), 10.0)
    # a few planted blobs so that some windows correlate
    for i0, d0 in [(200, 100), (700, 300), (1200, 50), (1500, 380), (900, 200)]:
        ii, jj = np.indices((21, 21))
        blob = 1.0 + 2.5 * np.exp(-((ii - 10) ** 2 + (jj - 10) ** 2) / 30.0)
        for r in range(21):
            for c in range(21):
                i, j = i0 - 10 + r, i0 + d0 - 10 + c
                if 0 <= j - i < bw:
                    band[i, j - i] *= blob[r, c]
    miss = (rng.random(n) < 0.03).astype(np.uint8)
    band[miss.astype(bool)] = 0
    for i in range(n):
        cols = i + d
        band[i, :bw][(cols >= n) | miss[np.minimum(cols, n - 1)].astype(bool)] = 0
    d_sig, d_miss = dev.to_device(band), dev.to_device(miss)
    sig = CsMatrix(d_sig.ptr, np_dtype_code(np.float32), LAYOUT_BAND, ld, 0, bw)
    kw = dict(pearson=0.25, lo_diag=0, hi_diag=md, inter=False, full=True, sym_upper=True, max_dist=md, mask_mode=MASK_BINS,
              miss_row=d_miss, miss_col=d_miss, missing_tol=0.5, precision="f32")
    for window in ((0, n), (517, 1290)):
        got = engine.run_candidates(dev, sig, (n, n), engine.KernelSpec(kern), window, **kw)
        assert last_kernel() == KERNEL_MFMA_WIDE
        monkeypatch.setenv("CHROMOSIGHT_HIP_NO_WIDE", "1")
        ref = engine.run_candidates(dev, sig, (n, n), engine.KernelSpec(kern), window, **kw)
        assert last_kernel() == KERNEL_GENERIC
        monkeypatch.delenv("CHROMOSIGHT_HIP_NO_WIDE")
        assert len(ref[0]) > 10
        a = np.lexsort((got[1], got[0]))
        b = np.lexsort((ref[1], ref[0]))
        assert np.array_equal(got[0][a], ref[0][b]) and np.array_equal(got[1][a], ref[1][b])
        assert np.abs(got[2][a] - ref[2][b]).max() < 1e-12           # float64 re-scoring: the same function on the same pixels
        # the short-list form of cs_candidates (re-scored unsorted, thresholded and ordered on the host) against the device form
        assert np.array_equal(a, np.arange(len(a)))                  # row-major as it comes
        monkeypatch.setenv("CHROMOSIGHT_HIP_NO_SMALL_KEEP", "1")
        dev_form = engine.run_candidates(dev, sig, (n, n), engine.KernelSpec(kern), window, **kw)
        monkeypatch.delenv("CHROMOSIGHT_HIP_NO_SMALL_KEEP")
        assert all(np.array_equal(x, y) for x, y in zip(got, dev_form))
    # a list longer than the heads that come back with the count (8192): the rest is decoded and re-scored from the same key list
    for kw["pearson"] in (0.2, 0.18, 0.16, 0.15, 0.14, 0.13, 0.12, 0.11, 0.1):
        got = engine.run_candidates(dev, sig, (n, n), engine.KernelSpec(kern), (0, n), **kw)
        if len(got[0]) > 8192:
            break
    assert last_kernel() == KERNEL_MFMA_WIDE and 8192 < len(got[0]) < 60000, (kw["pearson"], len(got[0]), last_kernel())
    # (the one launch is sized by the previous call's count: after a call with a handful of candidates the long list takes the
    # second round -- decoded and re-scored from the same key list --, the call after that does not)
    big = kw["pearson"]
    kw["pearson"] = 0.6
    few = engine.run_candidates(dev, sig, (n, n), engine.KernelSpec(kern), (0, n), **kw)
    assert len(few[0]) < 2000
    kw["pearson"] = big
    for _ in range(2):
        again = engine.run_candidates(dev, sig, (n, n), engine.KernelSpec(kern), (0, n), **kw)
        assert all(np.array_equal(x, y) for x, y in zip(got, again))
    monkeypatch.setenv("CHROMOSIGHT_HIP_NO_SMALL_KEEP", "1")
    dev_form = engine.run_candidates(dev, sig, (n, n), engine.KernelSpec(kern), (0, n), **kw)
    monkeypatch.delenv("CHROMOSIGHT_HIP_NO_SMALL_KEEP")
    assert all(np.array_equal(x, y) for x, y in zip(got, dev_form))
