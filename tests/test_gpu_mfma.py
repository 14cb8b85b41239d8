"""The matrix-core kernels (chromosight_amd/csrc/cs_corr_mfma.hip) against the C oracle.

Default dispatch sends unmasked dense float32 maps with large or odd-shaped templates to the
persistent dense kernel; CHROMOSIGHT_HIP_MFMA=1 sends every float32 call with a template of up to
17 x 17 to the matrix cores (general kernel: masks, bands, n_obs, float64 containers) and
CHROMOSIGHT_HIP_NO_MFMA=1 none -- all three must give the oracle's map at the 1e-5 bar, and the two
float32 kernels must agree with each other far below it."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

import chromosight_amd
import chromosight_amd.kernels as ck
from chromosight_amd import engine
from chromosight_amd._lib import LAYOUT_BAND, LAYOUT_DENSE, MASK_BINS, MASK_NONE, CsMatrix, get_device, np_dtype_code
from chromosight_amd.utils import detection as cud
from chromosight_amd.utils import preprocessing as cup
from oracle import c_oracle
from parity_util import assert_parity

# maps with missing bins: windows that lose (almost) all their present pixels, or whose present template pixels are all
# equal (piecewise-constant templates), are ill-defined; how many a test may hold, what they may do: tests/parity_util.py
ILL_MASKED = 0.1

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def f32_precision():
    old = chromosight_amd.get_precision()
    chromosight_amd.set_precision("f32")
    yield
    chromosight_amd.set_precision(old)


KERNEL_GENERIC, KERNEL_STREAM, KERNEL_MFMA, KERNEL_MFMA_DENSE, KERNEL_MFMA_REG, KERNEL_SEPARABLE, KERNEL_MFMA_WIDE = 1, 2, 3, 4, 5, 6, 7


def last_kernel():
    """Which native correlation kernel served the last call on the default context (cs_last_kernel)."""
    dev = get_device()
    return dev.lib.cs_last_kernel(dev.ctx)


def loops():
    return np.asarray(ck.loops["kernels"][0], dtype=np.float64)


def _signal(rng, shape, kind):
    if kind == "gamma":
        return rng.gamma(2.0, 1.0, size=shape)
    if kind == "hic":                       # detrended-map like: around 1, capped
        return np.minimum(rng.gamma(20, 0.05, size=shape), 10.0)
    if kind == "tiny":                      # scale far from 1: the per-tile power-of-two scale must absorb it
        return rng.gamma(2.0, 1.0, size=shape) * 3e-7
    if kind == "huge":
        return rng.gamma(2.0, 1.0, size=shape) * 7e8
    if kind == "signed":
        return rng.normal(size=shape)
    raise ValueError(kind)


@pytest.mark.parametrize("shape,kshape,kind", [
    ((300, 300), (17, 17), "gamma"), ((513, 777), (17, 17), "hic"), ((130, 64), (17, 17), "signed"),
    ((64, 70), (17, 17), "gamma"),          # row length not a multiple of 4: 4-byte transfers, scalar stores
    ((200, 333), (5, 9), "gamma"), ((150, 150), (15, 11), "hic"), ((97, 201), (13, 15), "gamma"),
    ((256, 256), (17, 17), "tiny"), ((256, 256), (17, 17), "huge"), ((90, 90), (3, 3), "gamma"),
])
@pytest.mark.parametrize("full", [True, False])
def test_dense_kernel_matches_oracle(shape, kshape, kind, full, monkeypatch):
    """Default dispatch (persistent dense kernel; forced for the small templates), every edge case of
    the tiling: partial tiles, frames narrower than a tile, valid-mode margins, rectangular and even
    templates, data far from unit scale."""
    if kshape[0] * kshape[1] < 169:
        monkeypatch.setenv("CHROMOSIGHT_HIP_MFMA", "1")
    rng = np.random.default_rng(shape[0] * 1000 + shape[1] + kshape[0])
    sig = _signal(rng, shape, kind).astype(np.float32)
    kern = loops() if kshape == (17, 17) else rng.normal(size=kshape) + 0.3
    if min(shape) < max(kshape) and not full:
        pytest.skip("no valid window")
    got, _ = cud.normxcorr2(sig, kern, full=full)
    assert last_kernel() == KERNEL_MFMA_DENSE
    want, cond = c_oracle.normxcorr2_rows(sig.astype(np.float64), kern, 0, shape[0], full=full)
    assert_parity(got, want, cond, "f32", f"mfma dense {shape} {kshape} {kind} full={full}")


def test_dense_kernel_equals_streaming_kernel(monkeypatch):
    """The two float32 kernels (matrix cores / packed FMA) on one map: same coefficients to ~1e-6."""
    rng = np.random.default_rng(5)
    sig = rng.gamma(2.0, 1.0, size=(1000, 1100)).astype(np.float32)
    a, _ = cud.normxcorr2(sig, loops(), full=True)
    assert last_kernel() == KERNEL_MFMA_DENSE
    monkeypatch.setenv("CHROMOSIGHT_HIP_NO_MFMA", "1")
    b, _ = cud.normxcorr2(sig, loops(), full=True)
    assert last_kernel() == KERNEL_STREAM
    assert np.abs(a - b).max() < 3e-6
    assert not np.array_equal(a, b)          # they are different kernels


def test_sym_upper_dense(monkeypatch):
    rng = np.random.default_rng(9)
    sig = np.triu(rng.gamma(2.0, 1.0, size=(260, 260))).astype(np.float32)
    for full in (True, False):
        got, _ = cud.normxcorr2(sig, loops(), sym_upper=True, full=full)
        assert last_kernel() == KERNEL_MFMA_DENSE
        want, cond = c_oracle.normxcorr2_rows(sig.astype(np.float64), loops(), 0, 260, sym_upper=True, full=full)
        assert_parity(got, want, cond, "f32", f"mfma dense sym_upper full={full}")
        assert np.all(np.tril(got, -1) == 0)


def test_general_kernel_masks_bands_nobs(monkeypatch):
    """CHROMOSIGHT_HIP_MFMA=1: the general matrix-core kernel on banded maps with per-bin masks (detect
    configuration, incl. n_obs), an explicit mask in valid mode and an inter-chromosomal block."""
    monkeypatch.setenv("CHROMOSIGHT_HIP_MFMA", "1")
    rng = np.random.default_rng(21)
    n, md = 900, 120
    ii, jj = np.indices((n, n))
    sig = np.triu(np.minimum(rng.gamma(20, 0.05, size=(n, n)), 10.0))
    sig[jj - ii > md + 17] = 0
    valid = np.flatnonzero(rng.random(n) > 0.04)
    miss = np.ones(n, bool)
    miss[valid] = False
    sig[miss, :] = 0
    sig[:, miss] = 0
    mask = cup.make_missing_mask((n, n), valid, valid, max_dist=md, sym_upper=True)
    for kern in (loops(), np.asarray(ck.borders["kernels"][0], dtype=np.float64)):
        c, _ = cud.normxcorr2(sp.csr_matrix(sig), kern, max_dist=md, sym_upper=True, full=True, missing_mask=mask,
                              missing_tol=0.75, pval=True)
        assert last_kernel() == KERNEL_MFMA
        want, cond = c_oracle.normxcorr2_rows(sig, kern, 0, n, max_dist=md, sym_upper=True, full=True, miss_row=miss,
                                              miss_col=miss)
        band = (jj - ii >= 0) & (jj - ii <= md)
        assert_parity(c.toarray()[band], want[band], cond[band], "f32", f"mfma general band {kern.shape}", max_ill_frac=ILL_MASKED)
    # inter-chromosomal block: dense, masks on both axes, no diagonal limits
    shape = (300, 420)
    inter = rng.gamma(2.0, 1.0, size=shape)
    vr, vc = np.flatnonzero(rng.random(300) > 0.05), np.flatnonzero(rng.random(420) > 0.05)
    mr, mc = np.ones(300, bool), np.ones(420, bool)
    mr[vr] = False
    mc[vc] = False
    inter[mr, :] = 0
    inter[:, mc] = 0
    mask = cup.make_missing_mask(shape, vr, vc, max_dist=None, sym_upper=False)
    c, _ = cud.normxcorr2(sp.csr_matrix(inter), loops(), max_dist=None, sym_upper=False, full=True, missing_mask=mask,
                          missing_tol=0.75)
    assert last_kernel() == KERNEL_MFMA
    want, cond = c_oracle.normxcorr2_rows(inter, loops(), 0, 300, full=True, miss_row=mr, miss_col=mc)
    assert_parity(c.toarray(), want, cond, "f32", "mfma general inter", max_ill_frac=ILL_MASKED)


@pytest.mark.parametrize("n,md,ksize", [(900, 120, 17), (1500, 400, 17), (700, 60, 9), (400, 399, 13)])
def test_tile_kernel_with_bin_masks(n, md, ksize, monkeypatch):
    """Per-bin masks on the persistent tile kernel (the default for the mirrored 17 x 17 loops template;
    CHROMOSIGHT_HIP_MFMA_REG=1 sends the other template sizes there too) (factorised mask tables of
    cs_mask_prep.hip, column flags x flagged-row cross term, edge / frame corrections), band in / band
    out, coefficients and n_obs (through the p-values) against the oracle."""
    monkeypatch.setenv("CHROMOSIGHT_HIP_MFMA_REG", "1")
    rng = np.random.default_rng(n + md)
    ii, jj = np.indices((n, n))
    sig = np.triu(np.minimum(rng.gamma(20, 0.05, size=(n, n)), 10.0))
    sig[jj - ii > md + ksize] = 0
    valid = np.flatnonzero(rng.random(n) > 0.04)
    miss = np.ones(n, bool)
    miss[valid] = False
    sig[miss, :] = 0
    sig[:, miss] = 0
    sig = sig.astype(np.float32)             # float32 band in HBM: the tile kernel's input type
    kern = loops() if ksize == 17 else rng.normal(size=(ksize, ksize)) + 0.3
    mask = cup.make_missing_mask((n, n), valid, valid, max_dist=md, sym_upper=True)
    c, p = cud.normxcorr2(sp.csr_matrix(sig), kern, max_dist=md, sym_upper=True, full=True, missing_mask=mask,
                          missing_tol=0.75, pval=True)
    assert last_kernel() == KERNEL_MFMA_REG
    want, cond = c_oracle.normxcorr2_rows(sig.astype(np.float64), kern, 0, n, max_dist=md, sym_upper=True, full=True,
                                          miss_row=miss, miss_col=miss)
    band = (jj - ii >= 0) & (jj - ii <= md)
    assert_parity(c.toarray()[band], want[band], cond[band], "f32", f"mfma tile kernel, bin masks n={n} md={md} k={ksize}", max_ill_frac=ILL_MASKED)
    # same call on the streaming kernel: coefficients and p-values (n_obs) agree
    monkeypatch.setenv("CHROMOSIGHT_HIP_MFMA_REG", "0")
    c2, p2 = cud.normxcorr2(sp.csr_matrix(sig), kern, max_dist=md, sym_upper=True, full=True, missing_mask=mask,
                            missing_tol=0.75, pval=True)
    assert last_kernel() == KERNEL_STREAM
    well = band & (cond >= 1e-3)
    assert np.abs(c.toarray() - c2.toarray())[well].max() < 5e-6
    assert np.abs(p.toarray() - p2.toarray())[well].max() < 5e-3


def test_f64_containers_are_narrowed_on_the_device():
    """float64 arrays (numpy's default) with float32 arithmetic: the rows are rounded to float32 by a device
    pass and take the persistent kernel, for normxcorr2 and xcorr2 alike."""
    rng = np.random.default_rng(33)
    sig = rng.gamma(2.0, 1.0, size=(210, 190))
    k = rng.normal(size=(7, 11))
    got = cud.xcorr2(sig, k, threshold=1e-4)                 # float64 container in, plain cross-correlation
    assert last_kernel() == KERNEL_MFMA_DENSE                # narrowed to float32 rows on the device first
    from oracle import pearson_oracle as orc
    want = orc.xcorr2_oracle(sig, k, threshold=0)
    near = np.abs(np.abs(want) - 1e-4) < 1e-6
    ref = np.where(np.abs(want) < 1e-4, 0.0, want)
    assert np.abs(got - ref)[~near].max() < 3e-6 * np.abs(want).max()
    c, _ = cud.normxcorr2(sig, loops(), full=True)            # float64 ndarray in
    assert last_kernel() == KERNEL_MFMA_DENSE
    w, cond = c_oracle.normxcorr2_rows(sig, loops(), 0, 210, full=True)
    assert_parity(c, w, cond, "f32", "mfma general f64 container")


def test_dense_kernel_row_windows():
    """Row windows through the persistent kernel (slab inputs): equal to the rows of the whole map."""
    dev = get_device()
    rng = np.random.default_rng(4)
    n, cols = 700, 900
    sig_h = rng.gamma(2.0, 1.0, size=(n, cols)).astype(np.float32)
    kspec = engine.KernelSpec(loops(), None)
    code = np_dtype_code(np.float32)

    def run(sig, out, window=None):
        params = engine._corr_params((n, cols), kspec, True, False, None, MASK_NONE, None, None, None, 0.75,
                                     engine.compute_code("f32"), window)
        dev._check(dev.lib.cs_normxcorr2(dev.ctx, None, C.byref(sig), C.byref(kspec.struct), C.byref(params), C.byref(out), None))

    d_sig = dev.to_device(sig_h)
    d_full = dev.zeros(n * cols, np.float32)
    run(CsMatrix(d_sig.ptr, code, LAYOUT_DENSE, cols, 0, 0), CsMatrix(d_full.ptr, code, LAYOUT_DENSE, cols, 0, 0))
    full = d_full.download().reshape(n, cols)
    assert last_kernel() == KERNEL_MFMA_DENSE
    want, cond = c_oracle.normxcorr2_rows(sig_h.astype(np.float64), loops(), 0, n, full=True)
    assert_parity(full, want, cond, "f32", "mfma dense whole map")
    for a, b in [(0, 37), (37, 200), (200, 201), (201, 695), (695, 700)]:
        ra, rb = max(0, a - 8), min(n, b + 8)
        d_slab = dev.to_device(np.ascontiguousarray(sig_h[ra:rb]))
        d_out = dev.zeros((b - a) * cols, np.float32)
        run(CsMatrix(d_slab.ptr, code, LAYOUT_DENSE, cols, 0, 0, ra), CsMatrix(d_out.ptr, code, LAYOUT_DENSE, cols, 0, 0, a), (a, b))
        got = d_out.download().reshape(b - a, cols)
        assert np.abs(got - full[a:b]).max() <= 2e-6, (a, b)


@pytest.mark.parametrize("shape,full", [((1500, 1111), True), ((1029, 1024), False), ((700, 2100), True)])
def test_host_pipelined_call(shape, full):
    """cs_normxcorr2_host: host map in, host map out, row slabs pipelined over PCIe -- equal to the oracle,
    with strided inputs / outputs and both output types; the Python surface takes this route for large
    float32 arrays."""
    dev = get_device()
    rng = np.random.default_rng(shape[0])
    ms, ns = shape
    kspec = engine.KernelSpec(loops(), None)
    backing = rng.gamma(2.0, 1.0, size=(ms, ns + 5)).astype(np.float32)
    sig = backing[:, :ns]                                    # row pitch ns + 5
    want, cond = c_oracle.normxcorr2_rows(sig.astype(np.float64), loops(), 0, ms, full=full)
    params = engine._corr_params((ms, ns), kspec, full, False, None, MASK_NONE, None, None, None, 0.75, engine.compute_code("f32"))
    for dt, code in ((np.float64, 1), (np.float32, 0)):
        out = np.full((ms, ns + 3), -7.0, dtype=dt)
        dev._check(dev.lib.cs_normxcorr2_host(dev.ctx, backing.ctypes.data, 0, ns + 5, C.byref(kspec.struct), C.byref(params),
                                              out.ctypes.data, code, ns + 3))
        assert last_kernel() == KERNEL_MFMA_DENSE
        assert_parity(out[:, :ns], want, cond, "f32", f"host pipelined {shape} full={full} {np.dtype(dt).name}")
        assert np.all(out[:, ns:] == -7.0)                   # the padding of the caller's rows is untouched
    # the Python surface: same numbers (float64 container), p-values from the plain template size
    big = np.ascontiguousarray(sig)
    got, pv = cud.normxcorr2(big, loops(), full=full, pval=True)
    assert got.dtype == np.float64 and last_kernel() == KERNEL_MFMA_DENSE
    assert_parity(got, want, cond, "f32", "python surface through the pipelined call")
    assert pv.shape == got.shape
    # masks are not served by this entry
    params_m = engine._corr_params((ms, ns), kspec, True, False, None, MASK_BINS, None, None, None, 0.75, engine.compute_code("f32"))
    out = np.zeros((ms, ns))
    assert dev.lib.cs_normxcorr2_host(dev.ctx, big.ctypes.data, 0, ns, C.byref(kspec.struct), C.byref(params_m), out.ctypes.data, 1, ns) == -3
    # float64 host map (numpy's default type): uploaded as it is, narrowed on the device slab by slab
    big64 = big.astype(np.float64)
    got64, _ = cud.normxcorr2(big64, loops(), full=full)
    assert last_kernel() == KERNEL_MFMA_DENSE
    assert np.array_equal(got64, got)


def test_dense_kernel_unaligned_rows():
    """16-byte tile transfers from rows that are only 4-byte aligned (odd row pitch, shifted base): same map
    as with 4-byte transfers, bit for bit, and within 1e-5 of the oracle."""
    dev = get_device()
    rng = np.random.default_rng(8)
    n, cols = 520, 516
    kspec = engine.KernelSpec(loops(), None)
    code = np_dtype_code(np.float32)
    for shift in (1, 3):
        ld = cols + 1
        host = rng.gamma(2.0, 1.0, size=n * ld + 8).astype(np.float32)
        d = dev.to_device(host)
        sig_h = host[shift:shift + n * ld].reshape(n, ld)[:, :cols]
        maps = {}
        for novec in (False, True):
            out = dev.zeros(n * cols, np.float32)
            params = engine._corr_params((n, cols), kspec, True, False, None, MASK_NONE, None, None, None, 0.75,
                                         engine.compute_code("f32"))
            import os
            if novec:
                os.environ["CHROMOSIGHT_HIP_MFMA_NOVEC"] = "1"
            try:
                dev._check(dev.lib.cs_normxcorr2(dev.ctx, None, C.byref(CsMatrix(d.ptr + 4 * shift, code, LAYOUT_DENSE, ld, 0, 0)),
                                                 C.byref(kspec.struct), C.byref(params),
                                                 C.byref(CsMatrix(out.ptr, code, LAYOUT_DENSE, cols, 0, 0)), None))
            finally:
                os.environ.pop("CHROMOSIGHT_HIP_MFMA_NOVEC", None)
            assert last_kernel() == KERNEL_MFMA_DENSE
            maps[novec] = out.download().reshape(n, cols)
        assert np.array_equal(maps[False], maps[True])
        want, cond = c_oracle.normxcorr2_rows(sig_h.astype(np.float64), loops(), 0, n, full=True)
        assert_parity(maps[False], want, cond, "f32", f"unaligned rows, shift {shift}")


@pytest.mark.parametrize("shape,kshape", [((700, 900), (17, 17)), ((300, 257), (7, 11)), ((130, 1500), (9, 9))])
def test_dense_kernel_xcorr2(shape, kshape):
    """Plain xcorr2 of a float32 array on the persistent kernel (no box sums, thresholded sum): equal to
    the numpy oracle up to float32 rounding of the sum, zero margins, same threshold semantics."""
    from oracle import pearson_oracle as orc
    rng = np.random.default_rng(shape[1])
    sig = rng.gamma(2.0, 1.0, size=shape).astype(np.float32)
    k = rng.normal(size=kshape)
    got = cud.xcorr2(sig, k, threshold=1e-4)
    assert last_kernel() == KERNEL_MFMA_DENSE
    want = orc.xcorr2_oracle(sig.astype(np.float64), k, threshold=0)
    near = np.abs(np.abs(want) - 1e-4) < 1e-5
    ref = np.where(np.abs(want) < 1e-4, 0.0, want)
    assert np.abs(got - ref)[~near].max() < 3e-6 * np.abs(want).max()
    kh, kw = (kshape[0] - 1) // 2, (kshape[1] - 1) // 2
    assert np.all(got[:kh] == 0) and np.all(got[:, :kw] == 0) and np.all(got[-kh:] == 0) and np.all(got[:, -kw:] == 0)


def test_mirrored_row_instance_equals_the_17_fragment_instance(monkeypatch):
    """The default masked tile kernel for the loops template keeps 9 weight head fragments (rows s and 16 - s of the
    template are equal); CHROMOSIGHT_HIP_MFMA_NORSYM=1 runs the general instance with 17: bit-identical maps
    (same MFMA operands in the same order).  float64 band containers are narrowed on the way in."""
    rng = np.random.default_rng(5)
    n, md = 1300, 300
    ii, jj = np.indices((n, n))
    sig = np.triu(np.minimum(rng.gamma(20, 0.05, size=(n, n)), 10.0))
    sig[jj - ii > md + 17] = 0
    valid = np.flatnonzero(rng.random(n) > 0.03)
    miss = np.ones(n, bool)
    miss[valid] = False
    sig[miss, :] = 0
    sig[:, miss] = 0
    mask = cup.make_missing_mask((n, n), valid, valid, max_dist=md, sym_upper=True)
    out = {}
    for tag, dtype in (("f32", np.float32), ("f64", np.float64)):
        for norsym in (False, True):
            if norsym:
                monkeypatch.setenv("CHROMOSIGHT_HIP_MFMA_NORSYM", "1")
            c, p = cud.normxcorr2(sp.csr_matrix(sig.astype(dtype)), loops(), max_dist=md, sym_upper=True, full=True,
                                  missing_mask=mask, missing_tol=0.75, pval=True)
            assert last_kernel() == KERNEL_MFMA_REG
            out[tag, norsym] = (c.toarray(), p.toarray())
            if norsym:
                monkeypatch.delenv("CHROMOSIGHT_HIP_MFMA_NORSYM")
        assert np.array_equal(out[tag, False][0], out[tag, True][0])
        assert np.array_equal(out[tag, False][1], out[tag, True][1])
    # the float64 container holds float32-representable values here?  no: it is rounded on the way in
    want, cond = c_oracle.normxcorr2_rows(sig, loops(), 0, n, max_dist=md, sym_upper=True, full=True, miss_row=miss, miss_col=miss)
    band = (jj - ii >= 0) & (jj - ii <= md)
    assert_parity(out["f64", False][0][band], want[band], cond[band], "f32", "masked tile kernel, float64 container", max_ill_frac=ILL_MASKED)


@pytest.mark.parametrize("name", ["stripes_left", "stripes_right"])
def test_separable_kernel_on_rank1_templates(name, monkeypatch):
    """The 31 x 31 stripes templates are outer products u v^T: cs_corr_sep.hip forms every window sum from a
    horizontal and a vertical 1-D pass (62 instead of 961 products per sum).  Banded map with per-bin masks and
    n_obs (detect configuration) and a dense unmasked map, against the float64 oracle and against the runtime-size
    kernel (CHROMOSIGHT_HIP_NO_SEPARABLE=1)."""
    kern = np.asarray(getattr(ck, name)["kernels"][0], dtype=np.float64)
    assert kern.shape == (31, 31) and np.linalg.matrix_rank(kern, tol=1e-12) == 1
    rng = np.random.default_rng(31)
    n, md = 700, 150
    ii, jj = np.indices((n, n))
    sig = np.triu(np.minimum(rng.gamma(20, 0.05, size=(n, n)), 10.0))
    sig[jj - ii > md + 31] = 0
    valid = np.flatnonzero(rng.random(n) > 0.03)
    miss = np.ones(n, bool)
    miss[valid] = False
    sig[miss, :] = 0
    sig[:, miss] = 0
    mask = cup.make_missing_mask((n, n), valid, valid, max_dist=md, sym_upper=True)
    c, p = cud.normxcorr2(sp.csr_matrix(sig), kern, max_dist=md, sym_upper=True, full=True, missing_mask=mask,
                          missing_tol=0.75, pval=True)
    assert last_kernel() == KERNEL_SEPARABLE
    want, cond = c_oracle.normxcorr2_rows(sig, kern, 0, n, max_dist=md, sym_upper=True, full=True, miss_row=miss, miss_col=miss)
    band = (jj - ii >= 0) & (jj - ii <= md)
    assert_parity(c.toarray()[band], want[band], cond[band], "f32", f"separable kernel, {name}, band + masks", max_ill_frac=ILL_MASKED)
    dense = rng.gamma(4.0, 0.25, size=(300, 411)).astype(np.float32)
    cd, _ = cud.normxcorr2(dense, kern)
    assert last_kernel() == KERNEL_SEPARABLE
    wd, cond_d = c_oracle.normxcorr2_rows(dense.astype(np.float64), kern, 0, 300, full=False)
    assert_parity(cd, wd, cond_d, "f32", f"separable kernel, {name}, dense", max_ill_frac=ILL_MASKED)
    monkeypatch.setenv("CHROMOSIGHT_HIP_NO_SEPARABLE", "1")
    c3, p3 = cud.normxcorr2(sp.csr_matrix(sig), kern, max_dist=md, sym_upper=True, full=True, missing_mask=mask,
                            missing_tol=0.75, pval=True)
    assert last_kernel() == KERNEL_MFMA_WIDE            # (the two-pass matrix-core kernel: full-rank evaluation of the same template)
    well = band & (cond >= 1e-3)
    assert np.abs(c.toarray() - c3.toarray())[well].max() < 5e-6
    assert np.abs(p.toarray() - p3.toarray())[well].max() < 5e-3
    monkeypatch.setenv("CHROMOSIGHT_HIP_NO_WIDE", "1")
    c2, p2 = cud.normxcorr2(sp.csr_matrix(sig), kern, max_dist=md, sym_upper=True, full=True, missing_mask=mask,
                            missing_tol=0.75, pval=True)
    assert last_kernel() == KERNEL_GENERIC
    # the runtime-size kernel adds 961 float32 products per sum (observed 2.3e-5 from the oracle on this map, where the
    # separable kernel's 31 + 31 stay within 2.2e-6): the cross-check is correspondingly loose
    assert np.abs(c.toarray() - c2.toarray())[well].max() < 5e-5
    assert np.abs(p.toarray() - p2.toarray())[well].max() < 5e-3


def test_default_dispatch_by_template():
    """Which kernel serves a masked, banded detect-style call by default (cs_api.cpp launch_corr): the masked matrix-core
    tile kernel for 15 x 15 and 17 x 17 templates, the streaming kernel below, the separable kernel for outer-product
    templates without an unrolled instance (on a narrow band), the two-pass matrix-core kernel for templates with a side of
    18 .. 33, the runtime-size kernel for the rest; dense unmasked maps take the dense tile kernel."""
    rng = np.random.default_rng(8)
    n, md = 500, 90
    sig = np.triu(np.minimum(rng.gamma(20, 0.05, size=(n, n)), 10.0))
    valid = np.flatnonzero(rng.random(n) > 0.03)
    miss = np.ones(n, bool)
    miss[valid] = False
    sig[miss, :] = 0
    sig[:, miss] = 0
    mask = cup.make_missing_mask((n, n), valid, valid, max_dist=md, sym_upper=True)
    u, v = rng.normal(size=21) + 0.5, rng.normal(size=21) + 0.5
    cases = [(loops(), KERNEL_MFMA_REG), (np.asarray(ck.borders["kernels"][0], dtype=np.float64), KERNEL_MFMA_REG),
             (np.asarray(ck.hairpins["kernels"][0], dtype=np.float64), KERNEL_MFMA_REG),
             (rng.normal(size=(13, 13)) + 0.3, KERNEL_STREAM), (rng.normal(size=(7, 7)) + 0.3, KERNEL_STREAM),
             (np.asarray(ck.stripes_left["kernels"][0], dtype=np.float64), KERNEL_SEPARABLE),
             (np.outer(u, v), KERNEL_SEPARABLE), (rng.normal(size=(21, 21)) + 0.3, KERNEL_MFMA_WIDE),
             (rng.normal(size=(19, 33)) + 0.3, KERNEL_MFMA_WIDE), (rng.normal(size=(35, 35)) + 0.3, KERNEL_GENERIC)]
    for kern, expected in cases:
        cud.normxcorr2(sp.csr_matrix(sig), kern, max_dist=md, sym_upper=True, full=True, missing_mask=mask, missing_tol=0.75)
        assert last_kernel() == expected, (kern.shape, last_kernel(), expected)
    cud.normxcorr2(rng.gamma(4.0, 0.25, size=(300, 320)).astype(np.float32), loops())
    assert last_kernel() == KERNEL_MFMA_DENSE
