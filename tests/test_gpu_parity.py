"""GPU parity tests: the HIP path (through the C ABI) against the golden vectors generated from
the reference and against the numpy oracle on seeded inputs.  Tolerances: float32 compute
1e-5 absolute on the coefficient (BASELINE.json north_star), float64 compute 1e-10."""
import numpy as np
import pytest
import scipy.sparse as sp

import chromosight_amd
from chromosight_amd.utils import detection as cud
from chromosight_amd.utils import preprocessing as cup
from oracle import c_oracle
from oracle import pearson_oracle as orc
from parity_util import assert_parity

pytestmark = pytest.mark.gpu

TOL = {"f32": 1e-5, "f64": 1e-10}


@pytest.fixture(params=["f32", "f64"])
def precision(request):
    old = chromosight_amd.get_precision()
    chromosight_amd.set_precision(request.param)
    yield request.param
    chromosight_amd.set_precision(old)


def coo(g, prefix):
    shape = tuple(g[f"{prefix}_shape"])
    return sp.coo_matrix((g[f"{prefix}_val"], (g[f"{prefix}_row"], g[f"{prefix}_col"])), shape=shape)


def oracle_cond(signal, kernel, **kw):
    """Conditioning of every output pixel (oracle/oracle.c), for assert_parity."""
    sig = np.asarray(signal.toarray() if sp.issparse(signal) else signal, dtype=np.float64)
    return c_oracle.normxcorr2_rows(sig, np.asarray(kernel, dtype=np.float64), 0, sig.shape[0], **kw)[1]


# ----------------------------------------------------------------------------------------------
def test_xcorr2_golden(golden, precision):
    g = golden("xcorr2")
    tol = 2e-6 if precision == "f32" else 1e-12
    for c in range(3):
        sig = g[f"sig{c}"]
        d = cud.xcorr2(sig, g["gauss_kernel"], threshold=1e-4)
        s = cud.xcorr2(sp.csr_matrix(sig), g["gauss_kernel"], threshold=1e-4)
        assert isinstance(d, np.ndarray) and sp.issparse(s)
        # values right at the 1e-4 zeroing threshold may flip in float32
        ref = g[f"dense{c}"]
        near = np.abs(np.abs(orc.xcorr2_oracle(sig, g["gauss_kernel"], threshold=0)) - 1e-4) < 1e-6
        assert np.abs(d - ref)[~near].max() < tol
        assert np.abs(s.toarray() - g[f"sparse{c}"])[~near].max() < tol
        k1 = np.ones((11, 11)) / 121
        cst = cud.xcorr2(sp.csr_matrix(sig), k1)
        near = np.abs(np.abs(orc.xcorr2_oracle(sig, k1, threshold=0)) - 1e-4) < 1e-6
        assert np.abs(cst.toarray() - g[f"const{c}"])[~near].max() < tol
    rtol = 2e-6 if precision == "f32" else 1e-13
    r = cud.xcorr2(g["rand"], g["rect_kernel_5x9"])
    assert np.abs(r - g["rand_rect_5x9"]).max() < rtol * np.abs(g["rand_rect_5x9"]).max()
    t = cud.xcorr2(sp.csr_matrix(g["rand"]), chromosight_amd.kernels.loops["kernels"][0], tsvd=0.999)
    assert np.abs(t.toarray() - g["rand_loops_tsvd999"]).max() < rtol * np.abs(g["rand_loops_tsvd999"]).max()


def test_normxcorr2_dense_golden(golden, templates, precision):
    g = golden("normxcorr2_dense")
    # sig_b holds an all-zero 30 x 30 region and a constant 20 x 23 patch (make_golden.py): flat windows by design
    flat = {"a": 0.0, "b": 0.35}
    for name in "ab":
        sig = g[f"sig_{name}"]
        for kname in ("loops", "small", "hairpin"):
            k = templates[kname]
            for full in (False, True):
                tag = f"{name}_{kname}_{'full' if full else 'valid'}"
                cond = oracle_cond(sig, k, full=full)
                cd, pd_ = cud.normxcorr2(sig, k, full=full, pval=not full)
                cs_, ps = cud.normxcorr2(sp.csr_matrix(sig), k, full=full, pval=True)
                assert isinstance(cd, np.ndarray) and sp.issparse(cs_)
                assert_parity(cd, g[f"dense_{tag}_corr"], cond, precision, f"dense {tag}", max_ill_frac=flat[name])
                assert_parity(cs_.toarray(), g[f"sparse_{tag}_corr"], cond, precision, f"sparse {tag}", max_ill_frac=flat[name])
                ref_p = g[f"sparse_{tag}_pval"]
                err = np.abs(ps.toarray() - ref_p)[cond >= 1e-3]
                # log10 p amplifies coefficient errors by up to ~ n / ln(10)
                assert err.max() < (5e-3 if precision == "f32" else 1e-7), tag
        sq = sig[:80, :80]
        cond = oracle_cond(np.triu(sq), templates["loops"], sym_upper=True, full=True)
        c, p = cud.normxcorr2(sp.csr_matrix(np.triu(sq)), templates["loops"], sym_upper=True, full=True, pval=True)
        assert_parity(c.toarray(), g[f"sparse_{name}_loops_symfull_corr"], cond, precision, f"{name} symfull", max_ill_frac=flat[name])
        assert np.all(np.tril(c.toarray(), -1) == 0)
        cond = oracle_cond(np.triu(sq), templates["loops"], sym_upper=True, full=False)
        c, _ = cud.normxcorr2(np.triu(sq), templates["loops"], sym_upper=True, full=False)
        assert_parity(c, g[f"dense_{name}_loops_symvalid_corr"], cond, precision, f"{name} symvalid", max_ill_frac=flat[name])
    c, _ = cud.normxcorr2(sp.csr_matrix(g["sig_a"]), templates["loops"], full=True, tsvd=0.999)
    assert np.abs(c.toarray() - g["sparse_a_loops_full_tsvd999_corr"]).max() < TOL[precision]


def test_normxcorr2_mask_golden(golden, templates, precision):
    g = golden("normxcorr2_mask")
    tol = TOL[precision]
    for i in range(int(g["n_intra"])):
        sig, k, valid = g[f"intra{i}_sig"], g[f"intra{i}_kernel"], g[f"intra{i}_valid"]
        md, mtol = int(g[f"intra{i}_max_dist"]), float(g[f"intra{i}_tol"])
        n = sig.shape[0]
        mask = cup.make_missing_mask((n, n), valid, valid, max_dist=md, sym_upper=True)
        c, p = cud.normxcorr2(sp.csr_matrix(sig), k, max_dist=md, sym_upper=True, full=True,
                              missing_mask=mask, missing_tol=mtol, pval=True)
        err = np.abs(c.toarray() - g[f"intra{i}_corr"])
        assert err.max() < tol, (i, err.max())
        perr = np.abs(p.toarray() - g[f"intra{i}_pval"])
        assert perr.max() < (5e-3 if precision == "f32" else 1e-7), (i, perr.max())
    n = g["nomd_sig"].shape[0]
    mask = cup.make_missing_mask((n, n), g["nomd_valid"], g["nomd_valid"], max_dist=None, sym_upper=True)
    c, _ = cud.normxcorr2(sp.csr_matrix(g["nomd_sig"]), templates["small"], max_dist=None, sym_upper=True,
                          full=True, missing_mask=mask, missing_tol=0.75)
    assert np.abs(c.toarray() - g["nomd_corr"]).max() < tol
    shape = g["inter_sig"].shape
    mask = cup.make_missing_mask(shape, g["inter_valid_rows"], g["inter_valid_cols"], max_dist=None, sym_upper=False)
    for kn in ("loops", "b11"):
        c, p = cud.normxcorr2(sp.csr_matrix(g["inter_sig"]), g[f"inter_{kn}_kernel"], max_dist=None,
                              sym_upper=False, full=True, missing_mask=mask, missing_tol=0.75, pval=True)
        assert np.abs(c.toarray() - g[f"inter_{kn}_corr"]).max() < tol, kn
    c, _ = cud.normxcorr2(sp.csr_matrix(g["valid_sig"]), templates["small"], max_dist=20, sym_upper=True,
                          full=False, missing_mask=sp.csr_matrix(g["valid_mask"]), missing_tol=0.75)
    assert np.abs(c.toarray() - g["valid_corr"]).max() < tol


def test_distance_law_detrend_golden(golden):
    g = golden("example_blocks")
    for ci in range(3):
        block = coo(g, f"chr{ci}_balanced")
        det = g[f"chr{ci}_det"]
        for pname in ("loops", "borders", "hairpins"):
            keep = int(g[f"chr{ci}_{pname}_keep"])
            law = cup.distance_law(block.tocsr(), detectable_bins=det, max_dist=keep, smooth=False)
            ref = g[f"chr{ci}_{pname}_law"]
            assert np.array_equal(np.isnan(law), np.isnan(ref))
            assert np.nanmax(np.abs(law - ref) / np.maximum(np.abs(ref), 1e-300)) < 1e-12
            m = cup.detrend(block, max_dist=keep, smooth=False, detectable_bins=det, max_val=10)
            m = cup.diag_trim(m.tocsr(), keep)
            m.data[np.isnan(m.data)] = 0
            m.eliminate_zeros()
            ref_m = coo(g, f"chr{ci}_{pname}_prepared").toarray()
            assert np.abs(m.toarray() - ref_m).max() < 1e-11


def test_reference_distance_law_example():
    """Known answer of the reference's own test (tests/test_preprocessing.py:202-213)."""
    m = np.ones((3, 3)) + np.array([1, 2, 3])
    assert np.all(cup.distance_law(sp.csr_matrix(m), smooth=False) == np.array([3.0, 3.5, 4.0]))
    assert np.all(cup.distance_law(sp.csr_matrix(m), smooth=True) == np.array([3.5, 3.5, 3.5]))


def test_example_blocks_maps_and_tables(golden, templates, precision):
    g = golden("example_blocks")
    tol = TOL[precision]
    pats = {
        "loops": (dict(pearson=0.3, max_perc_undetected=50.0, max_perc_zero=10.0, max_dist=2000000), [templates["loops"]]),
        "borders": (dict(pearson=0.15, max_perc_undetected=75.0, max_perc_zero=10.0, max_dist=0), templates["borders"]),
        "hairpins": (dict(pearson=0.1, max_perc_undetected=75.0, max_perc_zero=10.0, max_dist=0), [templates["hairpin"]]),
    }

    class Map:
        pass

    for ci in range(3):
        det = g[f"chr{ci}_det"]
        for pname, (cfg, kernels) in pats.items():
            m = coo(g, f"chr{ci}_{pname}_prepared").tocsr()
            max_dist = int(g[f"chr{ci}_{pname}_max_dist"])
            for ki, kern in enumerate(kernels):
                tag = f"chr{ci}_{pname}{ki}"
                mask = cup.make_missing_mask(m.shape, det, det, max_dist=max_dist, sym_upper=True)
                c, p = cud.normxcorr2(m, kern, max_dist=max_dist, sym_upper=True, full=True,
                                      missing_mask=mask, pval=True,
                                      missing_tol=cfg["max_perc_undetected"] / 100)
                ref = coo(g, f"{tag}_corr").toarray()
                assert np.abs(c.toarray() - ref).max() < tol, tag
                cmap = Map()
                cmap.matrix, cmap.detectable_bins = m.copy(), (det.copy(), det.copy())
                cmap.max_dist, cmap.inter = max_dist, False
                tab, wins = cud.pattern_detector(cmap, cfg, kern, full=True)
                ref_tab = g[f"{tag}_table"]
                if ref_tab.shape[0] == 0:
                    assert tab is None or len(tab) == 0
                    continue
                got = tab[["bin1", "bin2", "score", "pvalue"]].to_numpy(dtype=np.float64)
                # bit-exact coordinates, in the reference's order
                assert np.array_equal(got[:, :2], ref_tab[:, :2]), tag
                assert np.abs(got[:, 2] - ref_tab[:, 2]).max() < 1e-9, tag
                assert np.allclose(got[:, 3], ref_tab[:, 3], rtol=1e-6, atol=1e-300), tag
                assert np.allclose(wins, g[f"{tag}_windows"], equal_nan=True, rtol=0, atol=1e-12), tag


def test_quantify_and_inter(golden, templates):
    g = golden("example_blocks")

    class Map:
        pass

    cfg = dict(pearson=0.3, max_perc_undetected=50.0, max_perc_zero=10.0, max_dist=2000000)
    cmap = Map()
    cmap.matrix = coo(g, "quant_prepared").tocsr()
    det = g["quant_det"]
    cmap.detectable_bins, cmap.max_dist, cmap.inter = (det.copy(), det.copy()), int(g["quant_max_dist"]), False
    coords = g["quant_coords"].copy()
    tab, wins = cud.pattern_detector(cmap, cfg, templates["loops"], coords=coords, full=True)
    got = tab[["bin1", "bin2", "score", "pvalue"]].to_numpy(dtype=np.float64)
    ref = g["quant_table"]
    assert np.array_equal(got[:, :2], ref[:, :2])
    assert np.allclose(got[:, 2], ref[:, 2], equal_nan=True, rtol=0, atol=1e-9)
    assert np.allclose(got[:, 3], ref[:, 3], equal_nan=True, rtol=1e-6, atol=1e-300)
    assert np.allclose(wins, g["quant_windows"], equal_nan=True, rtol=0, atol=1e-12)
    # the reference shifts the caller's coords in place (detection.py:297-298)
    assert np.array_equal(coords, g["quant_coords"] + 8)
    cmap = Map()
    cmap.matrix = coo(g, "inter_prepared").tocsr()
    cmap.detectable_bins = (g["inter_det_rows"].copy(), g["inter_det_cols"].copy())
    cmap.max_dist, cmap.inter = None, True
    tab, wins = cud.pattern_detector(cmap, cfg, templates["loops"], coords=g["inter_coords"].copy(), full=True)
    got = tab[["bin1", "bin2", "score", "pvalue"]].to_numpy(dtype=np.float64)
    ref = g["inter_table"]
    assert np.array_equal(got[:, :2], ref[:, :2])
    assert np.allclose(got[:, 2], ref[:, 2], equal_nan=True, rtol=0, atol=1e-9)
    assert np.allclose(got[:, 3], ref[:, 3], equal_nan=True, rtol=1e-6, atol=1e-300)
    assert np.allclose(wins, g["inter_windows"], equal_nan=True, rtol=0, atol=1e-12)


def test_errors():
    k = chromosight_amd.kernels.loops["kernels"][0]
    sig = sp.csr_matrix(np.random.default_rng(0).random((40, 40)))
    with pytest.raises(ValueError):
        cud.normxcorr2(sig, np.ones((5, 5)))                       # flat kernel
    with pytest.raises(ValueError):
        cud.normxcorr2(sig, sp.csr_matrix(k))                      # sparse kernel
    with pytest.raises(ValueError):
        cud.normxcorr2(sig, k, missing_mask=np.zeros((40, 40), bool))   # dense mask
    with pytest.raises(ValueError):
        cud.normxcorr2(sig, k, missing_mask=sp.csr_matrix((30, 30), dtype=bool))
    with pytest.raises(ValueError):
        cud.normxcorr2(sig, k, missing_mask=sp.csr_matrix((40, 40), dtype=int))
    bad = sp.csr_matrix(np.ones((40, 40), dtype=bool))
    with pytest.raises(ValueError):
        cud.normxcorr2(sig, k, missing_mask=bad)                   # signal under the mask
    small = sp.csr_matrix(np.random.default_rng(0).random((10, 10)))
    with pytest.raises(ValueError):
        cud.normxcorr2(small, k, missing_mask=sp.csr_matrix((10, 10), dtype=bool))


# ----------------------------------------------------------------------------------------------
# seeded inputs spanning several strips / waves of the streaming kernel, against the C oracle
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(200, 300), (517, 261), (1100, 150)])
@pytest.mark.parametrize("ksize", [7, 11, 17])
def test_dense_multi_strip_vs_oracle(shape, ksize, precision):
    from oracle import c_oracle
    rng = np.random.default_rng(shape[0] + ksize)
    sig = rng.gamma(4, 0.25, size=shape)
    kern = chromosight_amd.kernels.loops["kernels"][0]
    if ksize != 17:
        kern = cup.resize_kernel(kern, factor=ksize / 17, quiet=True)
    assert kern.shape == (ksize, ksize)
    for full in (False, True):
        got, _ = cud.normxcorr2(sig.astype(np.float32) if precision == "f32" else sig, kern, full=full)
        ref_in = sig.astype(np.float32).astype(np.float64) if precision == "f32" else sig
        want, _ = c_oracle.normxcorr2(ref_in, kern, full=full)
        assert np.abs(got - want).max() < TOL[precision], (shape, ksize, full)


@pytest.mark.parametrize("n,max_dist,ksize,tol", [(700, 60, 17, 0.5), (1500, 300, 17, 0.5), (900, 40, 15, 0.75),
                                                  (400, 500, 7, 0.75), (650, 1, 17, 0.75)])
def test_banded_masked_multi_strip_vs_oracle(n, max_dist, ksize, tol, precision):
    """Upper-band maps with missing bins (clusters included), full mode, sym_upper: the production
    configuration of pattern_detector, at sizes that span many strips."""
    from oracle import c_oracle
    rng = np.random.default_rng(n + max_dist)
    kern = {17: chromosight_amd.kernels.loops["kernels"][0], 15: chromosight_amd.kernels.hairpins["kernels"][0],
            7: chromosight_amd.kernels.loops_small["kernels"][0]}[ksize]
    keep = min(max_dist, n) + ksize
    ii, jj = np.indices((n, n))
    a = rng.gamma(4, 0.25, size=(n, n)) * (rng.random((n, n)) > 0.15)
    a[(jj - ii < 0) | (jj - ii > keep)] = 0
    miss = np.zeros(n, dtype=bool)
    miss[rng.choice(n, size=n // 25, replace=False)] = True
    miss[n // 3:n // 3 + 6] = True
    miss[0] = miss[n - 1] = True
    a[miss, :] = 0
    a[:, miss] = 0
    valid = np.flatnonzero(~miss)
    mask = cup.make_missing_mask((n, n), valid, valid, max_dist=max_dist, sym_upper=True)
    got, _ = cud.normxcorr2(sp.csr_matrix(a), kern, max_dist=max_dist, sym_upper=True, full=True,
                            missing_mask=mask, missing_tol=tol)
    want, _ = c_oracle.normxcorr2(a, kern, max_dist=max_dist, sym_upper=True, full=True, miss_row=miss,
                                  miss_col=miss, missing_tol=tol)
    err = np.abs(got.toarray() - want)
    assert err.max() < TOL[precision], (n, max_dist, err.max())

    # the same map through pattern_detector's device path (per-bin flags instead of an explicit
    # mask): thresholded pixels must be identical to the oracle's
    class Map:
        pass
    cmap = Map()
    cmap.matrix, cmap.detectable_bins, cmap.max_dist, cmap.inter = sp.csr_matrix(a), (valid, valid.copy()), max_dist, False
    cfg = dict(pearson=0.2, max_perc_undetected=tol * 100, max_perc_zero=100.0, max_dist=5 * max_dist)
    tab, _ = cud.pattern_detector(cmap, cfg, kern, full=True)
    trimmed = np.where((jj - ii >= 0) & (jj - ii <= max_dist), want, 0.0)
    from oracle import foci_oracle
    want_tab = foci_oracle.detect_table(a, trimmed, miss, miss, kern.shape, pearson=0.2, zero_tol=1.0,
                                        missing_tol=tol, inter=False)
    if want_tab.shape[0] == 0:
        assert tab is None or len(tab) == 0
    else:
        got_tab = tab[["bin1", "bin2", "score"]].to_numpy(dtype=np.float64)
        assert np.array_equal(got_tab[:, :2], want_tab[:, :2]), (n, max_dist)     # same foci, same order
        assert np.abs(got_tab[:, 2] - want_tab[:, 2]).max() < 1e-9


def test_api_calls_from_several_threads_take_turns():
    """The reference's functions are plain numpy / scipy: callable from several threads at once.  Here calls that share the
    process-wide Device take turns (engine._one_call_per_context; a context owns its template weights, mask tables and
    pools) -- four threads with their own maps, templates and masks get the single-threaded results.  Without the lock all
    four got wrong ones (tools/stress_api_threads.py)."""
    import threading
    import chromosight_amd.kernels as ck
    from chromosight_amd.utils import preprocessing as cup
    jobs = []
    for t in range(4):
        rng = np.random.default_rng(100 + t)
        n = 300 + 60 * t
        a = np.triu(rng.gamma(4, 0.25, size=(n, n)))
        kern = np.asarray([ck.loops, ck.borders, ck.hairpins, ck.loops][t]["kernels"][0], dtype=np.float64)
        valid = np.flatnonzero(rng.random(n) > 0.03)
        mask = cup.make_missing_mask((n, n), valid, valid, max_dist=120, sym_upper=True)
        miss = np.ones(n, bool)
        miss[valid] = False
        a[miss, :] = 0
        a[:, miss] = 0
        jobs.append((sp.csr_matrix(a), kern, mask, rng.gamma(4, 0.25, size=(200 + 30 * t, 320))))

    def work(job):
        s, kern, mask, dense = job
        c1, p1 = cud.normxcorr2(s, kern, max_dist=120, sym_upper=True, full=True, missing_mask=mask, missing_tol=0.6, pval=True)
        c2, _ = cud.normxcorr2(dense, kern, full=False)
        c3 = cud.xcorr2(dense, kern)
        return c1.toarray(), p1.toarray(), c2, c3

    want = [work(j) for j in jobs]
    bad = []

    def thread(k):
        for _ in range(12):
            for g, w in zip(work(jobs[k]), want[k]):
                if not np.array_equal(g, w, equal_nan=True):
                    bad.append(k)
                    return

    threads = [threading.Thread(target=thread, args=(k,)) for k in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not bad, sorted(set(bad))


def test_centromeres_81x81_template_on_an_inter_block(precision):
    """The 81 x 81 `centromeres` template shipped in chromosight_amd/kernels (outside north_star's <= 21 x 21, but reachable
    through the boundary): an inter-chromosomal 600 x 520 block with missing bins, full mode, against the numpy oracle.
    Served by the runtime-size kernel (cs_last_kernel() == 1).  float64 arithmetic to 1e-10; float32 sums have 6561 terms:
    the 1e-5 bar belongs to the kernels that serve the BASELINE configurations, here the bound is 2e-4."""
    from chromosight_amd._lib import get_device
    kern = np.asarray(chromosight_amd.kernels.centromeres["kernels"][0], dtype=np.float64)
    assert kern.shape == (81, 81)
    rng = np.random.default_rng(81)
    ms, ns = 600, 520
    a = rng.gamma(4, 0.25, size=(ms, ns)) * (rng.random((ms, ns)) > 0.3)
    # a planted copy of the template, so that the map holds high coefficients too
    a[200:281, 300:381] = kern / kern.mean() * rng.gamma(50, 0.02, size=kern.shape)
    miss_r, miss_c = np.array([7, 100, 101, 433, 599]), np.array([0, 250, 251, 252, 519])
    a[miss_r, :] = 0
    a[:, miss_c] = 0
    valid_r, valid_c = np.setdiff1d(np.arange(ms), miss_r), np.setdiff1d(np.arange(ns), miss_c)
    mask = cup.make_missing_mask((ms, ns), valid_r, valid_c, sym_upper=False)
    got, _ = cud.normxcorr2(sp.csr_matrix(a), kern, full=True, missing_mask=mask, missing_tol=0.5)
    dev = get_device()
    assert dev.lib.cs_last_kernel(dev.ctx) == 1, "expected the runtime-size kernel for an 81 x 81 template"
    fr, fc = np.ones(ms, bool), np.ones(ns, bool)
    fr[valid_r] = False
    fc[valid_c] = False
    pred = orc.framed_missing_predicate((ms, ns), kern.shape, fr, fc, False, None)
    want, _ = orc.normxcorr2_oracle(a, kern, full=True, missing=pred, missing_tol=0.5)
    got = got.toarray() if sp.issparse(got) else np.asarray(got)
    err = float(np.abs(got - want).max())
    assert want.max() > 0.5                      # the planted copy is found
    assert err < (2e-4 if precision == "f32" else 1e-10), err
