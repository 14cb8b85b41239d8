"""Pins the oracle (oracle/pearson_oracle.py and oracle/oracle.c) to the reference: every vector
in tests/golden/*.npz was produced by importing /root/reference (tests/golden/make_golden.py).
CPU only."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import c_oracle
from oracle import pearson_oracle as orc

TIGHT = 1e-12


def coo(g, prefix):
    shape = tuple(g[f"{prefix}_shape"])
    return sp.coo_matrix((g[f"{prefix}_val"], (g[f"{prefix}_row"], g[f"{prefix}_col"])), shape=shape)


def flags(valid, n):
    f = np.ones(n, dtype=bool)
    f[valid] = False
    return f


def test_xcorr2(golden):
    g = golden("xcorr2")
    for c in range(3):
        o = orc.xcorr2_oracle(g[f"sig{c}"], g["gauss_kernel"])
        assert np.abs(o - g[f"dense{c}"]).max() < TIGHT
        assert np.abs(o - g[f"sparse{c}"]).max() < TIGHT
        # the reference's own test oracle: scipy.signal.correlate2d "valid"
        # (reference tests/test_detection.py:251-270)
        sc = np.zeros_like(o)
        sc[3:-3, 3:-3] = g[f"scipy_valid{c}"]
        sc[sc < 1e-4] = 0
        assert np.allclose(o, sc)
        assert np.abs(orc.xcorr2_oracle(g[f"sig{c}"], np.ones((11, 11)) / 121) - g[f"const{c}"]).max() < TIGHT
    assert np.abs(orc.xcorr2_oracle(g["rand"], g["rect_kernel_5x9"]) - g["rand_rect_5x9"]).max() < 1e-11
    k_tsvd = g["loops_tsvd999_u"] @ g["loops_tsvd999_v"]
    assert np.abs(orc.xcorr2_oracle(g["rand"], k_tsvd) - g["rand_loops_tsvd999"]).max() < 1e-10


def test_normxcorr2_dense(golden, templates):
    g = golden("normxcorr2_dense")
    for name, tol in (("a", TIGHT), ("b", 1e-6)):   # sig_b holds a constant patch: 0/0 windows
        s = g[f"sig_{name}"]
        for kname in ("loops", "small", "hairpin"):
            k = templates[kname]
            for full in (False, True):
                tag = f"{name}_{kname}_{'full' if full else 'valid'}"
                r, nobs = orc.normxcorr2_oracle(s, k, full=full)
                assert np.abs(r - g[f"dense_{tag}_corr"]).max() < tol, tag
                assert np.abs(r - g[f"sparse_{tag}_corr"]).max() < tol, tag
                rc, _ = c_oracle.normxcorr2(s, k, full=full)
                assert np.abs(rc - g[f"dense_{tag}_corr"]).max() < tol, tag
                lp = np.where(r != 0, orc.corr_to_pval_oracle(r, nobs), 0.0)
                assert np.nanmax(np.abs(lp - g[f"sparse_{tag}_pval"])) < max(tol * 1e3, 1e-9), tag
        sq = np.triu(s[:80, :80])
        r, _ = orc.normxcorr2_oracle(sq, templates["loops"], sym_upper=True, full=True)
        assert np.abs(r - g[f"sparse_{name}_loops_symfull_corr"]).max() < tol
        r, _ = orc.normxcorr2_oracle(sq, templates["loops"], sym_upper=True, full=False)
        assert np.abs(r - g[f"dense_{name}_loops_symvalid_corr"]).max() < tol


def test_missing_predicate(golden):
    g = golden("masks")
    for i in range(int(g["n_sym"])):
        n = int(g[f"sym{i}_n"])
        ks = tuple(int(x) for x in g[f"sym{i}_kshape"])
        md = int(g[f"sym{i}_max_dist"])
        md = None if md < 0 else md
        miss = flags(g[f"sym{i}_valid"], n)
        H, W = n + 2 * (ks[0] - 1), n + 2 * (ks[1] - 1)
        ref = np.unpackbits(g[f"sym{i}_framed"])[:H * W].reshape(H, W).astype(bool)
        assert np.array_equal(orc.framed_missing_predicate((n, n), ks, miss, miss, True, md), ref), i
    for i in range(int(g["n_inter"])):
        shape = tuple(int(x) for x in g[f"inter{i}_shape"])
        ks = tuple(int(x) for x in g[f"inter{i}_kshape"])
        mr, mc = flags(g[f"inter{i}_valid_rows"], shape[0]), flags(g[f"inter{i}_valid_cols"], shape[1])
        H, W = shape[0] + 2 * (ks[0] - 1), shape[1] + 2 * (ks[1] - 1)
        ref = np.unpackbits(g[f"inter{i}_framed"])[:H * W].reshape(H, W).astype(bool)
        assert np.array_equal(orc.framed_missing_predicate(shape, ks, mr, mc, False, None), ref), i


def test_normxcorr2_masked(golden, templates):
    g = golden("normxcorr2_mask")
    for i in range(int(g["n_intra"])):
        sig, k, valid = g[f"intra{i}_sig"], g[f"intra{i}_kernel"], g[f"intra{i}_valid"]
        md, tol = int(g[f"intra{i}_max_dist"]), float(g[f"intra{i}_tol"])
        miss = flags(valid, sig.shape[0])
        M = orc.framed_missing_predicate(sig.shape, k.shape, miss, miss, True, md)
        r, nobs = orc.normxcorr2_oracle(sig, k, max_dist=md, sym_upper=True, full=True, missing=M, missing_tol=tol)
        assert np.abs(r - g[f"intra{i}_corr"]).max() < TIGHT, i
        lp = np.where(r != 0, orc.corr_to_pval_oracle(r, nobs), 0.0)
        assert np.nanmax(np.abs(lp - g[f"intra{i}_pval"])) < 1e-9, i
        rc, _ = c_oracle.normxcorr2(sig, k, max_dist=md, sym_upper=True, full=True, miss_row=miss,
                                    miss_col=miss, missing_tol=tol)
        assert np.abs(rc - g[f"intra{i}_corr"]).max() < TIGHT, i
    miss = flags(g["nomd_valid"], g["nomd_sig"].shape[0])
    M = orc.framed_missing_predicate(g["nomd_sig"].shape, (7, 7), miss, miss, True, None)
    r, _ = orc.normxcorr2_oracle(g["nomd_sig"], templates["small"], sym_upper=True, full=True, missing=M)
    assert np.abs(r - g["nomd_corr"]).max() < TIGHT
    shape = g["inter_sig"].shape
    mr, mc = flags(g["inter_valid_rows"], shape[0]), flags(g["inter_valid_cols"], shape[1])
    for kn in ("loops", "b11"):
        k = g[f"inter_{kn}_kernel"]
        M = orc.framed_missing_predicate(shape, k.shape, mr, mc, False, None)
        r, _ = orc.normxcorr2_oracle(g["inter_sig"], k, full=True, missing=M)
        assert np.abs(r - g[f"inter_{kn}_corr"]).max() < TIGHT, kn
        rc, _ = c_oracle.normxcorr2(g["inter_sig"], k, full=True, miss_row=mr, miss_col=mc)
        assert np.abs(rc - g[f"inter_{kn}_corr"]).max() < TIGHT, kn
    r, _ = orc.normxcorr2_oracle(g["valid_sig"], templates["small"], max_dist=20, sym_upper=True, full=False,
                                 missing=g["valid_mask"])
    assert np.abs(r - g["valid_corr"]).max() < TIGHT


@pytest.mark.parametrize("ci", [0, 2])
def test_example_cool_blocks(golden, templates, ci):
    """Real data: balanced block -> distance law -> detrended block -> coefficient map."""
    g = golden("example_blocks")
    det = g[f"chr{ci}_det"]
    n = int(g[f"chr{ci}_n"])
    block = coo(g, f"chr{ci}_balanced")
    dense = block.toarray()
    stored = np.zeros((n, n), dtype=bool)
    stored[block.row, block.col] = True
    for pname, kern, tol in (("loops", templates["loops"], 0.5), ("borders", templates["borders"][1], 0.75)):
        keep = int(g[f"chr{ci}_{pname}_keep"])
        md = int(g[f"chr{ci}_{pname}_max_dist"])
        det_ref = orc.distance_law_oracle(np.where(stored, dense, 0.0), det, keep)
        ref_law = g[f"chr{ci}_{pname}_law"]
        assert np.array_equal(np.isnan(det_ref), np.isnan(ref_law))
        assert np.nanmax(np.abs(det_ref - ref_law)) < 1e-12
        out, _ = orc.detrend_oracle(dense, stored, det, keep, max_val=10)
        ii, jj = np.indices((n, n))
        out[(jj - ii < 0) | (jj - ii > keep)] = 0
        out[np.isnan(out)] = 0
        prepared = coo(g, f"chr{ci}_{pname}_prepared").toarray()
        assert np.abs(out - prepared).max() < 1e-11
        miss = flags(det, n)
        ki = 0 if pname == "loops" else 1
        rc, _ = c_oracle.normxcorr2(prepared, kern, max_dist=md, sym_upper=True, full=True, miss_row=miss,
                                    miss_col=miss, missing_tol=tol)
        assert np.abs(rc - coo(g, f"chr{ci}_{pname}{ki}_corr").toarray()).max() < 1e-11


def test_corr_to_pval(golden):
    g = golden("stats")
    lp = orc.corr_to_pval_oracle(g["r"], g["n"].astype(float))
    ref = g["logp_vec"]
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(lp), fin)
    assert np.abs(lp[fin] - ref[fin]).max() < 1e-9


# ------------------------------------------------------------------------------------------------
# oracle/foci_oracle.py (focus picking + window validation) pinned to the reference's captures
# ------------------------------------------------------------------------------------------------
def _coo(g, prefix):
    import scipy.sparse as sp
    return sp.coo_matrix((g[f"{prefix}_val"], (g[f"{prefix}_row"], g[f"{prefix}_col"])),
                         shape=tuple(g[f"{prefix}_shape"]))


def test_foci_oracle_matches_reference_pick_foci(golden):
    from oracle import foci_oracle
    g = golden("nms")
    tags = sorted({k[:-len("_pearson")] for k in g if k.endswith("_pearson")})
    assert len(tags) == 15
    for tag in tags:
        conv = _coo(g, f"{tag}_conv").toarray()
        foci = foci_oracle.pick_foci_dense(conv, float(g[f"{tag}_pearson"]))
        assert np.array_equal(foci, g[f"{tag}_foci"].reshape(-1, 2)), tag
        # the same map in band coordinates (diagonals 0 .. width-1)
        n = conv.shape[0]
        width = int(np.max(_coo(g, f"{tag}_conv").col - _coo(g, f"{tag}_conv").row)) + 1
        band = np.zeros((n, width))
        for d in range(width):
            band[:n - d, d] = np.diagonal(conv, d)
        assert np.array_equal(foci_oracle.pick_foci_band(band, 0, float(g[f"{tag}_pearson"])), foci), tag


def test_foci_oracle_matches_reference_tables(golden, templates):
    """pattern_detector tables and windows of the three example chromosomes x five templates,
    rebuilt by the oracle from the reference's own coefficient maps."""
    from oracle import foci_oracle
    g = golden("example_blocks")
    pats = {"loops": (0.3, 50.0, 10.0, False, [templates["loops"]]),
            "borders": (0.15, 75.0, 10.0, True, templates["borders"]),
            "hairpins": (0.1, 75.0, 10.0, True, [templates["hairpin"]])}
    n_rows = 0
    for ci in range(3):
        det = g[f"chr{ci}_det"]
        for pname, (pearson, pu, pz, diag_only, kernels) in pats.items():
            m = _coo(g, f"chr{ci}_{pname}_prepared").toarray()
            n = m.shape[0]
            max_dist = int(g[f"chr{ci}_{pname}_max_dist"])
            miss = np.ones(n, dtype=bool)
            miss[det] = False
            ii, jj = np.indices((n, n))
            for ki, kern in enumerate(kernels):
                tag = f"chr{ci}_{pname}{ki}"
                corr = _coo(g, f"{tag}_corr").toarray()
                trimmed = np.where((jj - ii >= 0) & (jj - ii <= max_dist), corr, 0.0)
                tab = foci_oracle.detect_table(m, trimmed, miss, miss, np.shape(kern), pearson, pz / 100, pu / 100,
                                               inter=False, diag_only=diag_only)
                ref = g[f"{tag}_table"]
                assert tab.shape[0] == ref.shape[0], tag
                if ref.shape[0]:
                    assert np.array_equal(tab[:, :2], ref[:, :2]), tag
                    assert np.abs(tab[:, 2] - ref[:, 2]).max() < 1e-12, tag
                    # windows of the validated patterns
                    foci = tab[:, :2].astype(int)
                    _, wins = foci_oracle.validate(foci, lambda p, q: m[p, q], m.shape, miss, miss, np.shape(kern),
                                                   pz / 100, pu / 100, False)
                    assert np.allclose(wins, g[f"{tag}_windows"], equal_nan=True, rtol=0, atol=1e-12), tag
                n_rows += ref.shape[0]
    assert n_rows > 200


def test_detrend_oracle_matches_reference_blocks(golden):
    """oracle/detrend_oracle.py (balanced band, distance law, detrend, trim) against the laws and
    prepared blocks the reference produced from data_test/example.cool."""
    from oracle import detrend_oracle
    cool = golden("example_cool")
    g = golden("example_blocks")
    for ci in range(3):
        for pname in ("loops", "borders", "hairpins"):
            keep = int(g[f"chr{ci}_{pname}_keep"])
            band, det = detrend_oracle.balanced_band(cool, ci, keep)
            assert np.array_equal(np.flatnonzero(det), g[f"chr{ci}_det"])
            prepared, law = detrend_oracle.prepare_band(band, det)
            ref_law = g[f"chr{ci}_{pname}_law"]
            w = min(band.shape[1], band.shape[0])
            assert np.array_equal(np.isnan(law[:w]), np.isnan(ref_law[:w]))
            assert np.nanmax(np.abs(law[:w] - ref_law[:w]) / np.maximum(np.abs(ref_law[:w]), 1e-300)) < 1e-12
            ref = _coo(g, f"chr{ci}_{pname}_prepared").toarray()
            n = ref.shape[0]
            dense = np.zeros((n, n))
            for d in range(min(band.shape[1], n)):
                dense[np.arange(n - d), np.arange(n - d) + d] = prepared[:n - d, d]
            assert np.abs(dense - ref).max() < 1e-11, (ci, pname)


def _nonsquare_case(g, tag):
    import scipy.sparse as sp
    n = int(g[f"{tag}_n"])
    m = sp.coo_matrix((g[f"{tag}_prepared_val"], (g[f"{tag}_prepared_row"], g[f"{tag}_prepared_col"])), shape=(n, n)).toarray()
    miss = np.ones(n, dtype=bool)
    miss[g[f"{tag}_det"]] = False
    pearson, pu, pz, md_bp = g[f"{tag}_cfg"]
    return m, miss, g[f"{tag}_kernel"], float(pearson), pu / 100, pz / 100, int(g[f"{tag}_max_dist"]), md_bp == 0


@pytest.mark.parametrize("tag", ["d2_59", "d1_37"])
def test_oracles_match_reference_nonsquare_templates(golden, tag):
    """Non-square templates in full mode (reference detection.py:287-345): the padding is (kw rows, kh columns), the
    coordinate shift (kh, kw), so windows, scores and the row of 1-D patterns are offset by kh - kw.  The oracle
    pipeline (pearson_oracle + foci_oracle) reproduces the reference's tables and windows."""
    from oracle import foci_oracle
    g = golden("nonsquare")
    m, miss, kern, pearson, missing_tol, zero_tol, max_dist, diag_only = _nonsquare_case(g, tag)
    n = m.shape[0]
    pred = orc.framed_missing_predicate((n, n), kern.shape, miss, miss, True, max_dist)
    corr, _ = orc.normxcorr2_oracle(m, kern, max_dist=max_dist, sym_upper=True, full=True, missing=pred, missing_tol=missing_tol)
    ii, jj = np.indices((n, n))
    trimmed = np.where((jj - ii >= 0) & (jj - ii <= max_dist), np.nan_to_num(corr), 0.0)
    tab, wins = foci_oracle.detect_table(m, trimmed, miss, miss, kern.shape, pearson, zero_tol, missing_tol,
                                         diag_only=diag_only, return_windows=True)
    ref = g[f"{tag}_table"]
    assert ref.shape[0] > 20 and tab.shape[0] == ref.shape[0]
    assert np.array_equal(tab[:, :2], ref[:, :2])
    assert np.abs(tab[:, 2] - ref[:, 2]).max() < 1e-12
    assert np.allclose(wins, g[f"{tag}_windows"], equal_nan=True, rtol=0, atol=1e-12)
    kh, kw = (kern.shape[0] - 1) // 2, (kern.shape[1] - 1) // 2
    if diag_only:
        assert np.array_equal(ref[:, 0], ref[:, 1] + kw - kh)          # the quirk is in the fixture


@pytest.mark.parametrize("tag", ["inter59", "inter95"])
def test_oracles_match_reference_nonsquare_inter(golden, tag):
    import scipy.sparse as sp
    from oracle import foci_oracle
    g = golden("nonsquare")
    shape = tuple(int(x) for x in g["inter_shape"])
    m = sp.coo_matrix((g["inter_prepared_val"], (g["inter_prepared_row"], g["inter_prepared_col"])), shape=shape).toarray()
    mr, mc = np.ones(shape[0], bool), np.ones(shape[1], bool)
    mr[g["inter_det_rows"]] = False
    mc[g["inter_det_cols"]] = False
    kern = g[f"{tag}_kernel"]
    pearson, pu, pz, _ = g["inter_cfg"]
    pred = orc.framed_missing_predicate(shape, kern.shape, mr, mc, False, None)
    corr, _ = orc.normxcorr2_oracle(m, kern, max_dist=None, sym_upper=False, full=True, missing=pred, missing_tol=pu / 100)
    tab, wins = foci_oracle.detect_table(m, np.nan_to_num(corr), mr, mc, kern.shape, float(pearson), pz / 100, pu / 100,
                                         inter=True, return_windows=True)
    ref = g[f"{tag}_table"]
    assert ref.shape[0] > 5 and tab.shape[0] == ref.shape[0]
    assert np.array_equal(tab[:, :2], ref[:, :2])
    assert np.abs(tab[:, 2] - ref[:, 2]).max() < 1e-12
    assert np.allclose(wins, g[f"{tag}_windows"], equal_nan=True, rtol=0, atol=1e-12)


@pytest.mark.parametrize("tag", ["nan", "inf", "ninf"])
def test_oracle_with_nonfinite_pixels(golden, tag):
    """The C oracle on the reference's captures with one NaN / infinite pixel (tests/golden/nonfinite.npz): the sums of every
    window that holds it are non-finite there as in the reference, and come out 0 (detection.py:1088-1101)."""
    g = golden("nonfinite")
    a, want, valid = g[f"{tag}_in"], g[f"{tag}_corr"], g["valid"]
    n = a.shape[0]
    miss = np.ones(n, bool)
    miss[valid] = False
    got, _ = c_oracle.normxcorr2_rows(a, g["kernel"], 0, n, max_dist=40, sym_upper=True, full=True, miss_row=miss, miss_col=miss,
                                      missing_tol=0.75)
    got = np.where(np.isfinite(got), got, 0.0)
    ii, jj = np.indices((n, n))
    band = (jj >= ii) & (jj - ii <= 40)
    assert np.abs(got - want)[band].max() < 1e-12               # (the reference's map is not trimmed to max_dist yet: in-band pixels)
