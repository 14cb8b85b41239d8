"""Host-side logic of the API mirror (no GPU): masks, trims, padding, foci labelling and picking,
window validation, neighbour removal, template editing, statistics -- against the vectors
generated from the reference (tests/golden/make_golden.py) and the known answers of the
reference's own tests."""
import numpy as np
import pandas as pd
import pytest
import scipy.sparse as sp

from chromosight_amd.utils import detection as cud
from chromosight_amd.utils import preprocessing as cup
from chromosight_amd.utils import stats as cus


def coo(g, prefix):
    shape = tuple(g[f"{prefix}_shape"])
    return sp.coo_matrix((g[f"{prefix}_val"], (g[f"{prefix}_row"], g[f"{prefix}_col"])), shape=shape)


# ------------------------------------------------------------------------------------------ masks
def test_masks_match_reference(golden):
    g = golden("masks")
    for i in range(int(g["n_sym"])):
        n = int(g[f"sym{i}_n"])
        ks = tuple(int(x) for x in g[f"sym{i}_kshape"])
        md = int(g[f"sym{i}_max_dist"])
        md = None if md < 0 else md
        valid = g[f"sym{i}_valid"]
        mask = cup.make_missing_mask((n, n), valid, valid, max_dist=md, sym_upper=True)
        ref = np.unpackbits(g[f"sym{i}_mask"])[:n * n].reshape(n, n).astype(bool)
        assert mask.dtype == bool and np.array_equal(mask.toarray(), ref), i
        framed = cup.frame_missing_mask(mask, ks, sym_upper=True, max_dist=md)
        H, W = n + 2 * (ks[0] - 1), n + 2 * (ks[1] - 1)
        ref = np.unpackbits(g[f"sym{i}_framed"])[:H * W].reshape(H, W).astype(bool)
        assert np.array_equal(framed.toarray(), ref), i
    for i in range(int(g["n_inter"])):
        shape = tuple(int(x) for x in g[f"inter{i}_shape"])
        ks = tuple(int(x) for x in g[f"inter{i}_kshape"])
        mask = cup.make_missing_mask(shape, g[f"inter{i}_valid_rows"], g[f"inter{i}_valid_cols"])
        ref = np.unpackbits(g[f"inter{i}_mask"])[:shape[0] * shape[1]].reshape(shape).astype(bool)
        assert np.array_equal(mask.toarray(), ref)
        framed = cup.frame_missing_mask(mask, ks, sym_upper=False, max_dist=None)
        H, W = shape[0] + 2 * (ks[0] - 1), shape[1] + 2 * (ks[1] - 1)
        ref = np.unpackbits(g[f"inter{i}_framed"])[:H * W].reshape(H, W).astype(bool)
        assert np.array_equal(framed.toarray(), ref)


def test_make_missing_mask_reference_example():
    """Same expectations as the reference's test (tests/test_preprocessing.py:49-97)."""
    missing = np.array([0, 4, 9])
    valid = np.array([i for i in range(10) if i not in missing])
    valid_cols = np.array([i for i in range(15) if i not in missing])
    exp_sym = np.zeros((10, 10), dtype=bool)
    exp_sym[:, missing] = True
    exp_sym[missing, :] = True
    exp_asym = np.zeros((10, 15), dtype=bool)
    exp_asym[:, missing] = True
    exp_asym[missing, :] = True
    assert np.array_equal(cup.make_missing_mask((10, 10), valid, valid, sym_upper=False).toarray(), exp_sym)
    assert np.array_equal(cup.make_missing_mask((10, 10), valid, valid, sym_upper=True).toarray(), np.triu(exp_sym))
    assert np.array_equal(cup.make_missing_mask((10, 15), valid, valid_cols).toarray(), exp_asym)
    with pytest.raises(ValueError):
        cup.make_missing_mask((10, 15), valid, valid, sym_upper=True)
    trimmed = cup.diag_trim(np.triu(exp_sym), 3 + 1)
    got = cup.make_missing_mask((10, 10), valid, valid, sym_upper=True, max_dist=3)
    assert np.array_equal(got.toarray(), trimmed)


def test_check_missing_mask():
    sig = sp.csr_matrix(np.array([[0.0, 1.0], [2.0, 0.0]]))
    ok = sp.csr_matrix(np.array([[True, False], [False, True]]))
    cup.check_missing_mask(sig, ok)
    with pytest.raises(ValueError):
        cup.check_missing_mask(sig, sp.csr_matrix(np.array([[True, True], [False, False]])))
    with pytest.raises(ValueError):
        cup.check_missing_mask(sig.toarray(), np.array([[1, 1], [0, 0]]))


def test_diag_trim_and_pad():
    rng = np.random.default_rng(0)
    m = sp.csr_matrix(rng.random((30, 30)))
    for d in (0, 1, 5, 29):
        t = cup.diag_trim(m, d)
        assert t.shape == m.shape
        dense = t.toarray()
        ii, jj = np.indices(dense.shape)
        assert np.all(dense[(jj - ii < 0) | (jj - ii > d)] == 0)
        assert np.array_equal(dense[(jj - ii >= 0) & (jj - ii <= d)], m.toarray()[(jj - ii >= 0) & (jj - ii <= d)])
    with pytest.raises(ValueError):
        cup.diag_trim(m.tocoo(), 3)
    base = sp.coo_matrix(np.ones((10, 10)))
    for hpad in range(4):
        for vpad in range(4):
            padded = cup.zero_pad_sparse(base, margin_h=hpad, margin_v=vpad)
            assert padded.shape == (10 + 2 * vpad, 10 + 2 * hpad)
            assert np.all(padded.toarray()[vpad:padded.shape[0] - vpad, hpad:padded.shape[1] - hpad] == 1)
            assert padded.sum() == 100


# ------------------------------------------------------------------------------------------ foci
def test_label_foci_reference_example(golden):
    g = golden("nms")
    nf, lab = cud.label_foci(sp.coo_matrix(g["spec_in"]))
    assert nf == int(g["spec_num"])
    assert np.array_equal(lab.toarray(), g["spec_labels"])
    n2, _ = cud.filter_foci(sp.coo_matrix(lab.toarray()), min_size=2)
    n3, _ = cud.filter_foci(sp.coo_matrix(lab.toarray()), min_size=3)
    assert (n2, n3) == (int(g["spec_num_min2"]), int(g["spec_num_min3"]))


def test_pick_foci_matches_reference(golden):
    g = golden("nms")
    tags = sorted({k[:-len("_pearson")] for k in g if k.endswith("_pearson")})
    assert len(tags) == 15
    n_with_foci = 0
    for tag in tags:
        conv = coo(g, f"{tag}_conv")
        foci, lab = cud.pick_foci(conv, float(g[f"{tag}_pearson"]))
        ref = g[f"{tag}_foci"]
        if ref.shape[0] == 0:
            assert foci is None
            continue
        n_with_foci += 1
        assert np.array_equal(foci, ref), tag          # same coordinates, same order
        assert np.array_equal(lab.toarray(), coo(g, f"{tag}_labels").toarray()), tag
    assert n_with_foci >= 10


def test_pick_foci_speckles_and_index():
    point = np.zeros((10, 10))
    point[5, 5] = point[2, 2] = 0.3
    coords, _ = cud.pick_foci(sp.coo_matrix(point), 0.1)
    assert coords is None                               # isolated pixels are dropped
    m = np.zeros((12, 12))
    m[3, 4] = 0.5
    m[3, 5] = 0.9
    m[4, 5] = 0.9                                       # tie: first in row-major order wins
    coords, _ = cud.pick_foci(sp.coo_matrix(m), 0.4)
    assert coords.tolist() == [[3, 5]]


def test_remove_neighbours_matches_reference(golden):
    g = golden("nms")
    pts = pd.DataFrame(g["rn_patterns"], columns=["bin1", "bin2", "score"])
    assert np.array_equal(cud.remove_neighbours(pts, win_size=5), g["rn_keep_w5"])
    assert np.array_equal(cud.remove_neighbours(pts, win_size=1), g["rn_keep_w1"])


def test_validate_patterns_rules():
    rng = np.random.default_rng(4)
    mat = sp.csr_matrix(np.triu(rng.random((40, 40)) + 0.1))
    conv = sp.csr_matrix(np.triu(rng.random((40, 40))))
    det = (np.ones(40, dtype=bool), np.ones(40, dtype=bool))
    kernel = np.ones((7, 7))
    coords = np.array([[20, 25], [2, 30], [36, 39], [10, 10]])
    tab, wins = cud.validate_patterns(coords, mat, conv, det, kernel, zero_tol=0.9, missing_tol=0.5)
    # windows leaving the matrix are dropped (strict bound: low < shape)
    assert tab[["bin1", "bin2"]].to_numpy().tolist() == [[20, 25], [10, 10]]
    assert np.allclose(tab["score"].to_numpy(), [conv[20, 25], conv[10, 10]])
    assert np.array_equal(wins[0], mat[17:24, 22:29].toarray())
    tab, wins = cud.validate_patterns(coords, mat, conv, det, kernel, drop=False, zero_tol=0.9, missing_tol=0.5)
    assert len(tab) == 4 and np.isnan(tab["score"].to_numpy()[[1, 2]]).all()
    # missing bins -> NaN rows/cols, rejected above missing_tol
    det2 = (np.setdiff1d(np.arange(40), [19, 20, 21, 22]), np.arange(40))
    tab, _ = cud.validate_patterns(coords[:1], mat, conv, det2, kernel, zero_tol=0.9, missing_tol=0.5)
    assert len(tab) == 0
    # window made only of missing bins: 0/0 proportion of zeros -> rejected, no exception
    det3 = (np.setdiff1d(np.arange(40), np.arange(14, 28)), np.setdiff1d(np.arange(40), np.arange(19, 33)))
    tab, _ = cud.validate_patterns(coords[:1], mat, conv, det3, kernel, drop=False, zero_tol=0.9, missing_tol=1.5)
    assert np.isnan(tab["score"].to_numpy()).all()
    assert np.isnan(cud.pileup_patterns(np.full((2, 3, 3), np.nan))).all()


# ------------------------------------------------------------------------------------------ templates / stats
def test_factorise_kernel(golden, templates):
    g = golden("xcorr2")
    u, v = cup.factorise_kernel(templates["loops"].copy(), prop_info=0.999)
    assert u.shape == g["loops_tsvd999_u"].shape == (17, 2)
    assert np.allclose(u @ v, g["loops_tsvd999_u"] @ g["loops_tsvd999_v"], atol=1e-12)


def test_resize_and_crop_kernel():
    m = 15
    point = np.zeros((m, m))
    point[m // 2, m // 2] = 10
    for kernel_res in (3, 4, 6, 10):
        for signal_res in (3, 4, 6, 10):
            exp = int(m * kernel_res / signal_res)
            exp -= 0 if exp % 2 else 1
            a = cup.resize_kernel(point, kernel_res=kernel_res, signal_res=signal_res, min_size=5, quiet=True)
            b = cup.resize_kernel(point, factor=kernel_res / signal_res, min_size=5, quiet=True)
            assert a.shape == b.shape == (max(exp, 5),) * 2
            assert a.max() == a[a.shape[0] // 2, a.shape[0] // 2]
    for targ in range(20):
        exp = targ if targ % 2 else targ + 1
        assert cup.crop_kernel(point, (targ, targ)).shape[0] == min(exp, m)
    with pytest.raises(ValueError):
        cup.resize_kernel(np.ones((4, 4)), factor=1)


def test_isotonic_smoothing():
    assert np.allclose(cup._isotonic_non_increasing([3.0, 3.5, 4.0]), [3.5, 3.5, 3.5])
    assert np.allclose(cup._isotonic_non_increasing([5, 4, 4.5, 1]), [5, 4.25, 4.25, 1])
    sk = pytest.importorskip("sklearn.isotonic")
    rng = np.random.default_rng(1)
    y = np.sort(rng.random(200))[::-1] + rng.normal(0, 0.05, 200)
    ref = sk.IsotonicRegression(increasing=False).fit_transform(range(200), y)
    assert np.allclose(cup._isotonic_non_increasing(y), ref, atol=1e-12)


def test_stats(golden):
    g = golden("stats")
    assert np.allclose(cus.fdr_correction(g["pvals"]), g["qvals"], rtol=0, atol=1e-15)
    # Benjamini-Hochberg known answer of the reference's test (tests/test_stats.py:5-12)
    assert np.allclose(cus.fdr_correction(np.array([0.01, 0.02, 0.03, 0.5])), [0.04, 0.04, 0.04, 0.5])
    lp = cus.corr_to_pval(g["r"], g["n"].astype(float))
    fin = np.isfinite(g["logp_vec"])
    assert np.allclose(lp[fin], g["logp_vec"][fin], rtol=0, atol=1e-12)
    assert np.allclose(cus.corr_to_pval(g["r"], 289)[fin], g["logp_289"][fin], rtol=0, atol=1e-12)
    with pytest.raises(ValueError):
        cus.corr_to_pval(np.zeros(3), np.zeros(4))
    assert cus.fdr_correction(None) is None


def test_kernels_module():
    import chromosight_amd.kernels as ck
    assert set(ck.kernel_names) >= {"loops", "borders", "hairpins", "loops_small", "centromeres"}
    assert ck.loops["kernels"][0].shape == (17, 17) and len(ck.borders["kernels"]) == 3
    assert ck.hairpins["kernels"][0].shape == (15, 15) and ck.loops["pearson"] == 0.3
    assert ck.borders["max_dist"] == 0 and ck.loops["max_perc_undetected"] == 50.0


def test_mask_as_bins_recognises_only_structured_masks():
    n = 300
    miss = np.zeros(n, bool)
    miss[[3, 50, 51, 299]] = True
    valid = np.flatnonzero(~miss)
    m = cup.make_missing_mask((n, n), valid, valid, max_dist=40, sym_upper=True)
    got = cud._mask_as_bins(m, True, 40)
    assert got is not None and np.array_equal(got[0], miss) and np.array_equal(got[1], miss)
    assert cud._mask_as_bins(m, True, 39) is None                 # another max_dist: another mask
    odd = m.tolil()
    odd[10, 12] = True
    assert cud._mask_as_bins(odd.tocsr(), True, 40) is None
    mr = np.zeros(200, bool)
    mc = np.zeros(310, bool)
    mr[[1, 7]] = True
    mc[[0, 100, 309]] = True
    mi = cup.make_missing_mask((200, 310), np.flatnonzero(~mr), np.flatnonzero(~mc), sym_upper=False)
    got = cud._mask_as_bins(mi, False, None)
    assert got is not None and np.array_equal(got[0], mr) and np.array_equal(got[1], mc)
    empty = sp.csr_matrix((50, 50), dtype=bool)
    got = cud._mask_as_bins(empty, True, 10)
    assert got is not None and not got[0].any()


def test_validate_on_virtual_padded_map_equals_materialised():
    """_validate(pad=, stripe_k=) must see exactly the map the reference builds with
    zero_pad_sparse + NaN sub-diagonals (detection.py:287-310)."""
    rng = np.random.default_rng(4)
    n, k = 120, 9
    kh = (k - 1) // 2
    a = np.triu(rng.random((n, n)) * (rng.random((n, n)) > 0.5))
    csr = sp.csr_matrix(a)
    coords = np.column_stack([rng.integers(0, n, 60), rng.integers(0, n, 60)])
    coords[:5] = [[0, 0], [n - 1, n - 1], [2, n - 1], [kh, kh], [50, 50]]
    scores = rng.random(60)
    miss = np.array([7, 8, 60])
    padded = cup.zero_pad_sparse(csr, kh, kh, fmt="csr").astype(np.float64)
    stripes = sp.diags([np.full(padded.shape[0], np.nan)] * k, -np.arange(1, k + 1), shape=padded.shape, format="csr")
    ref_tab, ref_win = cud._validate(coords + kh, padded + stripes, scores, miss + kh, miss + kh, (k, k), drop=False,
                                     zero_tol=0.6, missing_tol=0.5)
    tab, win = cud._validate(coords + kh, csr, scores, miss + kh, miss + kh, (k, k), drop=False, zero_tol=0.6,
                             missing_tol=0.5, pad=(kh, kh), stripe_k=k)
    np.testing.assert_array_equal(np.isnan(tab.score), np.isnan(ref_tab.score))
    np.testing.assert_allclose(tab.score.dropna(), ref_tab.score.dropna())
    np.testing.assert_array_equal(np.isnan(win), np.isnan(ref_win))
    np.testing.assert_allclose(np.nan_to_num(win), np.nan_to_num(ref_win))
    assert np.isfinite(tab.score).any() and np.isnan(tab.score).any()


def test_upper_band_block_equals_symmetric_block(golden):
    """pipeline.balanced_upper_band (built straight from the pixel table) = upper band of the
    symmetric balanced block cooler would return."""
    from chromosight_amd import pipeline
    cool = dict(golden("example_cool"))
    for ci in range(len(cool["chrom_offset"]) - 1):
        full = pipeline.balanced_intra_block(cool, ci).tocsr()
        for keep in (5, 40, 10_000):
            band = pipeline.balanced_upper_band(cool, ci, keep).toarray()
            want = full.toarray()
            ii, jj = np.indices(want.shape)
            want[(jj - ii < 0) | (jj - ii > keep)] = 0
            np.testing.assert_array_equal(np.isnan(band), np.isnan(want))
            np.testing.assert_allclose(np.nan_to_num(band), np.nan_to_num(want), rtol=0, atol=0)


def test_remove_neighbours_grid_equals_quadratic_scan():
    """cs_remove_neighbours (grid of win x win cells) against the reference's O(n^2) loop
    (detection.py:348-384) restated here, on random clouds with ties in position."""
    rng = np.random.default_rng(12)
    for n, span, win in ((0, 10, 3), (1, 10, 3), (300, 40, 5), (2000, 300, 8), (500, 30, 1), (800, 25, 12)):
        pts = pd.DataFrame({"bin1": rng.integers(0, span, n), "bin2": rng.integers(0, span, n), "score": rng.random(n)})
        got = cud.remove_neighbours(pts, win_size=win)
        ordered = pts.sort_values("score", ascending=False)
        black = set()
        for i, p in ordered.iterrows():
            if i in black:
                continue
            close = np.flatnonzero((np.abs(ordered.bin1 - p.bin1) < win) & (np.abs(ordered.bin2 - p.bin2) < win))
            for idx in ordered.index.values[close]:
                if idx != i:
                    black.add(idx)
        want = np.ones(n, dtype=bool)
        want[list(black)] = False
        assert np.array_equal(got, want), (n, span, win)


def test_sub_matrix_order_and_bins():
    from chromosight_amd import pipeline

    class G:
        n_chrom = 3
    assert pipeline.sub_matrices(G, False) == [(0, 0), (1, 1), (2, 2)]
    assert pipeline.sub_matrices(G, True) == [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]   # contacts_map.py:274-279


def test_load_cool_decodes_the_reference_fixture(golden, monkeypatch):
    """chromosight_amd.io.load_cool on data_test/example.cool (the reference's own test file, kept as a data fixture)
    == the decoded arrays every pipeline test uses: through the package's own HDF5 reader (hdf5_lite: no external tool,
    never skipped) and, where the HDF5 command line tool exists, through the h5dump route as well."""
    from chromosight_amd import io as cio
    from conftest import GOLDEN
    ref = golden("example_cool")
    routes = [None]
    try:
        cio.find_h5dump()
        routes.append("1")
    except RuntimeError:
        pass
    for only in routes:
        if only:
            monkeypatch.setenv("CHROMOSIGHT_H5DUMP_ONLY", only)
        cool = cio.load_cool(GOLDEN / "example.cool")
        for key in ("bin1_id", "bin2_id", "count", "bin_start", "bin_end", "chrom_offset", "binsize"):
            assert np.array_equal(np.asarray(cool[key]).astype(np.int64), np.asarray(ref[key]).astype(np.int64)), (only, key)
        assert np.array_equal(np.isnan(cool["weight"]), np.isnan(ref["weight"]))
        assert np.allclose(cool["weight"], ref["weight"], equal_nan=True, rtol=0, atol=0)
        assert [str(n) for n in cool["chrom_names"]] == [str(n) for n in ref["chrom_names"]]


def test_hdf5_lite_reads_what_a_cool_holds(golden):
    """The reader on the fixture: group listing, chunked + shuffled + deflated columns, the enumerated chromosome column,
    fixed-length strings, numeric attributes; load_cool's rules for the weight column (refused when absent unless
    norm='raw'; read errors are not swallowed)."""
    from chromosight_amd import hdf5_lite
    from chromosight_amd import io as cio
    from conftest import GOLDEN
    f = hdf5_lite.File(GOLDEN / "example.cool")
    assert sorted(f.links(f.root)) == ["bins", "chroms", "indexes", "pixels"]
    assert f.exists("bins/weight") and not f.exists("bins/KR")
    attrs = f.attrs("/")
    assert int(attrs["bin-size"]) == 1000 and int(attrs["nbins"]) == 720 and int(attrs["nnz"]) == 109975
    chrom = f.dataset("bins/chrom")
    assert f.enum_names("bins/chrom") == {0: "chr1", 1: "chr2", 2: "chr3"}
    off = golden("example_cool")["chrom_offset"]
    assert np.array_equal(chrom, np.repeat(np.arange(3), np.diff(off)))
    assert [n.decode() for n in f.dataset("chroms/name")] == ["chr1", "chr2", "chr3"]
    assert int(f.dataset("pixels/count").sum()) == int(attrs["sum"])
    with pytest.raises(ValueError, match="no balancing weights"):
        cio.load_cool(GOLDEN / "example.cool", balance="KR")
    raw = cio.load_cool(GOLDEN / "example.cool", balance="KR", norm="raw")
    assert np.all(raw["weight"] == 1.0)
    keep = cio.load_cool(GOLDEN / "example.cool", norm="raw")            # detectable bins of the stored weights, raw counts
    assert np.array_equal(np.isnan(keep["weight"]), np.isnan(golden("example_cool")["weight"]))
    with pytest.raises(hdf5_lite.Hdf5Unsupported):
        hdf5_lite.File(GOLDEN / "example_loops.tsv")


def test_distance_law_other_reducers_on_host():
    """fun other than (nan)mean is evaluated diagonal by diagonal on the host (no device needed)."""
    rng = np.random.default_rng(3)
    m = np.triu(rng.gamma(2, 1, size=(30, 30))) * (rng.random((30, 30)) > 0.3)
    det = np.setdiff1d(np.arange(30), [4, 17])
    got = cup.distance_law(sp.csr_matrix(m), detectable_bins=det, max_dist=12, smooth=False, fun=np.nanmedian)
    flag = np.zeros(30, bool)
    flag[det] = True
    for d in range(13):
        diag = np.diagonal(m, d)[flag[:30 - d] & flag[d:]]
        v = diag[diag > 0]
        assert (np.isnan(got[d]) and v.size == 0) or got[d] == np.median(v)
    assert np.all(got[13:] == 0)


def test_writers_and_bed2d(tmp_path):
    """chromosight_amd.io output side (reference io.py:208-326): formats and the bed2d conventions."""
    import json
    import pandas as pd
    from chromosight_amd import io as cio
    table = pd.DataFrame({"chrom1": ["chr1", "chr2"], "start1": [1000, 5000], "end1": [2000, 6000],
                          "chrom2": ["chr1", "chr2"], "start2": [34000, 3000], "end2": [35000, 4000],
                          "bin1": [1, 5], "bin2": [34, 3], "kernel_id": [0, 0], "iteration": [0, 0],
                          "score": [0.48979314123, 1 / 3], "pvalue": [1e-10, 0.5], "qvalue": [1.2e-10, 0.5]})
    cio.write_patterns(table, str(tmp_path / "p"))
    lines = (tmp_path / "p.tsv").read_text().splitlines()
    assert lines[0].split("\t") == list(table.columns)
    assert lines[1].split("\t")[-3:] == ["0.4897931412", "0.0000000001", "0.0000000001"]
    assert lines[2].split("\t")[-3:] == ["0.3333333333", "0.5000000000", "0.5000000000"]
    cio.write_patterns(table, str(tmp_path / "q"), dec=3)
    assert (tmp_path / "q.tsv").read_text().splitlines()[2].endswith("0.333\t0.500\t0.500")
    # header detected; the second pair is stored right anchor first and comes back oriented
    bed = cio.load_bed2d(str(tmp_path / "p.tsv"))
    assert list(bed.columns) == cio.BED2D_COLUMNS
    assert bed.iloc[1].tolist() == ["chr2", 3000, 4000, "chr2", 5000, 6000]
    # no header, numeric chromosome names become strings, inter-chromosomal pairs are left alone
    (tmp_path / "n.bed2d").write_text("1\t100\t200\t1\t50\t60\n1\t900\t1000\t2\t10\t20\n")
    bed = cio.load_bed2d(str(tmp_path / "n.bed2d"))
    assert bed.iloc[0].tolist() == ["1", 50, 60, "1", 100, 200]
    assert bed.iloc[1].tolist() == ["1", 900, 1000, "2", 10, 20]
    wins = np.arange(8.0).reshape(2, 2, 2)
    cio.save_windows(wins, str(tmp_path / "w"))
    assert json.loads((tmp_path / "w.json").read_text()) == {"0": [[0.0, 1.0], [2.0, 3.0]], "1": [[4.0, 5.0], [6.0, 7.0]]}
    with pytest.raises(ValueError):
        cio.save_windows(wins, str(tmp_path / "w"), fmt="csv")
    with pytest.raises(OSError):
        cio.check_prefix_dir(str(tmp_path / "missing_dir" / "prefix"))
    cio.check_prefix_dir(str(tmp_path / "prefix"))


def test_native_acceptance_rules_equal_the_numpy_ones():
    """cs_accept_records (host C++, what the genome drivers call) against _accept_records (numpy restatement of reference
    detection.py:121-141, 269-270, 332-336 + stats.py:43-81) on random records with every edge the rules have."""
    import types
    from chromosight_amd._lib import FOCUS_DTYPE
    from chromosight_amd.utils import detection as cid
    rng = np.random.default_rng(5)
    kspec = types.SimpleNamespace(km=17, kn=17)
    cfg = {"max_perc_undetected": 50.0, "max_perc_zero": 10.0}
    shapes, max_dists, counts = [(900, 900), (400, 400), (50, 50), (700, 700)], [300, 80, None, 5], np.array([2000, 700, 0, 300])
    n = int(counts.sum())
    rec = np.zeros(n, FOCUS_DTYPE)
    rec["bin1"] = rng.integers(-3, 905, n)
    rec["bin2"] = rec["bin1"] + rng.integers(-2, 320, n)
    rec["inside"] = rng.random(n) < 0.95
    rec["n_missing"] = rng.integers(0, 290, n)
    rec["n_missing"][rng.random(n) < 0.02] = 289
    rec["n_zero"] = rng.integers(0, 60, n)
    rec["score"] = rng.uniform(-1, 1, n)
    rec["score"][:8] = [0.0, 1.0, -1.0, np.nan, 1e-300, 0.999999999, -0.0, 1.5]
    rec["n_obs"] = 289 - rec["n_missing"]
    rec["n_obs"][8:12] = [0, 2, 3, 4]
    per = lambda v: np.repeat(np.asarray(v, dtype=np.int64), counts)
    for full in (True, False):
        for compact in (True, False):
            got, ok, kept = cid.accept_native(rec, counts, shapes, max_dists, kspec, cfg, inter=False, full=full, compact=compact)
            rr, cc = rec["bin1"].astype(np.int64), rec["bin2"].astype(np.int64)
            md = per([10 ** 9 if m is None else m for m in max_dists])
            want, _, want_ok = cid._accept_records(rec, None, rr, cc, "detect" if compact else "quantify",
                                                   (per([s[0] for s in shapes]), per([s[1] for s in shapes])), kspec, cfg, inter=False,
                                                   max_dist=md, full=full, raw=True, return_ok=True)
            assert np.array_equal(ok, want_ok)
            assert np.array_equal(kept, [want_ok[a:b].sum() for a, b in zip(np.cumsum(counts) - counts, np.cumsum(counts))])
            assert got.shape == want.shape
            assert np.array_equal(got[:, :3], want[:, :3], equal_nan=True)          # coordinates and scores: the same bits
            assert np.allclose(got[:, 3], want[:, 3], rtol=1e-12, atol=1e-300, equal_nan=True)   # libm vs scipy's erfc
    got, ok, kept = cid.accept_native(rec[:0], np.zeros(0, np.int64), [], [], kspec, cfg, inter=False, full=True, compact=True)
    assert got.shape == (0, 4) and ok.size == 0 and kept.size == 0


def test_native_acceptance_single_pass_equals_the_two_pass_form(monkeypatch):
    """Records that carry their p-values (cs_focus.pval, flags bit 1) with the accepted rows packed take ONE pass on the calling
    thread: the same table, mask and counts as the two-pass form given the same p-values (the unpacked call's rows of the accepted
    records), on lists with empty blocks, rejected records at both ends and out-of-band pixels."""
    import types
    from chromosight_amd._lib import FOCUS_DTYPE
    from chromosight_amd.utils import detection as cid
    monkeypatch.delenv("CHROMOSIGHT_HIP_HOST_PVALUES", raising=False)
    rng = np.random.default_rng(23)
    kspec = types.SimpleNamespace(km=17, kn=17)
    cfg = {"max_perc_undetected": 50.0, "max_perc_zero": 10.0}
    counts = np.array([0, 5000, 1, 0, 12000, 300, 0])
    shapes, max_dists = [(3000, 3000)] * 7, [700, 700, 700, None, 20, 700, 700]
    n = int(counts.sum())
    rec = np.zeros(n, FOCUS_DTYPE)
    rec["bin1"] = rng.integers(-2, 3003, n)
    rec["bin2"] = rec["bin1"] + rng.integers(-3, 760, n)
    rec["inside"] = rng.random(n) < 0.9
    rec["inside"][[0, n - 1]] = 0
    rec["n_missing"] = rng.integers(0, 290, n)
    rec["n_zero"] = rng.integers(0, 60, n)
    rec["score"] = rng.uniform(-1, 1, n)
    rec["n_obs"] = 289 - rec["n_missing"]
    rec["pval"] = rng.uniform(0, 1, n)
    # the limits of the two rules exactly: 20 / 200 = 0.1 is not < 0.1, 144 / 289 < 0.5 <= 145 / 289, an all-missing window (0 / 0)
    for k, (nm, nz) in enumerate([(89, 20), (89, 19), (144, 0), (145, 0), (289, 0), (0, 28), (0, 29), (288, 0), (288, 1)]):
        rec["n_missing"][10 + k], rec["n_zero"][10 + k], rec["inside"][10 + k] = nm, nz, 1
    for full in (True, False):
        loose, ok_l, kept_l = cid.accept_native(rec, counts, shapes, max_dists, kspec, cfg, inter=False, full=full, compact=False, pvals=True)
        packed, ok_p, kept_p = cid.accept_native(rec, counts, shapes, max_dists, kspec, cfg, inter=False, full=full, compact=True, pvals=True)
        assert 0.05 < ok_l.mean() < 0.9 and np.array_equal(ok_l, ok_p) and np.array_equal(kept_l, kept_p)
        assert np.array_equal(packed, loose[ok_l])
        assert np.array_equal(packed[:, 3], rec["pval"][ok_l])                 # copied, not recomputed
        assert list(ok_p[10:19]) == [False, True, True, False, False, True, False, False, False]
    got, ok, kept = cid.accept_native(rec[:0], np.zeros(0, np.int64), [], [], kspec, cfg, inter=False, full=True, compact=True, pvals=True)
    assert got.shape == (0, 4) and ok.size == 0 and kept.size == 0


def test_native_acceptance_rules_pool_and_concurrent_callers():
    """cs_accept_records on lists long enough for the library's worker pool (pieces of 1024 records taken from a counter by
    the workers and the caller), from several threads at once (a caller that finds the pool busy runs its pieces itself):
    every call gives the table of the single-piece calls, many times over."""
    import threading
    import types
    from chromosight_amd._lib import FOCUS_DTYPE
    from chromosight_amd.utils import detection as cid
    rng = np.random.default_rng(11)
    kspec = types.SimpleNamespace(km=17, kn=17)
    cfg = {"max_perc_undetected": 50.0, "max_perc_zero": 10.0}
    counts = np.array([9000, 0, 30000, 1500, 700, 12000])
    shapes, max_dists = [(5000, 5000)] * 6, [1000] * 6
    n = int(counts.sum())
    rec = np.zeros(n, FOCUS_DTYPE)
    rec["bin1"] = rng.integers(0, 5000, n)
    rec["bin2"] = rec["bin1"] + rng.integers(0, 1100, n)
    rec["inside"] = rng.random(n) < 0.95
    rec["n_missing"] = rng.integers(0, 200, n)
    rec["n_zero"] = rng.integers(0, 40, n)
    rec["score"] = rng.uniform(-1, 1, n)
    rec["n_obs"] = 289 - rec["n_missing"]
    # reference: block by block, each below the pool's threshold in pieces of 200 records
    want_rows, want_kept = [], []
    at = 0
    for c in counts:
        kept_b = 0
        for o in range(0, int(c), 200):
            m = min(200, int(c) - o)
            g, _, k = cid.accept_native(rec[at + o:at + o + m], np.array([m]), shapes[:1], max_dists[:1], kspec, cfg, inter=False,
                                        full=True, compact=True)
            want_rows.append(g)
            kept_b += int(k[0])
        want_kept.append(kept_b)
        at += int(c)
    want = np.concatenate(want_rows)
    results, errors = {}, []

    def worker(i):
        try:
            for rep in range(6):
                g, _, k = cid.accept_native(rec, counts, shapes, max_dists, kspec, cfg, inter=False, full=True, compact=True)
                if not (np.array_equal(g, want, equal_nan=True) and np.array_equal(k, want_kept)):
                    errors.append((i, rep))
            results[i] = True
        except Exception as exc:                                 # noqa: BLE001
            errors.append((i, repr(exc)))
    threads = [threading.Thread(target=worker, args=(i,)) for i in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors and len(results) == 3, errors


def test_map_pitch_never_lands_on_a_power_of_two_stride():
    """engine.map_pitch: the row pitch the engine gives device-resident dense maps (bench.py C2)."""
    from chromosight_amd import engine
    assert engine.map_pitch(4096, 4) == 4160 and engine.map_pitch(4096, 8) == 4160
    assert engine.map_pitch(4000, 4) == 4000 and engine.map_pitch(4001, 4) == 4016
    assert engine.map_pitch(512, 4) == 576 and engine.map_pitch(500, 4) == 576 and engine.map_pitch(500, 8) == 576
    for width in (1, 17, 255, 256, 1000, 1024, 8192, 16384, 50_000):
        for itemsize in (4, 8):
            ld = engine.map_pitch(width, itemsize)
            assert ld >= width and ld % 16 == 0 and (ld * itemsize) % 2048 != 0 and ld - width < 16 + 64


def test_accept_records_rejects_bad_arguments():
    import types
    from chromosight_amd._lib import FOCUS_DTYPE, load_library
    lib = load_library()
    rec = np.zeros(4, FOCUS_DTYPE)
    counts = np.array([5], dtype=np.int64)            # more records than the buffer holds is the caller's business; negative is not
    geo = np.array([10, 10, 3], dtype=np.int32)
    table, ok, kept = np.zeros((4, 4)), np.zeros(4, np.uint8), np.zeros(1, np.int64)
    args = lambda cnt, km: (rec.ctypes.data, 1, cnt.ctypes.data, geo[0:1].ctypes.data, geo[1:2].ctypes.data, geo[2:3].ctypes.data, 0, km, 17,
                            0.5, 0.1, 1, 1, table.ctypes.data, ok.ctypes.data, kept.ctypes.data)
    assert lib.cs_accept_records(*args(np.array([-1], dtype=np.int64), 17)) != 0
    assert lib.cs_accept_records(*args(np.array([4], dtype=np.int64), 0)) != 0
    assert lib.cs_accept_records(*args(np.array([4], dtype=np.int64), 17)) == 0


def test_resize_and_crop_kernel_match_the_reference(golden):
    """resize_kernel / crop_kernel against the reference's outputs for the built-in templates (tests/golden/resize.npz:
    8 factors, 4 resolution pairs, 5 crop targets each): what --win-size and another resolution feed the hot path."""
    from chromosight_amd.utils import preprocessing as cup
    import chromosight_amd.kernels as ck
    g = golden("resize")
    kernels = {"loops": ck.loops["kernels"][0], "borders0": ck.borders["kernels"][0], "borders2": ck.borders["kernels"][2],
               "hairpin": ck.hairpins["kernels"][0]}
    checked = 0
    for name, kern in kernels.items():
        kern = np.asarray(kern, dtype=np.float64)
        for key in [k for k in g if k.startswith(name + "_") and not k.endswith("_value")]:
            value = g[key + "_value"]
            if "_factor" in key:
                got = cup.resize_kernel(kern, factor=float(value), quiet=True)
            elif "_res" in key:
                got = cup.resize_kernel(kern, kernel_res=int(value[0]), signal_res=int(value[1]), quiet=True)
            else:
                got = cup.crop_kernel(kern, (int(value[0]), int(value[1])))
            assert got.shape == g[key].shape, (key, got.shape, g[key].shape)
            assert np.abs(got - g[key]).max() < 1e-12, key
            checked += 1
    assert checked == 4 * 17


def test_win_size_option_resizes_every_template():
    """--win-size (cli/chromosight.py:365-370, 689-695): every template resized to win_size / its own size, even sizes
    rejected, "auto" / None leaves the config alone; the caller's config is not modified."""
    import chromosight_amd.kernels as ck
    from chromosight_amd import pipeline
    cfg = dict(ck.borders)
    before = [np.array(k) for k in cfg["kernels"]]
    assert pipeline.with_win_size(cfg, None) is cfg and pipeline.with_win_size(cfg, "auto") is cfg
    for win in (9, "11", 21):
        out = pipeline.with_win_size(cfg, win)
        assert [np.shape(k) for k in out["kernels"]] == [(int(win), int(win))] * len(before)
        for k, src in zip(out["kernels"], before):
            assert np.array_equal(k, cup.resize_kernel(src.astype(np.float64), factor=int(win) / src.shape[0], quiet=True))
    assert all(np.array_equal(a, b) for a, b in zip(before, cfg["kernels"]))
    with pytest.raises(ValueError):
        pipeline.with_win_size(cfg, 10)
