"""Round-3 parity pins against captures of the reference itself (tests/golden/make_golden_r3.py, run where
/root/reference exists): iterated template refinement, non-square templates in full mode, `detect` on the
17-chromosome yeast map in float32 mode, and the selection block of `cmd_quantify`."""
import copy

import numpy as np
import pandas as pd
import pytest
import scipy.sparse as sp

import chromosight_amd
import chromosight_amd.kernels as ck
from chromosight_amd import parallel, pipeline
from chromosight_amd.utils import detection as cud

pytestmark = pytest.mark.gpu

FINAL_COLS = ["bin1", "bin2", "kernel_id", "iteration", "score", "pvalue", "qvalue"]


@pytest.fixture(params=["f32", "f64"])
def precision(request):
    old = chromosight_amd.get_precision()
    chromosight_amd.set_precision(request.param)
    yield request.param
    chromosight_amd.set_precision(old)


def assert_final(table, ref, what):
    """Output table against the reference's: coordinates, template and iteration columns bit-exact and in the
    reference's order; scores 1e-9; p / q-values relative 1e-6."""
    got = table[FINAL_COLS].to_numpy(dtype=np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert np.array_equal(got[:, :4], ref[:, :4]), what
    assert np.abs(got[:, 4] - ref[:, 4]).max() < 1e-9, what
    assert np.allclose(got[:, 5:], ref[:, 5:], rtol=1e-6, atol=1e-300), what


# ------------------------------------------------------------------------------------------------
# (a) iterations: the template of iteration i + 1 is the pileup of ALL sub-matrices' windows
#     (cli/chromosight.py:731-791, detection.py:158-174)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag,overrides", [
    ("loops", dict(max_iterations=2)),
    ("loops3", dict(pearson=0.25, max_iterations=3, max_dist=100000, min_dist=5000)),
])
def test_iterated_template_matches_reference(golden, tag, overrides):
    cool = golden("example_cool")
    g = golden("iterations")
    cfg = copy.deepcopy(ck.loops)
    cfg.update(overrides)
    table, windows = pipeline.detect(cool, cfg, return_windows=True)
    assert_final(table, g[f"{tag}_final"], tag)
    assert np.allclose(windows, g[f"{tag}_final_windows"], equal_nan=True, rtol=0, atol=1e-9)
    # the sharded driver: per-iteration tables (whole-genome bins)
    dcool = pipeline.DeviceCool(cool)
    rec = parallel.detect_genome(dcool, cfg)
    off = dcool.offsets
    for it in range(cfg["max_iterations"]):
        ref = g[f"{tag}_k0_i{it}_table"]
        sel = rec[rec[:, 6] == it]
        got = np.column_stack([sel[:, 1] + off[sel[:, 0].astype(int)], sel[:, 2] + off[sel[:, 0].astype(int)], sel[:, 3], sel[:, 4]])
        assert got.shape == ref.shape, (tag, it)
        assert np.array_equal(got[:, :2], ref[:, :2]), (tag, it)
        assert np.abs(got[:, 2] - ref[:, 2]).max() < 1e-9
        assert np.allclose(got[:, 3], ref[:, 3], rtol=1e-6, atol=1e-300)


def test_iterating_a_1d_pattern_raises_like_the_reference(golden):
    """borders / hairpins with max_iterations > 1: the reference stops with ValueError("Cannot have flat kernel.")
    -- the pileup of intra windows holds NaN (captured in iterations.npz: borders_error)."""
    cool = golden("example_cool")
    g = golden("iterations")
    assert str(g["borders_error"]) == "Cannot have flat kernel."
    cfg = copy.deepcopy(ck.borders)
    cfg["max_iterations"] = 2
    with pytest.raises(ValueError, match="Cannot have flat kernel."):
        pipeline.detect(cool, cfg)
    with pytest.raises(ValueError, match="Cannot have flat kernel."):
        parallel.detect_genome(pipeline.DeviceCool(cool), cfg)


# ------------------------------------------------------------------------------------------------
# (c) non-square templates, full mode (detection.py:287-345, preprocessing.py:636-676)
# ------------------------------------------------------------------------------------------------
class _Map:
    pass


def _intra_map(g, tag):
    n = int(g[f"{tag}_n"])
    cmap = _Map()
    cmap.matrix = sp.coo_matrix((g[f"{tag}_prepared_val"], (g[f"{tag}_prepared_row"], g[f"{tag}_prepared_col"])),
                                shape=(n, n)).tocsr()
    det = g[f"{tag}_det"]
    cmap.detectable_bins, cmap.max_dist, cmap.inter = (det.copy(), det.copy()), int(g[f"{tag}_max_dist"]), False
    pearson, pu, pz, md = g[f"{tag}_cfg"]
    cfg = dict(pearson=float(pearson), max_perc_undetected=float(pu), max_perc_zero=float(pz), max_dist=int(md))
    return cmap, cfg


def _check_table(tab, wins, g, tag):
    ref = g[f"{tag}_table"]
    assert ref.shape[0] > 0
    got = tab[["bin1", "bin2", "score", "pvalue"]].to_numpy(dtype=np.float64)
    assert got.shape == ref.shape, tag
    assert np.array_equal(got[:, :2], ref[:, :2]), tag
    assert np.allclose(got[:, 2], ref[:, 2], equal_nan=True, rtol=0, atol=1e-9), tag
    assert np.allclose(got[:, 3], ref[:, 3], equal_nan=True, rtol=1e-6, atol=1e-300), tag
    assert np.allclose(wins, g[f"{tag}_windows"], equal_nan=True, rtol=0, atol=1e-12), tag


@pytest.mark.parametrize("tag", ["d2_59", "d1_37"])
def test_nonsquare_detect_matches_reference(golden, tag, precision):
    g = golden("nonsquare")
    cmap, cfg = _intra_map(g, tag)
    tab, wins = cud.pattern_detector(cmap, cfg, g[f"{tag}_kernel"], full=True)
    _check_table(tab, wins, g, tag)


@pytest.mark.parametrize("tag", ["q2_59", "q1_37"])
def test_nonsquare_quantify_matches_reference(golden, tag):
    g = golden("nonsquare")
    cmap, cfg = _intra_map(g, tag)
    coords = g[f"{tag}_coords"].copy()
    tab, wins = cud.pattern_detector(cmap, cfg, g[f"{tag}_kernel"], coords=coords, full=True)
    _check_table(tab, wins, g, tag)


@pytest.mark.parametrize("tag", ["inter59", "inter95"])
def test_nonsquare_inter_matches_reference(golden, tag):
    g = golden("nonsquare")
    cmap = _Map()
    cmap.matrix = sp.coo_matrix((g["inter_prepared_val"], (g["inter_prepared_row"], g["inter_prepared_col"])),
                                shape=tuple(int(x) for x in g["inter_shape"])).tocsr()
    cmap.detectable_bins, cmap.max_dist, cmap.inter = (g["inter_det_rows"].copy(), g["inter_det_cols"].copy()), None, True
    pearson, pu, pz, md = g["inter_cfg"]
    cfg = dict(pearson=float(pearson), max_perc_undetected=float(pu), max_perc_zero=float(pz), max_dist=int(md))
    tab, wins = cud.pattern_detector(cmap, cfg, g[f"{tag}_kernel"], full=True)
    _check_table(tab, wins, g, tag)


def test_tall_template_on_intra_map_raises_like_the_reference(golden):
    g = golden("nonsquare")
    cmap, cfg = _intra_map(g, "d2_59")
    assert str(g["intra_95_error"]) == "There are 1473 non-zero elements reported as missing."
    with pytest.raises(ValueError) as err:
        cud.pattern_detector(cmap, cfg, g["intra_95_kernel"], full=True)
    assert " ".join(str(a) for a in err.value.args) == str(g["intra_95_error"])


# ------------------------------------------------------------------------------------------------
# (b) detect on real data beyond the 720-bin example: 17 yeast chromosomes, default float32 mode
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("pattern", ["loops", "borders", "hairpins"])
def test_yeast_detect_matches_reference(golden, pattern):
    assert chromosight_amd.engine.get_precision() == "f32"
    cool = golden("yeast_cool")
    g = golden("yeast_detect")
    cfg = copy.deepcopy(getattr(ck, pattern))
    table = pipeline.detect(cool, cfg)
    assert_final(table, g[f"{pattern}_final"], pattern)
    # per-template raw tables through the sharded driver
    dcool = pipeline.DeviceCool(cool)
    rec = parallel.detect_genome(dcool, cfg)
    off = dcool.offsets
    for kid in range(len(cfg["kernels"])):
        ref = g[f"{pattern}_k{kid}_i0_table"]
        sel = rec[rec[:, 5] == kid]
        got = np.column_stack([sel[:, 1] + off[sel[:, 0].astype(int)], sel[:, 2] + off[sel[:, 0].astype(int)], sel[:, 3], sel[:, 4]])
        assert got.shape == ref.shape and np.array_equal(got[:, :2], ref[:, :2]), (pattern, kid)
        assert np.abs(got[:, 2] - ref[:, 2]).max() < 1e-9
    print(f"yeast {pattern}: {len(table)} patterns, coordinates bit-exact")


def test_yeast_float32_prefilter_margin(golden):
    """How much of the float32 pre-filter's per-pixel bound real data uses: on every yeast chromosome, the float32
    coefficient map against the float64 oracle on the staged (detrended) block; for the pixels whose exact coefficient
    passes the loops threshold, the float32 error against the model behind the pre-filter's conditioning screen
    (|r32 - r64| <= 2 n 2^-24 / conditioning, csrc/cs_device.h cand_screen_*)."""
    from oracle import c_oracle
    from chromosight_amd import engine
    from chromosight_amd._lib import CsMatrix, LAYOUT_BAND, LAYOUT_DENSE, MASK_BINS, np_dtype_code
    cool = golden("yeast_cool")
    cfg = copy.deepcopy(ck.loops)
    dcool = pipeline.DeviceCool(cool)
    dev = dcool.dev
    max_dist = max(cfg["max_dist"] // dcool.binsize, 1)
    kern = cfg["kernels"][0]
    kspec = engine.KernelSpec(kern)
    tol = cfg["max_perc_undetected"] / 100
    used, short, n_pass, n_total, worst_err = 0.0, 0.0, 0, 0, 0.0
    for ci in range(dcool.n_chrom):
        blk = dcool.stage_intra(ci, max_dist, 17, resident=True)
        n = blk.shape[0]
        assert blk.sig.layout == LAYOUT_DENSE            # yeast chromosomes are shorter than twice the scanned band
        ld = (n + 15) // 16 * 16
        dev.sync()
        m = blk.buffer.download().view(np.float64).reshape(-1)[:n * ld].reshape(n, ld)[:, :n]
        buf = dev.zeros((n, ld), np.float32)
        out = CsMatrix(buf.ptr, np_dtype_code(np.float32), LAYOUT_DENSE, ld, 0, 0)
        engine.run_normxcorr2(dev, blk.sig, (n, n), kspec, out, full=True, sym_upper=True, max_dist=max_dist,
                              mask_mode=MASK_BINS, miss_row=blk.miss_row, miss_col=blk.miss_col, missing_tol=tol, precision="f32")
        r32 = buf.download()[:, :n].astype(np.float64)
        miss = dcool.miss_host[dcool.offsets[ci]:dcool.offsets[ci + 1]]
        want, cond = c_oracle.normxcorr2_rows(np.triu(m), kern, 0, n, max_dist=max_dist, sym_upper=True, full=True,
                                              miss_row=miss, miss_col=miss, missing_tol=tol)
        ii, jj = np.indices((n, n))
        band = (jj - ii >= 0) & (jj - ii <= max_dist)
        passing = band & (want >= cfg["pearson"])
        n_pass += int(passing.sum())
        n_total += int(band.sum())
        if passing.any():
            err = np.abs(r32 - want)[passing]
            worst_err = max(worst_err, float(err.max()))
            used = max(used, float((err * np.maximum(cond[passing], 1e-12) / (2 * 289 * 2.0 ** -24)).max()))
            short = max(short, float((cfg["pearson"] - r32[passing]).max()))
    print(f"yeast loops: {n_pass} of {n_total} scanned pixels pass the exact threshold; float32 error on them <= {worst_err:.1e}, "
          f"the lowest float32 value sits {short:.1e} under the threshold (margin {engine.RESCORE_MARGIN:g}); at most "
          f"{100 * used:.2f} % of the error model 2 n 2^-24 / conditioning is used")
    assert n_pass > 100
    assert used < 0.5


# ------------------------------------------------------------------------------------------------
# (d) cmd_quantify's selection among templates, captured from the reference's own tables
# ------------------------------------------------------------------------------------------------
def test_quantify_selection_matches_reference_capture(golden):
    cool = golden("yeast_cool")
    q = golden("yeast_quantify")
    g = golden("yeast_quantify_select")
    num = g["positions_num"]
    positions = pd.DataFrame({"chrom1": g["positions_chrom1"], "start1": num[:, 0], "end1": num[:, 1],
                              "chrom2": g["positions_chrom2"], "start2": num[:, 2], "end2": num[:, 3]})
    cfg = dict(pearson=0.15, max_perc_undetected=75.0, max_perc_zero=10.0, max_dist=0, min_dist=0,
               kernels=[q[f"kernel{ki}"] for ki in range(3)], max_iterations=1, min_separation=5000)
    table, windows = pipeline.quantify(cool, positions, cfg, inter=True, max_dist_bp=int(q["cfg_max_dist_bp"]))
    assert len(table) == g["final_num"].shape[0]
    assert table["chrom1"].tolist() == g["final_chrom1"].tolist() and table["chrom2"].tolist() == g["final_chrom2"].tolist()
    got_num = table[["start1", "end1", "start2", "end2", "bin1", "bin2"]].to_numpy(dtype=np.int64)
    assert np.array_equal(got_num, g["final_num"])
    got = table[["score", "pvalue", "qvalue"]].to_numpy(dtype=np.float64)
    ref = g["final_val"]
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    assert np.nanmax(np.abs(got[:, 0] - ref[:, 0])) < 1e-9
    assert np.allclose(got[:, 1:], ref[:, 1:], equal_nan=True, rtol=1e-6, atol=1e-300)
    wide = (num[:, 1] - num[:, 0]) > int(cool["binsize"])
    assert wide.sum() > 100          # intervals whose reported bins differ from the scored midpoints
    assert windows.shape == (len(table), 11, 11)


@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_nonfinite_pixels_zero_their_windows_like_the_reference(golden, precision):
    """normxcorr2 on maps with a NaN / +inf / -inf pixel (tests/golden/nonfinite.npz, from the reference): every window that
    holds the pixel is 0 (detection.py:1088-1101), the others as if the pixel were 0 -- sparse with the pipeline's mask, and
    dense.  (The device's running box sums would carry such a pixel into other windows: it is staged as 0 and its windows are
    zeroed at the Python boundary.)"""
    from chromosight_amd.utils import preprocessing as cup
    g = golden("nonfinite")
    kern, valid = g["kernel"], g["valid"]
    tol = 1e-5 if precision == "f32" else 1e-10
    chromosight_amd.set_precision(precision)
    try:
        for tag in ("nan", "inf", "ninf"):
            a, want = g[f"{tag}_in"], g[f"{tag}_corr"]
            n = a.shape[0]
            mask = cup.make_missing_mask((n, n), valid, valid, max_dist=40, sym_upper=True)
            got, logp = cud.normxcorr2(sp.csr_matrix(a), kern, max_dist=40, sym_upper=True, full=True, missing_mask=mask,
                                       missing_tol=0.75, pval=True)
            got = got.toarray()
            assert np.isfinite(got).all() and np.isfinite(logp.toarray()).all()
            p, q = (int(x[0]) for x in np.nonzero(~np.isfinite(a)))
            assert not got[max(p - 8, 0):p + 9, max(q - 8, 0):q + 9].any() and not want[max(p - 8, 0):p + 9, max(q - 8, 0):q + 9].any()
            assert np.abs(got - want).max() <= tol, (tag, np.abs(got - want).max())
            d, want_d = g[f"{tag}_dense_in"], g[f"{tag}_dense_corr"]
            got_d, _ = cud.normxcorr2(d, kern, full=False)
            assert np.isfinite(got_d).all()
            assert np.abs(got_d - want_d).max() <= tol, (tag, "dense", np.abs(got_d - want_d).max())
    finally:
        chromosight_amd.set_precision("f32")


def test_nonfinite_pixels_in_a_large_host_map():
    """The pipelined host call (maps of >= 1 Mpixel: cs_normxcorr2_host) finds a non-finite pixel or an out-of-range
    magnitude on the device, beside its kernels (CS_ERR_RANGE), and the Python boundary then applies the rule pinned above:
    the windows that hold the pixel are 0, the others as if it were 0; a map scaled by 1e20 goes through float64."""
    kern = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
    a = np.random.default_rng(0).gamma(4.0, 0.25, size=(1200, 1100)).astype(np.float32)
    for bad in (np.nan, -np.inf):
        b = a.copy()
        b[600, 500] = bad
        zeroed = a.copy()
        zeroed[600, 500] = 0.0
        got, _ = cud.normxcorr2(b, kern, full=False)
        want, _ = cud.normxcorr2(zeroed, kern, full=False)
        want = want.copy()
        want[592:609, 492:509] = 0.0
        assert np.isfinite(got).all() and np.abs(got - want).max() < 2e-6
    # a map beyond the float32 range as a whole (every pixel x 1e20): float64 arithmetic, the coefficients of the unscaled map
    # (nothing here sits near the 1e-4 zeroing thresholds)
    clean, _ = cud.normxcorr2(a, kern, full=False)
    got, _ = cud.normxcorr2(a.astype(np.float64) * 1e20, kern, full=False)
    assert np.isfinite(got).all() and np.abs(got - clean).max() < 2e-6


def test_range_guard_refuses_a_violating_device_map():
    """The direct C-ABI path: a device-resident map handed to cs_normxcorr2 must be finite (and below 1e15 in float32) --
    the kernels keep running box sums where the reference sums every window on its own (detection.py:1002-1018, 1088-1101).
    With cs_ctx_set_range_check the entry reduces the map first and refuses a violating one with CS_ERR_RANGE instead of
    returning silently different windows; a clean map passes and gives the unguarded call's result; off is the default."""
    import ctypes as C
    from chromosight_amd import engine
    from chromosight_amd._lib import LAYOUT_DENSE, MASK_NONE, CsMatrix, HipLibraryError, get_device, np_dtype_code
    dev = get_device()
    kspec = engine.KernelSpec(np.asarray(ck.loops["kernels"][0], dtype=np.float64))
    a = np.random.default_rng(3).gamma(4.0, 0.25, size=(300, 320)).astype(np.float32)

    def call(host, precision):
        sig = dev.to_device(host)
        out = dev.zeros(host.shape, np.float32)
        engine.run_normxcorr2(dev, CsMatrix(sig.ptr, np_dtype_code(host.dtype), LAYOUT_DENSE, host.shape[1], 0, 0), host.shape, kspec,
                              CsMatrix(out.ptr, np_dtype_code(np.float32), LAYOUT_DENSE, host.shape[1], 0, 0),
                              full=False, sym_upper=False, max_dist=None, mask_mode=MASK_NONE, precision=precision)
        return out.download()

    plain = call(a, "f32")
    assert dev.lib.cs_ctx_set_range_check(dev.ctx, 1) == 0
    try:
        assert np.array_equal(call(a, "f32"), plain)
        for bad, precisions in ((np.nan, ("f32", "f64")), (np.inf, ("f32", "f64")), (3e16, ("f32",))):
            b = a.copy()
            b[150, 160] = bad
            for precision in precisions:
                with pytest.raises(HipLibraryError, match="non-finite"):
                    call(b, precision)
        b = a.astype(np.float64)
        b[150, 160] = 3e16                    # float64 arithmetic squares it without overflow: accepted
        assert np.isfinite(call(b, "f64")).all()
    finally:
        assert dev.lib.cs_ctx_set_range_check(dev.ctx, 0) == 0


@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_inter_detect_matches_reference(golden, precision):
    """`detect --inter`: pattern_detector in detect mode on six inter-chromosomal blocks of the yeast map (rectangular, dense,
    median scaling: contacts_map.py:598-601), captured from the reference (tests/golden/inter_detect.npz): same patterns in
    the same order, scores to 1e-9, windows of the small blocks."""
    g = golden("inter_detect")
    cool = golden("yeast_cool")
    dcool = pipeline.DeviceCool(cool)
    kern = g["kernel"]
    pearson, undetected, zero = (float(x) for x in g["cfg"])
    cfg = dict(pearson=pearson, max_perc_undetected=undetected, max_perc_zero=zero, max_dist=2000000, min_dist=20000,
               min_separation=5000, max_iterations=1)
    chromosight_amd.set_precision(precision)
    try:
        total = 0
        for ca, cb in g["pairs"]:
            blk = dcool.stage_inter(int(ca), int(cb), resident=True)
            tab, wins = pipeline.detect_block(dcool, blk, cfg, kern, raw=True)
            want = g[f"b{ca}_{cb}_table"]
            assert tab is not None and tab.shape == want.shape, (ca, cb, None if tab is None else tab.shape, want.shape)
            assert np.array_equal(tab[:, :2], want[:, :2]), (ca, cb)
            assert np.abs(tab[:, 2] - want[:, 2]).max() < 1e-9, (ca, cb)
            assert np.allclose(tab[:, 3], want[:, 3], rtol=1e-6, atol=1e-300)
            if f"b{ca}_{cb}_windows" in g:
                assert np.allclose(wins, g[f"b{ca}_{cb}_windows"], rtol=0, atol=1e-9, equal_nan=True)
            total += len(want)
        assert total > 2000
    finally:
        chromosight_amd.set_precision("f32")


@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_smooth_trend_and_tsvd_match_reference(golden, precision):
    """`detect --smooth-trend` (isotonic fit of the distance law inside the detrend) and `detect --tsvd` (0.999, the CLI's value)
    on the three example blocks, loops and the three borders templates: per-block raw tables captured from the reference
    (tests/golden/options.npz) -- same patterns in the same order, scores to 1e-9."""
    g = golden("options")
    dcool = pipeline.DeviceCool(golden("example_cool"))
    chromosight_amd.set_precision(precision)
    try:
        total = 0
        for name in ("loops", "borders"):
            cfg = copy.deepcopy(getattr(ck, name))
            max_dist = max(cfg["max_dist"] // dcool.binsize, 1)
            kernels = [np.asarray(k, dtype=np.float64) for k in cfg["kernels"]]
            largest = max(k.shape[0] for k in kernels)
            for ci in range(dcool.n_chrom):
                plain = dcool.stage_intra(ci, max_dist, largest, resident=True)
                smooth = dcool.stage_intra(ci, max_dist, largest, smooth=True, resident=True)
                for ki, kern in enumerate(kernels):
                    for tag, blk, tsvd in (("smooth", smooth, None), ("tsvd", plain, 0.999)):
                        want = g[f"{name}_{tag}_c{ci}_k{ki}"]
                        tab, _ = pipeline.detect_block(dcool, blk, cfg, kern, tsvd=tsvd, raw=True)
                        got = np.zeros((0, 4)) if tab is None else tab
                        assert got.shape == want.shape, (name, tag, ci, ki, got.shape, want.shape)
                        if len(want):
                            assert np.array_equal(got[:, :2], want[:, :2]), (name, tag, ci, ki)
                            assert np.abs(got[:, 2] - want[:, 2]).max() < 1e-9, (name, tag, ci, ki)
                        total += len(want)
        assert total > 400
    finally:
        chromosight_amd.set_precision("f32")


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_win_size_matches_reference(golden, precision):
    """`detect --win-size` (cli/chromosight.py:689-695) on the three example blocks: loops resized to 9 and 21, the borders
    templates to 23 (pipeline.with_win_size), blocks staged with the resized template's keep distance: per-block raw tables
    captured from the reference (tests/golden/winsize.npz) -- same patterns in the same order, scores to 1e-9."""
    g = golden("winsize")
    dcool = pipeline.DeviceCool(golden("example_cool"))
    chromosight_amd.set_precision(precision)
    try:
        total = 0
        for name, pattern, win in (("loops9", "loops", 9), ("loops21", "loops", 21), ("borders23", "borders", 23)):
            cfg = pipeline.with_win_size(copy.deepcopy(getattr(ck, pattern)), win)
            max_dist = max(cfg["max_dist"] // dcool.binsize, 1)
            kernels = [np.asarray(k, dtype=np.float64) for k in cfg["kernels"]]
            for ci in range(dcool.n_chrom):
                blk = dcool.stage_intra(ci, max_dist, win, resident=True)
                for ki, kern in enumerate(kernels):
                    want = g[f"{name}_c{ci}_k{ki}"]
                    tab, _ = pipeline.detect_block(dcool, blk, cfg, kern, raw=True)
                    got = np.zeros((0, 4)) if tab is None else tab
                    assert got.shape == want.shape, (name, ci, ki, got.shape, want.shape)
                    if len(want):
                        assert np.array_equal(got[:, :2], want[:, :2]), (name, ci, ki)
                        assert np.abs(got[:, 2] - want[:, 2]).max() < 1e-9, (name, ci, ki)
                    total += len(want)
        assert total > 500
    finally:
        chromosight_amd.set_precision("f32")
