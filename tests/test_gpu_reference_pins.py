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
    """How close the float32 pre-filter comes to losing a pixel on real data: on every yeast chromosome, the pixels
    whose exact (float64) coefficient passes the loops threshold, and their float32 coefficients."""
    from chromosight_amd import engine
    from chromosight_amd._lib import CsMatrix, LAYOUT_BAND, MASK_BINS, np_dtype_code
    cool = golden("yeast_cool")
    cfg = copy.deepcopy(ck.loops)
    dcool = pipeline.DeviceCool(cool)
    dev = dcool.dev
    max_dist = max(cfg["max_dist"] // dcool.binsize, 1)
    kspec = engine.KernelSpec(cfg["kernels"][0])
    worst, n_pass, n_total = 0.0, 0, 0
    for ci in range(dcool.n_chrom):
        blk = dcool.stage_intra(ci, max_dist, 17, resident=True)
        n = blk.shape[0]
        if blk.sig.layout != LAYOUT_BAND:
            continue
        w = min(max_dist, n - 1) + 1
        ld = (w + 63) // 64 * 64
        outs = {}
        for prec, dt in (("f32", np.float32), ("f64", np.float64)):
            buf = dev.zeros((n, ld), dt)
            out = CsMatrix(buf.ptr, np_dtype_code(dt), LAYOUT_BAND, ld, 0, w)
            engine.run_normxcorr2(dev, blk.sig, (n, n), kspec, out, full=True, sym_upper=True, max_dist=max_dist,
                                  mask_mode=MASK_BINS, miss_row=blk.miss_row, miss_col=blk.miss_col,
                                  missing_tol=cfg["max_perc_undetected"] / 100, precision=prec)
            outs[prec] = buf.download()[:, :w].astype(np.float64)
        passing = outs["f64"] >= cfg["pearson"]
        n_pass += int(passing.sum())
        n_total += passing.size
        if passing.any():
            worst = max(worst, float((cfg["pearson"] - outs["f32"][passing]).max()))
    print(f"yeast loops: {n_pass} of {n_total} pixels pass the exact threshold; the lowest float32 value among them sits "
          f"{worst:.2e} below it (pre-filter margin {engine.RESCORE_MARGIN:g})")
    assert n_pass > 100
    assert worst < 0.1 * engine.RESCORE_MARGIN


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
