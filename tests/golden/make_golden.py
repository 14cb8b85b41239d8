#!/usr/bin/env python3
"""Generate the golden input/output vectors under tests/golden/ by IMPORTING
THE REFERENCE (authoring container only; /root/reference does not exist on the
GPU box, and nothing else in this repository reads it at run time).

    PYTHONPATH=/root/reference python tests/golden/make_golden.py

What is captured:
  xcorr2.npz            reference xcorr2 dense/sparse on the reference tests' Gaussian
                        blobs + 7x7 Gaussian kernel, constant 11x11 kernel, and the
                        scipy.signal.correlate2d values the reference test compares to
                        (tests/test_detection.py:241-291)
  normxcorr2_dense.npz  dense + sparse normxcorr2, no mask, full / not full / sym_upper
  masks.npz             make_missing_mask + frame_missing_mask for several geometries
  normxcorr2_mask.npz   sparse normxcorr2(full=True, mask) on random banded tiles with
                        missing bins, max_dist below and above N, and an inter block
  example_blocks.npz    per chromosome of data_test/example.cool: balanced block,
                        distance law, detrended+trimmed block, coefficient / log-p maps
                        for loops, borders x3, hairpins, pattern_detector tables
  nms.npz               pick_foci / label_foci / filter_foci / validate_patterns /
                        remove_neighbours inputs and outputs
  stats.npz             corr_to_pval, fdr_correction
  example_{loops,borders,hairpins}.tsv, example_bed2.txt
                        known-answer outputs / inputs copied from the reference's docs
                        and test data (data files, not source)
"""
import pathlib
import shutil
import sys

import numpy as np
import scipy.sparse as sp
import scipy.signal as sig
from scipy.stats import multivariate_normal

REF = pathlib.Path("/root/reference")
sys.path.insert(0, str(REF))
import chromosight.utils.detection as cud  # noqa: E402
import chromosight.utils.preprocessing as cup  # noqa: E402
import chromosight.utils.stats as cus  # noqa: E402

HERE = pathlib.Path(__file__).resolve().parent
KDIR = REF / "chromosight" / "kernels"


def load_template(name):
    return np.loadtxt(KDIR / name)


LOOPS = load_template("artificial_template_loops_type1.txt")
BORDERS = [load_template(f"artificial_template_borders_type{i}.txt") for i in (1, 2, 3)]
HAIRPIN = load_template("artificial_template_hairpin.txt")
LOOPS_SMALL = load_template("artificial_template_loops_small.txt")


def coo_fields(prefix, mat):
    mat = sp.coo_matrix(mat)
    return {
        f"{prefix}_row": mat.row.astype(np.int32),
        f"{prefix}_col": mat.col.astype(np.int32),
        f"{prefix}_val": mat.data,
        f"{prefix}_shape": np.array(mat.shape, dtype=np.int64),
    }


# --------------------------------------------------------------------------- #
def gauss_mat(meanx, meany, std, shape=(100, 100)):
    """Synthetic blob with the construction the reference tests use
    (tests/test_detection.py:18-37)."""
    k = multivariate_normal(mean=(meanx, meany), cov=np.eye(2) * std)
    x = np.linspace(-10, 10, shape[0])
    y = np.linspace(-10, 10, shape[1])
    xx, yy = np.meshgrid(x, y)
    return k.pdf(np.c_[xx.ravel(), yy.ravel()]).reshape(shape)


def make_xcorr2():
    gk = gauss_mat(0, 0, 5, shape=(7, 7))
    gk = gk + gk.T - np.diag(np.diag(gk))
    out = {"gauss_kernel": gk}
    cases = [(-1.5, -1.0, 0.3), (-0.5, 1.0, 1.5), (0.5, 1.0, 2.7)]
    for c, (mx, my, sd) in enumerate(cases):
        m = gauss_mat(mx, my, sd)
        out[f"sig{c}"] = m
        out[f"dense{c}"] = cud.xcorr2(m, gk, threshold=1e-4)
        out[f"sparse{c}"] = cud.xcorr2(sp.csr_matrix(m), gk, threshold=1e-4).toarray()
        out[f"scipy_valid{c}"] = sig.correlate2d(m, gk, "valid")
        k1 = np.ones((11, 11))
        out[f"const{c}"] = cud.xcorr2(sp.csr_matrix(m), k1 / 121).toarray()
        out[f"loops{c}"] = cud.xcorr2(m, LOOPS)
    rng = np.random.default_rng(11)
    m = rng.gamma(4, 0.25, size=(70, 90))
    rk = rng.random((5, 9))
    out["rand"] = m
    out["rect_kernel_5x9"] = rk
    out["rand_rect_5x9"] = cud.xcorr2(m, rk)
    # tsvd: correlation with the truncated-SVD reconstruction (detection.py:618-619).
    # Sparse input only: the reference's dense factorised branch raises
    # UnboundLocalError (detection.py:754-766 assign `out`, :801 reads `out_wo_margin`).
    out["rand_loops_tsvd999"] = cud.xcorr2(sp.csr_matrix(m), LOOPS, tsvd=0.999).toarray()
    u, v = cup.factorise_kernel(LOOPS.copy(), prop_info=0.999)
    out["loops_tsvd999_u"], out["loops_tsvd999_v"] = u, v
    np.savez_compressed(HERE / "xcorr2.npz", **out)


def make_normxcorr2_dense():
    rng = np.random.default_rng(5)
    out = {}
    sig_a = rng.gamma(4, 0.25, size=(96, 80))
    # sparse-ish variant with exact zeros and a flat region
    sig_b = sig_a * (rng.random(sig_a.shape) > 0.6)
    sig_b[30:60, 20:50] = 0.0
    sig_b[5:25, 55:78] = 2.5
    out["sig_a"], out["sig_b"] = sig_a, sig_b
    for name, s in (("a", sig_a), ("b", sig_b)):
        for kname, k in (("loops", LOOPS), ("small", LOOPS_SMALL), ("hairpin", HAIRPIN)):
            for full in (False, True):
                tag = f"{name}_{kname}_{'full' if full else 'valid'}"
                # dense + full + pval raises AttributeError in the reference
                # (detection.py:1263: int.flatten()), so p-values only when not full
                c, p = cud.normxcorr2(s, k, full=full, pval=not full)
                out[f"dense_{tag}_corr"] = c
                if p is not None:
                    out[f"dense_{tag}_pval"] = p
                c, p = cud.normxcorr2(sp.csr_matrix(s), k, full=full, pval=True)
                out[f"sparse_{tag}_corr"], out[f"sparse_{tag}_pval"] = c.toarray(), p.toarray()
        sq = s[:80, :80]
        c, p = cud.normxcorr2(sp.csr_matrix(np.triu(sq)), LOOPS, sym_upper=True, full=True, pval=True)
        out[f"sparse_{name}_loops_symfull_corr"] = c.toarray()
        out[f"sparse_{name}_loops_symfull_pval"] = p.toarray()
        c, _ = cud.normxcorr2(np.triu(sq), LOOPS, sym_upper=True, full=False)
        out[f"dense_{name}_loops_symvalid_corr"] = c
    # tsvd through normxcorr2 (sparse, no mask)
    c, _ = cud.normxcorr2(sp.csr_matrix(sig_a), LOOPS, full=True, tsvd=0.999)
    out["sparse_a_loops_full_tsvd999_corr"] = c.toarray()
    np.savez_compressed(HERE / "normxcorr2_dense.npz", **out)


def make_masks():
    out = {}
    rng = np.random.default_rng(3)
    cases = []
    for n in (40, 75):
        miss = np.sort(rng.choice(n, size=max(2, n // 10), replace=False))
        valid = np.setdiff1d(np.arange(n), miss)
        for kshape in ((7, 7), (17, 17), (5, 9)):
            for md in (None, 0, 1, 5, 20, 38, 60, 100):
                cases.append((n, valid, kshape, md))
    idx = 0
    for n, valid, kshape, md in cases:
        mask = cup.make_missing_mask((n, n), valid, valid, max_dist=md, sym_upper=True)
        framed = cup.frame_missing_mask(mask, kshape, sym_upper=True, max_dist=md)
        out[f"sym{idx}_n"] = np.int64(n)
        out[f"sym{idx}_valid"] = valid
        out[f"sym{idx}_kshape"] = np.array(kshape)
        out[f"sym{idx}_max_dist"] = np.int64(-1 if md is None else md)
        out[f"sym{idx}_mask"] = np.packbits(mask.toarray())
        out[f"sym{idx}_framed"] = np.packbits(framed.toarray())
        idx += 1
    out["n_sym"] = np.int64(idx)
    # inter (rectangular, non symmetric)
    idx = 0
    for shape in ((30, 50), (64, 33)):
        vr = np.setdiff1d(np.arange(shape[0]), rng.choice(shape[0], 4, replace=False))
        vc = np.setdiff1d(np.arange(shape[1]), rng.choice(shape[1], 5, replace=False))
        for kshape in ((7, 7), (17, 17)):
            mask = cup.make_missing_mask(shape, vr, vc, max_dist=None, sym_upper=False)
            framed = cup.frame_missing_mask(mask, kshape, sym_upper=False, max_dist=None)
            out[f"inter{idx}_shape"] = np.array(shape)
            out[f"inter{idx}_valid_rows"] = vr
            out[f"inter{idx}_valid_cols"] = vc
            out[f"inter{idx}_kshape"] = np.array(kshape)
            out[f"inter{idx}_mask"] = np.packbits(mask.toarray())
            out[f"inter{idx}_framed"] = np.packbits(framed.toarray())
            idx += 1
    out["n_inter"] = np.int64(idx)
    np.savez_compressed(HERE / "masks.npz", **out)


def banded_random(rng, n, band, zero_frac=0.2):
    """Upper-band matrix with values ~ gamma around 1 and some exact zeros."""
    a = rng.gamma(4, 0.25, size=(n, n))
    a *= rng.random((n, n)) > zero_frac
    ii, jj = np.indices((n, n))
    a[(jj - ii < 0) | (jj - ii > band)] = 0
    return a


def make_normxcorr2_mask():
    rng = np.random.default_rng(7)
    out = {}
    idx = 0
    for n, md, kern, tol in (
        (90, 30, LOOPS, 0.5),
        (90, 200, LOOPS, 0.5),       # max_dist > N
        (64, 12, LOOPS_SMALL, 0.75),
        (120, 50, HAIRPIN, 0.75),
        (70, 1, BORDERS[0], 0.75),   # borders-like: max_dist=1
        (70, 25, BORDERS[1], 0.25),
    ):
        keep = md + max(kern.shape)
        a = banded_random(rng, n, keep)
        miss = np.sort(rng.choice(n, size=max(3, n // 12), replace=False))
        # a cluster of adjacent missing bins too
        miss = np.unique(np.concatenate([miss, np.arange(n // 2, n // 2 + 4)]))
        a[miss, :] = 0
        a[:, miss] = 0
        valid = np.setdiff1d(np.arange(n), miss)
        mask = cup.make_missing_mask((n, n), valid, valid, max_dist=md, sym_upper=True)
        c, p = cud.normxcorr2(
            sp.csr_matrix(a), kern, max_dist=md, sym_upper=True, full=True,
            missing_mask=mask, missing_tol=tol, pval=True,
        )
        out[f"intra{idx}_sig"] = a
        out[f"intra{idx}_kernel"] = kern
        out[f"intra{idx}_valid"] = valid
        out[f"intra{idx}_max_dist"] = np.int64(md)
        out[f"intra{idx}_tol"] = np.float64(tol)
        out[f"intra{idx}_corr"] = c.toarray()
        out[f"intra{idx}_pval"] = p.toarray()
        idx += 1
    out["n_intra"] = np.int64(idx)
    # intra, max_dist=None
    n = 60
    a = np.triu(rng.gamma(4, 0.25, size=(n, n)))
    miss = np.array([3, 17, 18, 40])
    a[miss, :] = 0
    a[:, miss] = 0
    valid = np.setdiff1d(np.arange(n), miss)
    mask = cup.make_missing_mask((n, n), valid, valid, max_dist=None, sym_upper=True)
    c, p = cud.normxcorr2(sp.csr_matrix(a), LOOPS_SMALL, max_dist=None, sym_upper=True,
                          full=True, missing_mask=mask, missing_tol=0.75, pval=True)
    out["nomd_sig"], out["nomd_valid"] = a, valid
    out["nomd_corr"], out["nomd_pval"] = c.toarray(), p.toarray()
    # inter block
    shape = (50, 72)
    a = rng.gamma(4, 0.25, size=shape) * (rng.random(shape) > 0.3)
    mr, mc = np.array([0, 9, 10, 33]), np.array([5, 50, 51, 52, 71])
    a[mr, :] = 0
    a[:, mc] = 0
    vr = np.setdiff1d(np.arange(shape[0]), mr)
    vc = np.setdiff1d(np.arange(shape[1]), mc)
    mask = cup.make_missing_mask(shape, vr, vc, max_dist=None, sym_upper=False)
    for kname, kern in (("loops", LOOPS), ("b11", cup.resize_kernel(BORDERS[0], factor=11 / 17, quiet=True))):
        c, p = cud.normxcorr2(sp.csr_matrix(a), kern, max_dist=None, sym_upper=False,
                              full=True, missing_mask=mask, missing_tol=0.75, pval=True)
        out[f"inter_{kname}_kernel"] = kern
        out[f"inter_{kname}_corr"], out[f"inter_{kname}_pval"] = c.toarray(), p.toarray()
    out["inter_sig"], out["inter_valid_rows"], out["inter_valid_cols"] = a, vr, vc
    # non-full with a mask (mask used un-framed, detection.py:991-997)
    n = 50
    a = banded_random(rng, n, 30)
    miss = np.array([10, 11, 30])
    a[miss, :] = 0
    a[:, miss] = 0
    valid = np.setdiff1d(np.arange(n), miss)
    mask = cup.make_missing_mask((n, n), valid, valid, max_dist=20, sym_upper=True)
    c, p = cud.normxcorr2(sp.csr_matrix(a), LOOPS_SMALL, max_dist=20, sym_upper=True,
                          full=False, missing_mask=mask, missing_tol=0.75, pval=True)
    out["valid_sig"], out["valid_valid"] = a, valid
    out["valid_mask"] = mask.toarray()
    out["valid_corr"], out["valid_pval"] = c.toarray(), p.toarray()
    np.savez_compressed(HERE / "normxcorr2_mask.npz", **out)


# --------------------------------------------------------------------------- #
class RefMap:
    """Stand-in for the reference's ContactMap (the four attributes
    pattern_detector reads, tests/test_detection.py:88-100)."""

    def __init__(self, matrix, detectable_bins, max_dist, inter, name="blk"):
        self.matrix = matrix
        self.detectable_bins = detectable_bins
        self.max_dist = max_dist
        self.inter = inter
        self.name = name


def balanced_block(cool, ca, cb):
    """Emulates cooler's matrix(sparse=True, balance=True)[s1:e1, s2:e2]:
    count * w[bin1] * w[bin2], symmetric fill for intra blocks."""
    off = cool["chrom_offset"]
    s1, e1, s2, e2 = off[ca], off[ca + 1], off[cb], off[cb + 1]
    b1, b2, cnt, w = cool["bin1_id"], cool["bin2_id"], cool["count"], cool["weight"]
    val = cnt * w[b1] * w[b2]
    if ca == cb:
        sel = (b1 >= s1) & (b1 < e1) & (b2 >= s2) & (b2 < e2)
        r, c, v = b1[sel] - s1, b2[sel] - s2, val[sel]
        offd = r != c
        rows = np.concatenate([r, c[offd]])
        cols = np.concatenate([c, r[offd]])
        vals = np.concatenate([v, v[offd]])
    else:
        sel = (b1 >= s1) & (b1 < e1) & (b2 >= s2) & (b2 < e2)
        rows, cols, vals = b1[sel] - s1, b2[sel] - s2, val[sel]
    return sp.coo_matrix((vals, (rows, cols)), shape=(e1 - s1, e2 - s2))


def prepare_intra(block, det, max_dist, largest_kernel):
    """ContactMap.create_mat for a balanced intra block
    (reference contacts_map.py:527-548, 603-638)."""
    n = block.shape[0]
    keep = min(max_dist, n) + largest_kernel
    m = cup.detrend(block, max_dist=keep, smooth=False, detectable_bins=det, max_val=10)
    law = cup.distance_law(block.tocsr(), detectable_bins=det, max_dist=keep, smooth=False)
    m = cup.diag_trim(m.tocsr(), keep)
    m.data[np.isnan(m.data)] = 0
    m.eliminate_zeros()
    return m, law, keep


def make_example_blocks():
    cool = dict(np.load(HERE / "example_cool.npz"))
    binsize = int(cool["binsize"])
    off = cool["chrom_offset"]
    det_all = np.flatnonzero(np.isfinite(cool["weight"]))
    out = {}
    patterns = {
        "loops": (dict(pearson=0.3, max_perc_undetected=50.0, max_perc_zero=10.0, max_dist=2000000), [LOOPS]),
        "borders": (dict(pearson=0.15, max_perc_undetected=75.0, max_perc_zero=10.0, max_dist=0), BORDERS),
        "hairpins": (dict(pearson=0.1, max_perc_undetected=75.0, max_perc_zero=10.0, max_dist=0), [HAIRPIN]),
    }
    nms = {}
    for ci in range(3):
        s, e = off[ci], off[ci + 1]
        det = det_all[(det_all >= s) & (det_all < e)] - s
        block = balanced_block(cool, ci, ci)
        out[f"chr{ci}_n"] = np.int64(e - s)
        out[f"chr{ci}_det"] = det
        out.update(coo_fields(f"chr{ci}_balanced", block))
        for pname, (cfg, kernels) in patterns.items():
            max_dist = max(cfg["max_dist"] // binsize, 1)
            largest = max(k.shape[0] for k in kernels)
            m, law, keep = prepare_intra(block, det, max_dist, largest)
            out[f"chr{ci}_{pname}_law"] = law
            out[f"chr{ci}_{pname}_keep"] = np.int64(keep)
            out[f"chr{ci}_{pname}_max_dist"] = np.int64(max_dist)
            out.update(coo_fields(f"chr{ci}_{pname}_prepared", m))
            for ki, kern in enumerate(kernels):
                cmap = RefMap(m.copy(), (det.copy(), det.copy()), max_dist, False)
                mask = cup.make_missing_mask(m.shape, det, det, max_dist=max_dist, sym_upper=True)
                corr, pval = cud.normxcorr2(
                    m.tocsr(), kern, max_dist=max_dist, sym_upper=True, full=True,
                    missing_mask=mask, pval=True,
                    missing_tol=cfg["max_perc_undetected"] / 100,
                )
                tag = f"chr{ci}_{pname}{ki}"
                out.update(coo_fields(f"{tag}_corr", corr))
                out.update(coo_fields(f"{tag}_pval", pval))
                tab, wins = cud.pattern_detector(cmap, cfg, kern, full=True)
                if tab is None:
                    out[f"{tag}_table"] = np.zeros((0, 4))
                    out[f"{tag}_windows"] = np.zeros((0,) + kern.shape)
                else:
                    out[f"{tag}_table"] = tab[["bin1", "bin2", "score", "pvalue"]].to_numpy(dtype=np.float64)
                    out[f"{tag}_windows"] = wins
                # NMS stage in isolation: stage-04 map -> foci
                conv = corr.copy()
                conv.data[np.isnan(conv.data)] = 0
                conv = cup.diag_trim(conv.tocsr(), max_dist).tocoo()
                conv.eliminate_zeros()
                foci, lab = cud.pick_foci(conv, cfg["pearson"])
                nms.update(coo_fields(f"{tag}_conv", conv))
                nms[f"{tag}_pearson"] = np.float64(cfg["pearson"])
                if foci is None:
                    nms[f"{tag}_foci"] = np.zeros((0, 2), dtype=np.int64)
                else:
                    nms[f"{tag}_foci"] = foci
                    nms.update(coo_fields(f"{tag}_labels", lab))
    # quantify mode on given coordinates, chr1, loops, including out-of-bounds /
    # missing-bin coordinates
    ci = 1
    s, e = off[ci], off[ci + 1]
    det = det_all[(det_all >= s) & (det_all < e)] - s
    block = balanced_block(cool, ci, ci)
    cfg, kernels = patterns["loops"]
    max_dist = 60
    m, law, keep = prepare_intra(block, det, max_dist, 17)
    coords = np.array([[10, 40], [100, 130], [200, 205], [0, 3], [300, 360], [418, 421], [150, 150]])
    cmap = RefMap(m.copy(), (det.copy(), det.copy()), max_dist, False)
    qcfg = dict(cfg)
    tab, wins = cud.pattern_detector(cmap, qcfg, LOOPS, coords=coords.copy(), full=True)
    out["quant_coords"] = coords
    out["quant_max_dist"] = np.int64(max_dist)
    out.update(coo_fields("quant_prepared", m))
    out["quant_det"] = det
    out["quant_table"] = tab[["bin1", "bin2", "score", "pvalue"]].to_numpy(dtype=np.float64)
    out["quant_windows"] = wins
    # inter block chr0 x chr2 (reference contacts_map.py:598-601)
    blk = balanced_block(cool, 0, 2).tocoo()
    blk.data[np.isnan(blk.data)] = 0.0
    blk.data = blk.data / np.nanmedian(blk.data)
    blk.data[np.isnan(blk.data)] = 0
    blk.eliminate_zeros()
    det_r = det_all[(det_all >= off[0]) & (det_all < off[1])] - off[0]
    det_c = det_all[(det_all >= off[2]) & (det_all < off[3])] - off[2]
    cmap = RefMap(blk.copy(), (det_r.copy(), det_c.copy()), None, True)
    icoords = np.array([[20, 30], [60, 100], [5, 5], [120, 160], [64, 8]])
    icfg = dict(patterns["loops"][0])
    tab, wins = cud.pattern_detector(cmap, icfg, LOOPS, coords=icoords.copy(), full=True)
    out.update(coo_fields("inter_prepared", blk))
    out["inter_det_rows"], out["inter_det_cols"] = det_r, det_c
    out["inter_coords"] = icoords
    out["inter_table"] = tab[["bin1", "bin2", "score", "pvalue"]].to_numpy(dtype=np.float64)
    out["inter_windows"] = wins
    np.savez_compressed(HERE / "example_blocks.npz", **out)
    return nms


def make_nms(nms):
    out = dict(nms)
    # exact label example of the reference tests (tests/test_detection.py:204-238)
    spec = np.array(
        [[1, 0, 0, 0, 1, 1],
         [1, 0, 1, 0, 0, 0],
         [1, 0, 1, 1, 0, 1],
         [0, 0, 0, 0, 0, 1]])
    nf, lab = cud.label_foci(sp.coo_matrix(spec))
    out["spec_in"] = spec
    out["spec_labels"] = lab.toarray()
    out["spec_num"] = np.int64(nf)
    nf2, _ = cud.filter_foci(sp.coo_matrix(lab.toarray()), min_size=2)
    nf3, _ = cud.filter_foci(sp.coo_matrix(lab.toarray()), min_size=3)
    out["spec_num_min2"], out["spec_num_min3"] = np.int64(nf2), np.int64(nf3)
    # remove_neighbours
    import pandas as pd
    rng = np.random.default_rng(21)
    pts = pd.DataFrame({
        "bin1": rng.integers(0, 60, 80),
        "bin2": rng.integers(0, 60, 80),
        "score": np.round(rng.random(80), 3),
    })
    keep = cud.remove_neighbours(pts, win_size=5)
    out["rn_patterns"] = pts.to_numpy(dtype=np.float64)
    out["rn_keep_w5"] = keep
    out["rn_keep_w1"] = cud.remove_neighbours(pts, win_size=1)
    np.savez_compressed(HERE / "nms.npz", **out)


def make_stats():
    rng = np.random.default_rng(2)
    r = np.concatenate([rng.uniform(-1, 1, 200), [0.0, 1.0, -1.0, 0.999999, 1e-9]])
    n = np.concatenate([rng.integers(4, 290, 200), [289, 289, 100, 50, 289]])
    out = {
        "r": r, "n": n,
        "logp_vec": cus.corr_to_pval(r, n.astype(float)),
        "logp_289": cus.corr_to_pval(r, 289),
        "pvals": np.array([0.01, 0.02, 0.03, 0.5, 0.001, 0.2, 0.04]),
    }
    out["qvals"] = cus.fdr_correction(out["pvals"])
    np.savez_compressed(HERE / "stats.npz", **out)


def copy_known_answers():
    for name in ("loops", "borders", "hairpins"):
        shutil.copy(REF / "docs" / "notebooks" / "detect" / f"example_{name}.tsv", HERE / f"example_{name}.tsv")
    shutil.copy(REF / "data_test" / "example.bed2", HERE / "example_bed2.txt")


if __name__ == "__main__":
    make_xcorr2()
    make_normxcorr2_dense()
    make_masks()
    make_normxcorr2_mask()
    nms = make_example_blocks()
    make_nms(nms)
    make_stats()
    copy_known_answers()
    for f in sorted(HERE.glob("*")):
        print(f.name, f.stat().st_size)
