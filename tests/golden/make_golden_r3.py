#!/usr/bin/env python3
"""Round-3 golden vectors, produced by IMPORTING THE REFERENCE in the authoring container
(/root/reference does not exist on the GPU box; nothing else reads it at run time).

    PYTHONPATH=/root/reference python tests/golden/make_golden_r3.py

  iterations.npz        `detect` with max_iterations = 2 on the three blocks of data_test/example.cool,
                        loops and borders: the template of the second iteration is the pileup of the
                        windows of ALL sub-matrices (cli/chromosight.py:731-791, detection.py:158-174);
                        per (kernel_id, iteration) the whole-genome raw table and the template used, and
                        the table after the reference's post-processing (cli/chromosight.py:806-871)
  nonsquare.npz         pattern_detector(full=True) with 5x9 and 9x5 templates (detect 2-D, detect 1-D,
                        quantify): the reference pads the maps by (kw rows, kh columns) but shifts the
                        coordinates by (kh, kw) (detection.py:287-345, preprocessing.py:636-676), so
                        windows, scores and the 1-D row coordinate are offset by kh - kw
  yeast_detect.npz      `detect` (loops, borders, hairpins; default configs) on the 17 chromosomes of the
                        yeast map of the reference's docs: per-block raw tables and the final table
  yeast_quantify_select.npz
                        the selection block of cmd_quantify (cli/chromosight.py:395-470) on the
                        reference's own per-template tables of yeast_quantify.npz: best of the templates
                        per coordinate, bin columns from start1 / start2, q-values, output order; some
                        intervals are three bins wide so that bin1 / bin2 differ from the scored midpoints
  winsize.npz           detect --win-size 9 / 21 (loops) and 23 (borders) on the example blocks, per-block raw tables
  resize.npz            resize_kernel / crop_kernel of the built-in templates (factors, resolutions, target sizes)
  options.npz           `detect --smooth-trend` and `--tsvd 0.999` (loops, borders) on the example blocks: per-block raw tables
  inter_detect.npz      pattern_detector in detect mode on six inter-chromosomal blocks of the yeast map (median scaling)
  nonfinite.npz         normxcorr2 on maps with one NaN / +inf / -inf pixel (sparse + mask, dense): the windows that
                        hold the pixel are 0, the others as if it were 0
"""
import pathlib
import sys

import numpy as np
import pandas as pd
import scipy.sparse as sp

REF = pathlib.Path("/root/reference")
sys.path.insert(0, str(REF))
HERE = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import chromosight.utils.detection as cud  # noqa: E402
import chromosight.utils.preprocessing as cup  # noqa: E402
import chromosight.utils.stats as cus  # noqa: E402
from make_golden import BORDERS, HAIRPIN, LOOPS, RefMap, balanced_block, prepare_intra  # noqa: E402

OUT_COLS = ["bin1", "bin2", "kernel_id", "iteration", "score", "pvalue", "qvalue"]


def bins_table(cool):
    off = cool["chrom_offset"]
    names = [str(n) for n in cool["chrom_names"]]
    chrom = np.repeat(np.asarray(names, dtype=object), np.diff(off))
    return pd.DataFrame({"chrom": chrom, "start": cool["bin_start"], "end": cool["bin_end"]})


def detect_like_cli(cool, cfg, kernels, n_chrom):
    """The per-kernel / per-iteration loop and the post-processing of cmd_detect, with the reference's
    own functions on hand-assembled blocks (cooler is not installed)."""
    binsize = int(cool["binsize"])
    off = cool["chrom_offset"]
    det_all = np.flatnonzero(np.isfinite(cool["weight"]))
    max_dist = max(cfg["max_dist"] // binsize, 1)
    largest = max(k.shape[0] for k in kernels)
    blocks = []
    for ci in range(n_chrom):
        s, e = off[ci], off[ci + 1]
        det = det_all[(det_all >= s) & (det_all < e)] - s
        m, _, _ = prepare_intra(balanced_block(cool, ci, ci), det, max_dist, largest)
        blocks.append((m, det))
    all_coords, all_windows, log = [], [], {}
    for kernel_id, kernel in enumerate(kernels):
        for it in range(cfg["max_iterations"]):
            tabs, wins = [], []
            for ci, (m, det) in enumerate(blocks):
                cmap = RefMap(m.copy(), (det.copy(), det.copy()), max_dist, False)
                tab, win = cud.pattern_detector(cmap, cfg, kernel, full=True, tsvd=None)
                if tab is None:
                    continue
                tab = tab.copy()
                tab.bin1 += off[ci]
                tab.bin2 += off[ci]
                tabs.append(tab)
                wins.append(win)
            log[f"k{kernel_id}_i{it}_kernel"] = np.array(kernel, dtype=np.float64)
            if not tabs:
                log[f"k{kernel_id}_i{it}_table"] = np.zeros((0, 4))
                break
            kernel_windows = np.concatenate(wins, axis=0)
            tab = pd.concat(tabs, axis=0).reset_index(drop=True)
            log[f"k{kernel_id}_i{it}_table"] = tab[["bin1", "bin2", "score", "pvalue"]].to_numpy(dtype=np.float64)
            tab["kernel_id"] = kernel_id
            tab["iteration"] = it
            all_coords.append(tab)
            all_windows.append(kernel_windows)
            kernel = cud.pileup_patterns(kernel_windows)
    coords = pd.concat(all_coords, axis=0).reset_index(drop=True)
    windows = np.concatenate(all_windows, axis=0)
    separation = max(int(cfg["min_separation"] // binsize), 1)
    keep = cud.remove_neighbours(coords, win_size=separation)
    coords = coords.loc[keep, :]
    windows = windows[keep]
    bins = bins_table(cool)
    c1 = bins.iloc[coords.bin1.to_numpy(dtype=int), :].reset_index(drop=True)
    c2 = bins.iloc[coords.bin2.to_numpy(dtype=int), :].reset_index(drop=True)
    coords = coords.reset_index(drop=True)
    too_close = ((c1.chrom == c2.chrom) & (np.abs(c2.start - c1.start) < cfg["min_dist"])).to_numpy()
    coords, windows = coords.loc[~too_close, :], windows[~too_close]
    nanp = coords.pvalue.isnull().to_numpy()
    coords, windows = coords.loc[~nanp, :].copy(), windows[~nanp]
    coords["qvalue"] = cus.fdr_correction(coords["pvalue"])
    log["final"] = coords[OUT_COLS].to_numpy(dtype=np.float64)
    log["final_windows"] = windows
    return log


def make_iterations():
    cool = dict(np.load(HERE / "example_cool.npz", allow_pickle=True))
    out = {}
    loops = dict(pearson=0.3, max_perc_undetected=50.0, max_perc_zero=10.0, max_dist=2000000, min_dist=20000,
                 min_separation=5000, max_iterations=2)
    for k, v in detect_like_cli(cool, loops, [LOOPS], 3).items():
        out[f"loops_{k}"] = v
    loops3 = dict(loops, pearson=0.25, max_iterations=3, max_dist=100000, min_dist=5000)
    for k, v in detect_like_cli(cool, loops3, [LOOPS], 3).items():
        out[f"loops3_{k}"] = v
    # 1-D patterns cannot be iterated in the reference: the windows of intra maps carry NaN on the first
    # sub-diagonals (detection.py:300-310), the pileup keeps them, and normxcorr2 refuses the template
    # (kernel.std() is NaN -> "Cannot have flat kernel.", detection.py:888-889).  Captured as a known answer.
    borders = dict(pearson=0.15, max_perc_undetected=75.0, max_perc_zero=10.0, max_dist=0, min_dist=0,
                   min_separation=5000, max_iterations=2)
    try:
        detect_like_cli(cool, borders, BORDERS, 3)
        out["borders_error"] = np.array("")
    except ValueError as err:
        out["borders_error"] = np.array(str(err))
    print("borders, 2 iterations:", repr(str(out["borders_error"])))
    np.savez_compressed(HERE / "iterations.npz", **out)
    print("iterations:", {k: v.shape for k, v in out.items() if k.endswith("table") or k.endswith("final")})


def make_nonsquare():
    cool = dict(np.load(HERE / "example_cool.npz", allow_pickle=True))
    off = cool["chrom_offset"]
    det_all = np.flatnonzero(np.isfinite(cool["weight"]))
    rng = np.random.default_rng(41)
    out = {}
    ci = 1
    s, e = off[ci], off[ci + 1]
    det = det_all[(det_all >= s) & (det_all < e)] - s
    block = balanced_block(cool, ci, ci)
    # rectangular templates: a central crop of the loops template and a random one (both orientations)
    k59 = LOOPS[6:11, 4:13].copy()
    k95 = np.ascontiguousarray(rng.random((9, 5)) + LOOPS[4:13, 6:11])
    k37 = np.ascontiguousarray(BORDERS[0][7:10, 5:12])
    cases = [
        ("d2_59", k59, dict(pearson=0.3, max_perc_undetected=50.0, max_perc_zero=10.0, max_dist=60 * 1000), None),
        ("d1_37", k37, dict(pearson=0.2, max_perc_undetected=75.0, max_perc_zero=10.0, max_dist=0), None),
        ("q2_59", k59, dict(pearson=0.3, max_perc_undetected=50.0, max_perc_zero=10.0, max_dist=60 * 1000),
         np.array([[10, 40], [100, 130], [200, 205], [1, 3], [300, 360], [418, 421], [150, 150], [3, 1], [420, 421]])),
        ("q1_37", k37, dict(pearson=0.1, max_perc_undetected=75.0, max_perc_zero=10.0, max_dist=0),
         np.array([[10, 10], [100, 100], [5, 5], [200, 203], [419, 419], [300, 300], [1, 1]])),
    ]
    # taller than wide on an intra map: the framed mask flags sub-diagonals that hold signal and the reference
    # raises (detection.py:1022, preprocessing.py:501-526) -- captured as a known answer
    m, _, _ = prepare_intra(block, det, 60, 9)
    try:
        cud.pattern_detector(RefMap(m.copy(), (det.copy(), det.copy()), 60, False),
                             dict(pearson=0.3, max_perc_undetected=50.0, max_perc_zero=10.0, max_dist=60000), k95, full=True)
        out["intra_95_error"] = np.array("")
    except ValueError as err:
        out["intra_95_error"] = np.array(" ".join(str(a) for a in err.args))
    out["intra_95_kernel"] = k95
    print("9x5 on an intra map:", repr(str(out["intra_95_error"])))
    binsize = int(cool["binsize"])
    for tag, kern, cfg, coords in cases:
        max_dist = max(cfg["max_dist"] // binsize, 1)
        m, _, keep = prepare_intra(block, det, max_dist, max(kern.shape))
        cmap = RefMap(m.copy(), (det.copy(), det.copy()), max_dist, False)
        tab, wins = cud.pattern_detector(cmap, dict(cfg), kern, coords=None if coords is None else coords.copy(), full=True)
        out[f"{tag}_kernel"] = kern
        out[f"{tag}_max_dist"] = np.int64(max_dist)
        out[f"{tag}_cfg"] = np.array([cfg["pearson"], cfg["max_perc_undetected"], cfg["max_perc_zero"], cfg["max_dist"]])
        out[f"{tag}_det"] = det
        out[f"{tag}_prepared_row"] = m.tocoo().row.astype(np.int32)
        out[f"{tag}_prepared_col"] = m.tocoo().col.astype(np.int32)
        out[f"{tag}_prepared_val"] = m.tocoo().data
        out[f"{tag}_n"] = np.int64(m.shape[0])
        if coords is not None:
            out[f"{tag}_coords"] = coords
        out[f"{tag}_table"] = np.zeros((0, 4)) if tab is None else tab[["bin1", "bin2", "score", "pvalue"]].to_numpy(dtype=np.float64)
        out[f"{tag}_windows"] = np.zeros((0,) + kern.shape) if tab is None else wins
        print("nonsquare", tag, out[f"{tag}_table"].shape)
    # inter block, 5x9, detect
    blk = balanced_block(cool, 0, 2).tocoo()
    blk.data[np.isnan(blk.data)] = 0.0
    blk.data = blk.data / np.nanmedian(blk.data)
    blk.data[np.isnan(blk.data)] = 0
    blk.eliminate_zeros()
    det_r = det_all[(det_all >= off[0]) & (det_all < off[1])] - off[0]
    det_c = det_all[(det_all >= off[2]) & (det_all < off[3])] - off[2]
    cfg = dict(pearson=0.35, max_perc_undetected=50.0, max_perc_zero=50.0, max_dist=2000000)
    out["inter_cfg"] = np.array([cfg["pearson"], cfg["max_perc_undetected"], cfg["max_perc_zero"], cfg["max_dist"]])
    out["inter_det_rows"], out["inter_det_cols"] = det_r, det_c
    out["inter_prepared_row"], out["inter_prepared_col"] = blk.row.astype(np.int32), blk.col.astype(np.int32)
    out["inter_prepared_val"], out["inter_shape"] = blk.data, np.array(blk.shape)
    for tag, kern in (("inter59", k59), ("inter95", k95)):
        cmap = RefMap(blk.copy(), (det_r.copy(), det_c.copy()), None, True)
        tab, wins = cud.pattern_detector(cmap, dict(cfg), kern, full=True)
        out[f"{tag}_kernel"] = kern
        out[f"{tag}_table"] = np.zeros((0, 4)) if tab is None else tab[["bin1", "bin2", "score", "pvalue"]].to_numpy(dtype=np.float64)
        out[f"{tag}_windows"] = np.zeros((0,) + kern.shape) if tab is None else wins
        print("nonsquare", tag, out[f"{tag}_table"].shape)
    np.savez_compressed(HERE / "nonsquare.npz", **out)


def make_yeast_detect():
    cool = dict(np.load(HERE / "yeast_cool.npz", allow_pickle=True))
    n_chrom = len(cool["chrom_offset"]) - 1
    out = {}
    configs = {
        "loops": (dict(pearson=0.3, max_perc_undetected=50.0, max_perc_zero=10.0, max_dist=2000000, min_dist=20000,
                       min_separation=5000, max_iterations=1), [LOOPS]),
        "borders": (dict(pearson=0.15, max_perc_undetected=75.0, max_perc_zero=10.0, max_dist=0, min_dist=0,
                         min_separation=5000, max_iterations=1), BORDERS),
        "hairpins": (dict(pearson=0.1, max_perc_undetected=75.0, max_perc_zero=10.0, max_dist=0, min_dist=0,
                          min_separation=5000, max_iterations=1), [HAIRPIN]),
    }
    for name, (cfg, kernels) in configs.items():
        log = detect_like_cli(cool, cfg, kernels, n_chrom)
        for k, v in log.items():
            if k.endswith("_kernel") or k == "final_windows":
                continue
            out[f"{name}_{k}"] = v
        print("yeast", name, log["final"].shape)
    np.savez_compressed(HERE / "yeast_detect.npz", **out)


def coords_to_bins(bins, chroms, pos, binsize):
    """Whole-genome bin of (chrom, pos), NaN when no bin starts at floor(pos / binsize) * binsize
    (HicGenome.coords_to_bins, contacts_map.py:420-450)."""
    index = {(c, int(s)): i for i, (c, s) in enumerate(zip(bins.chrom, bins.start))}
    snapped = (np.asarray(pos) // binsize) * binsize
    return np.array([index.get((str(c), int(p)), np.nan) for c, p in zip(chroms, snapped)], dtype=np.float64)


def make_quantify_select():
    cool = dict(np.load(HERE / "yeast_cool.npz", allow_pickle=True))
    g = dict(np.load(HERE / "yeast_quantify.npz", allow_pickle=True))
    off = cool["chrom_offset"]
    names = [str(n) for n in cool["chrom_names"]]
    binsize = int(cool["binsize"])
    bins = bins_table(cool)
    rows, per_kernel = [], [[], [], []]
    for bi in range(int(g["n_blocks"])):
        ca, cb = (int(x) for x in g[f"b{bi}_chroms"])
        coords = g[f"b{bi}_coords"]
        sc = np.full((coords.shape[0], 3), np.nan)
        pv = np.full((coords.shape[0], 3), np.nan)
        for ki in range(3):
            key = f"b{bi}_k{ki}_table"
            if key in g and g[key].shape[0]:
                sc[:, ki], pv[:, ki] = g[key][:, 2], g[key][:, 3]
        for t, (r, c) in enumerate(coords):
            r, c = int(r), int(c)
            # inter positions: three-bin intervals centred on the scored bin where they fit
            wide = ca != cb and r >= 1 and c >= 1 and r + 2 <= off[ca + 1] - off[ca] and c + 2 <= off[cb + 1] - off[cb]
            a1, e1 = ((r - 1) * binsize, (r + 2) * binsize) if wide else (r * binsize, (r + 1) * binsize)
            a2, e2 = ((c - 1) * binsize, (c + 2) * binsize) if wide else (c * binsize, (c + 1) * binsize)
            rows.append((names[ca], a1, e1, names[cb], a2, e2))
            for ki in range(3):
                per_kernel[ki].append((sc[t, ki], pv[t, ki]))
    bed2d = pd.DataFrame(rows, columns=["chrom1", "start1", "end1", "chrom2", "start2", "end2"])
    bed2d_out = []
    for ki in range(3):
        b = bed2d.copy()
        b["score"] = [x[0] for x in per_kernel[ki]]
        b["pvalue"] = [x[1] for x in per_kernel[ki]]
        bed2d_out.append(b)
    # the selection of cmd_quantify (cli/chromosight.py:430-470): ascending sort by score, last row of each
    # (chrom1, start1, chrom2, start2) group, bins of the interval starts, q-values, sort by bins
    bed = pd.concat(bed2d_out, axis=0).reset_index(drop=True)
    bed = bed.sort_values("score", ascending=True).groupby(["chrom1", "start1", "chrom2", "start2"], sort=False).tail(1)
    picked = bed.index.to_numpy()
    bed = bed.reset_index(drop=True)
    bed["bin1"] = coords_to_bins(bins, bed.chrom1, bed.start1, binsize)
    bed["bin2"] = coords_to_bins(bins, bed.chrom2, bed.start2, binsize)
    bed["qvalue"] = cus.fdr_correction(bed["pvalue"])
    bad = np.isnan(bed.score)
    bed.loc[bad, "pvalue"] = np.nan
    bed.loc[bad, "qvalue"] = np.nan
    order = bed.sort_values(["bin1", "bin2"], ascending=True).index.to_numpy()
    bed = bed.loc[order].reset_index(drop=True)
    out = {
        "positions_chrom1": bed2d.chrom1.to_numpy(dtype=str), "positions_chrom2": bed2d.chrom2.to_numpy(dtype=str),
        "positions_num": bed2d[["start1", "end1", "start2", "end2"]].to_numpy(dtype=np.int64),
        "final_chrom1": bed.chrom1.to_numpy(dtype=str), "final_chrom2": bed.chrom2.to_numpy(dtype=str),
        "final_num": bed[["start1", "end1", "start2", "end2", "bin1", "bin2"]].to_numpy(dtype=np.int64),
        "final_val": bed[["score", "pvalue", "qvalue"]].to_numpy(dtype=np.float64),
        "picked_template": (picked[order] // len(bed2d)).astype(np.int64),
    }
    np.savez_compressed(HERE / "yeast_quantify_select.npz", **out)
    print("quantify selection:", len(bed2d), "positions ->", len(bed), "rows;",
          int(np.count_nonzero(bed2d.end1 - bed2d.start1 > binsize)), "wide intervals")


def make_inter_detect():
    """pattern_detector in DETECT mode on inter-chromosomal blocks of the yeast map (`detect --inter`, cli/chromosight.py:601-614
    with ContactMap.inter: NaN -> 0, divided by the median of the stored values, contacts_map.py:598-601): the loops template
    with a low threshold so that every block yields patterns; tables (block-local bins) and windows."""
    cool = dict(np.load(HERE / "yeast_cool.npz", allow_pickle=True))
    off = cool["chrom_offset"]
    det_all = np.flatnonzero(np.isfinite(cool["weight"]))
    cfg = dict(pearson=0.12, max_perc_undetected=60.0, max_perc_zero=95.0, max_dist=2000000, min_dist=20000, min_separation=5000,
               max_iterations=1)
    pairs = [(3, 6), (1, 12), (11, 14), (6, 15), (0, 16), (9, 10)]
    out_cfg = np.array([cfg["pearson"], cfg["max_perc_undetected"], cfg["max_perc_zero"]])
    out = {"pairs": np.array(pairs), "kernel": LOOPS, "cfg": out_cfg}
    for (ca, cb) in pairs:
        blk = balanced_block(cool, ca, cb).tocoo()
        blk.data[np.isnan(blk.data)] = 0.0
        blk.data = blk.data / np.nanmedian(blk.data)
        blk.data[np.isnan(blk.data)] = 0
        blk.eliminate_zeros()
        det_r = det_all[(det_all >= off[ca]) & (det_all < off[ca + 1])] - off[ca]
        det_c = det_all[(det_all >= off[cb]) & (det_all < off[cb + 1])] - off[cb]
        cmap = RefMap(blk.tocsr(), (det_r.copy(), det_c.copy()), None, True)
        tab, wins = cud.pattern_detector(cmap, cfg, LOOPS, full=True)
        tag = f"b{ca}_{cb}"
        out[f"{tag}_table"] = np.zeros((0, 4)) if tab is None else tab[["bin1", "bin2", "score", "pvalue"]].to_numpy(dtype=np.float64)
        if blk.shape[0] * blk.shape[1] < 60_000:             # (windows of the two small blocks only: fixture size)
            out[f"{tag}_windows"] = np.zeros((0,) + LOOPS.shape) if tab is None else wins
        print("inter detect", tag, blk.shape, out[f"{tag}_table"].shape)
    np.savez_compressed(HERE / "inter_detect.npz", **out)


def make_options():
    """`detect --smooth-trend` and `detect --tsvd` on the three blocks of data_test/example.cool, loops and borders, per-block
    raw tables: smooth = the isotonic fit of the distance law (preprocessing.py:192-195) inside detrend (contacts_map.py:603-621);
    tsvd = 0.999, the CLI's value (cli/chromosight.py:308, 644), through pattern_detector."""
    cool = dict(np.load(HERE / "example_cool.npz", allow_pickle=True))
    binsize = int(cool["binsize"])
    off = cool["chrom_offset"]
    det_all = np.flatnonzero(np.isfinite(cool["weight"]))
    configs = {
        "loops": (dict(pearson=0.3, max_perc_undetected=50.0, max_perc_zero=10.0, max_dist=2000000, min_dist=20000,
                       min_separation=5000, max_iterations=1), [LOOPS]),
        "borders": (dict(pearson=0.15, max_perc_undetected=75.0, max_perc_zero=10.0, max_dist=0, min_dist=0,
                         min_separation=5000, max_iterations=1), BORDERS),
    }
    out = {}
    for name, (cfg, kernels) in configs.items():
        max_dist = max(cfg["max_dist"] // binsize, 1)
        largest = max(k.shape[0] for k in kernels)
        for ci in range(len(off) - 1):
            s, e = off[ci], off[ci + 1]
            det = det_all[(det_all >= s) & (det_all < e)] - s
            block = balanced_block(cool, ci, ci)
            n = block.shape[0]
            keep = min(max_dist, n) + largest
            plain, _, _ = prepare_intra(block, det, max_dist, largest)
            smooth = cup.detrend(block, max_dist=keep, smooth=True, detectable_bins=det, max_val=10)
            smooth = cup.diag_trim(smooth.tocsr(), keep)
            smooth.data[np.isnan(smooth.data)] = 0
            smooth.eliminate_zeros()
            for ki, kern in enumerate(kernels):
                for tag, m, tsvd in (("smooth", smooth, None), ("tsvd", plain, 0.999)):
                    cmap = RefMap(m.copy(), (det.copy(), det.copy()), max_dist, False)
                    tab, _ = cud.pattern_detector(cmap, cfg, kern, full=True, tsvd=tsvd)
                    key = f"{name}_{tag}_c{ci}_k{ki}"
                    out[key] = np.zeros((0, 4)) if tab is None else tab[["bin1", "bin2", "score", "pvalue"]].to_numpy(dtype=np.float64)
                    print("options", key, out[key].shape)
    np.savez_compressed(HERE / "options.npz", **out)


def make_winsize():
    """`detect --win-size W` (cli/chromosight.py:689-695: every template through resize_kernel(factor=W / size) before the
    genome is split, so the kept distance grows with the template) on the three blocks of data_test/example.cool: loops at
    9 / 21 and borders at 23, per-block raw tables through pattern_detector."""
    cool = dict(np.load(HERE / "example_cool.npz", allow_pickle=True))
    binsize = int(cool["binsize"])
    off = cool["chrom_offset"]
    det_all = np.flatnonzero(np.isfinite(cool["weight"]))
    configs = {
        "loops9": (dict(pearson=0.3, max_perc_undetected=50.0, max_perc_zero=10.0, max_dist=2000000, min_dist=20000,
                        min_separation=5000, max_iterations=1), [LOOPS], 9),
        "loops21": (dict(pearson=0.3, max_perc_undetected=50.0, max_perc_zero=10.0, max_dist=2000000, min_dist=20000,
                         min_separation=5000, max_iterations=1), [LOOPS], 21),
        "borders23": (dict(pearson=0.15, max_perc_undetected=75.0, max_perc_zero=10.0, max_dist=0, min_dist=0,
                           min_separation=5000, max_iterations=1), BORDERS, 23),
    }
    out = {}
    for name, (cfg, kernels, win) in configs.items():
        kernels = [cup.resize_kernel(np.asarray(k, dtype=np.float64), factor=win / k.shape[0], quiet=True) for k in kernels]
        assert all(k.shape == (win, win) for k in kernels)
        max_dist = max(cfg["max_dist"] // binsize, 1)
        largest = max(k.shape[0] for k in kernels)
        for ci in range(len(off) - 1):
            s, e = off[ci], off[ci + 1]
            det = det_all[(det_all >= s) & (det_all < e)] - s
            block = balanced_block(cool, ci, ci)
            plain, _, _ = prepare_intra(block, det, max_dist, largest)
            for ki, kern in enumerate(kernels):
                cmap = RefMap(plain.copy(), (det.copy(), det.copy()), max_dist, False)
                tab, _ = cud.pattern_detector(cmap, cfg, kern, full=True)
                key = f"{name}_c{ci}_k{ki}"
                out[key] = np.zeros((0, 4)) if tab is None else tab[["bin1", "bin2", "score", "pvalue"]].to_numpy(dtype=np.float64)
                print("winsize", key, out[key].shape)
    np.savez_compressed(HERE / "winsize.npz", **out)


def make_resize():
    """resize_kernel / crop_kernel (preprocessing.py:679-807: what --win-size and a resolution other than the template's do to
    the templates before the hot path sees them): the reference's outputs for the built-in templates."""
    out = {}
    kernels = {"loops": LOOPS, "borders0": BORDERS[0], "borders2": BORDERS[2], "hairpin": HAIRPIN}
    for name, kern in kernels.items():
        for fi, factor in enumerate((0.3, 0.5, 0.65, 0.8, 1.0, 1.2, 1.7, 2.5)):
            out[f"{name}_factor{fi}"] = cup.resize_kernel(kern, factor=factor, quiet=True)
            out[f"{name}_factor{fi}_value"] = np.float64(factor)
        for ri, (kres, sres) in enumerate(((2000, 1000), (2000, 5000), (10000, 3200), (5000, 640))):
            out[f"{name}_res{ri}"] = cup.resize_kernel(kern, kernel_res=kres, signal_res=sres, quiet=True)
            out[f"{name}_res{ri}_value"] = np.array([kres, sres])
        for ti, target in enumerate(((11, 11), (7, 7), (13, 9), (3, 3), (17, 17))):
            out[f"{name}_crop{ti}"] = cup.crop_kernel(kern, target)
            out[f"{name}_crop{ti}_value"] = np.array(target)
    np.savez_compressed(HERE / "resize.npz", **out)
    print("resize.npz written", len(out))


def make_nonfinite():
    """normxcorr2 on maps that hold a NaN / an infinite pixel (API misuse: chromosight's own maps are zeroed first,
    contacts_map.py:539-540): every window that holds the pixel comes out 0 (detection.py:1088-1101), the others as if the
    pixel were 0.  Sparse + mask (the pipeline's call) and dense."""
    rng = np.random.default_rng(7)
    n = 90
    base = np.triu(rng.gamma(4, 0.25, size=(n, n)))
    valid = np.flatnonzero(rng.random(n) > 0.04)
    miss = np.ones(n, bool)
    miss[valid] = False
    base[miss, :] = 0
    base[:, miss] = 0
    mask = cup.make_missing_mask((n, n), valid, valid, max_dist=40, sym_upper=True)
    out = {"kernel": LOOPS, "valid": valid}
    for tag, bad, (p, q) in (("nan", np.nan, (30, 35)), ("inf", np.inf, (61, 70)), ("ninf", -np.inf, (12, 12))):
        a = base.copy()
        if miss[p] or miss[q]:
            raise SystemExit("pick another pixel")
        a[p, q] = bad
        c, lp = cud.normxcorr2(sp.csr_matrix(a), LOOPS, max_dist=40, sym_upper=True, full=True, missing_mask=mask,
                               missing_tol=0.75, pval=True)
        out[f"{tag}_in"] = a
        out[f"{tag}_corr"] = c.toarray()
        d = np.abs(rng.gamma(4, 0.25, size=(60, 70)))
        d[20, 33] = bad
        cd, _ = cud.normxcorr2(d, LOOPS, full=False)
        out[f"{tag}_dense_in"] = d
        out[f"{tag}_dense_corr"] = cd
    np.savez_compressed(HERE / "nonfinite.npz", **out)
    print("nonfinite.npz written")


if __name__ == "__main__":
    which = sys.argv[1:] or ["iterations", "nonsquare", "yeast_detect", "quantify_select", "nonfinite", "inter_detect", "options", "resize", "winsize"]
    for name in which:
        globals()[f"make_{name}"]()
