#!/usr/bin/env python3
"""Golden vectors for config C5: `quantify --pattern borders --win-size 11 --inter`
on the real 17-chromosome yeast map of the reference's docs
(/root/reference/docs/notebooks/input/scer_w303_g1_2kb_SRR8769554.cool, positions from
scer_cohesin_peaks.bed2d), produced by IMPORTING THE REFERENCE in the authoring container.

    PYTHONPATH=/root/reference python tests/golden/make_golden_yeast.py

Writes  yeast_cool.npz      the decoded .cool (pixels, ICE weights, offsets; data, not source)
        yeast_quantify.npz  per block: requested coordinates and the reference's
                            pattern_detector(coords=..., full=True) table for the three 11x11 kernels
No --subsample (the committed rad21_g1.tsv used unseeded random subsampling, preprocessing.py:382).
"""
import pathlib
import sys

import numpy as np
import pandas as pd
import scipy.sparse as sp

REF = pathlib.Path("/root/reference")
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REF))
sys.path.insert(0, str(ROOT / "tools"))
import chromosight.utils.detection as cud  # noqa: E402
import chromosight.utils.preprocessing as cup  # noqa: E402
from dump_cool import dump_cool  # noqa: E402

HERE = pathlib.Path(__file__).resolve().parent
INPUT = REF / "docs" / "notebooks" / "input"


class RefMap:
    def __init__(self, matrix, detectable_bins, max_dist, inter):
        self.matrix, self.detectable_bins, self.max_dist, self.inter = matrix, detectable_bins, max_dist, inter
        self.name = "blk"


def block(cool, ca, cb):
    off = cool["chrom_offset"]
    s1, e1, s2, e2 = off[ca], off[ca + 1], off[cb], off[cb + 1]
    b1, b2, w = cool["bin1_id"], cool["bin2_id"], cool["weight"]
    sel = (b1 >= s1) & (b1 < e1) & (b2 >= s2) & (b2 < e2)
    r, c = b1[sel] - s1, b2[sel] - s2
    v = cool["count"][sel] * w[b1[sel]] * w[b2[sel]]
    if ca == cb:
        offd = r != c
        r, c, v = np.concatenate([r, c[offd]]), np.concatenate([c, r[offd]]), np.concatenate([v, v[offd]])
    return sp.coo_matrix((v, (r, c)), shape=(e1 - s1, e2 - s2))


def main():
    cool = dump_cool(INPUT / "scer_w303_g1_2kb_SRR8769554.cool")
    np.savez_compressed(HERE / "yeast_cool.npz", **cool)
    binsize = int(cool["binsize"])
    off = cool["chrom_offset"]
    names = [str(n) for n in cool["chrom_names"]]
    det_all = np.flatnonzero(np.isfinite(cool["weight"]))
    bed = pd.read_csv(INPUT / "scer_cohesin_peaks.bed2d", sep="\t", header=None,
                      names=["chrom1", "start1", "end1", "chrom2", "start2", "end2"])
    furthest = int(np.max(bed.start2 - bed.start1))
    max_diag = int(off[-1]) * binsize
    cfg = dict(pearson=0.15, max_perc_undetected=75.0, max_perc_zero=10.0,
               max_dist=min(furthest, max_diag), min_dist=0)
    max_dist = max(cfg["max_dist"] // binsize, 1)
    tmpl = [np.loadtxt(REF / "chromosight" / "kernels" / f"artificial_template_borders_type{i}.txt") for i in (1, 2, 3)]
    kernels = [cup.resize_kernel(k, factor=11 / 17, quiet=True) for k in tmpl]
    assert all(k.shape == (11, 11) for k in kernels)
    largest = 11
    out = {"max_dist": np.int64(max_dist), "cfg_max_dist_bp": np.int64(cfg["max_dist"])}
    for ki, k in enumerate(kernels):
        out[f"kernel{ki}"] = k
    bed["pos1"] = (bed.start1 + bed.end1) // 2
    bed["pos2"] = (bed.start2 + bed.end2) // 2
    blocks = []
    rng = np.random.default_rng(17)
    for ci, name in enumerate(names):
        rows = bed[(bed.chrom1 == name) & (bed.chrom2 == name)]
        coords = np.stack([rows.pos1.to_numpy() // binsize, rows.pos2.to_numpy() // binsize], axis=1)
        blocks.append((ci, ci, coords))
    for ca, cb in [(0, 9), (0, 16), (9, 12), (12, 15), (15, 16), (3, 10), (9, 16)]:
        n1, n2 = off[ca + 1] - off[ca], off[cb + 1] - off[cb]
        coords = np.stack([rng.integers(0, n1, 40), rng.integers(0, n2, 40)], axis=1)
        blocks.append((ca, cb, coords))
    out["n_blocks"] = np.int64(len(blocks))
    for bi, (ca, cb, coords) in enumerate(blocks):
        s1, e1, s2, e2 = off[ca], off[ca + 1], off[cb], off[cb + 1]
        det_r = det_all[(det_all >= s1) & (det_all < e1)] - s1
        det_c = det_all[(det_all >= s2) & (det_all < e2)] - s2
        m = block(cool, ca, cb)
        if ca == cb:
            keep = min(max_dist, m.shape[0]) + largest
            m = cup.detrend(m, max_dist=keep, smooth=False, detectable_bins=det_r, max_val=10)
            m = cup.diag_trim(m.tocsr(), keep)
            m.data[np.isnan(m.data)] = 0
            m.eliminate_zeros()
            cmap_args = (max_dist, False)
        else:
            m = m.tocoo()
            m.data[np.isnan(m.data)] = 0.0
            m.data = m.data / np.nanmedian(m.data)
            m.data[np.isnan(m.data)] = 0
            m.eliminate_zeros()
            cmap_args = (None, True)
        out[f"b{bi}_chroms"] = np.array([ca, cb])
        out[f"b{bi}_coords"] = coords
        for ki, k in enumerate(kernels):
            if coords.shape[0] == 0:
                continue
            cmap = RefMap(m.copy(), (det_r.copy(), det_c.copy()), *cmap_args)
            tab, _ = cud.pattern_detector(cmap, dict(cfg), k, coords=coords.copy(), full=True)
            if tab is None:
                out[f"b{bi}_k{ki}_table"] = np.zeros((0, 4))
            else:
                out[f"b{bi}_k{ki}_table"] = tab[["bin1", "bin2", "score", "pvalue"]].to_numpy(dtype=np.float64)
    np.savez_compressed(HERE / "yeast_quantify.npz", **out)
    print(len(blocks), "blocks,", max_dist, "max_dist bins")


if __name__ == "__main__":
    main()
