"""Quantify mode against detect mode on the same staged blocks (GPU): the score quantify reports for a position is the
coefficient detect reported there (the reference's pattern_detector takes both from one coefficient map:
detection.py:849-916), for the built-in templates and for templates resized with --win-size (pipeline.with_win_size),
both precisions.  usage: python tools/check_quantify_vs_detect.py [seed]"""
import copy
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import chromosight_amd
import chromosight_amd.kernels as ck
from chromosight_amd import pipeline
from tools.synthetic_genome import make_cool


def main(seed=7):
    sizes = [2500, 900, 300, 40]
    template = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
    cool, _ = make_cool(sum(sizes), 150, 2000, seed=seed, template=template, chrom_sizes=sizes)
    dcool = pipeline.DeviceCool(cool)
    worst, n_checked = 0.0, 0
    for precision in ("f32", "f64"):
        chromosight_amd.set_precision(precision)
        for pattern in ("loops", "borders", "hairpins"):
            for win in (None, 9, 15, 21, 33):
                cfg = copy.deepcopy(getattr(ck, pattern))
                if pattern == "loops":
                    cfg["max_dist"] = 150 * 2000
                cfg = pipeline.with_win_size(cfg, win)
                md = max(cfg["max_dist"] // 2000, 1)
                largest = max(np.shape(k)[0] for k in cfg["kernels"])
                for ci in range(dcool.n_chrom):
                    if sizes[ci] <= largest:
                        continue
                    blk = dcool.stage_intra(ci, md, largest, resident=True)
                    for kern in cfg["kernels"]:
                        kern = np.asarray(kern, dtype=np.float64)
                        tab, wins = pipeline.detect_block(dcool, blk, cfg, kern, raw=True)
                        if tab is None or not len(tab):
                            continue
                        coords = tab[:, :2].astype(int)
                        q, qwins = pipeline.detect_block(dcool, blk, cfg, kern, coords=coords.copy(), raw=True)
                        assert q.shape[0] == coords.shape[0], (pattern, win, ci)
                        assert np.array_equal(q[:, :2], tab[:, :2]), (pattern, win, ci)
                        err = float(np.abs(q[:, 2] - tab[:, 2]).max())
                        tol = 1e-9 if precision == "f64" else 2e-5       # quantify scores in float32 mode are the float32 map's
                        assert err < tol, (precision, pattern, win, ci, err)
                        assert np.allclose(qwins, wins, rtol=0, atol=1e-9, equal_nan=True), (pattern, win, ci)
                        worst = max(worst, err)
                        n_checked += coords.shape[0]
    chromosight_amd.set_precision("f32")
    print(f"quantify == detect at {n_checked} positions, worst score deviation {worst:.2e}")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
