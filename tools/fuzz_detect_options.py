"""pipeline.detect through its option combinations (GPU): --inter, --smooth-trend, --tsvd, --subsample, --win-size,
--iterations on the yeast fixture, loops / borders / hairpins: every combination runs twice (same patterns, scores to 1e-11) and the
float32 screen reports the patterns of the float64 mode (same coordinates, scores <= 1e-9).
usage: python tools/fuzz_detect_options.py [n_combinations] [seed]"""
import copy
import itertools
import os
import random
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import chromosight_amd
import chromosight_amd.kernels as ck
from chromosight_amd import pipeline


def main(n=24, seed=0):
    cool = dict(np.load("tests/golden/yeast_cool.npz", allow_pickle=True))
    grid = list(itertools.product(("loops", "borders", "hairpins"), (False, True), (False, True), (None, 0.999),
                                  (None, 0.6), (None, 9, 21), (1, 2)))
    random.Random(seed).shuffle(grid)
    done, rows = 0, 0
    for pattern, inter, smooth, tsvd, sub, win, iters in grid:
        if done >= n:
            break
        cfg = copy.deepcopy(getattr(ck, pattern))
        if cfg["max_dist"] == 0 and iters > 1:
            continue                                # (the reference raises "Cannot have flat kernel." there: DESIGN 2a)
        cfg["max_iterations"] = iters
        opts = dict(inter=inter, smooth=smooth, tsvd=tsvd, subsample=sub, seed=5, win_size=win)
        t0 = time.perf_counter()
        tabs = []
        for precision in ("f32", "f32", "f64"):
            chromosight_amd.set_precision(precision)
            try:
                tabs.append(pipeline.detect(cool, cfg, **opts))
            except ValueError as exc:               # an iterated template with NaN (windows over the first sub-diagonals)
                tabs.append(str(exc))
        chromosight_amd.set_precision("f32")
        if any(isinstance(t, str) for t in tabs):
            assert tabs[0] == tabs[1] == tabs[2] == "Cannot have flat kernel.", (pattern, opts, [t if isinstance(t, str) else len(t) for t in tabs])
            done += 1
            print(f"{pattern:9s} inter={inter!s:5s} smooth={smooth!s:5s} tsvd={tsvd} sub={sub} win={win} it={iters}: {tabs[0]!r} in all modes",
                  flush=True)
            continue
        a, b, c = tabs
        key = ["bin1", "bin2", "kernel_id", "iteration"]
        # (scores to the last bits only: the distance law is a float64 sum whose order varies from run to run)
        assert len(a) == len(b) and (a[key].to_numpy() == b[key].to_numpy()).all(), (pattern, opts, "two float32 runs differ")
        if len(a):
            assert np.abs(a["score"].to_numpy(dtype=float) - b["score"].to_numpy(dtype=float)).max() < 1e-11, (pattern, opts)
        assert len(a) == len(c) and (a[key].to_numpy() == c[key].to_numpy()).all(), (pattern, opts, len(a), len(c))
        if len(a):
            assert np.abs(a["score"].to_numpy(dtype=float) - c["score"].to_numpy(dtype=float)).max() < 1e-9, (pattern, opts)
        done += 1
        rows += len(a)
        print(f"{pattern:9s} inter={inter!s:5s} smooth={smooth!s:5s} tsvd={tsvd} sub={sub} win={win} it={iters}: "
              f"{len(a)} patterns, {time.perf_counter() - t0:.1f} s", flush=True)
    print(f"{done} combinations, {rows} patterns: float32 runs repeat exactly and equal the float64 mode")


if __name__ == "__main__":
    main(*(int(x) for x in sys.argv[1:3]))
