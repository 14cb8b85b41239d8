#!/usr/bin/env python3
"""Timing of DeviceCool.stage_blocks alone on the C4 genome (23 blocks, keep = 1017): python tools/time_stage.py [max_dist]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import chromosight_amd.kernels as ck  # noqa: E402
from chromosight_amd import pipeline  # noqa: E402
from tools.synthetic_genome import make_cool  # noqa: E402

template = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
cool, _ = make_cool(200_000, 1000, 2000, seed=2, template=template)
dcool = pipeline.DeviceCool(cool)
dev = dcool.dev
chroms = list(range(dcool.n_chrom))
for max_dist in ([int(sys.argv[1])] if len(sys.argv) > 1 else [1000, 1]):
    ts = []
    for it in range(8):
        dev.sync()
        t0 = time.perf_counter()
        blocks = dcool.stage_blocks(chroms, max_dist, 17, **({"lazy64": True} if os.environ.get("STAGE_LAZY") else {}))
        dev.sync()
        ts.append((time.perf_counter() - t0) * 1e3)
        del blocks
    keep = max_dist + 17
    nbytes = min(dcool.nnz, dcool.n_bins * (keep + 1)) * 8 * 2 + dcool.n_bins * ((keep + 64) // 64 * 64) * 12
    print(f"max_dist {max_dist}: stage_blocks {np.mean(ts[2:]):.3f} ms (min {min(ts):.3f}); ~{nbytes / 1e9:.2f} GB moved -> {nbytes / min(ts) / 1e6:.0f} GB/s")
