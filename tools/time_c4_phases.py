#!/usr/bin/env python3
"""Host-side timing of the phases of one C4 genome step (bench.py --workload c4) with a device synchronisation after each:
staging and detection of the loops pass, then of the borders pass.  Usage: python tools/time_c4_phases.py [steps]"""
import copy
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import chromosight_amd.kernels as ck  # noqa: E402
from chromosight_amd import pipeline  # noqa: E402
from tools.synthetic_genome import make_cool  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    template = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
    cool, _ = make_cool(200_000, 1000, 2000, seed=2, template=template)
    dcool = pipeline.DeviceCool(cool)
    dev = dcool.dev
    loops = copy.deepcopy(ck.loops)
    loops["max_dist"] = 1000 * 2000
    borders = copy.deepcopy(ck.borders)
    chroms = list(range(dcool.n_chrom))
    acc = {}

    def lap(name, t0):
        dev.sync()
        acc.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)

    for step in range(steps + 2):
        for cfg, tag in ((loops, "loops"), (borders, "borders")):
            max_dist = max(cfg["max_dist"] // dcool.binsize, 1)
            t0 = time.perf_counter()
            blocks = dcool.stage_blocks(chroms, max_dist, 17)
            lap(f"{tag}: stage", t0)
            for ki, kern in enumerate(cfg["kernels"]):
                t0 = time.perf_counter()
                res = pipeline.detect_blocks(dcool, blocks, cfg, kern, raw=True, want_windows=False)
                lap(f"{tag}: detect", t0)
            del blocks
    for k, v in acc.items():
        v = v[2 * (len(v) // (steps + 2)):]          # drop the two warm-up steps
        print(f"{k:20s} {np.mean(v):8.3f} ms  (min {np.min(v):.3f}, n {len(v)})")


if __name__ == "__main__":
    main()
