#!/usr/bin/env python3
"""A rank's share of the ONE 200 000-bin block of the north star (bench.py SplitC4P: the block row-split over the ranks), each
share timed alone on one GPU, without the two collectives: cs_candidates on the rank's row window (the masked tile kernel with the
candidate epilogue -- no coefficient map, no compaction pass -- and the float64 re-scoring of its candidates) -- what
`north_star_c4p_split` does per step between its exchanges.  CS_BENCH_SPLIT_MAP=1: the map + compaction form of round 5.

    python tools/time_c4p_split_share.py            # 1 / 2 / 4 / 8 shares, every rank"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402
import chromosight_amd  # noqa: E402
from chromosight_amd._lib import get_device  # noqa: E402


def main():
    chromosight_amd.set_precision("f32")
    dev = get_device(0)
    steps = int(os.environ.get("CS_STEPS", "40"))
    worst = {}
    for world in (1, 2, 4, 8):
        for rank in range(world):
            w = bench.SplitC4P(dev, rank, world, "f32")
            corr, cand = w.scan.correlate, w.scan.candidates
            for _ in range(5):
                corr()
                n_cand = len(cand())
            dev.sync()
            w.kernel_ms.clear()
            ts = []
            for _ in range(steps):
                t0 = time.perf_counter()
                corr()
                cand()
                ts.append((time.perf_counter() - t0) * 1e3)
            kern = float(np.mean(w.kernel_ms))
            print(f"{world} shares, rank {rank}: rows {w.rows[0]}..{w.rows[1]}, tile kernel {kern:.4f} ms, kernel + candidates "
                  f"(compaction, download) {np.median(ts):.4f} ms (mean {np.mean(ts):.4f}), {n_cand} candidates", flush=True)
            worst[world] = max(worst.get(world, (0, 0)), (float(np.median(ts)), kern))
            del w
    for world in sorted(worst):
        print(f"slowest share at {world}: kernel + candidates {worst[world][0]:.4f} ms ({worst[1][0] / worst[world][0]:.2f}x), "
              f"tile kernel alone {worst[world][1]:.4f} ms ({worst[1][1] / worst[world][1]:.2f}x)")


if __name__ == "__main__":
    main()
