#!/usr/bin/env python3
"""Kernel time of the dense (C2-class) correlation call against the number of 64 x 64 tiles: T(tiles) = fixed + per-tile cost.
The fixed part (launch, weight set-up, pipeline fill and drain of the persistent workgroups) is what separates 4096^2 from
16384^2 in the roofline fraction.

    python tools/time_dense_sizes.py [size ...]
"""
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench                                                        # noqa: E402


def main():
    import chromosight_amd
    from chromosight_amd._lib import get_device
    chromosight_amd.set_precision("f32")
    dev = get_device(0)
    sizes = [int(a) for a in sys.argv[1:]] or [512, 1024, 1536, 2048, 3072, 4096, 5120, 6144, 8192, 12288, 16384]
    rows = []
    for n in sizes:
        wl = bench.Workload("c2", dev, 0, "f32", n)
        bench.prewarm(wl.step, dev.sync, 0.1)
        steps = max(10, min(300, int(4e9 / (n * n))))
        best = 1e9
        for _ in range(3):
            _, ms = bench.time_steps(dev, wl.step, dev.sync, steps, 5)
            best = min(best, ms)
        tiles = ((n + 63) // 64) ** 2
        rows.append((n, tiles, best))
        print(f"{n:6d}^2  {tiles:6d} tiles ({tiles / 512:7.2f} per workgroup)  {best * 1e3:9.2f} us  {n * n / best / 1e6:8.1f} Gpixel/s  "
              f"{714.0 * n * n / (best * 1e-3) / 1e12 / 157.3:6.3f} of the FP32 roof", flush=True)
        del wl
    t = np.array([r[1] for r in rows], dtype=np.float64)
    y = np.array([r[2] for r in rows]) * 1e3
    big = t >= 1024
    if big.sum() >= 2:
        a, b = np.polyfit(t[big], y[big], 1)
        print(f"fit over >= 1024 tiles: T = {b:.2f} us + {a * 512:.3f} us per (tile per workgroup)")


if __name__ == "__main__":
    main()
