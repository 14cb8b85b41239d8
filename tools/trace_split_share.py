#!/usr/bin/env python3
"""A rank's share of the ONE 200 000-bin block (bench.SplitC4P), `steps` steps without collectives: what rocprofv3 --kernel-trace
turns into a device timeline of the chain behind `north_star_c4p_split` (tools/kernel_timeline.py <dir> mask_prep).
    python tools/trace_split_share.py [world] [rank] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402
import chromosight_amd  # noqa: E402
from chromosight_amd._lib import get_device  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 1
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
chromosight_amd.set_precision("f32")
dev = get_device(0)
w = bench.SplitC4P(dev, rank, world, "f32")
ts = []
for _ in range(steps):
    dev.sync()
    t0 = time.perf_counter()
    w.scan.correlate()
    w.scan.candidates()
    ts.append((time.perf_counter() - t0) * 1e3)
print(f"{world} shares, rank {rank}: step median {np.median(ts[2:]):.4f} ms, device chain {np.mean(w.kernel_ms[2:]):.4f} ms")
