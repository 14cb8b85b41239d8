#!/usr/bin/env python3
"""`chromosight detect --win-size K` end to end on the resident C4 genome (200 000 bins, 23 chromosomes): pipeline.detect with the
loops template zoomed to K x K (reference cli/chromosight.py:365-370 -> preprocessing.py:731-807), final table included -- on the
two-pass matrix-core kernel (default) and on the runtime-size kernel these templates took before round 6 (CHROMOSIGHT_HIP_NO_WIDE=1).
    python tools/time_detect_win_size.py [K ...]"""
import copy
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import chromosight_amd.kernels as ck  # noqa: E402
from chromosight_amd import pipeline  # noqa: E402
from tools.synthetic_genome import make_cool  # noqa: E402


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [17, 21, 33]
    template = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
    cool, _ = make_cool(200_000, 1000, 2000, seed=2, template=template)
    dcool = pipeline.DeviceCool(cool)
    loops = copy.deepcopy(ck.loops)
    loops["max_dist"] = 2_000_000
    for k in sizes:
        for env in ({}, {"CHROMOSIGHT_HIP_NO_WIDE": "1"}):
            if k <= 17 and env:
                continue
            os.environ.update(env)
            try:
                for _ in range(2):
                    table = pipeline.detect(dcool, loops, win_size=None if k == 17 else k)
                ts = []
                for _ in range(5):
                    dcool.dev.sync()
                    t0 = time.perf_counter()
                    table = pipeline.detect(dcool, loops, win_size=None if k == 17 else k)
                    ts.append((time.perf_counter() - t0) * 1e3)
            finally:
                for a in env:
                    del os.environ[a]
            served = dcool.dev.lib.cs_last_kernel(dcool.dev.ctx)
            print(f"detect --pattern loops --win-size {k:2d}: median {np.median(ts):8.2f} ms (min {min(ts):8.2f}), {len(table)} rows"
                  f"   [{' '.join(f'{a}={b}' for a, b in env.items()) or 'default'}; last kernel id on the genome's context {served}]", flush=True)


if __name__ == "__main__":
    main()
