#!/usr/bin/env python3
"""Fast (compile-time-size) against general wave-per-window functions on one genome: which record columns differ, by how much."""
import copy, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import chromosight_amd.kernels as ck
from chromosight_amd import parallel, pipeline
from tools.synthetic_genome import make_cool

template = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
cool, _ = make_cool(12_000, 200, 2000, seed=5, template=template)
loops = copy.deepcopy(ck.loops)
loops["max_dist"] = 200 * 2000
for twin in (False, True):
    if twin:
        os.environ["CHROMOSIGHT_HIP_F64_TWIN"] = "1"
    dcool = pipeline.DeviceCool(cool)
    cfgs = [loops, copy.deepcopy(ck.borders)]
    staged = parallel.stage_genome(dcool, cfgs)            # ONE staging (a staging's law is summed with LDS atomics)
    fast = parallel.detect_patterns(dcool, cfgs, staged=staged)
    os.environ["CHROMOSIGHT_HIP_NO_FAST_WINDOWS"] = "1"
    slow = parallel.detect_patterns(dcool, cfgs, staged=staged)
    del os.environ["CHROMOSIGHT_HIP_NO_FAST_WINDOWS"]
    for name, f, s in zip(("loops", "borders"), fast, slow):
        print(f"twin={twin} {name}: shapes {f.shape} {s.shape}")
        if f.shape == s.shape:
            for c in range(f.shape[1]):
                neq = ~((f[:, c] == s[:, c]) | (np.isnan(f[:, c]) & np.isnan(s[:, c])))
                if neq.any():
                    i = np.flatnonzero(neq)[:5]
                    print(f"   column {c}: {neq.sum()} differ, max |diff| {np.nanmax(np.abs(f[:, c] - s[:, c])):.3e}; first rows {i.tolist()}: fast {f[i, c].tolist()} slow {s[i, c].tolist()} bins {f[i, 1:3].tolist()}")
