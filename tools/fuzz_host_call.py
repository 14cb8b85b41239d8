#!/usr/bin/env python3
"""The pipelined host call (cs_normxcorr2_host, arrays of >= 1 Mpixel) on random shapes, dtypes, modes and templates against
the C oracle on three row windows each.  python tools/fuzz_host_call.py [cases]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import chromosight_amd.kernels as ck
from chromosight_amd.utils import detection as cud
from oracle import c_oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(11)
loops = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
worst = 0.0
for case in range(cases):
    ms = int(rng.integers(700, 2600)); ns = int(rng.integers(max(1 << 20, 1) // ms + 1, 2600))
    k = int(rng.choice([7, 9, 11, 13, 15, 17]))
    kern = loops[:k, :k] + rng.normal(0, 0.05, size=(k, k)) if rng.random() < 0.5 else loops[(17 - k) // 2:(17 - k) // 2 + k, (17 - k) // 2:(17 - k) // 2 + k]
    if rng.random() < 0.2 and k >= 9:
        kern = kern[:, :k - 2]                                     # rectangular
    dtype = np.float32 if rng.random() < 0.6 else np.float64
    full = bool(rng.random() < 0.4)
    sym = bool(full and ms == ns and rng.random() < 0.5)
    a = rng.gamma(4.0, 0.25, size=(ms, ns)).astype(dtype)
    a *= float(rng.choice([1.0, 1e-3, 1e4]))
    assert a.size >= (1 << 20)
    got, _ = cud.normxcorr2(a, kern, full=full, sym_upper=sym)
    assert got.shape == a.shape and np.isfinite(got).all()
    for r0 in (0, ms // 2 - 20, ms - 41):
        want, cond = c_oracle.normxcorr2_rows(a.astype(np.float64), kern, r0, r0 + 40, full=full, sym_upper=sym)
        ok = cond > 1e-3
        d = float(np.abs(got[r0:r0 + 40] - want)[ok].max())
        worst = max(worst, d)
        assert d < 1e-5, (case, (ms, ns), kern.shape, dtype, full, sym, r0, d)
print(f"{cases} cases pass, worst deviation {worst:.2e}")
