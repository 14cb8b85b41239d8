import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np
import chromosight_amd
from chromosight_amd.utils import detection as cud
from oracle import pearson_oracle as orc
rng = np.random.default_rng(0)
kern = chromosight_amd.kernels.loops["kernels"][0]
for shape in ((96, 80), (200, 300)):
    sig = rng.gamma(4, 0.25, size=shape).astype(np.float32)
    for full in (False, True):
        got, _ = cud.normxcorr2(sig, kern, full=full)
        want, _ = orc.normxcorr2_oracle(sig.astype(np.float64), kern, full=full)
        err = np.abs(got - want)
        rows = np.flatnonzero(err.max(axis=1) > 1e-4)
        cols = np.flatnonzero(err.max(axis=0) > 1e-4)
        print(shape, "full", full, "max", err.max(), "bad rows", rows[:20], len(rows), "bad cols", cols[:10], len(cols))
