"""Seeded random sweep of normxcorr2 configurations on the GPU against the C oracle: template
sizes (fast and generic kernels, rectangular), dense / banded / inter maps, full / valid mode,
max_dist below, around and above N, missing-bin clusters at the matrix ends, explicit masks versus
per-bin flags, both precisions."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import chromosight_amd
from chromosight_amd.utils import detection as cud
from chromosight_amd.utils import preprocessing as cup
from oracle import c_oracle
from parity_util import assert_parity

# 25-40 % of the pixels of these maps are exact zeros and 5-6 % of the bins are missing: a few windows have (almost) no
# signal or no present template pixel left.  What is allowed of them: tests/parity_util.py.
ILL_MASKED = 0.05
ILL_SPARSE = 0.01

pytestmark = pytest.mark.gpu


def random_kernel(rng, km, kn):
    k = rng.random((km, kn)) + 0.5
    if rng.random() < 0.3:          # piecewise-constant, borders-like
        k = np.where(rng.random((km, kn)) < 0.5, 0.5, 1.5)
        k[0, 0], k[-1, -1] = 0.5, 1.5
    return k


# CS_SWEEP_FROM / CS_SWEEP_TO widen the sweep for an occasional long run (seeds 24 .. 1523 were run at the end of round 3: all pass)
@pytest.mark.parametrize("seed", range(int(os.environ.get("CS_SWEEP_FROM", "0")), int(os.environ.get("CS_SWEEP_TO", "24"))))
def test_random_configuration(seed):
    rng = np.random.default_rng(1000 + seed)
    precision = "f64" if seed % 3 == 0 else "f32"
    old = chromosight_amd.get_precision()
    chromosight_amd.set_precision(precision)
    try:
        ksz = [7, 9, 11, 13, 15, 17, 5, 19][seed % 8]
        km, kn = (ksz, ksz) if seed % 5 else (ksz, ksz + 2)     # every 5th: rectangular -> generic kernel
        kern = random_kernel(rng, km, kn)
        if seed % 7 in (1, 2, 5):       # vertically symmetric (loops-like): folded template rows on the device
            kern = (kern + kern[::-1, :]) / 2
        mode = seed % 4
        if mode == 0:      # dense, no mask
            shape = (int(rng.integers(60, 400)), int(rng.integers(60, 400)))
            sig = rng.gamma(3, 0.4, size=shape) * (rng.random(shape) > 0.3)
            full = bool(seed & 1)
            got, _ = cud.normxcorr2(sig, kern, full=full)
            want, cond = c_oracle.normxcorr2_rows(sig, kern, 0, shape[0], full=full)
            assert_parity(got, want, cond, precision, f"seed {seed} dense", max_ill_frac=ILL_SPARSE)
        elif mode in (1, 2):   # intra band with missing bins
            n = int(rng.integers(80, 900))
            max_dist = int([rng.integers(1, 8), rng.integers(8, 60), rng.integers(60, n + 50)][seed % 3])
            keep = min(max_dist, n) + max(km, kn)
            ii, jj = np.indices((n, n))
            a = rng.gamma(3, 0.4, size=(n, n)) * (rng.random((n, n)) > 0.25)
            a[(jj - ii < 0) | (jj - ii > keep)] = 0
            miss = rng.random(n) < 0.06
            miss[:3] = True
            miss[-2:] = True
            a[miss, :] = 0
            a[:, miss] = 0
            tol = float(rng.choice([0.25, 0.5, 0.75]))
            valid = np.flatnonzero(~miss)
            mask = cup.make_missing_mask((n, n), valid, valid, max_dist=max_dist, sym_upper=True)
            got, _ = cud.normxcorr2(sp.csr_matrix(a), kern, max_dist=max_dist, sym_upper=True, full=True,
                                    missing_mask=mask, missing_tol=tol)
            want, cond = c_oracle.normxcorr2_rows(a, kern, 0, n, max_dist=max_dist, sym_upper=True, full=True,
                                                  miss_row=miss, miss_col=miss, missing_tol=tol)
            assert_parity(got.toarray(), want, cond, precision, f"seed {seed} band n={n} max_dist={max_dist}", max_ill_frac=ILL_MASKED)
        else:              # inter block
            shape = (int(rng.integers(40, 300)), int(rng.integers(40, 300)))
            a = rng.gamma(3, 0.4, size=shape) * (rng.random(shape) > 0.4)
            mr, mc = rng.random(shape[0]) < 0.05, rng.random(shape[1]) < 0.05
            a[mr, :] = 0
            a[:, mc] = 0
            mask = cup.make_missing_mask(shape, np.flatnonzero(~mr), np.flatnonzero(~mc), sym_upper=False)
            got, _ = cud.normxcorr2(sp.csr_matrix(a), kern, sym_upper=False, full=True, missing_mask=mask)
            want, cond = c_oracle.normxcorr2_rows(a, kern, 0, shape[0], sym_upper=False, full=True, miss_row=mr,
                                                  miss_col=mc)
            assert_parity(got.toarray(), want, cond, precision, f"seed {seed} inter", max_ill_frac=ILL_MASKED)
    finally:
        chromosight_amd.set_precision(old)
