"""The float32 pre-filter of detect mode must never lose a pixel whose exact coefficient passes the threshold
(coordinates are bit-exact with the reference only if the candidate set is a superset of the passing pixels).
A fixed margin below the threshold does not promise that on low-variance windows, where the float32 error of
the quotient grows like 1 / conditioning; the kernels' candidate mode stores an upper bound instead
(cs_device.h: cand_upper_*).  Here: plateaus (value c +- a few 1e-4) carrying a faint copy of the template, so
that the exact coefficients straddle the threshold while float32 evaluates them with errors of 1e-2 and more."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import chromosight_amd
from chromosight_amd.utils import detection as cud

pytestmark = pytest.mark.gpu


def plateau_map(seed, n=700, max_dist=80, k=17, level=1.0, amp=1e-3, small_scale=None):
    """Upper-band map of gamma noise with flat patches: value `level` + amp * (template - mean) / std + noise whose
    size puts the exact coefficient of the patch centre near 0.3.  Returns (dense map, missing flags, centres)."""
    rng = np.random.default_rng(seed)
    kern = chromosight_amd.kernels.loops["kernels"][0]
    kz = (kern - kern.mean()) / kern.std()
    keep = max_dist + k
    ii, jj = np.indices((n, n))
    a = rng.gamma(4, 0.25, size=(n, n)) * (rng.random((n, n)) > 0.1)
    centres = []
    for t, i0 in enumerate(range(40, n - 120, 60)):
        d = 25 + (t * 7) % 40                       # patch centre on diagonal d
        j0 = i0 + d
        half = 20
        lvl = level if small_scale is None or t % 2 == 0 else level * small_scale
        sigma = amp * (2.2 + 0.25 * (t % 9))        # r = 1 / sqrt(1 + (sigma / amp)^2): 0.41 ... 0.23
        patch = lvl * (1.0 + sigma * rng.standard_normal((2 * half + 1, 2 * half + 1)))
        patch[half - 8:half + 9, half - 8:half + 9] += lvl * amp * kz
        a[i0 - half:i0 + half + 1, j0 - half:j0 + half + 1] = patch
        centres.append((i0, j0))
    a[(jj - ii < 0) | (jj - ii > keep)] = 0
    miss = np.zeros(n, dtype=bool)
    miss[rng.choice(n, size=n // 40, replace=False)] = True
    miss[:2] = True
    a[miss, :] = 0
    a[:, miss] = 0
    return a, miss, kern, np.array(centres)


_PLATEAU_CASES = [(1, 1e-3, None), (2, 3e-4, None), (3, 1e-3, 2e-3), (4, 3e-3, None)]
# CS_MARGIN_EXTRA=n: n more seeded cases for an occasional long run (120 were run at the end of round 3: 117 pass, 3 maps are no stress case and skip)
_PLATEAU_CASES += [(100 + k, [1e-3, 3e-4, 3e-3, 1e-4][k % 4], [None, 2e-3, None, 5e-4][k % 4])
                   for k in range(int(os.environ.get("CS_MARGIN_EXTRA", "0")))]


@pytest.mark.parametrize("seed,amp,small", _PLATEAU_CASES)
def test_plateaus_around_the_threshold_give_the_oracles_foci(seed, amp, small):
    from oracle import c_oracle, foci_oracle
    n, max_dist, pearson, tol = 700, 80, 0.3, 0.5
    a, miss, kern, centres = plateau_map(seed, n=n, max_dist=max_dist, amp=amp, small_scale=small)
    want, cond = c_oracle.normxcorr2_rows(a, kern, 0, n, max_dist=max_dist, sym_upper=True, full=True, miss_row=miss,
                                          miss_col=miss, missing_tol=tol)
    ii, jj = np.indices((n, n))
    band = (jj - ii >= 0) & (jj - ii <= max_dist)
    trimmed = np.where(band, want, 0.0)
    # the plateaus do what they are for: passing pixels on windows conditioned 1e-4 and worse
    passing = band & (trimmed >= pearson)
    low = passing & (cond < 1e-4)
    if seed >= 100 and low.sum() <= 10:
        pytest.skip("this seeded map has too few passing pixels on badly conditioned windows to be a stress case")
    assert low.sum() > 10, int(low.sum())
    want_tab = foci_oracle.detect_table(a, trimmed, miss, miss, kern.shape, pearson=pearson, zero_tol=1.0, missing_tol=tol)

    class Map:
        pass
    cmap = Map()
    valid = np.flatnonzero(~miss)
    cmap.matrix, cmap.detectable_bins, cmap.max_dist, cmap.inter = sp.csr_matrix(a), (valid, valid.copy()), max_dist, False
    cfg = dict(pearson=pearson, max_perc_undetected=tol * 100, max_perc_zero=100.0, max_dist=5 * max_dist)
    assert chromosight_amd.get_precision() == "f32"
    tab, _ = cud.pattern_detector(cmap, cfg, kern, full=True)
    got = tab[["bin1", "bin2", "score"]].to_numpy(dtype=np.float64)
    assert got.shape == want_tab.shape, (got.shape, want_tab.shape)
    assert np.array_equal(got[:, :2], want_tab[:, :2])                    # same foci, same maxima, same order
    # scores: float64 on both sides, but on windows conditioned 1e-5 two summation orders differ by ~1e-16 / 1e-5
    assert np.abs(got[:, 2] - want_tab[:, 2]).max() < 1e-7
    # what the float32 map alone would have decided on these windows
    mask = chromosight_amd.utils.preprocessing.make_missing_mask((n, n), valid, valid, max_dist=max_dist, sym_upper=True)
    c32, _ = cud.normxcorr2(sp.csr_matrix(a), kern, max_dist=max_dist, sym_upper=True, full=True, missing_mask=mask,
                            missing_tol=tol)
    c32 = c32.toarray()
    short = passing & (c32 < pearson - 2e-3)
    print(f"seed {seed}: {int(passing.sum())} passing pixels, {int(low.sum())} of them on windows with cond < 1e-4; the float32 "
          f"map puts {int(short.sum())} of the passing pixels more than 2e-3 under the threshold (max float32 error on passing "
          f"pixels {np.abs(c32 - want)[passing].max():.1e}); {len(got)} foci bit-exact")


def test_candidate_mode_is_an_upper_bound():
    """cs_candidates on a float32 map returns exactly the pixels whose float64 coefficient passes (it re-scores the
    candidates): equal to the oracle's thresholded set on a plateau map, row window by row window."""
    from oracle import c_oracle
    from chromosight_amd import engine
    from chromosight_amd._lib import CsMatrix, LAYOUT_DENSE, MASK_BINS, get_device, np_dtype_code
    n, max_dist, pearson, tol = 500, 60, 0.3, 0.5
    a, miss, kern, _ = plateau_map(7, n=n, max_dist=max_dist, amp=1e-3)
    want = c_oracle.normxcorr2(a, kern, max_dist=max_dist, sym_upper=True, full=True, miss_row=miss, miss_col=miss,
                               missing_tol=tol)[0]
    ii, jj = np.indices((n, n))
    band = (jj - ii >= 0) & (jj - ii <= max_dist)
    dev = get_device()
    ld = (n + 15) // 16 * 16
    host = np.zeros((n, ld))
    host[:, :n] = a
    buf = dev.to_device(host)
    sig = CsMatrix(buf.ptr, np_dtype_code(np.float64), LAYOUT_DENSE, ld, 0, 0)
    flags = dev.to_device(miss.astype(np.uint8))
    rows, cols, vals = engine.run_candidates(dev, sig, (n, n), engine.KernelSpec(kern), (0, n), pearson=pearson, lo_diag=0,
                                             hi_diag=max_dist, inter=False, full=True, sym_upper=True, max_dist=max_dist,
                                             mask_mode=MASK_BINS, miss_row=flags, miss_col=flags, missing_tol=tol)
    got = np.zeros((n, n), dtype=bool)
    got[rows, cols] = True
    exp = band & (want >= pearson) & (want != 0)
    edge = band & (np.abs(want - pearson) < 1e-9)              # two float64 summation orders may disagree here
    assert np.array_equal(got & ~edge, exp & ~edge), (int(got.sum()), int(exp.sum()))
    assert np.abs(vals - want[rows, cols]).max() < 1e-7
