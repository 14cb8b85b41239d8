"""Statistical helpers of the detection path (reference chromosight/utils/stats.py)."""
import numpy as np
from scipy.special import ndtr


def fdr_correction(pvals):
    """Benjamini-Hochberg q-values (reference stats.py:7-40)."""
    if pvals is None:
        return None
    pvals = np.asarray(pvals, dtype=np.float64)
    n = pvals.size
    order = np.argsort(pvals)[::-1]          # largest p first
    ranks = np.arange(n, 0, -1)              # rank of each sorted p (n .. 1)
    stepped = np.minimum.accumulate(pvals[order] * (float(n) / ranks))
    q_sorted = np.minimum(1, stepped)
    qvals = np.empty(n)
    qvals[order] = q_sorted
    return qvals


def corr_to_pval(corr, n, rho0=0):
    """log10 of the two-sided p-value of Pearson coefficients through Fisher's z:
    log10(2 * Phi(-|atanh(r) - atanh(rho0)| * sqrt(n - 3))) (reference stats.py:43-81)."""
    corr = np.asarray(corr, dtype=np.float64)
    if isinstance(n, (int, np.integer)):
        n = np.repeat(int(n), corr.shape)
    elif isinstance(n, np.ndarray):
        if n.shape != corr.shape:
            raise ValueError("corr and n must have identical shapes.")
    with np.errstate(all="ignore"):
        z = np.arctanh(corr) - np.arctanh(rho0)
        pvals = 2 * ndtr(-np.abs(z * np.sqrt(n - 3)))
        return np.log10(pvals)
