"""Shared parity assertion of the GPU tests: float32 coefficients within the north-star 1e-5 of the
float64 oracle (float64 compute: 1e-10) on every well-defined pixel.

A pixel is ill-defined when one factor of its denominator -- the variance of the window, or the
variance of the template over the present pixels -- is below COND_EPS of its scale
(oracle/oracle.c pixel(): `cond`).  There the float64 value is itself cancellation noise: the
reference's own dense and sparse paths disagree on such windows (tests/test_oracle_golden.py).
Ill-defined pixels are neither dropped nor free: the error of a quotient grows like 1 / cond, so their
error is bounded by tol * COND_EPS / cond (1e-5 at cond = 1e-3, 1e-2 at 1e-6, ...; only windows whose
variance is below 1e-8 of their mean square -- constant to float32 rounding -- may hold anything in
[-1, 1]), and their fraction is bounded in every test (max_ill_frac, default none at all: a test on a
map with flat patches states how many it expects)."""
import numpy as np

TOL = {"f32": 1e-5, "f64": 1e-10}
COND_EPS = 1e-3


def assert_parity(got, want, cond, precision="f32", what="", tol=None, max_ill_frac=0.0):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    cond = np.asarray(cond, dtype=np.float64)
    tol = TOL[precision] if tol is None else tol
    err = np.abs(got - want)
    ill = cond < COND_EPS
    n_ill = int(ill.sum())
    worst = float(err[~ill].max()) if (~ill).any() else 0.0
    worst_ill = float(err[ill].max()) if n_ill else 0.0
    print(f"[parity] {what}: {err.size} px, max|err| {worst:.2e} (tol {tol:g}), "
          f"{n_ill} ill-defined px (cond < {COND_EPS:g}), max|err| there {worst_ill:.2e}")
    assert worst < tol, (what, worst)
    assert n_ill <= max_ill_frac * err.size, (what, n_ill, err.size)
    if n_ill:
        allowed = tol * COND_EPS / np.maximum(cond[ill], 1e-300)
        over = err[ill] > allowed
        assert not over.any(), (what, "ill-defined pixels beyond tol / cond", int(over.sum()), float(err[ill][over].max()))
    assert np.all(np.abs(got) <= 1.0 + 1e-6), what
    return worst
