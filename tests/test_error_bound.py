"""The constant behind the bound-and-verify line search, checked on the CPU: the
reference's sequentially rounded f64 sum s (oracle) and any other summation of the same products lie within
gamma_D * T of the real-number sum, T = sum_j |x_j * w_j|, and the per-candidate constant the host uses
(column maxima instead of the document's own |x_j|) dominates it.  A statistical check of the analysis, not
its proof."""
import numpy as np

from oracle import pyoracle as o

U = 2.0 ** -53


def _real_sum(X, w):
    # exact products (f32 * f64 fits long double only approximately; use Python fractions-free two-sum in longdouble)
    return (X.astype(np.longdouble) * w.astype(np.longdouble)).sum(axis=1)


def test_sequential_sum_error_is_within_gamma_T():
    rng = np.random.default_rng(5)
    n, d = 4000, 136
    X = rng.lognormal(0, 2, (n, d)).astype(np.float32) * rng.choice([-1, 1], (n, d)).astype(np.float32)
    # heavy cancellation: weights of both signs and very different magnitudes
    w = rng.normal(0, 1, d) * 10.0 ** rng.integers(-6, 7, d)
    ds = o.Dataset(X, np.zeros(n), np.zeros(n, dtype=np.int64))
    s = ds.score_linear(w)                      # the reference's ordered, unfused sum
    real = _real_sum(X, w)
    T = (np.abs(X.astype(np.float64)) * np.abs(w)).sum(axis=1)
    err = np.abs(s.astype(np.longdouble) - real).astype(np.float64)
    gamma = (d + 1) * U                         # the half of 2.5 (D + 1) u allotted to the reference's own sum
    assert (err <= gamma * T).all()
    # another association order (numpy's pairwise dot) obeys the same kind of bound
    other = X.astype(np.float64) @ w
    assert (np.abs(other.astype(np.longdouble) - real).astype(np.float64) <= gamma * T).all()
    # so two evaluations differ by at most 2 gamma T <= eps as the host computes it from column maxima
    colmax = np.abs(X).max(axis=0).astype(np.float64)
    T_hat = float((np.abs(w) * colmax).sum())
    eps = 2.5 * (d + 1) * U * T_hat * (1 + 1e-6)
    assert (np.abs(s - other) <= eps).all() and (T <= T_hat * (1 + 1e-12)).all()


def test_bound_is_tight_enough_to_be_useful():
    """eps must stay far below typical score gaps, or everything would go to the exact kernel."""
    rng = np.random.default_rng(6)
    n, d = 2000, 136
    X = rng.uniform(0, 1, (n, d)).astype(np.float32)
    w = rng.uniform(-1, 1, d)
    w /= np.abs(w).sum()
    s = np.sort(X.astype(np.float64) @ w)
    gaps = np.diff(s)
    eps = 2.5 * (d + 1) * U * float((np.abs(w) * np.abs(X).max(axis=0)).sum())
    assert np.median(gaps) > 1e6 * eps
