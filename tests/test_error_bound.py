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


# ---- the resident-sum recurrence ---------------------------------
def _bound(which, a, b, c=0.0):
    from fastrank_amd import clib

    return float(clib._load().fr_debug_resident_bound(int(which), float(a), float(b), float(c)))


def _replay(seed, n, d, updates, normalize, col_gen, weight_gen, cand_gen, refresh=None):
    """Replays what the trainer and the verify kernel do to ONE restart's resident sums over a chain of accepted
    candidates -- best_w -> base = best_w / sum|best_w| (each entry divided separately, coordinate_ascent.rs:72-82)
    -> best_w' = base with [f] = cand, R' = fma(x_f, cand, fma(-x_f, base_f, R * fl(1/norm))) -- and holds the host's
    bound E (the product's own constants, through fr_debug_resident_bound) against |R - sum_j x_j best_w_j| in
    extended precision at every step.  Returns (max observed error / E, final E / T)."""
    rng = np.random.default_rng(seed)
    X = col_gen(rng, n, d).astype(np.float32)
    XL = X.astype(np.longdouble)
    colmax = np.abs(X).max(axis=0).astype(np.float64)
    w = weight_gen(rng, d)
    ds = o.Dataset(X, np.zeros(n), np.zeros(n, dtype=np.int64))

    def exact_refresh(w):
        T = float((np.abs(w) * colmax).sum())
        return ds.score_linear(w).copy(), _bound(0, d, T)

    R, E = exact_refresh(w)
    worst = 0.0
    for step in range(updates):
        real = (XL * w.astype(np.longdouble)).sum(axis=1)
        err = float(np.abs(R.astype(np.longdouble) - real).max())
        assert err <= E, (step, err, E)
        worst = max(worst, err / E if E > 0 else 0.0)
        norm = 1.0
        base = w.copy()
        if normalize:
            s = float(np.abs(w).sum())
            if s > 0.0:
                base = w / s          # elementwise IEEE division, like l1_normalize
                norm = s
        f = int(rng.integers(0, d))
        cand = cand_gen(rng, base[f])
        w2 = base.copy()
        w2[f] = cand
        o.resident_update(R, np.ascontiguousarray(X[:, f]), cand, base[f], 1.0 / norm)
        T = abs(base[f]) * colmax[f] + float((np.abs(w2) * colmax).sum())
        E = _bound(1, E, norm, T)
        w = w2
        if refresh and (step + 1) % refresh == 0:
            R, E = exact_refresh(w)
    real = (XL * w.astype(np.longdouble)).sum(axis=1)
    err = float(np.abs(R.astype(np.longdouble) - real).max())
    assert err <= E
    T = float((np.abs(w) * colmax).sum())
    return max(worst, err / E), E / T


def _signed_heavy_tail(rng, n, d):
    X = rng.lognormal(0.0, 2.5, (n, d)) * rng.choice([-1.0, 1.0], (n, d))
    X[:, ::5] = np.floor(rng.exponential(2.0, (n, len(range(0, d, 5)))))   # small-integer columns
    X[:, 3::7] *= 1e-4                                                        # columns of very different scale
    return X


def test_resident_recurrence_long_accept_chain_signed_heavy_tail():
    """>= 1000 accepted updates without any exact refresh (FR_RESIDENT_REFRESH -> infinity), signed heavy-tail
    columns, line-search-sized steps: the bound holds at every step and stays a small multiple of u * T."""
    def weights(rng, d):
        w = rng.uniform(-1, 1, d)
        return w / np.abs(w).sum()

    def cand(rng, orig):
        step = 0.05 * (2.0 ** int(rng.integers(0, 12))) * (abs(orig) if abs(orig) > 0 else 1.0)
        return float(orig + rng.choice([-1.0, 1.0]) * step) if rng.random() > 0.1 else 0.0  # (dir 0: w_f = 0)

    worst, rel = _replay(101, 600, 48, 1200, True, _signed_heavy_tail, weights, cand)
    assert worst <= 1.0
    assert rel < 1e-11, "the bound must stay useful: a few thousand ulps of T after 1200 updates"


def test_resident_recurrence_tiny_and_huge_norms():
    """norm = sum |best_w| far from 1: not normalised runs never rescale (norm = 1 with weights of size 1e6), and a
    chain of dir-0 candidates shrinks the weights until the normaliser divides by ~1e-6."""
    def small(rng, d):
        return rng.uniform(-1, 1, d) * 1e-6 / d

    def big(rng, d):
        return rng.uniform(-1, 1, d) * 1e6

    def cand_small(rng, orig):
        return float(orig * rng.uniform(-2, 2)) * (1e-6 if rng.random() < 0.5 else 1.0)

    def cand_big(rng, orig):
        return float(orig + rng.normal() * 1e6)

    w1, _ = _replay(103, 300, 24, 1000, True, _signed_heavy_tail, small, cand_small)
    w2, _ = _replay(104, 300, 24, 1000, False, _signed_heavy_tail, big, cand_big)
    w3, _ = _replay(105, 300, 24, 1000, True, _signed_heavy_tail, big, cand_big, refresh=256)  # the product's default refresh
    assert max(w1, w2, w3) <= 1.0


def test_candidate_key_error_within_eps_in_resident_form():
    """One level up: the key the verify kernel forms, fma(x_f, w_c, A) with A = fma(-x_f, base_f, R * fl(1/norm)),
    against the reference's ordered sum of the candidate's weights -- within eps_c = gamma * T_c + extra (the numbers
    compute_eps2 uploads, class bits left out here), after a drifted chain of updates."""
    rng = np.random.default_rng(107)
    n, d = 500, 40
    X = _signed_heavy_tail(rng, n, d).astype(np.float32)
    colmax = np.abs(X).max(axis=0).astype(np.float64)
    ds = o.Dataset(X, np.zeros(n), np.zeros(n, dtype=np.int64))
    w = rng.uniform(-1, 1, d)
    R = ds.score_linear(w).copy()
    E = _bound(0, d, float((np.abs(w) * colmax).sum()))
    for step in range(300):
        s = float(np.abs(w).sum())
        base, norm = w / s, s
        f = int(rng.integers(0, d))
        # the candidates of this line search, checked BEFORE one of them is accepted
        A = R * (1.0 / norm)
        A = np.array([np.float64(np.longdouble(a) - np.longdouble(x) * np.longdouble(base[f])) for a, x in zip(A, X[:, f].astype(np.float64))])
        T = float((np.abs(base) * colmax).sum())
        extra = _bound(2, E, norm, T)
        for cand in (0.0, base[f] + 0.05, base[f] - 3.2, base[f] * 1e3):
            wc = base.copy()
            wc[f] = cand
            ref = ds.score_linear(wc)                                  # the reference's number
            key = (np.longdouble(1) * X[:, f].astype(np.longdouble) * np.longdouble(cand) + A.astype(np.longdouble)).astype(np.float64)
            Tc = float((np.abs(np.delete(base, f)) * np.delete(colmax, f)).sum()) + abs(base[f]) * colmax[f] + abs(cand) * colmax[f]
            eps = 2.5 * (d + 1) * U * Tc * (1 + 1e-6) + extra
            assert float(np.abs(key - ref).max()) <= eps, (step, cand)
        cand = base[f] + float(rng.normal()) * 0.1
        w2 = base.copy()
        w2[f] = cand
        o.resident_update(R, np.ascontiguousarray(X[:, f]), cand, base[f], 1.0 / norm)
        E = _bound(1, E, norm, abs(base[f]) * colmax[f] + float((np.abs(w2) * colmax).sum()))
        w = w2
