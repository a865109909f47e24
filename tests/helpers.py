import numpy as np


def relerr(a, b):
    """Norm-wise relative error per SURVEY.md section 8c: max|a-b| / max|b| (intercept included)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = max(np.abs(b).max(), 1e-300)
    return np.abs(a - b).max() / den


def synth_lasso(n, p, m, seed=123, sd=2.0):
    """README.md:195-201 recipe scaled: X ~ N(0, sd^2), beta* = U(0,1) on the first m, y = X beta* + N(0,1)."""
    rng = np.random.default_rng(seed)
    b = np.concatenate([rng.uniform(size=m), np.zeros(p - m)])
    x = rng.standard_normal((n, p)) * sd
    y = x @ b + rng.standard_normal(n)
    return x, y
