"""GPU: randomised small problems for the two solvers built in round 3 from the reference's unbuilt sources -- admm_hip_parbp
(column-block sharing basis pursuit) and admm_hip_dantzig -- against their oracle restatements decision by decision.  Double
arithmetic on both sides: identical iteration counts, every recorded threshold / residual to 1e-7, coefficients to 1e-8 of
their largest entry.  Shapes, block counts, scalings, flags, tolerances and iteration caps are drawn per case (fixed seeds)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rel(a, b, floor=1e-300):
    """Largest difference relative to the largest reference entry (or to `floor`: a residual column that is exactly zero in one
    execution and 1e-16 in the other -- the null model at lambda_max -- has no scale of its own)."""
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(float(np.abs(np.asarray(b)).max()), floor))


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_fuzz_sharing_basis_pursuit(seed):
    import admm_amd
    from oracle import entry
    rng = np.random.default_rng(seed)
    worst = dict(trace=0.0, beta=0.0)
    ncase = 14
    for c in range(ncase):
        n = int(rng.integers(8, 140))
        p = int(rng.integers(n + 3, 4 * n + 8))
        N = int(rng.integers(2, 7))
        scale = float(rng.choice([0.05, 1.0, 1.0, 30.0]))
        A = np.asfortranarray(rng.standard_normal((n, p)) * scale + (rng.uniform(-1, 1) * scale if rng.uniform() < 0.3 else 0.0))
        k = int(rng.integers(1, max(2, n // 3)))
        b0 = np.zeros(p); b0[rng.choice(p, k, replace=False)] = rng.standard_normal(k) * rng.choice([0.1, 1.0, 10.0])
        b = A @ b0 + (1e-3 * rng.standard_normal(n) if rng.uniform() < 0.3 else 0.0)
        eps = float(rng.choice([1e-3, 1e-4, 1e-6]))
        maxit = int(rng.choice([60, 400, 3000]))
        ratio = float(rng.choice([0.5, 1.0, 1.0, 3.0]))
        fit = admm_amd.admm_bp(A, b).parallel(N).opts(maxit=maxit, eps_abs=eps, eps_rel=eps, rho=ratio).fit(trace=True)
        d = {"trace": []}
        ref = entry.admm_parbp(A, b, N, dict(maxit=maxit, eps_abs=eps, eps_rel=eps, rho_ratio=ratio), d)
        tr = np.asarray(d["trace"], dtype=np.float64)
        t = fit.trace[1:]
        label = f"seed {seed} case {c}: n={n} p={p} N={N} scale={scale} eps={eps} maxit={maxit} rho_ratio={ratio}"
        assert fit.niter == ref["niter"], (label, fit.niter, ref["niter"])
        assert len(t) == len(tr), label
        e = max(_rel(t[:, 2], tr[:, 1]), _rel(t[:, 3], tr[:, 2]), _rel(t[:, 4], tr[:, 3]), float(np.abs(t[:, 5] - tr[:, 4]).max() / max(tr[:, 4].max(), 1e-300)))
        eb = _rel(fit.beta.toarray().ravel(), ref["beta"])
        assert e < 1e-7 and eb < 1e-8, (label, e, eb)
        assert np.array_equal(t[:, 11], tr[:, 5]) and np.array_equal(t[:, 8] == 0, tr[:, 6] == 1), label
        worst["trace"], worst["beta"] = max(worst["trace"], e), max(worst["beta"], eb)
    print(f"[fuzz parbp seed {seed}] {ncase} cases: iteration counts identical, trace within {worst['trace']:.1e}, coefficients within {worst['beta']:.1e}")


@pytest.mark.parametrize("seed", [21, 22, 23])
def test_fuzz_dantzig(seed):
    import admm_amd
    from oracle import entry
    rng = np.random.default_rng(seed)
    worst = dict(trace=0.0, beta=0.0)
    ncase = 10
    conv = 0
    for c in range(ncase):
        p = int(rng.integers(3, 60))
        n = int(rng.integers(4, 6 * p + 20))                     # tall and wide: the operator switches form at n > p
        scale = float(rng.choice([0.1, 1.0, 5.0]))
        x = np.asfortranarray(rng.standard_normal((n, p)) * scale + (rng.uniform(-2, 2) if rng.uniform() < 0.3 else 0.0))
        k = int(rng.integers(1, max(2, p // 3)))
        b0 = np.zeros(p); b0[rng.choice(p, k, replace=False)] = rng.standard_normal(k)
        y = x @ b0 + 0.2 * rng.standard_normal(n) + rng.uniform(-1, 1)
        std_, icpt = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        nl = int(rng.integers(1, 5))
        user = rng.uniform() < 0.3
        lam = sorted(rng.uniform(0.01, 1.0, nl).tolist(), reverse=True) if user else None
        ratio = float(rng.choice([0.3, 0.05]))
        maxit = int(rng.choice([40, 250]))
        rho = float(rng.choice([0.01, 1.0])) if rng.uniform() < 0.3 else None
        m = admm_amd.admm_dantzig(x, y, icpt, std_)
        m.penalty(lam, nlambda=nl, lambda_min_ratio=ratio) if user else m.penalty(nlambda=nl, lambda_min_ratio=ratio)
        fit = m.opts(maxit=maxit, rho=rho).fit(trace=True)
        d = {"trace": []}
        ref = entry.admm_dantzig(x, y, lam, nl, ratio, std_, icpt, dict(maxit=maxit, eps_abs=1e-5, eps_rel=1e-5, rho=-1.0 if rho is None else rho), d)
        tr = np.asarray(d["trace"], dtype=np.float64)
        t = fit.trace[1:]
        label = f"seed {seed} case {c}: n={n} p={p} std={int(std_)} icpt={int(icpt)} nl={nl} user={user} maxit={maxit} rho={rho}"
        assert list(fit.niter) == list(ref["niter"]), (label, list(fit.niter), list(ref["niter"]))
        assert len(t) == len(tr) and np.array_equal(t[:, 0], tr[:, 0]) and np.array_equal(t[:, 1], tr[:, 1]) and np.array_equal(t[:, 8], tr[:, 8]), label
        fl = float(np.abs(tr[:, 2]).max())                       # residuals are judged on the scale of the thresholds they are compared with
        # (at lambda_max the primal residual is the clipping of ONE entry by an internal lambda that is lambda_0 up to an ulp: 1e-16 of
        # the threshold, and the two lambda grids may differ in that ulp)
        e = max(_rel(t[:, 2], tr[:, 2]), _rel(t[:, 3], tr[:, 3]), _rel(t[:, 4], tr[:, 4], fl), _rel(t[:, 5], tr[:, 5], fl), _rel(t[:, 10], tr[:, 10]))
        eb = _rel(fit.beta_dense, ref["beta"])
        assert e < 1e-7 and eb < 1e-8, (label, e, eb)
        worst["trace"], worst["beta"] = max(worst["trace"], e), max(worst["beta"], eb)
        conv += int(max(fit.niter) <= maxit)
    print(f"[fuzz dantzig seed {seed}] {ncase} cases ({conv} converged on every lambda): iteration counts identical, trace within {worst['trace']:.1e}, "
          f"coefficients within {worst['beta']:.1e}")
