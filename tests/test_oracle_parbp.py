"""CPU: the oracle's restatement of `admm_parbp` (oracle/solvers.py SharingBP; the reference's unbuilt src/TODO/PADMMBP.h on the
current PADMMBase_Master loop).  The reference holds no vector for it (the symbol was never built), so it is pinned on the
problem itself: basis pursuit is a linear programme, and the reference's own serial solver (pinned on the README's numbers in
tests/test_oracle_readme.py) solves the same one."""
import numpy as np
import pytest

from oracle import entry, readme


def test_recovers_the_readme_signal_like_the_serial_solver():
    x, y, bt = readme.bp_data()                                  # README.md:217-246
    ser = entry.admm_bp(x, y, entry.BP_OPTS)
    for N in (1, 2, 3, 4, 7):
        d = {"trace": []}
        o = entry.admm_parbp(x, y, N, dict(entry.BP_OPTS, rho_ratio=1.0), d)
        tr = np.asarray(d["trace"])
        assert o["niter"] <= 1000 and tr[-1, 6] == 1                            # converged
        assert np.abs(o["beta"] - bt).max() < 2e-3                              # the README's BP error range is 1e-3
        assert np.abs(o["beta"] - ser["beta"]).max() < 3e-3
        assert np.array_equal(tr[:, 5] == 1, np.arange(len(tr)) % 10 == 0)      # regular step on 0, 10, 20, ... (PADMMBP.h:49)
        # the partition: N - 1 blocks of p div N columns, the last takes the remainder (PADMMBP.h:150-167)
        s = d["solver"]
        assert [a.shape[1] for a in s.A] == [100 // N] * (N - 1) + [100 // N + 100 % N]
        assert abs(s.rho * np.mean(s.sprad) - 1.0) < 1e-12                      # rho = 1 / (rho_ratio mean sprad), :181-186


def test_optimum_of_the_linear_programme():
    from scipy.optimize import linprog
    rng = np.random.default_rng(11)
    n, p = 40, 120
    A = rng.standard_normal((n, p))
    b = A @ (rng.standard_normal(p) * (rng.uniform(size=p) < 0.08))
    lp = linprog(np.ones(2 * p), A_eq=np.hstack([A, -A]), b_eq=b, bounds=[(0, None)] * (2 * p), method="highs")
    assert lp.status == 0
    for N in (2, 5):
        o = entry.admm_parbp(A, b, N, dict(maxit=20000, eps_abs=1e-6, eps_rel=1e-6, rho_ratio=1.0))
        assert o["niter"] <= 20000
        assert np.linalg.norm(A @ o["beta"] - b) < 1e-3 * np.linalg.norm(b)
        assert abs(np.abs(o["beta"]).sum() / lp.fun - 1) < 1e-3


def test_dual_residual_identity_and_maxit_exit():
    """resid_dual is evaluated as sum ||A_i dx_i||^2 - 2 dr'dS + N ||dr||^2; the direct form sum ||A_i dx_i - dr||^2 agrees."""
    x, y, _ = readme.bp_data()
    from oracle.solvers import SharingBP
    s2 = SharingBP(x, y, 3, 1e-4, 1e-4); s2.init(1.0)
    assert s2.solve(25) == 26                                                   # `return i + 1` after the loop (PADMMBase.h:236)
    # one step by hand
    s3 = SharingBP(x, y, 3, 1e-4, 1e-4); s3.init(1.0); s3.trace = []
    s3.solve(12)
    s4 = SharingBP(x, y, 3, 1e-4, 1e-4); s4.init(1.0); s4.trace = []
    s4.solve(11)
    dr = s3.r - s4.r
    direct = s3.rho * np.sqrt(sum(float(((s3.Ax[i] - s4.Ax[i]) - dr) @ ((s3.Ax[i] - s4.Ax[i]) - dr)) for i in range(3)))
    assert abs(direct / s3.trace[-1][4] - 1) < 1e-10
