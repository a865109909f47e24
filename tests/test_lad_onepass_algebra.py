"""CPU: the algebra of the one-pass LAD iteration (admm_amd/csrc/fadmm_dense.hip: dense_head_kernel's `lp` block + lad_rows_kernel),
restated in NumPy beside the reference's two-product iteration (ADMMLAD.h:62-107 under FADMMBase::solve, FADMMBase.h:185-265).

Reference:  vec = d - adj_y / rho + adj_z ;  x = X (X'X)^-1 X' vec ;  z, y elementwise ;  adj_* = accelerate / restart combination.
Library:    X'vec is never formed from vec: the head combines the p-vectors X'd (fixed) and X'z, X'y of the two latest iterates with
            the coefficients the adj vectors are combined with; the rows launch forms x = X s row by row and, from the same rows,
            X'z_new and X'y_new (direct products of the current iterates: nothing accumulates).
Same decisions (accelerate / restart / rho changes / stop) and the same iterates to rounding over a run with restarts and rho changes."""
import numpy as np
import scipy.linalg as sla


def _soft(v, pen):
    return np.where(v > pen, v - pen, np.where(v < -pen, v + pen, 0.0))


def _run(X, d, onepass, maxit=400, eps=1e-7, rho0=1.0):
    n, p = X.shape
    chol = sla.cho_factor(X.T @ X, lower=True)
    z = np.zeros(n); y = np.zeros(n); zo = z.copy(); yo = y.copy()
    adjz = z.copy(); adjy = y.copy()
    Xd = X.T @ d
    Xz = np.zeros(p); Xy = np.zeros(p); Xzo = np.zeros(p); Xyo = np.zeros(p)       # the library's two slots of each
    Xadjz = np.zeros(p); Xadjy = np.zeros(p)
    rho, a, c_old = rho0, 1.0, 9999.0
    codes, rhos, xs = [], [], []
    for i in range(maxit):
        eps_p = max(np.linalg.norm(X @ np.zeros(p)) if i == 0 else np.linalg.norm(x), np.linalg.norm(z), np.linalg.norm(d)) * eps + np.sqrt(n) * eps
        eps_d = np.linalg.norm(y) * eps + np.sqrt(n) * eps
        if onepass:
            u = Xd - Xadjy / rho + Xadjz                       # the head: p-vectors only
        else:
            u = X.T @ (d - adjy / rho + adjz)                  # the reference: a product with X'
        s = sla.cho_solve(chol, u)
        x = X @ s
        zn = _soft(x - d + adjy / rho, 1.0 / rho)
        r = x - d - zn
        yn = adjy + rho * r
        if onepass:                                            # the rows launch: from the same rows of X
            Xzo, Xyo = Xz, Xy
            Xz, Xy = X.T @ zn, X.T @ yn
        rp, rd = np.linalg.norm(r), rho * np.linalg.norm(zn - z)
        zo, yo, z, y = z, y, zn, yn
        xs.append(x)
        if rp < eps_p and rd < eps_d:
            codes.append(0)
            break
        c = rho * rp * rp + rho * np.sum((z - adjz) ** 2)
        if c < 0.999 * c_old:
            a_new = 0.5 + 0.5 * np.sqrt(1 + 4 * a * a)
            t = (a - 1.0) / a_new
            adjz, adjy = (1 + t) * z - t * zo, (1 + t) * y - t * yo
            Xadjz, Xadjy = (1 + t) * Xz - t * Xzo, (1 + t) * Xy - t * Xyo
            a, c_old = a_new, c
            codes.append(1)
        else:
            adjz, adjy = zo, yo
            Xadjz, Xadjy = Xzo, Xyo
            a, c_old = 1.0, c_old / 0.999
            codes.append(2)
        if i > 5:                                              # update_rho (FADMMBase.h:109-133)
            if rp / eps_p > 10 * rd / eps_d: rho *= 2
            elif rd / eps_d > 10 * rp / eps_p: rho /= 2
            if rp < eps_p: rho /= 1.2
            if rd < eps_d: rho *= 1.2
        rhos.append(rho)
    return codes, rhos, xs, sla.cho_solve(chol, X.T @ (d - adjy / rho + adjz))


def test_one_pass_lad_is_the_reference_iteration():
    rng = np.random.default_rng(4)
    for n, p in ((400, 30), (900, 120)):
        X = rng.standard_normal((n, p)) * 2 + 0.3
        d = X @ rng.uniform(size=p) + rng.standard_t(3, size=n) + 1.5
        c2, r2, x2, b2 = _run(X, d, False)
        c1, r1, x1, b1 = _run(X, d, True)
        k = next((i for i, (a, b) in enumerate(zip(c1, c2)) if a != b), min(len(c1), len(c2)))
        # the restart rule's structural ties may part two executions that differ in rounding: compare up to there
        assert k >= 20, (n, p, k)
        assert 2 in c2[:k] and len(set(r2[:k])) >= 2, "the compared stretch holds restarts and rho changes"
        print(f"[lad one-pass algebra] n={n} p={p}: {k} of {len(c2)} decisions compared, {c2[:k].count(2)} restarts, rho values {sorted(set(r2[:k]))}")
        err = max(np.abs(a - b).max() / np.abs(b).max() for a, b in zip(x1[:k], x2[:k]))
        assert err < 1e-10, (n, p, err)
        if k == len(c2) == len(c1):
            assert np.abs(b1 - b2).max() < 1e-9 * np.abs(b2).max()
