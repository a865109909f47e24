"""Stepwise parity -- TEST INFRASTRUCTURE (only tests/ and tests/tools/ import this).

The follow-mode rule of tests/helpers.py compares two EXECUTIONS and needs them to stay within rounding distance of each
other over a whole lambda path.  ADMM with Goldstein acceleration does not guarantee that: once a path sits at its
rounding floor (hundreds of iterations in a limit cycle of single-ulp flips of z, c ~ 1e-7) the iteration is not
contractive any more and two correct float executions drift apart -- the soak of profiles/r03_soak_summary.md has such
cases, every one of them late in a long run.  There is a stronger statement that does not care about drift: EVERY SINGLE
ITERATION the library made is the reference's iteration applied to the library's own previous iterates.  With the iterate
dump of admm_hip_lasso_plan_state_* (record s = the vectors trace record s judged) this module replays, for every s,

    FADMMBase::solve (FADMMBase.h:219-265) / PADMMBase_Master::solve (PADMMBase.h:222-237), one iteration at a time:
      adj_z, adj_y   from the decision of record s-1 and the dumped z, y of s-1 and s-2   (FADMMBase.h:243-256)   BIT-EXACT
      x              = (X'X + rho I)^-1 (X'y - adj_y + rho adj_z)                         (ADMMLassoTall.h:70-80)
                     the ONLY step that is not elementwise: compared with the float system solved in double, and its
                     error held against the error the reference's own float Cholesky solve makes on the same right-hand
                     side (consensus: PADMMLasso.h:17-31, Cholesky or Woodbury)
      z              = prox(x + adj_y / rho) from the library's own x                     (ADMMLassoTall.h:55-69,81-85;
                                                                                           ADMMEnet.h:24-45; PADMMLasso.h:99-108)   BIT-EXACT
      y              = adj_y + rho (x - z)                                                (FADMMBase.h:203-211; PADMMBase.h:70-78)  BIT-EXACT
      eps, r_p, r_d, c recomputed from the dumped vectors (ADMMLassoTall.h:141-161; PADMMBase.h:117-138) against the
                     values the library recorded, and the recorded decision against the reference's rule on them -- with
                     the reference's FLOAT norm accumulators (`Scalar r`, ADMMLassoTall.h:108) as well as the library's
                     double ones: a decision that differs between the two is counted (it is a tie inside the rounding
                     of one norm, the only way the reference could decide differently on the same iterates).

Round 4: the same for the solvers that were held by the follow rule alone --
    check_wide    ADMMBase::solve (ADMMBase.h:158-216) with ADMMLassoWide / ADMMEnetWide (ADMMLassoWide.h:70-186, ADMMEnet.h:62-141):
                  dump x | A x | z | y per decision; the two mat-vecs (X't in the x-update, X x for the cache) are held to the
                  first-order yardstick of a float dot product, everything else -- the prox's zero pattern on active-set steps,
                  z, y, the thresholds, residuals, the stopping test, the rho adaptation (ADMMBase.h:85-109) and the regular /
                  active-set schedule (ADMMLassoWide.h:121-155) -- exactly;
    check_dense   FADMMBase::solve with ADMMLAD (ADMMLAD.h:62-107,152-169) / ADMMBP (ADMMBP.h:48-93,138-153), float64: dump
                  x | z | y | adj_z | adj_y; the projection is held against a Householder-QR projection (yardstick: what the
                  reference's own normal-equation solve misses it by), the elementwise steps bit for bit, decisions incl. the rho
                  adaptation (FADMMBase.h:109-133) exactly.

Returns a report; `assert_*` raise AssertionError with the first offending record.
"""
import numpy as np
import scipy.linalg as sla

from .datastd import DataStd
from .entry import _lambda_grid
from .solvers import F, LassoTall, PADMMLasso, _enet_f, _rho_rule, _soft_d, is_regular_update


def _strip(trace):
    t = np.asarray(trace, dtype=np.float64)
    return t


def _rule_float_norm(v):
    """VectorXf::norm() / squaredNorm(): float accumulation (order unspecified: Eigen vectorises; any order is within the
    same rounding, so NumPy's pairwise float sum stands in)."""
    return np.float64(np.sqrt(np.sum(v.astype(F) * v.astype(F), dtype=F)))


def check_tall(problem, trace, state, x_factor=4.0, x_floor_ulps=8.0, label="", system=None):
    """problem: the oracle's arguments (dict x, y, lam, nlambda, lmin_ratio, standardize, intercept, opts, alpha).
    trace: libadmm_hip decision trace INCLUDING the cold-start record; state: (nrec, 5 p) iterate dump of the same run."""
    x = np.asarray(problem["x"], dtype=np.float64)
    y = np.asarray(problem["y"], dtype=np.float64)
    n, p = x.shape
    opts = problem["opts"]
    alpha = problem.get("alpha")
    datX = np.array(x, dtype=F, order="F")
    datY = np.array(y, dtype=F)
    std = DataStd(n, p, problem["standardize"], problem["intercept"], F)
    std.standardize(datX, datY)
    s = LassoTall(datX, datY, float(opts["eps_abs"]), float(opts["eps_rel"]), alpha)
    lam = np.atleast_1d(np.asarray(problem["lam"], dtype=np.float64)) if problem.get("lam") is not None else np.zeros(0)
    if lam.size < 1:
        lam = _lambda_grid(s.lambda0, n, std.scaleY, problem["nlambda"], problem["lmin_ratio"])
    lam_int = np.array([np.float64(F(l * n / np.float64(std.scaleY))) for l in lam])
    s.init(lam_int[0], float(opts["rho"]))
    t = _strip(trace)
    assert t[0, 8] == -1, "the trace must start with the cold-start record"
    rho = float(t[1, 9]) if len(t) > 1 else s.rho
    assert abs(rho - s.rho) <= 1e-5 * s.rho, (label, "rho", rho, s.rho)
    rho_f = F(rho)
    # the float system the x-update solves: the Gram in float with rho added on the diagonal IN FLOAT (XX.diagonal().array() += rho,
    # ADMMLassoTall.h:204).  `system`: that matrix as the LIBRARY formed it (admm_hip_lasso_plan_system_read) -- its Gram
    # rounds differently from NumPy's, and measured against NumPy's system that difference (cond(M) ulps of x) would be booked
    # as an error of the solve
    if system is not None:
        XX = np.asarray(system, dtype=np.float64)
    else:
        XX32 = (datX.T @ datX).astype(F)
        XX32[np.arange(p), np.arange(p)] += F(rho)
        XX = XX32.astype(np.float64)
    chol64 = sla.cho_factor(XX, lower=True, check_finite=False)
    chol32 = sla.cho_factor(XX.astype(F), lower=True, check_finite=False)
    S = np.asarray(state, dtype=F).reshape(len(state), 5, p)
    nrec = min(len(t), len(S))
    # record 0 carries X'y as the library holds it (its summation order differs from NumPy's: a few ulps, which the
    # right-hand side X'y - adj_y amplifies when the two nearly cancel -- early in a path x is 1e-3 of X'y)
    XYg = S[0, 0].copy()
    assert np.abs(XYg - s.XY).max() <= 64 * np.spacing(np.abs(s.XY).max()), (label, "X'y differs from the oracle's beyond summation rounding")
    Minv_abs = np.abs(np.linalg.inv(XX))
    G_abs = np.abs(XX)
    u = float(np.finfo(F).eps) / 2
    eps_abs, eps_rel, sqrt_p = float(opts["eps_abs"]), float(opts["eps_rel"]), np.sqrt(float(p))
    zero = np.zeros(p, F)
    a = 1.0
    rep = dict(records=nrec - 1, x_ratio_max=0.0, x_err_max=0.0, x_ref_err_at_max=0.0, bit_mismatch=[], accum_ties=[], decisions_checked=0,
               norm_rel_max=0.0)
    c_old = 9999.0

    def vecs(k):
        return (zero, zero, zero, zero, zero) if k <= 0 else tuple(S[k, j] for j in range(5))

    for k in range(1, nrec):
        xg, zg, yg, ajz, ajy = vecs(k)
        xp_, zp, yp, ajzp, ajyp = vecs(k - 1)
        _, zpp, ypp, _, _ = vecs(k - 2)
        li, it = int(t[k, 0]), int(t[k, 1])
        lam_k = float(t[k, 11])                              # the library's own internal lambda (its grid comes from ITS max|X'y|, scaleY)
        assert abs(lam_k - np.float64(F(lam_int[li]))) <= 4 * np.spacing(F(lam_int[li])), (label, "lambda", li, lam_k, lam_int[li])
        prev_code = int(t[k - 1, 8])
        reuse = False
        # ---- adj of this iteration from the previous decision (FADMMBase.h:243-256; none on a converged exit :237-238)
        if k == 1:
            ez, ey = zero, zero
        elif prev_code == 1:
            a_new = 0.5 + 0.5 * np.sqrt(1.0 + 4.0 * a * a)
            ratio = (a - 1.0) / a_new
            t1, tt = F(1.0 + ratio), F(ratio)
            ez = (t1 * zp - tt * zpp).astype(F)
            ey = (t1 * yp - tt * ypp).astype(F)
            a = a_new
        elif prev_code == 2:
            ez, ey, a = zpp, ypp, 1.0
        else:                                                # converged: next lambda starts from the stored adj (and re-solves the same system)
            ez, ey = ajzp, ajyp
            reuse = True
        if not (np.array_equal(ez, ajz) and np.array_equal(ey, ajy)):
            rep["bit_mismatch"].append((k, li, it, "adj", int((ez != ajz).sum() + (ey != ajy).sum())))
        # ---- x-update: the library's x against the exact solve of the float system on the float right-hand side
        rhs = (XYg - ajy).astype(F)
        rhs = (rhs.astype(np.float64) + rho * ajz.astype(np.float64)).astype(F)
        xe = sla.cho_solve(chol64, rhs.astype(np.float64), check_finite=False)
        xr = sla.cho_solve(chol32, rhs, check_finite=False).astype(F)
        e_gpu = float(np.linalg.norm(xg.astype(np.float64) - xe))
        e_ref = float(np.linalg.norm(xr.astype(np.float64) - xe))
        # first-order yardstick of ANY float solve of this system: every entry of the right-hand side and of the matrix
        # moved by one unit roundoff,  B = u || |M^-1| (|rhs| + |M| |x|) ||_2  (Higham, Accuracy and Stability, Thm 7.4)
        B = u * float(np.linalg.norm(Minv_abs @ (np.abs(rhs).astype(np.float64) + G_abs @ np.abs(xe))))
        ratio_x = e_gpu / max(B, 1e-300)
        rep["x_vs_ref_max"] = max(rep.get("x_vs_ref_max", 0.0), e_gpu / max(e_ref, B))
        rep["_sg"] = rep.get("_sg", 0.0) + e_gpu ** 2
        rep["_sr"] = rep.get("_sr", 0.0) + e_ref ** 2
        rep["_sb"] = rep.get("_sb", 0.0) + B ** 2
        if ratio_x > rep["x_ratio_max"]:
            rep.update(x_ratio_max=ratio_x, x_err_max=e_gpu, x_ref_err_at_max=e_ref, x_bound_at_max=B, x_worst=(k, li, it))
        if reuse and not np.array_equal(xg, xp_):
            rep["bit_mismatch"].append((k, li, it, "x reused after convergence", int((xg != xp_).sum())))
        # ---- z, y from the library's own x: elementwise, bit for bit
        vec = (xg + ajy / rho_f).astype(F)
        pen = lam_k / rho
        zn = _soft_d(vec, pen, F) if alpha is None else _enet_f(vec, pen, F(alpha))
        if not np.array_equal(zn, zg):
            rep["bit_mismatch"].append((k, li, it, "z", int((zn != zg).sum())))
        r = (xg - zg).astype(F)
        yn = (ajy + rho_f * r).astype(F)
        if not np.array_equal(yn, yg):
            rep["bit_mismatch"].append((k, li, it, "y", int((yn != yg).sum())))
        # ---- thresholds / residuals / decision
        d64 = lambda v: float(np.sqrt(np.sum(v.astype(np.float64) ** 2)))
        eps_p = max(d64(xp_), d64(zp)) * eps_rel + sqrt_p * eps_abs
        eps_d = d64(yp) * eps_rel + sqrt_p * eps_abs
        rp, rd = d64(r), rho * d64(zg - zp)
        c = rho * rp * rp + rho * float(np.sum((zg - ajz).astype(np.float64) ** 2))
        for got, want in ((t[k, 2], eps_p), (t[k, 3], eps_d), (t[k, 4], rp), (t[k, 5], rd)):
            rep["norm_rel_max"] = max(rep["norm_rel_max"], abs(got - want) / max(abs(want), 1e-300))
        code = int(t[k, 8])
        if code != 0:
            rep["norm_rel_max"] = max(rep["norm_rel_max"], abs(t[k, 6] - c) / max(abs(c), 1e-300))
        own = 0 if (rp < eps_p and rd < eps_d) else (1 if c < 0.999 * c_old else 2)
        assert own == code, (label, f"record {k} (lambda {li}, iteration {it}): the library decided {code}, the rule on its own iterates gives {own}",
                             dict(eps_p=eps_p, eps_d=eps_d, rp=rp, rd=rd, c=c, c_old=c_old))
        # the reference's float accumulators on the same iterates
        eps_pf = max(_rule_float_norm(xp_), _rule_float_norm(zp)) * eps_rel + sqrt_p * eps_abs
        eps_df = _rule_float_norm(yp) * eps_rel + sqrt_p * eps_abs
        rpf = _rule_float_norm(r)
        rdf = rho * np.float64(np.sqrt(np.sum((zg - zp).astype(F) ** 2, dtype=F)))
        cf = rho * rpf * rpf + rho * np.float64(np.sum((zg - ajz).astype(F) ** 2, dtype=F))
        ownf = 0 if (rpf < eps_pf and rdf < eps_df) else (1 if cf < 0.999 * c_old else 2)
        if ownf != code:
            rep["accum_ties"].append((k, li, it, code, ownf))
        rep["decisions_checked"] += 1
        if code == 1:
            c_old = c
        elif code == 2:
            c_old = c_old / 0.999
    return _finish(rep)


def check_consensus(problem, trace, state, x_factor=4.0, x_floor_ulps=8.0, label=""):
    """Consensus solver (problem["nthread"] = K row blocks): state records z | x_0 .. x_{K-1} | y_0 .. y_{K-1}."""
    x = np.asarray(problem["x"], dtype=np.float64)
    y = np.asarray(problem["y"], dtype=np.float64)
    n, p = x.shape
    K = int(problem["nthread"])
    opts = problem["opts"]
    datX = np.array(x, dtype=F, order="F")
    datY = np.array(y, dtype=F)
    std = DataStd(n, p, problem["standardize"], problem["intercept"], F)
    std.standardize(datX, datY)
    s = PADMMLasso(datX, datY, K, float(opts["eps_abs"]), float(opts["eps_rel"]))
    lam = np.atleast_1d(np.asarray(problem["lam"], dtype=np.float64)) if problem.get("lam") is not None else np.zeros(0)
    if lam.size < 1:
        lam = _lambda_grid(s.lambda0, n, std.scaleY, problem["nlambda"], problem["lmin_ratio"])
    lam_int = np.array([l * n / np.float64(std.scaleY) for l in lam])
    s.init(lam_int[0], float(opts["rho"]))
    t = _strip(trace)
    assert t[0, 8] == -1, "the trace must start with the cold-start record"
    rho = float(t[1, 9]) if len(t) > 1 else s.rho
    assert abs(rho - s.rho) <= 1e-5 * abs(s.rho), (label, "rho", rho, s.rho)
    rho_f = F(rho)
    m64 = []
    for A in s.A:
        A64 = A.astype(np.float64)
        M = A64.T @ A64
        M[np.arange(p), np.arange(p)] += np.float64(F(rho))
        m64.append(sla.cho_factor(M, lower=True, check_finite=False))
    S = np.asarray(state, dtype=F).reshape(len(state), 1 + 2 * K, p)
    nrec = min(len(t), len(S))
    Abg = [S[0, 1 + w].copy() for w in range(K)]            # record 0: A_k'b_k as the library's workers hold them
    for w in range(K):
        assert np.abs(Abg[w] - s.Ab[w]).max() <= 64 * np.spacing(np.abs(s.Ab[w]).max()), (label, "A_k'b_k differs from the oracle's beyond summation rounding", w)
    Minv_abs = [np.abs(sla.cho_solve(m, np.eye(p), check_finite=False)) for m in m64]
    G_abs = []
    for A in s.A:
        A64 = A.astype(np.float64)
        M = np.abs(A64).T @ np.abs(A64)                     # |A|'|A| >= |A'A|: the products inside the Gram / Woodbury form round too
        M[np.arange(p), np.arange(p)] += np.float64(F(rho))
        G_abs.append(M)
    u = float(np.finfo(F).eps) / 2
    eps_abs, eps_rel = float(opts["eps_abs"]), float(opts["eps_rel"])
    rep = dict(records=nrec - 1, x_ratio_max=0.0, x_err_max=0.0, x_ref_err_at_max=0.0, bit_mismatch=[], accum_ties=[], decisions_checked=0,
               norm_rel_max=0.0)
    zero = np.zeros((1 + 2 * K, p), F)
    d64sq = lambda v: float(np.sum(v.astype(np.float64) ** 2))
    for k in range(1, nrec):
        cur, prev = S[k], (S[k - 1] if k > 1 else zero)
        zg, xg, yg = cur[0], cur[1:1 + K], cur[1 + K:]
        zp, xp_, yp = prev[0], prev[1:1 + K], prev[1 + K:]
        li, it = int(t[k, 0]), int(t[k, 1])
        # ---- workers' x-updates (PADMMLasso.h:17-31): exact solve of the float system on the float right-hand side
        e2g = e2r = b2 = 0.0
        s.x = [v.copy() for v in xp_]
        for w in range(K):
            rhs = (Abg[w] - yp[w]).astype(F)
            rhs = (rhs.astype(np.float64) + rho * zp.astype(np.float64)).astype(F)
            xe = sla.cho_solve(m64[w], rhs.astype(np.float64), check_finite=False)
            A = s.A[w]
            if A.shape[0] >= A.shape[1]:
                xr = s._solve(w, rhs)
            else:
                tt = (A @ rhs).astype(F)
                sv = s._solve(w, tt)
                xr = ((rhs - (A.T @ sv).astype(F)) / rho_f).astype(F)
            e2g += float(np.sum((xg[w].astype(np.float64) - xe) ** 2))
            e2r += float(np.sum((xr.astype(np.float64) - xe) ** 2))
            b2 += float(np.sum((Minv_abs[w] @ (np.abs(rhs).astype(np.float64) + G_abs[w] @ np.abs(xe))) ** 2))
        e_gpu, e_ref, B = float(np.sqrt(e2g)), float(np.sqrt(e2r)), u * float(np.sqrt(b2))
        ratio_x = e_gpu / max(B, 1e-300)
        rep["x_vs_ref_max"] = max(rep.get("x_vs_ref_max", 0.0), e_gpu / max(e_ref, B))
        rep["_sg"] = rep.get("_sg", 0.0) + e_gpu ** 2
        rep["_sr"] = rep.get("_sr", 0.0) + e_ref ** 2
        rep["_sb"] = rep.get("_sb", 0.0) + B ** 2
        if ratio_x > rep["x_ratio_max"]:
            rep.update(x_ratio_max=ratio_x, x_err_max=e_gpu, x_ref_err_at_max=e_ref, x_bound_at_max=B, x_worst=(k, li, it))
        # ---- master next_z (PADMMLasso.h:99-108) and the workers' dual update (PADMMBase.h:70-78): bit for bit
        vec = np.zeros(p, F)
        for w in range(K):
            vec = (vec + (xg[w] + yp[w] / rho_f)).astype(F)
        vec = (vec / F(K)).astype(F)
        lam_k = float(t[k, 11])                              # the library's own internal lambda (its grid comes from ITS max|X'y|, scaleY)
        assert abs(lam_k - lam_int[li]) <= 1e-6 * lam_int[li], (label, "lambda", li, lam_k, lam_int[li])
        zn = _soft_d(vec, lam_k / (rho * K), F)
        if not np.array_equal(zn, zg):
            rep["bit_mismatch"].append((k, li, it, "z", int((zn != zg).sum())))
        coll = 0.0
        for w in range(K):
            r = (xg[w] - zg).astype(F)
            yn = (yp[w] + rho_f * r).astype(F)
            coll += d64sq(r)
            if not np.array_equal(yn, yg[w]):
                rep["bit_mismatch"].append((k, li, it, f"y_{w}", int((yn != yg[w]).sum())))
        # ---- thresholds / residuals / decision (PADMMBase.h:117-138, PADMMLasso.h:149-152)
        xn = sum(d64sq(v) for v in xp_)
        yn2 = sum(d64sq(v) for v in yp)
        eps_p = max(np.sqrt(xn), np.sqrt(d64sq(zp)) * np.sqrt(K)) * eps_rel + np.sqrt(float(p * K)) * eps_abs
        eps_d = np.sqrt(yn2) * eps_rel + np.sqrt(float(p * K)) * eps_abs
        rp = np.sqrt(coll)
        rd = rho * np.sqrt(K * d64sq(zg - zp))
        for got, want in ((t[k, 2], eps_p), (t[k, 3], eps_d), (t[k, 4], rp), (t[k, 5], rd)):
            rep["norm_rel_max"] = max(rep["norm_rel_max"], abs(got - want) / max(abs(want), 1e-300))
        code = int(t[k, 8])
        own = 0 if (rp < eps_p and rd < eps_d) else 1
        assert own == code, (label, f"record {k} (lambda {li}, iteration {it}): the library decided {code}, the rule on its own iterates gives {own}",
                             dict(eps_p=eps_p, eps_d=eps_d, rp=rp, rd=rd))
        fsq = lambda v: np.float64(np.sum(v.astype(F) ** 2, dtype=F))
        xnf, ynf = sum(fsq(v) for v in xp_), sum(fsq(v) for v in yp)
        eps_pf = max(np.sqrt(xnf), np.float64(F(np.sqrt(fsq(zp)))) * np.sqrt(K)) * eps_rel + np.sqrt(float(p * K)) * eps_abs
        eps_df = np.sqrt(ynf) * eps_rel + np.sqrt(float(p * K)) * eps_abs
        rpf = np.sqrt(sum(fsq((xg[w] - zg).astype(F)) for w in range(K)))
        rdf = rho * np.sqrt(K * fsq((zg - zp).astype(F)))
        ownf = 0 if (rpf < eps_pf and rdf < eps_df) else 1
        if ownf != code:
            rep["accum_ties"].append((k, li, it, code, ownf))
        rep["decisions_checked"] += 1
    return _finish(rep)


def _finish(rep):
    """Root-mean-square x-update error over the whole run against the yardstick's and the reference float solve's: the
    per-record maxima above are maxima of a RATIO of two noisy magnitudes (a record where the reference's error happens to
    be small makes it large); these are the errors themselves."""
    sg, sr, sb = rep.pop("_sg", 0.0), rep.pop("_sr", 0.0), rep.pop("_sb", 0.0)
    rep["x_rms_vs_ref"] = float(np.sqrt(sg / sr)) if sr > 0 else 0.0
    rep["x_rms_vs_yardstick"] = float(np.sqrt(sg / sb)) if sb > 0 else 0.0
    return rep


def assert_stepwise(rep, label="", x_factor=4.0, max_accum_tie_rate=0.01, norm_tol=1e-9, x_rms_factor=2.5):
    """The report of check_tall / check_consensus is clean: every elementwise step bit-exact; the x-update's error against
    the exact solve within `x_factor` x the first-order yardstick B = u || |M^-1| (|rhs| + |M| |x|) || of a float solve of
    that system (one unit roundoff on every entry of the data); every recorded threshold / residual equal to the value
    recomputed from the dumped iterates; every decision the rule's; and the decisions that the reference's float norm
    accumulators would have taken differently are rare."""
    assert not rep["bit_mismatch"], (label, "elementwise steps differ from the reference's arithmetic", rep["bit_mismatch"][:8], len(rep["bit_mismatch"]))
    assert rep["x_ratio_max"] <= x_factor, (label, f"x-update error {rep['x_err_max']:.3e} is {rep['x_ratio_max']:.2f} x the float-solve yardstick "
                                            f"({rep.get('x_bound_at_max', 0):.3e}; the reference's own float solve: {rep['x_ref_err_at_max']:.3e}) "
                                            f"at record {rep.get('x_worst')}")
    assert rep.get("x_rms_vs_ref", 0.0) <= x_rms_factor or rep.get("x_rms_vs_yardstick", 0.0) <= 1.0, (
        label, f"x-update error over the run: {rep.get('x_rms_vs_ref', 0):.2f} x the reference float solve's (rms), "
        f"{rep.get('x_rms_vs_yardstick', 0):.2f} x the yardstick")
    assert rep["norm_rel_max"] < norm_tol, (label, "recorded thresholds / residuals differ from the dumped iterates", rep["norm_rel_max"])
    allowed = max(2, int(np.ceil(max_accum_tie_rate * rep["decisions_checked"])))
    assert len(rep["accum_ties"]) <= allowed, (label, "decisions that float norm accumulation would flip", len(rep["accum_ties"]), rep["accum_ties"][:8])
    return rep


def _rounding_tie(own, code, rp, eps_p, rd, eps_d, c, c_old, rel=1e-12):
    """A decision that differs between the library's record and the rule applied to norms recomputed from its dump is excused
    ONLY where the deciding comparison is a tie to `rel` (1e-12: a few thousand double ulps, nine orders below anything a wrong
    iterate would produce): the two sums of squares add up in different orders."""
    if (own == 0) != (code == 0):
        return (abs(rp - eps_p) <= rel * abs(eps_p)) or (abs(rd - eps_d) <= rel * abs(eps_d))
    return abs(c - 0.999 * c_old) <= rel * abs(c_old)


# ---------------------------------------------------------------------------------------------------------------- wide
def check_wide(problem, trace, state, gamma, X=None, Y=None, label=""):
    """Wide Lasso / elastic net (n <= p).  problem: the oracle's arguments; trace: libadmm_hip's decision trace INCLUDING the
    cold-start record; state: (nrec, p + 3 n) iterate dump  x | A x | z | y;  gamma: the library's spectral-radius estimate
    (admm_stats.eig_est -- a float);  X, Y: the standardised data as the LIBRARY holds them (admm_hip_lasso_plan_data_read; its
    DataStd statistics accumulate in double, so its X differs from the oracle's in the last bit -- measured against the oracle's
    X that difference would be booked as an error of the mat-vecs); None: the oracle's own standardisation."""
    x = np.asarray(problem["x"], dtype=np.float64)
    y = np.asarray(problem["y"], dtype=np.float64)
    n, p = x.shape
    opts = problem["opts"]
    alpha = problem.get("alpha")
    datX = np.array(x, dtype=F, order="F")
    datY = np.array(y, dtype=F)
    std = DataStd(n, p, problem["standardize"], problem["intercept"], F)
    std.standardize(datX, datY)
    if X is not None:
        X = np.asarray(X, dtype=F)
        sc = np.abs(datX).max(axis=0) + 1e-30
        assert (np.abs(X - datX).max(axis=0) <= 64 * np.spacing(sc.astype(F))).all(), (label, "the library's standardised X differs from the oracle's beyond rounding")
        datX = X
    if Y is not None:
        Y = np.asarray(Y, dtype=F)
        assert np.abs(Y - datY).max() <= 64 * np.spacing(F(np.abs(datY).max())), (label, "the library's standardised y differs from the oracle's beyond rounding")
        datY = Y
    X64 = datX.astype(np.float64)
    Xabs = np.abs(X64)
    gam_f = F(gamma)
    gam = np.float64(gam_f)
    sqrt_gam = np.float64(F(np.sqrt(gam_f)))                    # (double)std::sqrt(sprad) of a float
    lambda0 = F(np.abs((datX.T @ datY).astype(F)).max())
    if alpha is not None:
        lambda0 = F(lambda0 / (np.float64(F(alpha)) + 0.0001))
    t = _strip(trace)
    assert t[0, 8] == -1, "the trace must start with the cold-start record"
    S = np.asarray(state, dtype=F)
    assert S.shape[1] == p + 3 * n, (label, "record size", S.shape, p, n)
    nrec = min(len(t), len(S))
    u = float(np.finfo(F).eps) / 2
    eps_abs, eps_rel = float(opts["eps_abs"]), float(opts["eps_rel"])
    sqrt_n, sqrt_p = np.sqrt(float(n)), np.sqrt(float(p))
    maxit = int(opts["maxit"])
    rep = dict(records=nrec - 1, bit_mismatch=[], accum_ties=[], decisions_checked=0, norm_rel_max=0.0, xt_ratio_max=0.0, ax_ratio_max=0.0,
               kinds={0: 0, 1: 0, 2: 0}, rho_changes=0)
    d64 = lambda v: float(np.sqrt(np.sum(v.astype(np.float64) ** 2)))
    zero = np.zeros(p + 3 * n, F)
    counter = 0
    sx = sa = sb_x = sb_a = 0.0

    def split(rec):
        return rec[:p], rec[p:p + n], rec[p + n:p + 2 * n], rec[p + 2 * n:]

    for k in range(1, nrec):
        xg, axg, zg, yg = split(S[k])
        xp_, axp, zp, yp = split(S[k - 1] if k > 1 else zero)
        li, it = int(t[k, 0]), int(t[k, 1])
        rho = float(t[k, 9])
        rho_f = F(rho)
        lam = F(t[k, 11])
        kind = int(t[k - 1, 7])
        # ---- the schedule (ADMMLassoWide.h:121-155 / ADMMEnet.h:124-141): what kind of x-update this iteration had to be
        if it == 0:
            counter = 0                                          # init / init_warm
        # (the first lambda of an automatic grid IS lambda_0 up to the rounding of lambda_0 / n * scaleY * n / scaleY, and the
        # library's lambda_0 = max|X'y| comes from ITS summation order: within a few ulps of it the comparison is a coin toss by
        # construction -- the recorded kind is taken, the counter follows it)
        tie = abs(np.float64(lam) - np.float64(lambda0)) <= 16 * np.spacing(np.float64(lambda0).astype(F)) + (1e-5 if alpha is None else 0.0)
        if alpha is None:
            if (np.float64(lam) > np.float64(lambda0) - 1e-5) if not tie else (kind == 0):
                want = 0
            else:
                want = 1 if is_regular_update(counter) else 2
                counter += 1
        else:
            below = (lam < lambda0) if not tie else (kind == 1)
            want = 1 if (is_regular_update(counter) and below) else 2
            counter += 1
        assert kind == want, (label, f"record {k} (lambda {li}, iteration {it}): x-update kind {kind}, the schedule gives {want}")
        rep["kinds"][kind] += 1
        # ---- x-update
        tvec = ((axp + zp).astype(F) + (yp / rho_f).astype(F)).astype(F)             # cache_Ax + aux_z + dual_y / Scalar(rho)
        pen_d = np.float64(lam) / (rho * gam)
        if kind == 0:
            if np.any(xg != 0):
                rep["bit_mismatch"].append((k, li, it, "x must be zero above lambda_0", int((xg != 0).sum())))
        else:
            if kind == 1:
                cols = np.arange(p)
                tt = tvec.astype(np.float64)
                d = X64.T @ tt
                vec = -d / gam + xp_.astype(np.float64)
                if alpha is None:
                    xe = np.where(vec > pen_d, vec - pen_d, np.where(vec < -pen_d, vec + pen_d, 0.0))
                else:
                    th = np.float64(F(np.float64(F(alpha)) * pen_d)); den = np.float64(F(1.0 + pen_d * (1.0 - np.float64(F(alpha)))))
                    xe = np.where(vec > th, (vec - th) / den, np.where(vec < -th, (vec + th) / den, 0.0))
                B = u * (Xabs.T @ np.abs(tt) / gam + 2.0 * np.abs(vec) + np.abs(xp_.astype(np.float64)))
            else:
                cols = np.nonzero(xp_)[0]
                off = np.ones(p, bool); off[cols] = False
                if np.any(xg[off] != 0):                         # active-set step: a zero stays zero (SparseVector: only stored entries are visited)
                    rep["bit_mismatch"].append((k, li, it, "a zero coordinate became non-zero on an active-set step", int((xg[off] != 0).sum())))
                tmp = (tvec / gam_f).astype(F).astype(np.float64)                     # tmp = (...) / gamma, a float vector (:90)
                pen_f = np.float64(F(pen_d))                                          # `const Scalar penalty`
                d = X64[:, cols].T @ tmp
                vec = xp_[cols].astype(np.float64) - d
                if alpha is None:
                    xe = np.where(vec > pen_f, vec - pen_f, np.where(vec < -pen_f, vec + pen_f, 0.0))
                else:
                    th = np.float64(F(F(alpha) * F(pen_d))); den = np.float64(F(1.0 + pen_f * (1.0 - np.float64(F(alpha)))))
                    xe = np.where(vec > th, (vec - th) / den, np.where(vec < -th, (vec + th) / den, 0.0))
                B = u * (Xabs[:, cols].T @ np.abs(tmp) + 2.0 * np.abs(vec) + np.abs(xp_[cols].astype(np.float64)))
            e = float(np.linalg.norm(xg[cols].astype(np.float64) - xe))
            b = float(np.linalg.norm(B))
            sx += e * e; sb_x += b * b
            if e / max(b, 1e-300) > rep["xt_ratio_max"]:
                rep.update(xt_ratio_max=e / max(b, 1e-300), xt_worst=(k, li, it, kind, e, b))
        # ---- cache_Ax = X x  (ADMMLassoWide.h:158-161)
        nz = np.nonzero(xg)[0]
        xa = xg[nz].astype(np.float64)
        axe = X64[:, nz] @ xa
        Ba = u * (Xabs[:, nz] @ np.abs(xa) + np.abs(axe))
        e, b = float(np.linalg.norm(axg.astype(np.float64) - axe)), float(np.linalg.norm(Ba))
        sa += e * e; sb_a += b * b
        if e / max(b, 1e-300) > rep["ax_ratio_max"] and (e > 0):
            rep.update(ax_ratio_max=e / max(b, 1e-300), ax_worst=(k, li, it, int(nz.size), e, b))
        if nz.size == 0 and np.any(axg != 0):
            rep["bit_mismatch"].append((k, li, it, "A x of a zero x", int((axg != 0).sum())))
        # ---- z, y from the library's own A x: elementwise, bit for bit (ADMMLassoWide.h:156-170, ADMMBase.h:176-184)
        zn = (((datY + yp).astype(F) + (rho_f * axg).astype(F)).astype(F) / F(-1.0 - rho)).astype(F)
        if not np.array_equal(zn, zg):
            rep["bit_mismatch"].append((k, li, it, "z", int((zn != zg).sum())))
        r = (axg + zg).astype(F)
        yn = (yp + (rho_f * r).astype(F)).astype(F)
        if not np.array_equal(yn, yg):
            rep["bit_mismatch"].append((k, li, it, "y", int((yn != yg).sum())))
        # ---- thresholds / residuals / decision / rho adaptation
        eps_p = max(d64(axp), d64(zp)) * eps_rel + sqrt_n * eps_abs
        eps_d = sqrt_gam * d64(yp) * eps_rel + sqrt_p * eps_abs
        rp = d64(r)
        rd = rho * sqrt_gam * d64((zg - zp).astype(F))
        for got, want_v in ((t[k, 2], eps_p), (t[k, 3], eps_d), (t[k, 4], rp), (t[k, 5], rd)):
            rep["norm_rel_max"] = max(rep["norm_rel_max"], abs(got - want_v) / max(abs(want_v), 1e-300))
        code = int(t[k, 8])
        own = 0 if (rp < eps_p and rd < eps_d) else 1
        if own != code and _rounding_tie(own, code, rp, eps_p, rd, eps_d, 0.0, 1.0):
            rep.setdefault("rounding_ties", []).append((k, it, code, own))
        else:
            assert own == code, (label, f"record {k} (lambda {li}, iteration {it}): the library decided {code}, the rule on its own iterates gives {own}",
                                 dict(eps_p=eps_p, eps_d=eps_d, rp=rp, rd=rd))
        rpf = _rule_float_norm(r)
        rdf = rho * sqrt_gam * _rule_float_norm((zg - zp).astype(F))
        eps_pf = max(_rule_float_norm(axp), _rule_float_norm(zp)) * eps_rel + sqrt_n * eps_abs
        eps_df = sqrt_gam * _rule_float_norm(yp) * eps_rel + sqrt_p * eps_abs
        if (0 if (rpf < eps_pf and rdf < eps_df) else 1) != code:
            rep["accum_ties"].append((k, li, it, code, 1 - code))
        class _R:
            pass
        q = _R()
        q.rho, q.eps_primal, q.eps_dual, q.resid_primal, q.resid_dual = rho, eps_p, eps_d, rp, rd
        if code != 0 and it > 3:
            _rho_rule(q)                                         # ADMMBase.h:85-109, from i > 3 (:209-210)
        assert abs(q.rho - t[k, 10]) <= 1e-12 * abs(q.rho), (label, f"record {k}: rho after the decision {t[k, 10]}, the rule gives {q.rho}")
        if q.rho != rho:
            rep["rho_changes"] += 1
        if k + 1 < nrec:                                         # what the next iteration runs with
            nxt = t[k + 1]
            fin = code == 0 or it + 1 >= maxit
            assert (int(nxt[0]), int(nxt[1])) == ((li + 1, 0) if fin else (li, it + 1)), (label, f"record {k + 1}: (lambda, iteration) after record {k}", nxt[:2], fin)
            assert abs(nxt[9] - q.rho) <= 1e-12 * abs(q.rho), (label, f"record {k + 1} ran with rho {nxt[9]}, the decision before it left {q.rho}")
        rep["decisions_checked"] += 1
    rep["xt_rms"] = float(np.sqrt(sx / sb_x)) if sb_x > 0 else 0.0
    rep["ax_rms"] = float(np.sqrt(sa / sb_a)) if sb_a > 0 else 0.0
    return rep


def assert_stepwise_wide(rep, label="", mv_factor=4.0, mv_rms=1.0, max_accum_tie_rate=0.01, norm_tol=1e-9):
    """check_wide's report is clean: zero pattern, z, y bit-exact; the two mat-vecs within `mv_factor` x (per record) and
    `mv_rms` x (over the run) the first-order yardstick of a float dot product (one unit roundoff on every term); recorded
    thresholds / residuals equal to the recomputed ones; decisions, rho adaptation and schedule the rule's (asserted inside)."""
    assert not rep["bit_mismatch"], (label, "elementwise steps differ from the reference's arithmetic", rep["bit_mismatch"][:8], len(rep["bit_mismatch"]))
    assert rep["xt_ratio_max"] <= mv_factor, (label, f"x-update (X't) error is {rep['xt_ratio_max']:.2f} x the float dot-product yardstick at {rep.get('xt_worst')}")
    assert rep["ax_ratio_max"] <= mv_factor, (label, f"A x error is {rep['ax_ratio_max']:.2f} x the float dot-product yardstick at {rep.get('ax_worst')}")
    assert rep["xt_rms"] <= mv_rms and rep["ax_rms"] <= mv_rms, (label, "mat-vec error over the run (rms, in yardsticks)", rep["xt_rms"], rep["ax_rms"])
    assert rep["norm_rel_max"] < norm_tol, (label, "recorded thresholds / residuals differ from the dumped iterates", rep["norm_rel_max"])
    allowed = max(2, int(np.ceil(max_accum_tie_rate * rep["decisions_checked"])))
    assert len(rep["accum_ties"]) <= allowed, (label, "decisions that float norm accumulation would flip", len(rep["accum_ties"]), rep["accum_ties"][:8])
    assert len(rep.get("rounding_ties", [])) <= 2, (label, "decisions that are exact ties up to the order of a sum", rep.get("rounding_ties"))
    return rep


# ---------------------------------------------------------------------------------------------------------------- LAD / BP
def check_dense(kind, x, y, opts, trace, state, intercept=True, label=""):
    """LAD (kind "lad": x n x p, n > p; dim n) / BP ("bp": x = A n x p, p > n; dim p), float64.  trace: libadmm_hip's decision
    trace INCLUDING the cold-start record; state: (nrec, 5, dim) iterate dump  x | z | y | adj_z | adj_y  (record 0: the data
    vector as the library holds it in the x slot)."""
    x = np.array(x, dtype=np.float64, order="F")
    y = np.array(y, dtype=np.float64)
    n, p = x.shape
    t = _strip(trace)
    assert t[0, 8] == -1, "the trace must start with the cold-start record"
    S = np.asarray(state, dtype=np.float64)
    dim = n if kind == "lad" else p
    S = S.reshape(len(S), 5, dim)
    nrec = min(len(t), len(S))
    dvec = S[0, 0].copy()
    if kind == "lad":
        std = DataStd(n, p, True, intercept, np.float64)        # LAD.cpp:34
        std.standardize(x, y)
        assert np.abs(dvec - y).max() <= 64 * np.spacing(np.abs(y).max()), (label, "the library's standardised y differs from the oracle's beyond rounding")
        Q, _ = np.linalg.qr(x)                                   # range(X): P = Q Q'
        chol = sla.cho_factor(x.T @ x, lower=True, check_finite=False)     # the reference's route (ADMMLAD.h:75-76,186-189)
        extra = float(np.linalg.norm(dvec))
        proj = lambda v: Q @ (Q.T @ v)
        ref_x = lambda v: x @ sla.cho_solve(chol, x.T @ v, check_finite=False)
    else:
        Q, R = np.linalg.qr(x.T)                                 # A' = Q R: null-space projector I - Q Q', A^+ b = Q R^-T b
        aaab = Q @ sla.solve_triangular(R, y, trans="T", check_finite=False)
        assert np.abs(dvec - aaab).max() <= 1e-9 * max(np.abs(aaab).max(), 1e-300), (label, "A'(AA')^-1 b differs from the oracle's", np.abs(dvec - aaab).max())
        chol = sla.cho_factor(x @ x.T, lower=True, check_finite=False)
        LinvA = sla.solve_triangular(np.tril(chol[0]), x, lower=True, check_finite=False)
        extra = 0.0
        proj = lambda v: v + dvec - Q @ (Q.T @ v)
        ref_x = lambda v: (v + dvec) - LinvA.T @ (LinvA @ v)     # ADMMBP.h:48-67
    eps_abs, eps_rel, sqrt_dim = float(opts["eps_abs"]), float(opts["eps_rel"]), np.sqrt(float(dim))
    maxit = int(opts["maxit"])
    nrm = lambda v: float(np.sqrt(np.sum(v * v)))
    zero = np.zeros(dim)
    rep = dict(records=nrec - 1, bit_mismatch=[], decisions_checked=0, norm_rel_max=0.0, x_vs_ref_max=0.0, x_rel_max=0.0, rho_changes=0)
    a, c_old = 1.0, 9999.0
    sg = sr = 0.0

    def vecs(k):
        return (zero,) * 5 if k <= 0 else tuple(S[k, j] for j in range(5))

    for k in range(1, nrec):
        xg, zg, yg, ajz, ajy = vecs(k)
        xp_, zp, yp, _, _ = vecs(k - 1)
        _, zpp, ypp, _, _ = vecs(k - 2)
        it = int(t[k, 1])
        rho = float(t[k, 9])
        prev_code = int(t[k - 1, 8])
        assert it == k - 1, (label, "iteration number", k, it)
        # ---- adj of this iteration from the previous decision (FADMMBase.h:243-256)
        if k == 1:
            ez, ey = zero, zero
        elif prev_code == 1:
            a_new = 0.5 + 0.5 * np.sqrt(1.0 + 4.0 * a * a)
            ratio = (a - 1.0) / a_new
            ez = (1.0 + ratio) * zp - ratio * zpp
            ey = (1.0 + ratio) * yp - ratio * ypp
            a = a_new
        else:
            assert prev_code == 2, (label, "a converged decision ends a LAD / BP run", k, prev_code)
            ez, ey, a = zpp, ypp, 1.0
        if not (np.array_equal(ez, ajz) and np.array_equal(ey, ajy)):
            rep["bit_mismatch"].append((k, it, "adj", int((ez != ajz).sum() + (ey != ajy).sum())))
        # ---- x-update: the projection (ADMMLAD.h:62-78 / ADMMBP.h:48-67) against Householder QR
        vec = (dvec if kind == "lad" else 0.0) - ajy / rho + ajz
        xe = proj(vec)
        xr = ref_x(vec)
        e_gpu, e_ref = nrm(xg - xe), nrm(xr - xe)
        floor = 64 * np.finfo(np.float64).eps * max(nrm(vec), nrm(xe))
        sg += e_gpu ** 2; sr += max(e_ref, floor) ** 2
        rep["x_rel_max"] = max(rep["x_rel_max"], e_gpu / max(nrm(xe), 1e-300))
        if e_gpu / max(e_ref, floor) > rep["x_vs_ref_max"]:
            rep.update(x_vs_ref_max=e_gpu / max(e_ref, floor), x_worst=(k, it, e_gpu, e_ref, floor))
        # ---- z, y from the library's own x: elementwise, bit for bit
        pen = 1.0 / rho
        if kind == "lad":
            v = xg - dvec + ajy / rho                            # ADMMLAD.h:94-98
            zn = np.where(v > pen, v - pen, np.where(v < -pen, v + pen, 0.0))
            r = xg - dvec - zg                                   # :99-107
        else:
            v = xg + ajy / rho                                   # ADMMBP.h:84-88
            zn = np.where(v > pen, v - pen, np.where(v < -pen, v + pen, 0.0))
            r = xg - zg
        if not np.array_equal(zn, zg):
            rep["bit_mismatch"].append((k, it, "z", int((zn != zg).sum())))
        yn = ajy + rho * r
        if not np.array_equal(yn, yg):
            rep["bit_mismatch"].append((k, it, "y", int((yn != yg).sum())))
        # ---- thresholds / residuals / decision / rho adaptation (FADMMBase.h:109-133,213-259)
        eps_p = max(nrm(xp_), nrm(zp), extra) * eps_rel + sqrt_dim * eps_abs
        eps_d = nrm(yp) * eps_rel + sqrt_dim * eps_abs
        rp, rd = nrm(r), rho * nrm(zg - zp)
        c = rho * rp * rp + rho * float(np.sum((zg - ajz) ** 2))
        code = int(t[k, 8])
        for got, want in ((t[k, 2], eps_p), (t[k, 3], eps_d), (t[k, 4], rp), (t[k, 5], rd)) + (((t[k, 6], c),) if code != 0 else ()):
            rep["norm_rel_max"] = max(rep["norm_rel_max"], abs(got - want) / max(abs(want), 1e-300))
        own = 0 if (rp < eps_p and rd < eps_d) else (1 if c < 0.999 * c_old else 2)
        if own != code and _rounding_tie(own, code, rp, eps_p, rd, eps_d, c, c_old):
            # the norms recomputed here add up in NumPy's order, the library's in its own: a comparison that is an EXACT tie in exact
            # arithmetic is decided by that last bit.  LAD has one by construction: while z = 0 the iterate repeats (x = P(2 y - x0) = x0),
            # so after the restart that follows c_2 = c_1 is tested against 0.999 (c_0 / 0.999) = c_0 (1 +- ulp) (README LAD n = 5000:
            # record 3, c = 2.984715452011392 on both sides)
            rep.setdefault("rounding_ties", []).append((k, it, code, own))
        else:
            assert own == code, (label, f"record {k} (iteration {it}): the library decided {code}, the rule on its own iterates gives {own}",
                                 dict(eps_p=eps_p, eps_d=eps_d, rp=rp, rd=rd, c=c, c_old=c_old))
        class _R:
            pass
        q = _R()
        q.rho, q.eps_primal, q.eps_dual, q.resid_primal, q.resid_dual = rho, eps_p, eps_d, rp, rd
        if code != 0 and it > 5:
            _rho_rule(q)
        assert abs(q.rho - t[k, 10]) <= 1e-12 * abs(q.rho), (label, f"record {k}: rho after the decision {t[k, 10]}, the rule gives {q.rho}")
        if q.rho != rho:
            rep["rho_changes"] += 1
        if k + 1 < nrec:
            assert code != 0 and abs(t[k + 1, 9] - q.rho) <= 1e-12 * abs(q.rho), (label, f"record {k + 1} ran with rho {t[k + 1, 9]}, the decision before it left {q.rho}")
        rep["decisions_checked"] += 1
        if code == 1:
            c_old = c
        elif code == 2:
            c_old = c_old / 0.999
    rep["x_rms_vs_ref"] = float(np.sqrt(sg / sr)) if sr > 0 else 0.0
    return rep


def assert_stepwise_dense(rep, label="", x_factor=8.0, x_rms_factor=4.0, norm_tol=1e-9, max_ties=8):
    """check_dense's report is clean: adj, z, y bit-exact; the projection within `x_factor` x (per record) / `x_rms_factor` x
    (rms over the run; measured over 3300 LAD / BP cases of the round-4 soaks: at most 3.01, an n = 29, p = 27 LAD problem) of what the
    reference's own normal-equation route misses the QR projection by (floored at 64 ulps of the operand); recorded thresholds / residuals / c equal to the recomputed ones; decisions and rho adaptation the rule's."""
    assert not rep["bit_mismatch"], (label, "elementwise steps differ from the reference's arithmetic", rep["bit_mismatch"][:8], len(rep["bit_mismatch"]))
    assert rep["x_vs_ref_max"] <= x_factor, (label, f"projection error is {rep['x_vs_ref_max']:.2f} x the reference route's at {rep.get('x_worst')}")
    assert rep["x_rms_vs_ref"] <= x_rms_factor, (label, "projection error over the run (rms) against the reference route's", rep["x_rms_vs_ref"])
    assert rep["norm_rel_max"] < norm_tol, (label, "recorded thresholds / residuals differ from the dumped iterates", rep["norm_rel_max"])
    # exact ties (to 1e-12) of the restart test come in RUNS: while z = 0 the iterate repeats, so every other iteration tests c against
    # 0.999 (c / 0.999) again (README LAD n = 5000: one; out-of-sample soak 829:67 / 840:24, BP: three in the first seven iterations)
    assert len(rep.get("rounding_ties", [])) <= max_ties, (label, "decisions that are exact ties up to the order of a sum", rep.get("rounding_ties"))
    return rep
