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

Returns a report; `assert_*` raise AssertionError with the first offending record.
"""
import numpy as np
import scipy.linalg as sla

from .datastd import DataStd
from .entry import _lambda_grid
from .solvers import F, LassoTall, PADMMLasso, _enet_f, _soft_d


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
