"""The five `.Call` entry points of the reference, restated (TEST INFRASTRUCTURE).

  admm_lasso     /root/reference/src/Lasso.cpp:32-137
  admm_enet      /root/reference/src/Enet.cpp:31-137
  admm_parlasso  /root/reference/src/ParLasso.cpp:33-110
  admm_lad       /root/reference/src/LAD.cpp:16-47
  admm_bp        /root/reference/src/BP.cpp:20-45
Arguments keep the reference's order and meaning; `opts` is the R list
{maxit, eps_abs, eps_rel, rho} as a dict.  x is n x p (any layout), y length n,
both float64 like R numerics.  Returns dicts shaped like the R lists, with
`beta` dense: (p+1) x nlambda float32 (row 0 = intercept) for the Lasso family.
"""
import numpy as np

from .datastd import DataStd
from .solvers import LassoTall, LassoWide, PADMMLasso, LAD, BP

F = np.float32


def _lambda_grid(lambda0, n, scaleY, nlambda, lmin_ratio):
    """Lasso.cpp:78-89: log-spaced from lambda0/n*scaleY down to lmin_ratio times that."""
    lmax = np.float64(lambda0) / n * np.float64(scaleY)
    lmin = lmin_ratio * lmax
    return np.exp(np.linspace(np.log(lmax), np.log(lmin), int(nlambda)))


def _attach(solver, detail):
    """Tests: collect the decision trace / follow another execution's decisions (oracle/solvers.py)."""
    if detail is None:
        return
    if detail.get("trace") is not None:
        solver.trace = detail["trace"]
    if detail.get("state") is not None:
        solver.state_log = detail["state"]
    if detail.get("types") is not None:
        solver.type_log = detail["types"]
    if detail.get("follow_x") is not None and hasattr(solver, "follow_x"):
        solver.follow_x = iter(detail["follow_x"])
    if detail.get("follow") is not None:
        solver.follow = iter(detail["follow"])
        solver.follow_band = float(detail.get("follow_band", 8.0))
        solver.forced = detail.setdefault("forced", [])


def _lasso_family(x, y, lam, nlambda, lmin_ratio, standardize, intercept, opts, alpha, nthread, detail=None):
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    n, p = x.shape
    datX = np.array(x, dtype=F, order="F")                  # Lasso.cpp:45-50 double -> float copy
    datY = np.array(y, dtype=F)
    lam = np.atleast_1d(np.asarray(lam, dtype=np.float64)) if lam is not None else np.zeros(0)
    maxit, eps_abs, eps_rel, rho = int(opts["maxit"]), float(opts["eps_abs"]), float(opts["eps_rel"]), float(opts["rho"])
    std = DataStd(n, p, standardize, intercept, F)
    std.standardize(datX, datY)
    if nthread is not None:
        solver = PADMMLasso(datX, datY, int(nthread), eps_abs, eps_rel)
    elif n > p:
        solver = LassoTall(datX, datY, eps_abs, eps_rel, alpha)
    else:
        solver = LassoWide(datX, datY, eps_abs, eps_rel, alpha)
    if lam.size < 1:
        lam = _lambda_grid(solver.lambda0, n, std.scaleY, nlambda, lmin_ratio)
    nl = lam.size
    beta = np.zeros((p + 1, nl), dtype=F)
    niter = np.zeros(nl, dtype=np.int32)
    _attach(solver, detail)
    for i in range(nl):
        ilambda = lam[i] * n / np.float64(std.scaleY)       # Lasso.cpp:99
        solver.lam_idx = i
        if i == 0:
            solver.init(ilambda, rho)
        else:
            solver.init_warm(ilambda)
        niter[i] = solver.solve(maxit)
        beta0, coef = std.recover(solver.get_coef())
        beta[0, i] = beta0
        beta[1:, i] = coef
    if detail is not None:
        detail.update(solver=solver, std=std)
    return {"lambda": lam, "beta": beta, "niter": niter}


def admm_lasso(x, y, lam, nlambda, lmin_ratio, standardize, intercept, opts, detail=None):
    return _lasso_family(x, y, lam, nlambda, lmin_ratio, standardize, intercept, opts, None, None, detail)


def admm_enet(x, y, lam, nlambda, lmin_ratio, standardize, intercept, alpha, opts, detail=None):
    return _lasso_family(x, y, lam, nlambda, lmin_ratio, standardize, intercept, opts, alpha, None, detail)


def admm_parlasso(x, y, lam, nlambda, lmin_ratio, standardize, intercept, nthread, opts, detail=None):
    return _lasso_family(x, y, lam, nlambda, lmin_ratio, standardize, intercept, opts, None, nthread, detail)


def admm_lad(x, y, intercept, opts, detail=None):
    x = np.array(x, dtype=np.float64, order="F")
    y = np.array(y, dtype=np.float64)
    n, p = x.shape
    std = DataStd(n, p, True, intercept, np.float64)        # LAD.cpp:34 standardize is always TRUE
    std.standardize(x, y)
    solver = LAD(x, y, float(opts["rho"]), float(opts["eps_abs"]), float(opts["eps_rel"]))
    _attach(solver, detail)
    niter = solver.solve(int(opts["maxit"]))
    beta0, coef = std.recover(solver.get_coef())
    if detail is not None:
        detail.update(solver=solver, std=std)
    return {"beta": np.concatenate([[beta0], coef]), "niter": niter}


def admm_bp(x, y, opts, detail=None):
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    solver = BP(x, y, float(opts["rho"]), float(opts["eps_abs"]), float(opts["eps_rel"]))
    _attach(solver, detail)
    niter = solver.solve(int(opts["maxit"]))
    if detail is not None:
        detail.update(solver=solver)
    return {"beta": solver.get_coef().copy(), "niter": niter}


def admm_parbp(x, y, nthread, opts, detail=None):
    """src/TODO/ParBP.cppp:26-71 (never built by the reference): opts carries rho_ratio (R passes its `rho` field under that
    name, R/10_admm_bp.R:115); returns the dense coefficient vector (the reference: a one-column dgCMatrix) and niter."""
    from .solvers import SharingBP
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    solver = SharingBP(x, y, int(nthread), float(opts["eps_abs"]), float(opts["eps_rel"]))
    solver.init(float(opts.get("rho_ratio", opts.get("rho", 1.0))))
    if detail is not None and detail.get("trace") is not None:
        solver.trace = detail["trace"]
    niter = solver.solve(int(opts["maxit"]))
    if detail is not None:
        detail.update(solver=solver)
    return {"beta": solver.get_x(), "niter": niter}


def admm_dantzig(x, y, lam, nlambda, lmin_ratio, standardize, intercept, opts, detail=None):
    """src/TODO/Dantzig.cpp:32-99 (never built by the reference): DataStd<double>, the automatic grid from lambda_0 = max|X'y|
    (:63-70), internal lambda = lambda n / scaleY (:79), warm-started loop, recover -> (p+1) x nlambda doubles."""
    from .solvers import Dantzig
    x = np.array(x, dtype=np.float64, order="F")
    y = np.array(y, dtype=np.float64)
    n, p = x.shape
    std = DataStd(n, p, standardize, intercept, np.float64)
    std.standardize(x, y)
    solver = Dantzig(x, y, float(opts["eps_abs"]), float(opts["eps_rel"]))
    lam = np.atleast_1d(np.asarray(lam, dtype=np.float64)) if lam is not None else np.zeros(0)
    if lam.size < 1:
        lam = _lambda_grid(solver.lambda0, n, std.scaleY, nlambda, lmin_ratio)
    if detail is not None and detail.get("trace") is not None:
        solver.trace = detail["trace"]
    beta = np.zeros((p + 1, lam.size))
    niter = np.zeros(lam.size, dtype=np.int32)
    for i in range(lam.size):
        solver.lam_idx = i
        il = lam[i] * n / np.float64(std.scaleY)
        if i == 0:
            solver.init(il, float(opts["rho"]))
        else:
            solver.init_warm(il)
        niter[i] = solver.solve(int(opts["maxit"]))
        b0, coef = std.recover(solver.get_coef())
        beta[0, i] = b0
        beta[1:, i] = coef
    if detail is not None:
        detail.update(solver=solver, std=std)
    return {"lambda": lam, "beta": beta, "niter": niter}


# Defaults of the R builders (R/30_admm_lasso.R:31-50, R/10_admm_bp.R:34-43, R/20_admm_lad.R:25-32)
LASSO_OPTS = {"maxit": 10000, "eps_abs": 1e-5, "eps_rel": 1e-5, "rho": -1.0}
BP_OPTS = {"maxit": 10000, "eps_abs": 1e-4, "eps_rel": 1e-4, "rho": 1.0}
LAD_OPTS = {"maxit": 10000, "eps_abs": 1e-4, "eps_rel": 1e-4, "rho": 1.0}
