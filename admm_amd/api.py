"""Host-side mirror of the reference's R interface for the GPU path.

R is not available in this image, so the builder chain of the package is mirrored in Python with
the same names, argument meaning, defaults and error behaviour:

    admm_lasso(x, y)$penalty(...)$parallel(...)$opts(...)$fit()     R/30_admm_lasso.R
    admm_enet(x, y)$penalty(..., alpha)$opts(...)$fit()             R/40_admm_enet.R
    admm_lad(x, y, intercept)$opts(...)$fit()                       R/20_admm_lad.R
    admm_bp(x, y)$opts(...)$fit()                                   R/10_admm_bp.R

`fit()` forwards to the C ABI of libadmm_hip.so exactly where the R `$fit()` does its
`.Call("admm_*", ...)` (R/30_admm_lasso.R:136-160 etc.).  All numerics run in the HIP library;
nothing here computes.  x may be a host array or a `DevicePtr` (column-major float64 in HBM).
"""
import ctypes

import numpy as np
import scipy.sparse as sp

from . import _lib
from ._lib import AdmmOpts, AdmmStats, DevicePtr, as_input, check


def _stop(msg):
    raise ValueError(msg)


def _shape(x, n, p):
    if isinstance(x, DevicePtr):
        if n is None or p is None:
            _stop("n and p are required with a DevicePtr")
        return int(n), int(p)
    a = np.asarray(x)
    if a.ndim != 2:
        _stop("x must be a matrix")
    return a.shape


def _beta_to_csc(beta_dense):
    """(p+1) x nlambda dense -> CSC holding row 0 always plus the non-zeros (Lasso.cpp:22-30)."""
    p1, nl = beta_dense.shape
    indptr = [0]
    indices, data = [], []
    for j in range(nl):
        col = beta_dense[:, j]
        nz = np.nonzero(col[1:])[0] + 1
        idx = np.concatenate([[0], nz])
        indices.append(idx)
        data.append(col[idx].astype(np.float64))
        indptr.append(indptr[-1] + idx.size)
    return sp.csc_matrix((np.concatenate(data), np.concatenate(indices), np.array(indptr)), shape=(p1, nl))


class ADMM_Lasso_fit:
    """Fields lambda, beta (dgCMatrix-like CSC, (p+1) x nlambda), niter (R/30_admm_lasso.R:18-22)."""

    def __init__(self, lam, beta_dense, niter, stats):
        self.lambda_ = lam
        self.beta_dense = beta_dense
        self._beta = None
        self.niter = niter
        self.stats = stats

    @property
    def beta(self):
        """dgCMatrix-like CSC, built on first use."""
        if self._beta is None:
            self._beta = _beta_to_csc(self.beta_dense)
        return self._beta

    def __repr__(self):
        return (f"{getattr(self, '_title', 'ADMM Lasso fitting result')}\n\n$lambda\n{self.lambda_}\n\n$beta\n<{self.beta.shape[0]} x "
                f"{self.beta.shape[1]}> sparse matrix\n\n$niter\n{self.niter}")

    def show(self):
        """ADMM_Lasso_fit$show (R/30_admm_lasso.R:181-186)."""
        print(repr(self))

    def path_data(self):
        """What ADMM_Lasso_fit$plot draws (R/30_admm_lasso.R:189-214): log(lambda) and, per variable that is non-zero for
        at least one lambda (intercept excluded), its coefficient along the path.  Returns (loglambda [nl], coef [nl, nvar])."""
        if self.lambda_.size < 2:
            _stop("need to have at least two lambda values")
        inc = np.any(self.beta_dense != 0, axis=1)
        inc[0] = False
        return np.log(self.lambda_), self.beta_dense[inc].T.astype(np.float64)

    def plot(self, ax=None):
        """Solution-path plot (the reference uses ggplot2; here matplotlib, same axes and title)."""
        loglambda, coef = self.path_data()
        import matplotlib
        if ax is None:
            matplotlib.use("Agg", force=False)
            import matplotlib.pyplot as plt
            _, ax = plt.subplots()
        ax.plot(loglambda, coef)
        ax.set_xlabel("log(lambda)")
        ax.set_ylabel("Coefficients")
        ax.set_title("Solution path")
        return ax


class ADMM_Lasso:
    _name = "ADMM Lasso model"

    def __init__(self, x, y, intercept=True, standardize=True, n=None, p=None):
        n_, p_ = _shape(x, n, p)
        ylen = n_ if isinstance(y, DevicePtr) else len(y)
        if n_ != ylen:
            _stop("nrow(x) should be equal to length(y)")                       # R/30_admm_lasso.R:34-35
        self.x, self.y, self.n, self.p = x, y, int(n_), int(p_)
        self.intercept = bool(intercept)
        self.standardize = bool(standardize)
        self.lambda_ = np.zeros(0)
        self.nlambda = 100
        self.lambda_min_ratio = 0.01 if n_ < p_ else 0.0001
        self.nthread = 1
        self.maxit = 10000
        self.eps_abs = 1e-5
        self.eps_rel = 1e-5
        self.rho = -1.0

    def penalty(self, lambda_=None, nlambda=100, lambda_min_ratio=None, **kw):
        if "lambda" in kw:
            lambda_ = kw["lambda"]
        lam = np.sort(np.atleast_1d(np.asarray(lambda_, dtype=np.float64)))[::-1] if lambda_ is not None else np.zeros(0)
        if np.any(lam <= 0):
            _stop("lambda must be positive")
        if nlambda <= 0:
            _stop("nlambda must be a positive integer")
        lmr = (0.01 if self.n < self.p else 0.0001) if lambda_min_ratio is None else float(lambda_min_ratio)
        if lmr >= 1 or lmr <= 0:
            _stop("lambda_min_ratio must be within (0, 1)")
        self.lambda_ = lam
        self.nlambda = int(nlambda)
        self.lambda_min_ratio = lmr
        return self

    def parallel(self, nthread=2):
        nt = int(nthread)
        if nt < 1:
            nt = 1
        if nt >= self.p / 5:
            _stop("nthread cannot exceed ncol(x)/5")                             # R/30_admm_lasso.R:105-106
        self.nthread = nt
        return self

    def opts(self, maxit=10000, eps_abs=1e-5, eps_rel=1e-5, rho=None):
        if maxit <= 0:
            _stop("maxit should be positive")
        if eps_abs < 0 or eps_rel < 0:
            _stop("eps_abs and eps_rel should be nonnegative")
        if rho is not None and rho <= 0:
            _stop("rho should be positive")
        self.maxit = int(maxit)
        self.eps_abs = float(eps_abs)
        self.eps_rel = float(eps_rel)
        self.rho = -1.0 if rho is None else float(rho)
        return self

    # -- .Call marshalling
    def _common(self):
        lib = _lib.load()
        xp, xmem, xk = as_input(self.x)
        yp, ymem, yk = as_input(self.y)
        if xmem != ymem:
            _stop("x and y must live in the same memory space")
        nl = self.lambda_.size if self.lambda_.size else self.nlambda
        lam_in = np.ascontiguousarray(self.lambda_, dtype=np.float64)
        o = AdmmOpts(self.maxit, self.eps_abs, self.eps_rel, self.rho)
        lam_out = np.zeros(nl, dtype=np.float64)
        beta = np.zeros((self.p + 1, nl), dtype=np.float32, order="F")
        niter = np.zeros(nl, dtype=np.int32)
        stats = AdmmStats()
        head = (xp, yp, self.n, self.p, xmem,
                ctypes.c_void_p(lam_in.ctypes.data if lam_in.size else 0), int(lam_in.size), self.nlambda,
                self.lambda_min_ratio, int(self.standardize), int(self.intercept))
        tail = (ctypes.byref(o), lam_out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                beta.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                niter.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), ctypes.byref(stats))
        keep = (xk, yk, lam_in)
        return lib, head, tail, lam_out, beta, niter, stats, keep

    def fit_responses(self, Y, m=None):
        """Fit this model for several responses of the same x at once (admm_hip_lasso_multi; not in the reference package):
        Y is n x m (a host matrix, or a DevicePtr to column-major doubles together with m); returns one ADMM_Lasso_fit per
        column, each bit-identical to a separate fit() with that column as y."""
        lib, head, tail, lam_out, beta, niter, stats, keep = self._common()
        if isinstance(Y, DevicePtr):
            if m is None:
                _stop("m (the number of responses) is needed with a device pointer")
            yp, ymem, yk = as_input(Y)
            m = int(m)
        else:
            Ya = np.asfortranarray(Y, dtype=np.float64)
            if Ya.ndim != 2 or Ya.shape[0] != self.n:
                _stop("Y should be a matrix with nrow(x) rows")
            m = Ya.shape[1]
            yp, ymem, yk = as_input(Ya)
        if ymem != head[4]:
            _stop("x and Y must live in the same memory space")
        nl = lam_out.size
        lam_all = np.zeros((m, nl))
        beta_all = np.zeros((m, nl, self.p + 1), dtype=np.float32)
        nit_all = np.zeros((m, nl), dtype=np.int32)
        st_all = (AdmmStats * m)()
        alpha = float(getattr(self, "alpha", -1.0))
        check(lib.admm_hip_lasso_multi(head[0], yp, head[2], head[3], m, head[4], *head[5:], alpha, tail[0],
                                       lam_all.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                       beta_all.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                       nit_all.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), st_all))
        return [ADMM_Lasso_fit(lam_all[j].copy(), np.asfortranarray(beta_all[j].T), nit_all[j].copy(), st_all[j].as_dict()) for j in range(m)]

    def cv(self, nfolds=10, fold_id=None, keep_fold_beta=False):
        """K-fold cross-validation of this model's lambda path (admm_hip_lasso_cv; not in the reference package).
        fold_id: integer array of length n with values in [0, nfolds) (default: i mod nfolds)."""
        lib, head, tail, lam_out, beta, niter, stats, keep = self._common()
        nl = lam_out.size
        fid = None if fold_id is None else np.ascontiguousarray(fold_id, dtype=np.int32)
        if fid is not None and fid.size != self.n:
            _stop("fold_id should have length nrow(x)")
        cvm, cvse = np.zeros(nl), np.zeros(nl)
        fmse = np.zeros((nfolds, nl))
        fnit = np.zeros((nfolds, nl), dtype=np.int32)
        fbeta = np.zeros((nfolds, nl, self.p + 1), dtype=np.float32) if keep_fold_beta else None
        imin, i1se = ctypes.c_int(0), ctypes.c_int(0)
        dp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        ip = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
        alpha = float(getattr(self, "alpha", -1.0))
        check(lib.admm_hip_lasso_cv(*head[:5], ip(fid) if fid is not None else None, int(nfolds), *head[5:], alpha, tail[0],
                                    tail[1], tail[2], tail[3], dp(cvm), dp(cvse), dp(fmse), ip(fnit),
                                    fbeta.ctypes.data_as(ctypes.POINTER(ctypes.c_float)) if keep_fold_beta else None,
                                    ctypes.byref(imin), ctypes.byref(i1se), tail[4]))
        fit = ADMM_Lasso_fit(lam_out, beta, niter, stats.as_dict())
        if fbeta is not None:
            fbeta = np.transpose(fbeta, (0, 2, 1))                               # [fold][p + 1][lambda]
        return ADMM_Lasso_cv(fit, cvm, cvse, fmse, fnit, fbeta, imin.value, i1se.value)

    def fit(self):
        lib, head, tail, lam_out, beta, niter, stats, keep = self._common()
        if self.nthread <= 1:
            check(lib.admm_hip_lasso(*head, *tail))                              # .Call("admm_lasso", ...)
        else:
            check(lib.admm_hip_parlasso(*head, self.nthread, *tail))             # .Call("admm_parlasso", ...)
        return ADMM_Lasso_fit(lam_out, beta, niter, stats.as_dict())


class ADMM_Lasso_cv:
    """Result of ADMM_Lasso.cv / ADMM_Enet.cv: the full-data fit plus the K-fold table (admm_hip_lasso_cv)."""

    def __init__(self, fit, cvm, cvse, fold_mse, fold_niter, fold_beta, idx_min, idx_1se):
        self.fit = fit
        self.lambda_ = fit.lambda_
        self.cvm, self.cvse = cvm, cvse
        self.fold_mse, self.fold_niter, self.fold_beta = fold_mse, fold_niter, fold_beta
        self.idx_min, self.idx_1se = int(idx_min), int(idx_1se)
        self.lambda_min, self.lambda_1se = float(fit.lambda_[idx_min]), float(fit.lambda_[idx_1se])

    def __repr__(self):
        return (f"ADMM cross-validation: {self.fold_mse.shape[0]} folds x {self.lambda_.size} lambdas\n"
                f"lambda.min = {self.lambda_min:.6g} (cvm {self.cvm[self.idx_min]:.6g}), lambda.1se = {self.lambda_1se:.6g}")


class ADMM_Enet(ADMM_Lasso):
    _name = "ADMM Elastic Net model"

    def parallel(self, nthread=2):
        """The reference's $parallel() on an elastic-net model only sets a field that ADMM_Enet$fit never reads
        (R/40_admm_enet.R:50-64 always calls admm_enet): kept as a no-op with the same validation."""
        super().parallel(nthread)
        self.nthread = 1
        return self

    def __init__(self, x, y, intercept=True, standardize=True, n=None, p=None):
        super().__init__(x, y, intercept, standardize, n, p)
        self.alpha = 1.0

    def penalty(self, lambda_=None, nlambda=100, lambda_min_ratio=None, alpha=1, **kw):
        super().penalty(lambda_, nlambda, lambda_min_ratio, **kw)
        if alpha < 0 or alpha > 1:
            _stop("alpha must be within [0, 1]")                                 # R/40_admm_enet.R:38-39
        self.alpha = float(alpha)
        return self

    def fit(self):
        lib, head, tail, lam_out, beta, niter, stats, keep = self._common()
        check(lib.admm_hip_enet(*head, self.alpha, *tail))                       # .Call("admm_enet", ...)
        return ADMM_Lasso_fit(lam_out, beta, niter, stats.as_dict())


class _Records(np.ndarray):
    """A record buffer that carries its own capacity (`cap`, 0 = disabled): a one-record request is not mistaken for 'off'
    (the capacity used to be inferred from shape[0] > 1)."""
    cap = 0


def _records(shape, cap):
    a = np.zeros(shape, dtype=np.float64).view(_Records)
    a.cap = int(cap)
    return a


def _trace_buffers(maxit):
    cap = maxit + 8 if maxit > 0 else 0
    return _records((max(cap, 1), _lib.TRACE_FIELDS), cap), ctypes.c_longlong(0)


def _trace_args(tr, ntr):
    if tr.cap == 0:
        return None, 0, None
    return tr.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), tr.cap, ctypes.byref(ntr)


def _state_buffers(nrec, dim):
    nrec = int(nrec) if nrec else 0
    return _records((max(nrec, 1), 5, dim), nrec), ctypes.c_longlong(0)


def _state_args(sbuf, nst):
    if sbuf.cap == 0:
        return None, 0, None
    return sbuf.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), sbuf.cap, ctypes.byref(nst)


class ADMM_BP_fit:
    def __init__(self, beta, niter, stats):
        self.beta = beta
        self.niter = niter
        self.stats = stats

    def __repr__(self):                                                      # ADMM_BP_fit$show (R/10_admm_bp.R:125-133)
        return f"ADMM Basis Pursuit fitting result\n\n$beta\n<{self.beta.shape[0]} x 1> sparse matrix\n\n$niter\n{self.niter}"

    def show(self):
        print(repr(self))


class ADMM_BP:
    def __init__(self, x, y, n=None, p=None):
        n_, p_ = _shape(x, n, p)
        if n_ >= p_:
            _stop("ncol(x) must be greater than nrow(x)")                        # R/10_admm_bp.R:30-31
        ylen = n_ if isinstance(y, DevicePtr) else len(y)
        if n_ != ylen:
            _stop("nrow(x) should be equal to length(y)")
        self.x, self.y, self.n, self.p = x, y, int(n_), int(p_)
        self.nthread = 1
        self.maxit = 10000
        self.eps_abs = 1e-4
        self.eps_rel = 1e-4
        self.rho = 1.0

    def parallel(self, nthread=2):
        """ADMM_BP$parallel (R/10_admm_bp.R:65-76) only stores nthread; $fit() with nthread > 1 then calls the C symbol
        `admm_parbp`, which the reference never builds (it lives in src/TODO/ParBP.cppp) -- the R call fails there; here it
        runs the column-block sharing solver (admm_hip_parbp).  The setter clamps to >= 1 and stores, exactly like R (no
        ncol(x)/5 check here: only ADMM_Lasso$parallel has one, R/30_admm_lasso.R:119-124).  ADMM_LAD inherits this method
        as in R (`contains = "ADMM_BP"`, R/20_admm_lad.R:4) and, as in R, its fit() ignores nthread."""
        self.nthread = max(1, int(nthread))
        return self

    def opts(self, maxit=10000, eps_abs=1e-4, eps_rel=1e-4, rho=1.0):
        if maxit <= 0:
            _stop("maxit should be positive")
        if eps_abs < 0 or eps_rel < 0:
            _stop("eps_abs and eps_rel should be nonnegative")
        if rho is not None and rho <= 0:
            _stop("rho should be positive")
        self.maxit, self.eps_abs, self.eps_rel = int(maxit), float(eps_abs), float(eps_rel)
        self.rho = 1.0 if rho is None else float(rho)
        return self

    def fit(self, trace=False, state=0):
        """trace=True also returns the decision trace (fit.trace, layout of include/admm_hip.h ADMM_TRACE_*); state = N > 0 (serial
        solver only) also the iterate dump of the first N decisions (fit.state: (records, 5, p) -- x | z | y | adj_z | adj_y,
        admm_hip_bp_state)."""
        lib = _lib.load()
        xp, xmem, xk = as_input(self.x)
        yp, ymem, yk = as_input(self.y)
        if xmem != ymem:
            _stop("x and y must live in the same memory space")
        o = AdmmOpts(self.maxit, self.eps_abs, self.eps_rel, self.rho)
        beta = np.zeros(self.p, dtype=np.float64)
        niter = np.zeros(1, dtype=np.int32)
        stats = AdmmStats()
        tr, ntr = _trace_buffers(self.maxit if trace else 0)
        if getattr(self, "nthread", 1) > 1:
            # .Call("admm_parbp", x, y, nthread, list(maxit, eps_abs, eps_rel, rho_ratio = rho)), R/10_admm_bp.R:111-116 -- the
            # symbol the reference never builds; here the column-block sharing solver of admm_amd/csrc/sharing_bp.hip
            check(lib.admm_hip_parbp_traced(xp, yp, self.n, self.p, xmem, int(self.nthread), ctypes.byref(o),
                                            beta.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                            niter.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), ctypes.byref(stats), *_trace_args(tr, ntr)))
        else:
            sbuf, nst = _state_buffers(state if trace else 0, self.p)
            check(lib.admm_hip_bp_state(xp, yp, self.n, self.p, xmem, ctypes.byref(o),
                                        beta.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                        niter.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), ctypes.byref(stats), *_trace_args(tr, ntr),
                                        *_state_args(sbuf, nst)))
        fit = ADMM_BP_fit(sp.csc_matrix(beta.reshape(-1, 1)), int(niter[0]), stats.as_dict())   # dgCMatrix p x 1 (BP.cpp:38-43)
        fit.trace = tr[:ntr.value].copy() if trace else None
        fit.state = sbuf[:nst.value].copy() if (trace and state and self.nthread <= 1) else None
        return fit


class ADMM_LAD_fit:
    def __init__(self, beta, niter, stats):
        self.beta = beta            # numeric p+1, intercept first (LAD.cpp:40-45)
        self.niter = niter
        self.stats = stats

    def __repr__(self):
        return f"ADMM LAD fitting result\n\n$beta\n{self.beta}\n\n$niter\n{self.niter}"

    def show(self):
        print(repr(self))


class ADMM_LAD(ADMM_BP):
    def __init__(self, x, y, intercept=True, n=None, p=None):
        n_, p_ = _shape(x, n, p)
        if n_ <= p_:
            _stop("nrow(x) must be greater than ncol(x)")                        # R/20_admm_lad.R:21-22
        ylen = n_ if isinstance(y, DevicePtr) else len(y)
        if n_ != ylen:
            _stop("nrow(x) should be equal to length(y)")
        self.x, self.y, self.n, self.p = x, y, int(n_), int(p_)
        self.maxit = 10000
        self.eps_abs = 1e-4
        self.eps_rel = 1e-4
        self.rho = 1.0
        self.intercept = bool(intercept)

    def fit(self, trace=False, state=0):
        """trace / state as ADMM_BP.fit (the iterate dump has dimension n: admm_hip_lad_state)."""
        lib = _lib.load()
        xp, xmem, xk = as_input(self.x)
        yp, ymem, yk = as_input(self.y)
        if xmem != ymem:
            _stop("x and y must live in the same memory space")
        o = AdmmOpts(self.maxit, self.eps_abs, self.eps_rel, self.rho)
        beta = np.zeros(self.p + 1, dtype=np.float64)
        niter = np.zeros(1, dtype=np.int32)
        stats = AdmmStats()
        tr, ntr = _trace_buffers(self.maxit if trace else 0)
        sbuf, nst = _state_buffers(state if trace else 0, self.n)
        check(lib.admm_hip_lad_state(xp, yp, self.n, self.p, xmem, int(self.intercept), ctypes.byref(o),
                                     beta.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                     niter.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), ctypes.byref(stats), *_trace_args(tr, ntr),
                                     *_state_args(sbuf, nst)))
        fit = ADMM_LAD_fit(beta, int(niter[0]), stats.as_dict())
        fit.trace = tr[:ntr.value].copy() if trace else None
        fit.state = sbuf[:nst.value].copy() if (trace and state) else None
        return fit


class LassoPlan:
    """Prepared problem (admm_hip_lasso_plan_*): setup once, run the lambda path repeatedly.

    Built from a configured ADMM_Lasso / ADMM_Enet object; used by bench.py to time the ADMM
    loop without the one-time Gram/factorisation."""

    def __init__(self, model):
        if isinstance(model, ADMM_Dantzig):
            _stop(ADMM_Dantzig._missing)
        lib = _lib.load()
        self._lib = lib
        self.model = model
        xp, xmem, xk = as_input(model.x)
        yp, ymem, yk = as_input(model.y)
        lam_in = np.ascontiguousarray(model.lambda_, dtype=np.float64)
        o = AdmmOpts(model.maxit, model.eps_abs, model.eps_rel, model.rho)
        alpha = float(model.alpha) if isinstance(model, ADMM_Enet) else -1.0
        h = ctypes.c_void_p()
        nl = ctypes.c_int()
        check(lib.admm_hip_lasso_plan_create(
            xp, yp, model.n, model.p, xmem, ctypes.c_void_p(lam_in.ctypes.data if lam_in.size else 0), int(lam_in.size),
            model.nlambda, model.lambda_min_ratio, int(model.standardize), int(model.intercept), alpha,
            int(model.nthread), ctypes.byref(o), ctypes.byref(h), ctypes.byref(nl)))
        self._h = h
        self.nlambda = nl.value

    def run(self):
        m = self.model
        lam_out = np.zeros(self.nlambda, dtype=np.float64)
        beta = np.zeros((m.p + 1, self.nlambda), dtype=np.float32, order="F")
        niter = np.zeros(self.nlambda, dtype=np.int32)
        stats = AdmmStats()
        check(self._lib.admm_hip_lasso_plan_run(self._h, lam_out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                                beta.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                                niter.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), ctypes.byref(stats)))
        return ADMM_Lasso_fit(lam_out, beta, niter, stats.as_dict())

    def enable_trace(self, capacity=1 << 18):
        """Record one decision record per ADMM iteration of the following run() calls (tall path only)."""
        check(self._lib.admm_hip_lasso_plan_trace_enable(self._h, int(capacity)))
        self._trace_cap = int(capacity)

    def read_trace(self):
        """(nrecords, TRACE_FIELDS) float64 array of the last run(): see include/admm_hip.h."""
        buf = np.zeros((self._trace_cap, _lib.TRACE_FIELDS), dtype=np.float64)
        n = ctypes.c_longlong()
        check(self._lib.admm_hip_lasso_plan_trace_read(self._h, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                                       self._trace_cap, ctypes.byref(n)))
        return buf[:n.value].copy()

    def enable_state(self, capacity):
        """Also dump the iterates of every iteration of the following run() calls (tall and consensus solvers; include/admm_hip.h,
        admm_hip_lasso_plan_state_*): record s = the iterates trace record s judged."""
        check(self._lib.admm_hip_lasso_plan_state_enable(self._h, int(capacity)))
        self._state_cap = int(capacity)

    def read_state(self):
        """(nrecords, record_floats) float32 array of the last run()."""
        n, rf = ctypes.c_longlong(), ctypes.c_longlong()
        check(self._lib.admm_hip_lasso_plan_state_read(self._h, None, 0, ctypes.byref(n), ctypes.byref(rf)))     # size query
        nrec = max(int(n.value), 1)
        buf = np.zeros((nrec, rf.value), dtype=np.float32)
        check(self._lib.admm_hip_lasso_plan_state_read(self._h, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), nrec,
                                                       ctypes.byref(n), ctypes.byref(rf)))
        return buf[:n.value]

    def read_data(self):
        """(X, Y): the standardised float32 data as the wide solver holds them (admm_hip_lasso_plan_data_read)."""
        m = self.model
        X = np.zeros((m.n, m.p), dtype=np.float32, order="F")
        Y = np.zeros(m.n, dtype=np.float32)
        check(self._lib.admm_hip_lasso_plan_data_read(self._h, X.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), m.n,
                                                      Y.ctypes.data_as(ctypes.POINTER(ctypes.c_float))))
        return X, Y

    def read_system(self):
        """(p, p) float32: the system matrix X'X + rho I as this library formed it (tall solver, ADMM_HIP_REFINE=1 only)."""
        p = self.model.p
        out = np.zeros((p, p), dtype=np.float32, order="F")
        check(self._lib.admm_hip_lasso_plan_system_read(self._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), p))
        return out

    def close(self):
        if self._h:
            check(self._lib.admm_hip_lasso_plan_destroy(self._h))
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def admm_lasso(x, y, intercept=True, standardize=True, **kw):
    return ADMM_Lasso(x, y, intercept, standardize, **kw)


def admm_enet(x, y, intercept=True, standardize=True, **kw):
    return ADMM_Enet(x, y, intercept, standardize, **kw)


def admm_lad(x, y, intercept=True, **kw):
    return ADMM_LAD(x, y, intercept, **kw)


def admm_bp(x, y, **kw):
    return ADMM_BP(x, y, **kw)


class ADMM_Dantzig(ADMM_Lasso):
    """`admm_dantzig` is exported by the reference (NAMESPACE:13, R/50_admm_dantzig.R) but its `.Call("admm_dantzig", ...)`
    names a symbol the package never builds: the solver lives in src/TODO/ (Dantzig.cpp, ADMMDantzig.h, written against an
    older ADMMBase) and is not compiled, so `$fit()` fails in R.  Here fit() runs that algorithm restated on the current
    ADMMBase::solve (admm_hip_dantzig; admm_amd/csrc/dantzig.hip, oracle/solvers.py Dantzig) -- double precision like its
    `typedef double Scalar`.  It converges on comfortably tall problems (n >= 5 p) and, as the restated algorithm itself, not for
    p > n (tests/test_oracle_dantzig.py): niter = maxit + 1 marks such a lambda.  The builder chain is ADMM_Lasso's
    (R/50_admm_dantzig.R:2 `contains = "ADMM_Lasso"`): $parallel stores nthread as there and, as there, $fit ignores it."""
    _name = "ADMM Dantzig Selector model"

    _missing = "not available for the Dantzig selector (the reference has no such entry point)"

    def fit(self, trace=False):
        lib, head, tail, lam_out, _, niter, stats, keep = self._common()
        beta = np.zeros((self.p + 1, lam_out.size), dtype=np.float64, order="F")
        tr, ntr = _trace_buffers(self.maxit * lam_out.size + 2 * lam_out.size if trace else 0)
        check(lib.admm_hip_dantzig_traced(*head, tail[0], tail[1], beta.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), tail[3], tail[4],
                                          *_trace_args(tr, ntr)))
        fit = ADMM_Lasso_fit(lam_out, beta, niter, stats.as_dict())
        fit._title = "ADMM Dantzig Selector fitting result"                     # ADMM_Dantzig_fit$show, R/50_admm_dantzig.R:52-58
        fit.trace = tr[:ntr.value].copy() if trace else None
        return fit

    # the extensions of this build that ride on ADMM_Lasso (cross-validation, several responses, prepared problems, row blocks)
    # must not run a plain Lasso under a Dantzig label
    def cv(self, *a, **kw):
        _stop(self._missing)

    def fit_responses(self, *a, **kw):
        _stop(self._missing)


def admm_dantzig(x, y, intercept=True, standardize=True, **kw):
    return ADMM_Dantzig(x, y, intercept, standardize, **kw)
