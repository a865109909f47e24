"""Python side of the C restatements of the wide / consensus / LAD / BP loops (oracle/c/admm_loops_cpu.c) -- TEST INFRASTRUCTURE.

Each `*_c` function mirrors the oracle.entry function of the same name: the one-time part (DataStd, Gram, the Spectra call,
Cholesky factors, L^-1 A ...) is the NumPy oracle's own constructor (oracle/solvers.py), the iteration loop runs in compiled C
on `nthreads` OpenMP threads.  Used by tests/test_oracle_cloops.py (C against NumPy) and by bench.py's cpu_baseline legs of
the configs other than the headline (timed only)."""
import ctypes
import os
import subprocess

import numpy as np
import scipy.linalg as sla

from .datastd import DataStd
from .entry import _lambda_grid
from .solvers import BP, LAD, LassoWide, PADMMLasso, SharingBP

F = np.float32
_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c")
_SO = os.path.join(_DIR, "liboracle_loops.so")
_lib = None
_fp, _dp, _ip = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)


def build(force=False):
    src = os.path.join(_DIR, "admm_loops_cpu.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        try:
            subprocess.run(["make", "-C", _DIR, "liboracle_loops.so", "-B"], check=True, capture_output=True)
        except Exception:
            if not os.path.exists(_SO):          # a prebuilt library that travelled with the tree is still usable
                raise
    return _SO


def load():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(_SO)
        c_int, c_dbl, c_long = ctypes.c_int, ctypes.c_double, ctypes.c_long
        lib.oracle_wide_path.argtypes = [_fp, c_long, c_int, c_int, _fp, ctypes.c_float, ctypes.c_float, _dp, c_int, c_dbl, c_dbl, c_dbl, c_int, c_int,
                                         _fp, _ip, _dp, ctypes.POINTER(ctypes.c_longlong), _dp, c_int, _ip, c_dbl]
        lib.oracle_wide_path.restype = c_int
        pp = ctypes.POINTER(_fp)
        lib.oracle_consensus_path.argtypes = [pp, _ip, pp, pp, c_int, c_int, _dp, c_int, c_dbl, c_dbl, c_dbl, c_int, c_int, _fp, _ip, _dp, _dp, c_int, _ip, c_dbl]
        lib.oracle_consensus_path.restype = c_int
        ppd = ctypes.POINTER(_dp)
        lib.oracle_consensus_path_exact.argtypes = [pp, _ip, pp, ppd, c_int, c_int, _dp, c_int, c_dbl, c_dbl, c_dbl, c_int, c_int, _fp, _ip, _dp, _dp, c_int, _ip, c_dbl]
        lib.oracle_consensus_path_exact.restype = c_int
        lib.oracle_dense_loop.argtypes = [c_int, _dp, c_int, c_int, _dp, _dp, c_dbl, c_dbl, c_dbl, c_int, c_int, _dp, _dp, _dp, _dp, _ip, _dp, _dp, c_int, _ip, c_dbl]
        lib.oracle_dense_loop.restype = c_int
        lib.oracle_sharing_loop.argtypes = [_dp, c_long, c_int, c_int, c_int, _dp, _dp, c_dbl, c_dbl, c_dbl, c_int, c_int, _dp, _ip, _dp, _dp, c_int, _ip, c_dbl]
        lib.oracle_sharing_loop.restype = c_int
        lib.oracle_loops_max_threads.restype = c_int
        _lib = lib
    return _lib


def max_threads():
    return int(load().oracle_loops_max_threads())


def _trace_buf(cap):
    return np.zeros((max(cap, 1), 12)), ctypes.c_int(0)


# ------------------------------------------------------------------------------------------------------------------- wide
def wide_loop(X, Y, sprad, lambda0, lam_int, rho, eps_abs, eps_rel, maxit, nthreads=1, trace=None, want_beta=True, budget_s=0.0):
    """The compiled ADMMLassoWide loop on prepared (standardised, float32, column-major) data.  Returns (x per lambda [nlam, p] or
    None, niter, loop seconds, sum over iterations of the non-zeros after the x-update)."""
    lib = load()
    X = np.asfortranarray(X, dtype=F)
    n, p = X.shape
    Y = np.ascontiguousarray(Y, dtype=F)
    lam_int = np.ascontiguousarray(lam_int, dtype=np.float64)
    nl = lam_int.size
    beta = np.zeros((nl, p), dtype=F) if want_beta else None
    niter = np.zeros(nl, dtype=np.int32)
    secs, nnz = ctypes.c_double(), ctypes.c_longlong()
    cap = nl * int(maxit) if trace is not None else 0
    tr, ntr = _trace_buf(cap)
    rc = lib.oracle_wide_path(X.ctypes.data_as(_fp), n, n, p, Y.ctypes.data_as(_fp), float(sprad), float(lambda0), lam_int.ctypes.data_as(_dp), nl,
                              float(rho), float(eps_abs), float(eps_rel), int(maxit), int(nthreads), beta.ctypes.data_as(_fp) if want_beta else None,
                              niter.ctypes.data_as(_ip), ctypes.byref(secs), ctypes.byref(nnz), tr.ctypes.data_as(_dp) if cap else None, cap, ctypes.byref(ntr), float(budget_s))
    if rc != 0:
        raise MemoryError("oracle_wide_path failed")
    if trace is not None:
        trace.extend(tr[:ntr.value].tolist())
    return beta, niter, secs.value, int(nnz.value)


def admm_lasso_wide_c(x, y, lam, nlambda, lmin_ratio, standardize, intercept, opts, nthreads=1, trace=None):
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    n, p = x.shape
    assert n <= p, "the wide solver (ADMMLassoWide) is the n <= p branch of Lasso.cpp"
    datX = np.array(x, dtype=F, order="F")
    datY = np.array(y, dtype=F)
    std = DataStd(n, p, standardize, intercept, F)
    std.standardize(datX, datY)
    s = LassoWide(datX, datY, float(opts["eps_abs"]), float(opts["eps_rel"]))
    lam = np.atleast_1d(np.asarray(lam, dtype=np.float64)) if lam is not None else np.zeros(0)
    if lam.size < 1:
        lam = _lambda_grid(s.lambda0, n, std.scaleY, nlambda, lmin_ratio)
    lam_int = np.array([np.float64(F(l * n / np.float64(std.scaleY))) for l in lam])
    b, niter, secs, nnz = wide_loop(datX, datY, s.sprad, s.lambda0, lam_int, float(opts["rho"]), opts["eps_abs"], opts["eps_rel"], opts["maxit"], nthreads, trace)
    beta = np.zeros((p + 1, lam.size), dtype=F)
    for i in range(lam.size):
        b0, coef = std.recover(b[i])
        beta[0, i] = b0
        beta[1:, i] = coef
    return {"lambda": lam, "beta": beta, "niter": niter, "loop_seconds": secs, "nnz_sum": nnz, "sprad": float(s.sprad)}


# ------------------------------------------------------------------------------------------------------------------- consensus
def consensus_loop(A, Ab, Lf, p, lam_int, rho, eps_abs, eps_rel, maxit, nthreads=1, trace=None, budget_s=0.0, exact=False):
    """A: list of row blocks (rows_k x p float32, column-major), Ab: their A_k'b_k, Lf: Cholesky factors (lower, float32,
    column-major) of A_k'A_k + rho I (tall block) or A_k A_k' + rho I (wide block).  Returns (z per lambda, niter, seconds).
    exact=True: Lf are the factors of the same float systems in DOUBLE and the workers' small solves run in double (the rounding
    variant "exact" of oracle/solvers.py PADMMLasso)."""
    lib = load()
    K = len(A)
    A = [np.asfortranarray(a, dtype=F) for a in A]
    Ab = [np.ascontiguousarray(v, dtype=F) for v in Ab]
    Lf = [np.asfortranarray(np.tril(l), dtype=np.float64 if exact else F) for l in Lf]
    rows = np.asarray([a.shape[0] for a in A], dtype=np.int32)
    arr = lambda xs: (_fp * K)(*[v.ctypes.data_as(_fp) for v in xs])
    lam_int = np.ascontiguousarray(lam_int, dtype=np.float64)
    nl = lam_int.size
    beta = np.zeros((nl, p), dtype=F)
    niter = np.zeros(nl, dtype=np.int32)
    secs = ctypes.c_double()
    cap = nl * int(maxit) if trace is not None else 0
    tr, ntr = _trace_buf(cap)
    fn, larr = (lib.oracle_consensus_path_exact, (_dp * K)(*[v.ctypes.data_as(_dp) for v in Lf])) if exact else (lib.oracle_consensus_path, arr(Lf))
    rc = fn(arr(A), rows.ctypes.data_as(_ip), arr(Ab), larr, K, int(p), lam_int.ctypes.data_as(_dp), nl, float(rho),
                                   float(eps_abs), float(eps_rel), int(maxit), int(nthreads), beta.ctypes.data_as(_fp), niter.ctypes.data_as(_ip),
                                   ctypes.byref(secs), tr.ctypes.data_as(_dp) if cap else None, cap, ctypes.byref(ntr), float(budget_s))
    if rc != 0:
        raise MemoryError("oracle_consensus_path failed")
    if trace is not None:
        trace.extend(tr[:ntr.value].tolist())
    return beta, niter, secs.value


def admm_parlasso_c(x, y, lam, nlambda, lmin_ratio, standardize, intercept, nthread, opts, nthreads=1, trace=None, exact=False):
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    n, p = x.shape
    datX = np.array(x, dtype=F, order="F")
    datY = np.array(y, dtype=F)
    std = DataStd(n, p, standardize, intercept, F)
    std.standardize(datX, datY)
    s = PADMMLasso(datX, datY, int(nthread), float(opts["eps_abs"]), float(opts["eps_rel"]))
    lam = np.atleast_1d(np.asarray(lam, dtype=np.float64)) if lam is not None else np.zeros(0)
    if lam.size < 1:
        lam = _lambda_grid(s.lambda0, n, std.scaleY, nlambda, lmin_ratio)
    lam_int = np.array([l * n / np.float64(std.scaleY) for l in lam])
    if exact:
        s.xmode = "exact"                                    # instance attribute: double factors of the same float systems
    s.init(lam_int[0], float(opts["rho"]))                   # rho = lambda / K, the workers' float LLT factors (PADMMLasso.h:48-63)
    Lf = [np.tril(c[0]) for c in s.chol]
    b, niter, secs = consensus_loop(s.A, s.Ab, Lf, p, lam_int, s.rho, opts["eps_abs"], opts["eps_rel"], opts["maxit"], nthreads, trace, exact=exact)
    beta = np.zeros((p + 1, lam.size), dtype=F)
    for i in range(lam.size):
        b0, coef = std.recover(b[i])
        beta[0, i] = b0
        beta[1:, i] = coef
    return {"lambda": lam, "beta": beta, "niter": niter, "loop_seconds": secs, "rho": s.rho}


# ------------------------------------------------------------------------------------------------------------------- LAD / BP
def dense_loop(prob, M, Lf, dvec, rho, eps_abs, eps_rel, maxit, nthreads=1, trace=None, budget_s=0.0):
    lib = load()
    M = np.asfortranarray(M, dtype=np.float64)
    n, p = M.shape
    dim = n if prob == 0 else p
    Lf = np.asfortranarray(np.tril(Lf), dtype=np.float64) if Lf is not None else None
    dvec = np.ascontiguousarray(dvec, dtype=np.float64)
    z, az, ay = np.zeros(dim), np.zeros(dim), np.zeros(dim)
    rho_out, secs, niter = ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
    cap = int(maxit) if trace is not None else 0
    tr, ntr = _trace_buf(cap)
    rc = lib.oracle_dense_loop(int(prob), M.ctypes.data_as(_dp), n, p, Lf.ctypes.data_as(_dp) if Lf is not None else None, dvec.ctypes.data_as(_dp),
                               float(rho), float(eps_abs), float(eps_rel), int(maxit), int(nthreads), z.ctypes.data_as(_dp), az.ctypes.data_as(_dp),
                               ay.ctypes.data_as(_dp), ctypes.byref(rho_out), ctypes.byref(niter), ctypes.byref(secs),
                               tr.ctypes.data_as(_dp) if cap else None, cap, ctypes.byref(ntr), float(budget_s))
    if rc != 0:
        raise MemoryError("oracle_dense_loop failed")
    if trace is not None:
        trace.extend(tr[:ntr.value].tolist())
    return z, az, ay, rho_out.value, niter.value, secs.value


def admm_lad_c(x, y, intercept, opts, nthreads=1, trace=None):
    """General branch X (X'X)^-1 X' only (the reference's n > 2000 branch, ADMMLAD.h:75-76) whatever n is."""
    x = np.array(x, dtype=np.float64, order="F")
    y = np.array(y, dtype=np.float64)
    n, p = x.shape
    std = DataStd(n, p, True, intercept, np.float64)
    std.standardize(x, y)
    chol = sla.cho_factor(x.T @ x, lower=True, check_finite=False)
    z, az, ay, rho, niter, secs = dense_loop(0, x, chol[0], y, float(opts["rho"]), opts["eps_abs"], opts["eps_rel"], opts["maxit"], nthreads, trace)
    coef = sla.cho_solve(chol, x.T @ (y - ay / rho + az), check_finite=False)          # get_x(): ADMMLAD.h:220-225
    beta0, coef = std.recover(coef)
    return {"beta": np.concatenate([[beta0], coef]), "niter": niter, "loop_seconds": secs, "rho": rho}


def admm_bp_c(x, y, opts, nthreads=1, trace=None):
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    s = BP(x, y, float(opts["rho"]), float(opts["eps_abs"]), float(opts["eps_rel"]))
    z, _, _, rho, niter, secs = dense_loop(1, s.LinvA, None, s.cache_AAAb, float(opts["rho"]), opts["eps_abs"], opts["eps_rel"], opts["maxit"], nthreads, trace)
    return {"beta": z, "niter": niter, "loop_seconds": secs, "rho": rho}


def sharing_loop(A, b, N, sprad, rho, eps_abs, eps_rel, maxit, nthreads=1, trace=None, budget_s=0.0):
    """The loop of oracle/solvers.py SharingBP.solve in C.  trace: list that receives the class's own 7-field records."""
    lib = load()
    A = np.asfortranarray(A, dtype=np.float64)
    n, p = A.shape
    b = np.ascontiguousarray(b, dtype=np.float64)
    sprad = np.ascontiguousarray(sprad, dtype=np.float64)
    x = np.zeros(p)
    secs, niter = ctypes.c_double(), ctypes.c_int()
    cap = int(maxit) if trace is not None else 0
    tr, ntr = np.zeros((max(cap, 1), 7)), ctypes.c_int(0)
    rc = lib.oracle_sharing_loop(A.ctypes.data_as(_dp), n, n, p, int(N), b.ctypes.data_as(_dp), sprad.ctypes.data_as(_dp), float(rho), float(eps_abs),
                                 float(eps_rel), int(maxit), int(nthreads), x.ctypes.data_as(_dp), ctypes.byref(niter), ctypes.byref(secs),
                                 tr.ctypes.data_as(_dp) if cap else None, cap, ctypes.byref(ntr), float(budget_s))
    if rc != 0:
        raise MemoryError("oracle_sharing_loop failed")
    if trace is not None:
        trace.extend(tuple(row) for row in tr[:ntr.value].tolist())
    return x, niter.value, secs.value


def admm_parbp_c(x, y, nthread, opts, nthreads=1, trace=None):
    """Mirror of oracle.entry.admm_parbp: the constructor (partition, exact spectral radii, rho) is the NumPy class's, the loop is C."""
    s = SharingBP(x, y, int(nthread), float(opts["eps_abs"]), float(opts["eps_rel"]))
    rho = 1.0 / (float(opts["rho_ratio"]) * float(np.mean(s.sprad)))
    beta, niter, secs = sharing_loop(x, y, int(nthread), s.sprad, rho, opts["eps_abs"], opts["eps_rel"], opts["maxit"], nthreads, trace)
    return {"beta": beta, "niter": niter, "loop_seconds": secs, "rho": rho}
