"""Python side of the C restatement of the tall loop (oracle/c/admm_tall_cpu.c) -- TEST INFRASTRUCTURE.

`admm_lasso_c` / `admm_enet_c` mirror oracle.entry.admm_lasso / admm_enet for n > p: the one-time part
(double -> float copy, DataStd, X'y, Gram, the Spectra call, rho, LAPACK Cholesky) is shared with the NumPy
oracle (Lasso.cpp:42-89, ADMMLassoTall.h:164-216); the warm-started lambda loop (Lasso.cpp:97-124 ->
FADMMBase::solve) runs in compiled C, in one of the two CPU configurations bench.py reports:
  mode 0  two triangular solves per iteration on one thread (the reference's configuration),
  mode 1  cached-inverse mat-vec over `nthreads` OpenMP threads (best effort).
"""
import ctypes
import os
import subprocess

import numpy as np
import scipy.linalg as sla

from .datastd import DataStd
from .solvers import LassoTall
from .entry import _lambda_grid

F = np.float32
_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c")
_SO = os.path.join(_DIR, "liboracle_tall.so")
_lib = None


def build(force=False):
    """gcc the C restatement (oracle/c/Makefile).  Called by __graft_entry__.build()."""
    src = os.path.join(_DIR, "admm_tall_cpu.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        try:
            subprocess.run(["make", "-C", _DIR, "-B"], check=True, capture_output=True)
        except Exception:
            if not os.path.exists(_SO):          # a prebuilt library that travelled with the tree is still usable
                raise
    return _SO


def load():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(_SO)
        fp, dp, ip = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
        lib.oracle_tall_path.argtypes = [fp, fp, ctypes.c_int, dp, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                         ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, fp, ip, dp]
        lib.oracle_tall_path.restype = ctypes.c_int
        lib.oracle_tall_path_traced.argtypes = lib.oracle_tall_path.argtypes + [dp, ctypes.c_int, ip]
        lib.oracle_tall_path_traced.restype = ctypes.c_int
        lib.oracle_max_threads.restype = ctypes.c_int
        _lib = lib
    return _lib


def max_threads():
    return int(load().oracle_max_threads())


def tall_loop(factor, XY, lam_int, rho, eps_abs, eps_rel, maxit, alpha=None, mode=0, nthreads=1, trace=None):
    """The compiled loop on prepared inputs.  factor: Cholesky factor (mode 0) or inverse (mode 1), p x p float32
    column-major.  Returns (beta [nlam, p] standardised scale, niter [nlam], loop seconds).  trace: a list that receives
    one row per decision (lambda index, iteration, eps_p, eps_d, r_p, r_d, c, outcome)."""
    lib = load()
    p = XY.shape[0]
    Fm = np.asfortranarray(factor, dtype=F)
    XY = np.ascontiguousarray(XY, dtype=F)
    lam_int = np.ascontiguousarray(lam_int, dtype=np.float64)
    nl = lam_int.size
    beta = np.zeros((nl, p), dtype=F)
    niter = np.zeros(nl, dtype=np.int32)
    secs = ctypes.c_double()
    fp, dp, ip = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
    cap = nl * int(maxit) if trace is not None else 0
    tr = np.zeros((max(cap, 1), 8))
    ntr = ctypes.c_int(0)
    rc = lib.oracle_tall_path_traced(Fm.ctypes.data_as(fp), XY.ctypes.data_as(fp), p, lam_int.ctypes.data_as(dp), nl, float(rho),
                                     float(eps_abs), float(eps_rel), int(maxit), -1.0 if alpha is None else float(alpha), int(mode),
                                     int(nthreads), beta.ctypes.data_as(fp), niter.ctypes.data_as(ip), ctypes.byref(secs),
                                     tr.ctypes.data_as(dp) if cap else None, cap, ctypes.byref(ntr))
    if rc != 0:
        raise MemoryError("oracle_tall_path failed")
    if trace is not None:
        trace.extend(tr[:ntr.value].tolist())
    return beta, niter, secs.value


def _family(x, y, lam, nlambda, lmin_ratio, standardize, intercept, opts, alpha, mode, nthreads, trace=None):
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    n, p = x.shape
    assert n > p, "the C restatement covers the tall solver only"
    datX = np.array(x, dtype=F, order="F")
    datY = np.array(y, dtype=F)
    std = DataStd(n, p, standardize, intercept, F)
    std.standardize(datX, datY)
    s = LassoTall(datX, datY, float(opts["eps_abs"]), float(opts["eps_rel"]), alpha)
    lam = np.atleast_1d(np.asarray(lam, dtype=np.float64)) if lam is not None else np.zeros(0)
    if lam.size < 1:
        lam = _lambda_grid(s.lambda0, n, std.scaleY, nlambda, lmin_ratio)
    lam_int = np.array([np.float64(F(l * n / np.float64(std.scaleY))) for l in lam])
    s.init(lam_int[0], float(opts["rho"]))                   # Gram, Spectra, rho, LAPACK Cholesky (shared with the NumPy oracle)
    L = np.tril(s.chol[0])
    if mode == 0:
        factor = L
    else:
        Li = sla.solve_triangular(L.astype(np.float64), np.eye(p), lower=True)
        factor = (Li.T @ Li).astype(F)
    b, niter, secs = tall_loop(factor, s.XY, lam_int, s.rho, opts["eps_abs"], opts["eps_rel"], opts["maxit"], alpha, mode, nthreads, trace)
    beta = np.zeros((p + 1, lam.size), dtype=F)
    for i in range(lam.size):
        b0, coef = std.recover(b[i])
        beta[0, i] = b0
        beta[1:, i] = coef
    return {"lambda": lam, "beta": beta, "niter": niter, "loop_seconds": secs, "rho": s.rho}


def admm_lasso_c(x, y, lam, nlambda, lmin_ratio, standardize, intercept, opts, mode=0, nthreads=1, trace=None):
    return _family(x, y, lam, nlambda, lmin_ratio, standardize, intercept, opts, None, mode, nthreads, trace)


def admm_enet_c(x, y, lam, nlambda, lmin_ratio, standardize, intercept, alpha, opts, mode=0, nthreads=1, trace=None):
    return _family(x, y, lam, nlambda, lmin_ratio, standardize, intercept, opts, alpha, mode, nthreads, trace)
