#!/usr/bin/env python
"""Measure the non-headline BASELINE.json configs (C3 wide, C4 consensus on one GPU, C5 LAD / BP) on
device-resident synthetic data: iterations, loop seconds, iterations/s and achieved algorithmic GB/s
(SURVEY.md section 8d byte counts).  Prints one JSON line per config.  Usage: bench_configs.py [c3 c4 c5lad c5bp c5parbp dantzig]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (before libadmm_hip)
import numpy as np  # noqa: E402
from admm_amd import DevicePtr, admm_bp, admm_dantzig, admm_lad, admm_lasso  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(123)


def gen(n, p, sd, m, dense_beta=False, noise=True):
    xt = torch.empty((p, n), dtype=torch.float64, device=dev)
    chunk = max(1, (1 << 27) // n)
    for c0 in range(0, p, chunk):
        c1 = min(p, c0 + chunk)
        xt[c0:c1] = torch.randn((c1 - c0, n), generator=g, device=dev, dtype=torch.float64) * sd
    b = torch.zeros(p, dtype=torch.float64, device=dev)
    if dense_beta:
        b[:] = torch.rand(p, generator=g, device=dev, dtype=torch.float64)
    else:
        idx = torch.randperm(p, generator=g, device=dev)[:m] if not noise else torch.arange(m, device=dev)
        b[idx] = torch.rand(m, generator=g, device=dev, dtype=torch.float64)
    y = b @ xt
    if noise:
        y = y + torch.randn(n, generator=g, device=dev, dtype=torch.float64)
    torch.cuda.synchronize()
    return xt, y, b


def report(name, fit, bytes_per_iter, extra=None):
    st = fit.stats
    it = int(st["total_iter"])
    out = {"config": name, "iterations": it, "loop_s": st["t_loop"], "iters_per_s": it / st["t_loop"],
           "alg_GB_per_iter": bytes_per_iter / 1e9, "achieved_GBps": bytes_per_iter * it / st["t_loop"] / 1e9,
           "frac_of_8TBps": bytes_per_iter * it / st["t_loop"] / 8e12,
           "setup_s": {k: round(st[k], 4) for k in ("t_standardize", "t_gram", "t_eigs", "t_factor")}, "total_s": st["t_total"]}
    if extra:
        out.update(extra)
    print(json.dumps(out), flush=True)


which = sys.argv[1:] or ["c3", "c4", "c5lad", "c5bp"]
if "c3" in which:       # wide Lasso n=2000 p=200000, 100-lambda path (lambda_min_ratio 0.01)
    n, p = 2000, 200000
    xt, y, _ = gen(n, p, 2.0, 100)
    fit = admm_lasso(DevicePtr(xt.data_ptr()), DevicePtr(y.data_ptr()), n=n, p=p).penalty(nlambda=100).fit()
    nnz = int(np.count_nonzero(fit.beta_dense[1:, -1]))
    # bytes: a regular iteration streams X once (4np); every iteration reads the support twice (8 n nS): nS unknown per
    # iteration, so only the regular-step stream is counted here (lower bound of the traffic)
    niter = fit.niter.astype(np.int64)
    reg = sum(int(sum(1 for c in range(k) if (c + 1) & c == 0 and ((c + 1) & 0x55555555))) for k in niter)
    scr = int(fit.stats["xupdate_variant"])                     # regular steps screened: 0 no (4np), 1 fp16 copy (2np), 2 8-bit code (np)
    report("C3 admm_lasso wide n=2000 p=200000 nlambda=100", fit, {0: 4.0, 1: 2.0, 2: 1.0}[scr] * n * p * reg / max(1, int(niter.sum())),
           {"regular_iterations": reg, "nnz_last_lambda": nnz, "niter_minmax": [int(niter.min()), int(niter.max())],
            "regular_steps_screened": {0: False, 1: "fp16 copy", 2: "8-bit code"}[scr]})
    del xt
    torch.cuda.empty_cache()
if "c4" in which:       # consensus Lasso n=10000 p=100000, 8 row blocks on ONE GPU (Woodbury branch), short path
    n, p, K = 10000, 100000, 8
    xt, y, _ = gen(n, p, 2.0, 100)
    # a lambda range on which the consensus iteration converges (like bench.py's child run): the rate is not that of runs cut off at maxit
    fit = admm_lasso(DevicePtr(xt.data_ptr()), DevicePtr(y.data_ptr()), n=n, p=p).penalty(nlambda=3, lambda_min_ratio=0.3).parallel(K).opts(maxit=4000).fit()
    report("C4 admm_lasso$parallel(8) n=10000 p=100000 (8 virtual workers on 1 GPU), 3 lambdas down to 0.3 lambda_max, run to convergence", fit, 4.0 * n * p + 4.0 * K * (n / K) ** 2,
           {"niter": [int(v) for v in fit.niter]})
    del xt
    torch.cuda.empty_cache()
if "c5lad" in which:    # LAD n=50000 p=5000 fp64
    n, p = 50000, 5000
    xt, y, _ = gen(n, p, 2.0, p, dense_beta=True)
    fit = admm_lad(DevicePtr(xt.data_ptr()), DevicePtr(y.data_ptr()), intercept=False, n=n, p=p).fit()
    onepass = int(fit.stats["xupdate_variant"]) == 1           # one pass over the rows of X per iteration (8np) instead of the reference's two products (16np)
    report("C5 admm_lad n=50000 p=5000 fp64", fit, (8.0 if onepass else 16.0) * n * p + 8.0 * p * p, {"one_pass": onepass})
    del xt
    torch.cuda.empty_cache()
if "c5bp" in which:     # BP n=5000 p=50000 fp64, 500 non-zeros, exact y
    n, p = 5000, 50000
    xt, y, b = gen(n, p, 1.0, 500, noise=False)
    fit = admm_bp(DevicePtr(xt.data_ptr()), DevicePtr(y.data_ptr()), n=n, p=p).fit()
    beta = np.asarray(fit.beta.todense()).ravel()
    err = beta - b.cpu().numpy()
    report("C5 admm_bp n=5000 p=50000 fp64", fit, 8.0 * n * p, {"recovery_error_range": [float(err.min()), float(err.max())]})
if "c5parbp" in which:  # the same problem by the column-block sharing solver (admm_bp$parallel(8): admm_hip_parbp), 8 blocks on ONE GPU
    n, p = 5000, 50000
    xt, y, b = gen(n, p, 1.0, 500, noise=False)
    for nb in (8,):
        fit = admm_bp(DevicePtr(xt.data_ptr()), DevicePtr(y.data_ptr()), n=n, p=p).parallel(nb).fit()
        beta = np.asarray(fit.beta.todense()).ravel()
        err = beta - b.cpu().numpy()
        it = int(fit.stats["total_iter"])
        # algorithmic bytes: a regular iteration (every 10th) reads the whole matrix once, 8np; an active-set iteration the non-zero columns twice
        reg = (it + 9) // 10
        scr = int(fit.stats["xupdate_variant"]) >= 4            # regular iterations screened through the fp16 copy (2np instead of 8np)
        report(f"C5 admm_bp$parallel({nb}) n=5000 p=50000 fp64 (sharing ADMM)", fit, (2.0 if scr else 8.0) * n * p * reg / max(it, 1),
               {"recovery_error_range": [float(err.min()), float(err.max())], "regular_iterations": reg, "nnz": int(np.count_nonzero(beta)),
                "lanczos_steps": int(fit.stats["xupdate_samples"]), "rho": fit.stats["rho"],
                "active_set_variant": int(fit.stats["xupdate_variant"]) & 3, "regular_iterations_screened": scr, "gram_space_stretches": int(fit.stats["xupdate_launches"]), "rebuilds_of_U": int(fit.stats["persist_iter"])})
if "dantzig" in which:  # Dantzig selector (admm_hip_dantzig, the reference's unbuilt TODO/ADMMDantzig.h), operator form: n = 50 000, p = 2000 fp64
    n, p = 50000, 2000
    xt, y, _ = gen(n, p, 2.0, 100)
    fit = admm_dantzig(DevicePtr(xt.data_ptr()), DevicePtr(y.data_ptr()), n=n, p=p).penalty(nlambda=5, lambda_min_ratio=0.05).fit()
    niter = [int(v) for v in fit.niter]
    # algorithmic bytes: A = X'X is applied twice per iteration (to rhs and to x), each time as X v then X't on the two stored layouts: 4 x 8 n p
    report("admm_dantzig n=50000 p=2000 fp64, 5 lambdas down to 0.05 lambda_max (operator form: 4 streaming products per iteration)", fit, 32.0 * n * p,
           {"niter": niter, "maxit_plus_one_marks_no_convergence": 10001, "nnz_last_lambda": int(np.count_nonzero(fit.beta_dense[1:, -1]))})
