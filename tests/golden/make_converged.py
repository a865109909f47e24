"""Writes the CONVERGED oracle fixtures at the FULL BASELINE shapes (round 6) -- every lambda run to the reference's own eps, no fixed maxit:

    c2_converged.npz   admm_lasso tall, n = 100 000, p = 10 000 (configs[1]): the first NL2 lambdas of the automatic 100-grid, warm-started,
                       by the compiled C restatement of the tall loop (oracle/c/admm_tall_cpu.c, mode 0: float Cholesky factor + two
                       triangular solves per iteration on one thread = FADMMBase.h:185-265, ADMMLassoTall.h:70-95)
    c3_converged.npz   admm_lasso wide, n = 2000, p = 200 000 (configs[2]): ten lambdas of the automatic 100-grid (every tenth), warm-started,
                       by oracle/c/admm_loops_cpu.c oracle_wide_path (ADMMBase.h:192-216, ADMMLassoWide.h:86-170)
    c4_converged.npz   admm_lasso$parallel(8), n = 10 000, p = 100 000 (configs[3]): lambda = 1, 0.55, 0.3 x lambda_max (the 3-point grid of
                       bench.py's c4 line), eight 1250 x 10^5 Woodbury workers, by oracle_consensus_path (PADMMBase.h:174-237, PADMMLasso.h:17-31)

on data the generators make from a seed (NumPy PCG64, whole-column chunks: the tests regenerate the same arrays).  Data only: the lambdas,
the coefficients (non-zeros: rows / columns / float32 values), niter, rho or the loose spectral radius, and the oracle's decision trace.  The
setup of each run (DataStd, Gram, Spectra call, Cholesky) is the NumPy oracle's constructor, as everywhere in oracle/ctall.py / cloops.py.

    python tests/golden/make_converged.py [c2] [c3] [c4]      (c2: ~10 min and ~25 GB of RAM on 8 cores; c3: ~2 min; c4: ~6 min, ~20 GB)

What a converged fixture can and cannot hold: a stopping or restart test that an ulp of rounding decides differently forks the trajectory,
and no fixed file can follow a fork (the live parity tests let the oracle FOLLOW the library through such near-ties, tests/helpers.py R1).
The script therefore prints, and stores as `margins`, the distance of every decision of the oracle's run from its threshold; the test that
uses the file requires the library to take every one of these decisions identically and reports the closest one."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

C2 = dict(n=100000, p=10000, m=1000, seed=123, nl=int(os.environ.get("C2C_NL", 20)))
C3 = dict(n=2000, p=200000, m=100, seed=3003, pick=tuple(range(5, 100, 10)))
C4 = dict(n=10000, p=100000, m=100, seed=4004, K=8, nl=3, lmin_ratio=0.3, maxit=4000)


def _sparse(beta):
    r, c = np.nonzero(beta)
    return dict(beta_rows=r.astype(np.int32), beta_cols=c.astype(np.int32), beta_vals=beta[r, c].astype(np.float32), beta_shape=np.asarray(beta.shape))


def dense_beta(g, prefix="beta"):
    b = np.zeros(tuple(int(v) for v in g[prefix + "_shape"]), dtype=np.float32)
    b[g[prefix + "_rows"], g[prefix + "_cols"]] = g[prefix + "_vals"]
    return b


def stop_margins(tr, accelerated):
    """Per decision: relative distance of the deciding quantity from its threshold (stopping test: max(r_p/eps_p, r_d/eps_d) against 1;
    accelerated solvers, when not converged: c / (0.999 c_old) against 1).  `tr` in the 12-column layout of include/admm_hip.h."""
    tr = np.asarray(tr, dtype=np.float64)
    stop = np.maximum(tr[:, 4] / tr[:, 2], tr[:, 5] / tr[:, 3])
    m = np.abs(stop - 1.0)
    if accelerated:
        nc = tr[:, 8] != 0
        with np.errstate(divide="ignore", invalid="ignore"):
            c = np.where(nc & (tr[:, 7] > 0), np.abs(tr[:, 6] / (0.999 * tr[:, 7]) - 1.0), np.inf)
        m = np.minimum(m, c)
    return m


def make_c2():
    from make_c2_short import c2_short_data
    from oracle import ctall, entry
    from oracle.datastd import DataStd
    c = C2
    t0 = time.time()
    x, y = c2_short_data(c["seed"], c["n"], c["p"], c["m"])
    print(f"[c2] data {time.time() - t0:.0f} s", flush=True)
    # the automatic grid (Lasso.cpp:78-89) needs lambda_0 = max|X'y| of the standardised float data (ADMMLassoTall.h:172-173)
    dx, dy = np.array(x, dtype=np.float32, order="F"), np.array(y, dtype=np.float32)
    std = DataStd(c["n"], c["p"], True, True, np.float32)
    std.standardize(dx, dy)
    lambda0 = np.float32(np.abs((dx.T @ dy).astype(np.float32)).max())
    lam = entry._lambda_grid(lambda0, c["n"], std.scaleY, 100, 1e-4)[:c["nl"]]
    del dx, dy
    trace = []
    ref = ctall.admm_lasso_c(x, y, lam, 100, 1e-4, True, True, dict(entry.LASSO_OPTS), mode=0, nthreads=1, trace=trace)
    t8 = np.asarray(trace)                                  # 8 columns: lambda, iteration, eps_p, eps_d, r_p, r_d, c, outcome
    # the tall C loop does not record c_old: rebuilt from the rule itself (FADMMBase.h:243-256: an accelerated step keeps its c, a
    # restart divides the old one by 0.999; init_warm keeps it across lambdas, the cold start has 9999)
    m = np.abs(np.maximum(t8[:, 4] / t8[:, 2], t8[:, 5] / t8[:, 3]) - 1.0)
    c_old = 9999.0
    for k, r in enumerate(t8):
        if r[7] == 0:
            continue
        m[k] = min(m[k], abs(r[6] / (0.999 * c_old) - 1.0))
        c_old = r[6] if r[7] == 1 else c_old / 0.999
    path = os.path.join(HERE, "c2_converged.npz")
    np.savez_compressed(path, n=c["n"], p=c["p"], m=c["m"], seed=c["seed"], lam=lam, niter=ref["niter"].astype(np.int64), rho=np.float64(ref["rho"]),
                        trace=t8, margins=m, **_sparse(ref["beta"]))
    print(f"[c2] wrote {path} {os.path.getsize(path)} bytes; niter {ref['niter'].tolist()} rho {ref['rho']} nnz {(ref['beta'][1:] != 0).sum(axis=0).tolist()}; "
          f"closest stopping / restart test {m.min():.2e}; loop {ref['loop_seconds']:.0f} s, total {time.time() - t0:.0f} s", flush=True)


def make_c3():
    from make_fullsize import lasso_data
    from oracle import cloops, entry
    from oracle.datastd import DataStd
    c = C3
    t0 = time.time()
    x, y = lasso_data(c["seed"], c["n"], c["p"], c["m"])
    dx, dy = np.array(x, dtype=np.float32, order="F"), np.array(y, dtype=np.float32)
    std = DataStd(c["n"], c["p"], True, True, np.float32)
    std.standardize(dx, dy)
    lambda0 = np.float32(np.abs((dx.T @ dy).astype(np.float32)).max())
    lam = entry._lambda_grid(lambda0, c["n"], std.scaleY, 100, 0.01)[list(c["pick"])]
    del dx, dy
    trace = []
    ref = cloops.admm_lasso_wide_c(x, y, lam, 100, 0.01, True, True, dict(entry.LASSO_OPTS), nthreads=cloops.max_threads(), trace=trace)
    tr = np.asarray(trace)
    m = stop_margins(tr, accelerated=False)
    path = os.path.join(HERE, "c3_converged.npz")
    np.savez_compressed(path, n=c["n"], p=c["p"], m=c["m"], seed=c["seed"], pick=np.asarray(c["pick"]), lam=lam, niter=ref["niter"].astype(np.int64),
                        sprad=np.float64(ref["sprad"]), trace=tr, margins=m, **_sparse(ref["beta"]))
    print(f"[c3] wrote {path} {os.path.getsize(path)} bytes; niter {ref['niter'].tolist()} sprad {ref['sprad']} nnz {(ref['beta'][1:] != 0).sum(axis=0).tolist()}; "
          f"closest stopping test {m.min():.2e}; loop {ref['loop_seconds']:.0f} s, total {time.time() - t0:.0f} s", flush=True)


def make_c4():
    from make_fullsize import lasso_data
    from oracle import cloops, entry
    c = C4
    t0 = time.time()
    x, y = lasso_data(c["seed"], c["n"], c["p"], c["m"])
    trace = []
    ref = cloops.admm_parlasso_c(x, y, None, c["nl"], c["lmin_ratio"], True, True, c["K"], dict(entry.LASSO_OPTS, maxit=c["maxit"]),
                                 nthreads=cloops.max_threads(), trace=trace)
    tr = np.asarray(trace)
    m = stop_margins(tr, accelerated=False)
    # The reference's OWN rounding sensitivity at convergence: the same run with the workers' small solves (A_k A_k' + rho I) s = t carried
    # out in double on the same float systems (rounding variant "exact" of oracle/solvers.py PADMMLasso; everything else -- the two float
    # products with A_k, rhs, z, y -- unchanged).  Stored beside the reference-arithmetic run: `beta_exact_*`, `niter_exact`.
    alt = cloops.admm_parlasso_c(x, y, None, c["nl"], c["lmin_ratio"], True, True, c["K"], dict(entry.LASSO_OPTS, maxit=c["maxit"]),
                                 nthreads=cloops.max_threads(), exact=True)
    ex = {"beta_exact_" + k[5:]: v for k, v in _sparse(alt["beta"]).items()}
    sys.path.insert(0, os.path.dirname(HERE))
    from helpers import col_err
    floor = 1e-2 * float(np.abs(ref["beta"]).max())
    drift = [float(col_err(alt["beta"][:, j], ref["beta"][:, j], floor)) for j in range(c["nl"])]
    print(f"[c4] exact-solve variant: niter {alt['niter'].tolist()}; distance from the reference-arithmetic run per column {drift}", flush=True)
    path = os.path.join(HERE, "c4_converged.npz")
    np.savez_compressed(path, **{k: v for k, v in c.items()}, lam=ref["lambda"], niter=ref["niter"].astype(np.int64), rho=np.float64(ref["rho"]),
                        trace=tr, margins=m, niter_exact=alt["niter"].astype(np.int64), drift_exact=np.asarray(drift), **ex, **_sparse(ref["beta"]))
    print(f"[c4] wrote {path} {os.path.getsize(path)} bytes; niter {ref['niter'].tolist()} rho {ref['rho']} nnz {(ref['beta'][1:] != 0).sum(axis=0).tolist()}; "
          f"closest stopping test {m.min():.2e}; loop {ref['loop_seconds']:.0f} s, total {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    for w in (sys.argv[1:] or ["c3", "c4", "c2"]):
        {"c2": make_c2, "c3": make_c3, "c4": make_c4}[w]()
