#!/usr/bin/env python
"""Soak of admm_hip_parbp's Gram-space active-set iterations against the oracle (oracle/solvers.py SharingBP), decision by
decision, on randomised problems -- the drawing of tests/test_gpu_fuzz_new.py with more seeds, larger shapes and, per case, a
random capacity of the Gram matrix (ADMM_HIP_SBP_GRAM_CAP) so that halts (support larger than the matrix), re-entries and
rebuilds of the column set are exercised as well as the plain path.  Prints one line per seed and a summary (markdown).

    python tests/tools/parbp_gram_soak.py [first_seed [n_seeds [cases_per_seed [shape_factor]]]]

shape_factor f > 1 draws n from 8f .. 400f (supports of several hundred columns: U near the default capacity of 1024).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def rel(a, b, floor=1e-300):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(float(np.abs(np.asarray(b)).max()), floor))


def main():
    import admm_amd
    from oracle import entry
    s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    ns = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    nc = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    sf = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    tot = dict(cases=0, fail=0, trace=0.0, beta=0.0, v0=0, v1=0, v2=0, stretches=0, rebuilds=0, iters=0)
    fails = []
    for seed in range(s0, s0 + ns):
        rng = np.random.default_rng(seed)
        worst = 0.0
        for c in range(nc):
            n = int(rng.integers(8 * sf, 400 * sf))
            p = int(rng.integers(n + 3, 5 * n + 8))
            N = int(rng.integers(2, 9))
            scale = float(rng.choice([0.05, 1.0, 1.0, 30.0]))
            A = np.asfortranarray(rng.standard_normal((n, p)) * scale + (rng.uniform(-1, 1) * scale if rng.uniform() < 0.3 else 0.0))
            k = int(rng.integers(1, max(2, n // 3)))
            b0 = np.zeros(p); b0[rng.choice(p, k, replace=False)] = rng.standard_normal(k) * rng.choice([0.1, 1.0, 10.0])
            b = A @ b0 + (1e-3 * rng.standard_normal(n) if rng.uniform() < 0.3 else 0.0)
            eps = float(rng.choice([1e-3, 1e-4, 1e-6]))
            maxit = int(rng.choice([60, 400, 3000]))
            ratio = float(rng.choice([0.5, 1.0, 1.0, 3.0]))
            cap = int(rng.choice([8, 16, 32, 64, 128, 1024, 1024])) if sf == 1 else int(rng.choice([256, 512, 1024, 1024]))
            os.environ["ADMM_HIP_SBP_GRAM_CAP"] = str(cap)
            fit = admm_amd.admm_bp(A, b).parallel(N).opts(maxit=maxit, eps_abs=eps, eps_rel=eps, rho=ratio).fit(trace=True)
            d = {"trace": []}
            ref = entry.admm_parbp(A, b, N, dict(maxit=maxit, eps_abs=eps, eps_rel=eps, rho_ratio=ratio), d)
            tr = np.asarray(d["trace"], dtype=np.float64)
            t = fit.trace[1:]
            label = f"seed {seed} case {c}: n={n} p={p} N={N} k={k} scale={scale} eps={eps} maxit={maxit} rho_ratio={ratio} cap={cap}"
            st = fit.stats
            tot["cases"] += 1
            tot["v%d" % st["xupdate_variant"]] += 1
            tot["stretches"] += int(st["xupdate_launches"]); tot["rebuilds"] += int(st["persist_iter"]); tot["iters"] += int(fit.niter)
            ok = fit.niter == ref["niter"] and len(t) == len(tr)
            e = eb = float("nan")
            if ok:
                e = max(rel(t[:, 2], tr[:, 1]), rel(t[:, 3], tr[:, 2]), rel(t[:, 4], tr[:, 3]), float(np.abs(t[:, 5] - tr[:, 4]).max() / max(tr[:, 4].max(), 1e-300)))
                eb = rel(fit.beta.toarray().ravel(), ref["beta"])
                ok = e < 1e-7 and eb < 1e-8 and np.array_equal(t[:, 11], tr[:, 5]) and np.array_equal(t[:, 8] == 0, tr[:, 6] == 1)
                tot["trace"], tot["beta"] = max(tot["trace"], e), max(tot["beta"], eb)
                worst = max(worst, e)
            if not ok:
                tot["fail"] += 1
                fails.append(f"{label}: niter {fit.niter} / {ref['niter']}, trace {e:.1e}, beta {eb:.1e}")
        print(f"seed {seed}: {nc} cases, worst trace difference {worst:.1e}", flush=True)
    print(f"\n| cases | failed | iterations | largest trace difference | largest coefficient difference | all direct / Gram space / Gram space with halts | Gram-space stretches | rebuilds of U |")
    print("|---|---|---|---|---|---|---|---|")
    print(f"| {tot['cases']} | {tot['fail']} | {tot['iters']} | {tot['trace']:.1e} | {tot['beta']:.1e} | {tot['v0']} / {tot['v1']} / {tot['v2']} | {tot['stretches']} | {tot['rebuilds']} |")
    for f in fails:
        print("FAIL", f)
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
