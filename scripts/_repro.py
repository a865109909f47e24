import sys, os; sys.path.insert(0, ".")
import numpy as np, admm_amd
from oracle import entry
def make(seed, case, sf=1):
    rng = np.random.default_rng(seed)
    for c in range(case + 1):
        n = int(rng.integers(8 * sf, 400 * sf)); p = int(rng.integers(n + 3, 5 * n + 8)); N = int(rng.integers(2, 9))
        scale = float(rng.choice([0.05, 1.0, 1.0, 30.0]))
        A = np.asfortranarray(rng.standard_normal((n, p)) * scale + (rng.uniform(-1, 1) * scale if rng.uniform() < 0.3 else 0.0))
        k = int(rng.integers(1, max(2, n // 3)))
        b0 = np.zeros(p); b0[rng.choice(p, k, replace=False)] = rng.standard_normal(k) * rng.choice([0.1, 1.0, 10.0])
        b = A @ b0 + (1e-3 * rng.standard_normal(n) if rng.uniform() < 0.3 else 0.0)
        eps = float(rng.choice([1e-3, 1e-4, 1e-6])); maxit = int(rng.choice([60, 400, 3000])); ratio = float(rng.choice([0.5, 1.0, 1.0, 3.0]))
        cap = int(rng.choice([8, 16, 32, 64, 128, 1024, 1024])) if sf == 1 else int(rng.choice([256, 512, 1024, 1024]))
    return A, b, N, eps, maxit, ratio, cap
for seed, case, sf in ((437, 7, 1), (405, 0, 1), (738, 2, 4)):
    A, b, N, eps, maxit, ratio, cap = make(seed, case, sf)
    os.environ["ADMM_HIP_SBP_GRAM_CAP"] = str(cap)
    d = {"trace": []}
    ref = entry.admm_parbp(A, b, N, dict(maxit=maxit, eps_abs=eps, eps_rel=eps, rho_ratio=ratio), d)
    tr = np.asarray(d["trace"], dtype=np.float64)
    for rep in range(8):
        for env in ({}, {"ADMM_HIP_SBP_GRAM_CARRY": "0"}):
            for k_, v_ in env.items(): os.environ[k_] = v_
            fit = admm_amd.admm_bp(A, b).parallel(N).opts(maxit=maxit, eps_abs=eps, eps_rel=eps, rho=ratio).fit(trace=True)
            for k_ in env: del os.environ[k_]
            t = fit.trace[1:]
            m = min(len(t), len(tr))
            bad = []
            for col, oc in ((2, 1), (3, 2), (4, 3), (5, 4)):
                dd = np.abs(t[:m, col] - tr[:m, oc]) / max(np.abs(tr[:m, oc]).max(), 1e-300)
                j = int(np.argmax(dd))
                if dd[j] > 1e-9: bad.append((col, j, float(dd[j]), float(t[j, col]), float(tr[j, oc]), int(np.count_nonzero(dd > 1e-9))))
            print(seed, case, "rep", rep, env, "niter", fit.niter, ref["niter"], "variant", fit.stats["xupdate_variant"], "bad", bad, flush=True)
