"""Dev tool (GPU box): one case of the small random stream in detail.   python tests/tools/fuzz_debug.py ncases seed case"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402
import numpy as np  # noqa: E402
from fuzz_cases import cases  # noqa: E402
from helpers import oracle_following, traced_fit, col_err  # noqa: E402
from admm_amd import admm_enet, admm_lasso  # noqa: E402
from oracle import entry  # noqa: E402

n_, seed, want = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
for cs in cases(n_, seed):
    if cs["c"] != want:
        continue
    kind, x, y, n, p, icpt, stdz = (cs[k] for k in ("kind", "x", "y", "n", "p", "icpt", "stdz"))
    print({k: v for k, v in cs.items() if k not in ("x", "y")})
    opts = dict(entry.LASSO_OPTS)
    lmr = 0.01 if n < p else 1e-4
    lam = None
    if cs["user_lam"]:
        ref0 = entry.admm_lasso(x, y, None, 3, 0.1, stdz, icpt, dict(opts, maxit=1), {})
        lam = np.sort(ref0["lambda"][0] * cs["ulam"])[::-1]
    nl = cs["nl"]
    if kind.startswith("enet"):
        m = admm_enet(x, y, icpt, stdz).penalty(lam, nlambda=nl, lambda_min_ratio=lmr, alpha=cs["alpha"])
    else:
        m = admm_lasso(x, y, icpt, stdz).penalty(lam, nlambda=nl, lambda_min_ratio=lmr)
    fit, trace = traced_fit(m, capacity=max(nl, 1) * (opts["maxit"] + 2) + 8)
    prob = dict(x=x, y=y, lam=lam, nlambda=nl, lmin_ratio=lmr, standardize=stdz, intercept=icpt, opts=opts, alpha=cs["alpha"])
    try:
        ref, forced, ndec = oracle_following(trace, band=8.0, **prob)
    except Exception as e:                                  # noqa: BLE001 -- show how far apart the two executions are at the mismatch
        print("FOLLOW FAILED:", e)
        d = {"trace": []}
        plain = (entry.admm_enet(x, y, lam, nl, lmr, stdz, icpt, cs["alpha"], opts, d) if kind.startswith("enet") else entry.admm_lasso(x, y, lam, nl, lmr, stdz, icpt, opts, d))
        to = np.asarray(d["trace"], dtype=float)
        tg = np.asarray(trace, dtype=float)
        tg = tg[1:] if tg[0, 8] == -1 else tg
        m_ = min(len(to), len(tg))
        same = (to[:m_, 0] == tg[:m_, 0]) & (to[:m_, 1] == tg[:m_, 1])
        k = int(np.argmin(same)) if not same.all() else m_
        print("records in lock step:", k, "of", m_)
        for r in (k - 3, k - 2, k - 1):
            if r >= 0:
                print("rec", r, "lam/iter", tg[r, :2], "gpu rp, rd, eps_p, eps_d", tg[r, 4], tg[r, 5], tg[r, 2], tg[r, 3], "| oracle", to[r, 4], to[r, 5], to[r, 2], to[r, 3],
                      "| rel diff rp %.2e rd %.2e" % (abs(tg[r, 4] / to[r, 4] - 1), abs(tg[r, 5] / max(to[r, 5], 1e-300) - 1)), "outcome gpu/oracle", tg[r, 8], to[r, 8])
        sys.exit(0)
    plain = (entry.admm_enet(x, y, lam, nl, lmr, stdz, icpt, cs["alpha"], opts) if kind.startswith("enet") else entry.admm_lasso(x, y, lam, nl, lmr, stdz, icpt, opts))
    print("lambda", fit.lambda_, "\nniter gpu", fit.niter, "\nniter followed", ref["niter"], "\nniter plain oracle", plain["niter"])
    print("forced", forced[:5], "ndec", ndec, "trace len", len(trace))
    floor = 1e-2 * float(np.abs(ref["beta"]).max())
    for j in range(nl):
        bg, br, bp = fit.beta_dense[:, j].astype(float), ref["beta"][:, j].astype(float), plain["beta"][:, j].astype(float)
        k = int(np.argmax(np.abs(bg - br)))
        print(f"lambda {j}: err vs followed {col_err(bg, br, floor):.2e} vs plain {col_err(bg, bp, floor):.2e} | max|beta| gpu {np.abs(bg).max():.4g} ref {np.abs(br).max():.4g} | worst coef {k}: gpu {bg[k]:.6g} ref {br[k]:.6g} plain {bp[k]:.6g}")
    t = np.asarray(trace)
    print("trace head\n", t[:4, :11])
    print("rho", fit.stats["rho"], "eig", fit.stats["eig_est"])
