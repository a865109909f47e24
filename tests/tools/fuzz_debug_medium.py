"""Dev tool (GPU box): one medium tall / elastic-net case in detail.   python tests/tools/fuzz_debug_medium.py ncases seed case"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402
import numpy as np  # noqa: E402
from fuzz_cases import medium_cases  # noqa: E402
from helpers import oracle_following, traced_fit, col_err, coef_scale  # noqa: E402
from admm_amd import admm_enet, admm_lasso  # noqa: E402
from oracle import entry  # noqa: E402

n_, seed, want = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
for cs in medium_cases(n_, seed):
    if cs["c"] != want:
        continue
    x, y, icpt, stdz = cs["x"], cs["y"], cs["icpt"], cs["stdz"]
    print({k: v for k, v in cs.items() if k not in ("x", "y")})
    opts = dict(maxit=cs["maxit"], eps_abs=cs["eps"], eps_rel=cs["eps"], rho=cs["rho"])
    lam = None
    if cs["user_lam"]:
        ref0 = entry.admm_lasso(x, y, None, 3, 0.1, stdz, icpt, dict(entry.LASSO_OPTS, maxit=1), {})
        lam = np.sort(ref0["lambda"][0] * cs["ulam"])[::-1]
    rho = None if cs["rho"] <= 0 else cs["rho"]
    if cs["kind"] == "enet_tall":
        m = admm_enet(x, y, icpt, stdz).penalty(lam, nlambda=cs["nl"], alpha=cs["alpha"])
    else:
        m = admm_lasso(x, y, icpt, stdz).penalty(lam, nlambda=cs["nl"])
    m.opts(cs["maxit"], cs["eps"], cs["eps"], rho)
    fit, trace = traced_fit(m, capacity=cs["nl"] * (cs["maxit"] + 2) + 8)
    problem = dict(x=x, y=y, lam=lam, nlambda=cs["nl"], lmin_ratio=1e-4, standardize=stdz, intercept=icpt, opts=opts, alpha=cs["alpha"])
    ref, forced, ndec = oracle_following(trace, band=8.0, **problem)
    print("niter gpu", fit.niter, "\nniter ref", ref["niter"], "forced", len(forced))
    floor = 1e-2 * max(float(np.abs(ref["beta"]).max()), coef_scale(problem))
    var = {}
    for mode in ("inv32", "exact"):
        v, _, _ = oracle_following(trace, band=1e9, mode=mode, **problem)
        var[mode] = v["beta"]
    for j in range(fit.beta_dense.shape[1]):
        bg, br = fit.beta_dense[:, j].astype(float), ref["beta"][:, j].astype(float)
        k = int(np.argmax(np.abs(bg - br)))
        print(f"lambda {j} ({fit.lambda_[j]:.4g}): err {col_err(bg, br, floor):.2e} | drift inv32 {col_err(var['inv32'][:, j], br, floor):.2e} exact {col_err(var['exact'][:, j], br, floor):.2e}"
              f" | max|beta| {np.abs(br).max():.4g} nnz gpu {np.count_nonzero(bg)} ref {np.count_nonzero(br)} | worst coef {k}: gpu {bg[k]:.7g} ref {br[k]:.7g} inv32 {var['inv32'][k, j]:.7g} exact {var['exact'][k, j]:.7g}")
    print("rho", fit.stats["rho"], "floor", floor)
