"""Dev tool (GPU box): medium-size randomised parity sweep of the Lasso family (matrix-core setup, symmetric
x-update, consensus with several blocks, random maxit / eps / rho).   python tests/tools/fuzz_medium.py [ncases] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import numpy as np
from admm_amd import admm_lasso, admm_enet
from admm_amd._lib import check
from oracle import entry
from fuzz_cases import medium_cases

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 3
for cs in medium_cases(ncases, seed):
    kind, x, y, n, p, icpt, stdz = (cs[k] for k in ("kind", "x", "y", "n", "p", "icpt", "stdz"))
    opts = dict(maxit=cs["maxit"], eps_abs=cs["eps"], eps_rel=cs["eps"], rho=cs["rho"])
    lmr = 0.01 if n < p else 1e-4
    lam = None
    if cs["user_lam"]:
        ref0 = entry.admm_lasso(x, y, None, 3, 0.1, stdz, icpt, dict(entry.LASSO_OPTS, maxit=1), {})
        lam = np.sort(ref0["lambda"][0] * cs["ulam"])[::-1]
    rho = None if cs["rho"] <= 0 else cs["rho"]
    t0 = time.time()
    if kind == "enet_tall":
        m = admm_enet(x, y, icpt, stdz).penalty(lam, nlambda=cs["nl"], alpha=cs["alpha"]).opts(cs["maxit"], cs["eps"], cs["eps"], rho)
        fit = m.fit(); bg, ng, st = fit.beta_dense, fit.niter, fit.stats
        t1 = time.time()
        ref = entry.admm_enet(x, y, lam, cs["nl"], lmr, stdz, icpt, cs["alpha"], opts)
    elif kind == "par":
        m = admm_lasso(x, y, icpt, stdz).penalty(lam, nlambda=cs["nl"]).opts(cs["maxit"], cs["eps"], cs["eps"], rho)
        m.nthread = cs["K"]
        lib, head, tail, lam_out, bg, ng, stats, keep = m._common()
        check(lib.admm_hip_parlasso(*head, cs["K"], *tail)); st = stats.as_dict()
        t1 = time.time()
        ref = entry.admm_parlasso(x, y, lam, cs["nl"], lmr, stdz, icpt, cs["K"], opts)
    else:
        m = admm_lasso(x, y, icpt, stdz).penalty(lam, nlambda=cs["nl"]).opts(cs["maxit"], cs["eps"], cs["eps"], rho)
        fit = m.fit(); bg, ng, st = fit.beta_dense, fit.niter, fit.stats
        t1 = time.time()
        ref = entry.admm_lasso(x, y, lam, cs["nl"], lmr, stdz, icpt, opts)
    t2 = time.time()
    floor = 1e-3 * float(np.abs(ref["beta"]).max())
    errs = [float(np.abs(bg[:, j].astype(np.float64) - ref["beta"][:, j]).max()) / max(float(np.abs(ref["beta"][:, j]).max()), floor, 1e-300)
            for j in range(ref["beta"].shape[1])]
    dn = np.abs(np.asarray(ng, int) - np.asarray(ref["niter"], int))
    first_flip = int(np.argmax(dn > 2)) if (dn > 2).any() else len(dn)
    e_before = max(errs[:first_flip]) if first_flip > 0 else 0.0
    flag = "SUSPECT" if (e_before > 2e-4 or not np.all(np.isfinite(bg))) else ""
    print(f"{cs['c']:3d} {kind:9s} n={n:5d} p={p:5d} icpt={int(icpt)} std={int(stdz)} scale={cs['scale']:<4g} nl={cs['nl']:2d} K={cs['K']} maxit={cs['maxit']:5d} eps={cs['eps']:g} rho={cs['rho']:g} "
          f"| max relerr {max(errs):.2e} (before first count flip {e_before:.2e}) max dniter {int(dn.max())} of {int(np.max(ref['niter']))} | gpu {t1 - t0:.2f}s oracle {t2 - t1:.1f}s {flag}", flush=True)
    if flag:
        print("     niter gpu", np.asarray(ng), "\n     niter ref", ref["niter"], "\n     errs", np.array(errs))
