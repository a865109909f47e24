"""Dev tool (GPU box): randomised parity sweep of the five entry points against the oracle on small problems.
Prints one line per case; a case is SUSPECT when beta differs by > 1e-3 although the iteration counts agree
(a count flip explains a larger difference: the stopping rule is loose).   python tests/tools/fuzz_parity.py [ncases] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401  (one HIP runtime per process)
import numpy as np
from admm_amd import admm_lasso, admm_enet, admm_lad, admm_bp, AdmmHipError
from oracle import entry
from fuzz_cases import cases


def relerr(a, b):
    return np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(np.abs(b).max(), 1e-300)


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    only = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else None     # verbose re-run of these cases
    suspects = 0
    for cs in cases(ncases, seed):
        c, kind, icpt, stdz, scale, n, p, x, y = (cs[k] for k in ("c", "kind", "icpt", "stdz", "scale", "n", "p", "x", "y"))
        tag = f"{c:3d} {kind:9s} n={n:4d} p={p:4d} icpt={int(icpt)} std={int(stdz)} scale={scale:<5g}"
        skip = only is not None and c not in only
        try:
            if kind in ("tall", "wide", "enet_tall", "enet_wide", "par"):
                user_lam, nl, alpha, ulam, Kdraw = cs["user_lam"], cs["nl"], cs["alpha"], cs["ulam"], cs["K"]
                if skip:
                    continue
                opts = dict(entry.LASSO_OPTS)
                if kind == "par":
                    opts["maxit"] = 500
                lam = None
                lmr = 0.01 if n < p else 1e-4
                if user_lam:
                    d = {}
                    ref0 = entry.admm_lasso(x, y, None, 3, 0.1, stdz, icpt, dict(opts, maxit=1), d)
                    lam = np.sort(ref0["lambda"][0] * ulam)[::-1]
                if kind.startswith("enet"):
                    mdl = admm_enet(x, y, icpt, stdz).penalty(lam, nlambda=nl, alpha=alpha)
                    mdl.opts(maxit=opts["maxit"]); fit = mdl.fit()
                    ref = entry.admm_enet(x, y, lam, nl, lmr, stdz, icpt, alpha, opts)
                elif kind == "par":
                    K = Kdraw
                    mdl = admm_lasso(x, y, icpt, stdz).penalty(lam, nlambda=nl).opts(maxit=opts["maxit"])
                    mdl.nthread = K
                    lib, head, tail, lam_out, beta, niter, stats, keep = mdl._common()
                    from admm_amd._lib import check
                    check(lib.admm_hip_parlasso(*head, K, *tail))
                    class F: pass
                    fit = F(); fit.beta_dense = beta; fit.niter = niter; fit.stats = stats.as_dict()
                    ref = entry.admm_parlasso(x, y, lam, nl, lmr, stdz, icpt, K, opts)
                    tag += f" K={K}"
                else:
                    fit = admm_lasso(x, y, icpt, stdz).penalty(lam, nlambda=nl).fit()
                    ref = entry.admm_lasso(x, y, lam, nl, lmr, stdz, icpt, opts)
                # a (near-)null column is compared on the scale of the path, and a path that is null altogether (one
                # automatic lambda = lambda_max) on the natural coefficient scale sd(y) / sd(x)
                floor = 1e-3 * max(float(np.abs(ref["beta"]).max()), float(np.std(y) / max(np.std(x), 1e-300)))
                e = max(np.abs(fit.beta_dense[:, j].astype(np.float64) - ref["beta"][:, j]).max() / max(float(np.abs(ref["beta"][:, j]).max()), float(floor), 1e-300)
                        for j in range(ref["beta"].shape[1]))
                dn = int(np.abs(np.asarray(fit.niter, int) - np.asarray(ref["niter"], int)).max())
                nit = int(np.max(ref["niter"]))
                if only is not None:
                    print("   lambda", ref["lambda"], "\n   niter gpu", np.asarray(fit.niter), "ref", ref["niter"], "\n   rho gpu", fit.stats["rho"], "eig", fit.stats["eig_est"], "alpha", alpha, "user_lam", user_lam)
                    print("   nan gpu/ref", np.isnan(fit.beta_dense).sum(), np.isnan(ref["beta"]).sum())
                    for j in range(ref["beta"].shape[1]):
                        print("   col", j, "relerr", relerr(fit.beta_dense[:, j], ref["beta"][:, j]), "max|ref|", np.abs(ref["beta"][:, j]).max(), "nnz gpu/ref", np.count_nonzero(fit.beta_dense[1:, j]), np.count_nonzero(ref["beta"][1:, j]))
            elif skip:
                continue
            elif kind == "lad":
                fit = admm_lad(x, y, icpt).fit()
                ref = entry.admm_lad(x, y, icpt, entry.LAD_OPTS)
                e = relerr(fit.beta, ref["beta"]); dn = abs(int(fit.niter) - int(ref["niter"])); nit = int(ref["niter"])
            else:
                fit = admm_bp(x, y).fit()
                ref = entry.admm_bp(x, y, entry.BP_OPTS)
                e = relerr(fit.beta.toarray().ravel(), ref["beta"]); dn = abs(int(fit.niter) - int(ref["niter"])); nit = int(ref["niter"])
            if only is not None and kind in ("lad", "bp"):
                bg = fit.beta.toarray().ravel() if kind == "bp" else np.asarray(fit.beta)
                print("   niter gpu", fit.niter, "ref", ref["niter"], " nan gpu/ref", np.isnan(bg).sum(), np.isnan(ref["beta"]).sum(),
                      " max|b| gpu/ref", np.abs(bg).max(), np.abs(ref["beta"]).max(), " |Xb-y| gpu/ref",
                      np.abs(x @ (bg[1:] if kind == "lad" else bg) + (bg[0] if kind == "lad" else 0) - y).max(),
                      np.abs(x @ (ref["beta"][1:] if kind == "lad" else ref["beta"]) + (ref["beta"][0] if kind == "lad" else 0) - y).max(),
                      " l1 gpu/ref", np.abs(bg).sum(), np.abs(ref["beta"]).sum())
            bad = (e > 1e-3 and dn <= 2) or not np.isfinite(e)
            suspects += bad
            print(f"{tag}  relerr={e:.2e} dniter={dn} (of {nit}) {'SUSPECT' if bad else ''}", flush=True)
        except AdmmHipError as ex:
            print(f"{tag}  library error {ex.code}: {ex}", flush=True)
        except Exception as ex:
            suspects += 1
            print(f"{tag}  EXCEPTION {type(ex).__name__}: {ex} SUSPECT", flush=True)
    print("suspects:", suspects)


if __name__ == "__main__":
    main()
