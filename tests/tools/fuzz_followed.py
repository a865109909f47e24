"""Dev tool (GPU box): the strict trace-based parity rule of tests/test_gpu_fuzz.py on another draw of random small
problems.   python tests/tools/fuzz_followed.py [ncases] [seed] [case,case,...|medium]"""
import os
import sys
import traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402  (one HIP runtime per process)
from fuzz_cases import cases  # noqa: E402
import test_gpu_fuzz as T  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 11
medium = len(sys.argv) > 3 and sys.argv[3] == "medium"
only = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 and not medium else None
bad = 0
if medium:
    from fuzz_cases import medium_cases
    import numpy as np
    from admm_amd import admm_lasso
    from oracle import entry
    from helpers import assert_followed_parity, traced_fit

    def medium_other(cs):                                   # wide / consensus with random maxit / eps / rho
        x, y, icpt, stdz = cs["x"], cs["y"], cs["icpt"], cs["stdz"]
        opts = dict(maxit=cs["maxit"], eps_abs=cs["eps"], eps_rel=cs["eps"], rho=cs["rho"])
        lam = None
        if cs["user_lam"]:
            ref0 = entry.admm_lasso(x, y, None, 3, 0.1, stdz, icpt, dict(entry.LASSO_OPTS, maxit=1), {})
            lam = np.sort(ref0["lambda"][0] * cs["ulam"])[::-1]
        lmr = 0.01 if cs["n"] < cs["p"] else 1e-4
        m = admm_lasso(x, y, icpt, stdz).penalty(lam, nlambda=cs["nl"], lambda_min_ratio=lmr)
        m.opts(cs["maxit"], cs["eps"], cs["eps"], None if cs["rho"] <= 0 else cs["rho"])
        prob = dict(x=x, y=y, lam=lam, nlambda=cs["nl"], lmin_ratio=lmr, standardize=stdz, intercept=icpt, opts=opts, alpha=None)
        if cs["kind"] == "par":
            m.nthread = cs["K"]
            prob["nthread"] = cs["K"]
        fit, trace = traced_fit(m, capacity=cs["nl"] * (cs["maxit"] + 2) + 8)
        label = f"medium {cs['c']} {cs['kind']} n={cs['n']} p={cs['p']} K={cs['K']} maxit={cs['maxit']} eps={cs['eps']:g} rho={cs['rho']:g}"
        return assert_followed_parity(fit.beta_dense, fit.niter, trace, prob, 1e-4, label=label)

    for cs in medium_cases(n, seed):
        try:
            if cs["kind"] in ("tall", "enet_tall"):
                if len(sys.argv) > 4 and sys.argv[4] == "other":
                    continue
                T._medium_tall(cs)
            else:
                if not (len(sys.argv) > 4 and sys.argv[4] == "other"):
                    continue
                if cs["kind"] == "wide" and cs["maxit"] > 300:
                    cs["maxit"] = 300                       # the NumPy oracle's wide loop is slow
                medium_other(cs)
        except Exception as e:                              # noqa: BLE001
            bad += 1
            print("FAIL medium case", cs["c"], cs["kind"], "n=%d p=%d" % (cs["n"], cs["p"]), type(e).__name__, str(e)[:300], flush=True)
    print("medium cases", n, "seed", seed, "failures", bad)
    sys.exit(0)
for cs in cases(n, seed):
    if only is not None and cs["c"] not in only:
        continue
    try:
        if cs["kind"] in ("lad", "bp"):
            T._run_dense_case(cs)
        else:
            T._run_lasso_case(cs)
    except Exception as e:                                  # noqa: BLE001
        bad += 1
        print("FAIL case", cs["c"], cs["kind"], "n=%d p=%d" % (cs["n"], cs["p"]), type(e).__name__, str(e)[:300], flush=True)
print("cases", n, "seed", seed, "failures", bad)
