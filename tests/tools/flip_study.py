"""Dev tool (GPU box): decision-trace comparison of the tall path against the oracle.

    python tests/tools/flip_study.py [case ...]      cases: std11 std10 std01 std00 enet smoke m3 m8 (default: all)

For every case: run the prepared problem with the decision trace enabled, run the oracle with its trace, walk both
traces in lock step and report the first decision where they differ together with how close the deciding quantity
was to its threshold on both sides.  ADMM_HIP_INVERSE / ADMM_HIP_XUPDATE select the x-update variant."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import numpy as np
from admm_amd import admm_lasso, admm_enet
from admm_amd.api import LassoPlan
from oracle import entry
from helpers import synth_lasso, compare_traces, relerr, traced_fit, assert_tall_parity
from fuzz_cases import medium_cases


def build(case):
    if case.startswith("std"):
        stdz, icpt = case[3] == "1", case[4] == "1"
        x, y = synth_lasso(2000, 300, 30, seed=7); x += 0.7
        return dict(x=x, y=y, stdz=stdz, icpt=icpt, nl=20, lam=None, alpha=None, opts=entry.LASSO_OPTS)
    if case == "enet":
        x, y = synth_lasso(1500, 200, 20, seed=11)
        return dict(x=x, y=y, stdz=True, icpt=True, nl=15, lam=None, alpha=0.6, opts=entry.LASSO_OPTS)
    if case == "smoke":
        rng = np.random.default_rng(0)
        X = rng.standard_normal((3000, 400)) * 2
        beta = np.concatenate([rng.uniform(size=40), np.zeros(360)])
        return dict(x=X, y=X @ beta + rng.standard_normal(3000), stdz=True, icpt=True, nl=10, lam=None, alpha=None, opts=entry.LASSO_OPTS)
    if case.startswith("m"):
        want = int(case[1:])
        for cs in medium_cases(want + 1, 3):
            if cs["c"] == want:
                break
        opts = dict(maxit=cs["maxit"], eps_abs=cs["eps"], eps_rel=cs["eps"], rho=cs["rho"])
        lam = None
        if cs["user_lam"]:
            ref0 = entry.admm_lasso(cs["x"], cs["y"], None, 3, 0.1, cs["stdz"], cs["icpt"], dict(entry.LASSO_OPTS, maxit=1), {})
            lam = np.sort(ref0["lambda"][0] * cs["ulam"])[::-1]
        print(f"  medium case {want}: {cs['kind']} n={cs['n']} p={cs['p']} maxit={cs['maxit']} eps={cs['eps']} rho={cs['rho']} scale={cs['scale']}")
        return dict(x=cs["x"], y=cs["y"], stdz=cs["stdz"], icpt=cs["icpt"], nl=cs["nl"], lam=lam, alpha=cs["alpha"], opts=opts)
    raise SystemExit("unknown case " + case)


names = sys.argv[1:] or ["std11", "std10", "std01", "std00", "enet", "smoke", "m3", "m8", "m9"]
for name in names:
    c = build(name)
    o = c["opts"]
    rho = None if o["rho"] <= 0 else o["rho"]
    if c["alpha"] is None:
        m = admm_lasso(c["x"], c["y"], c["icpt"], c["stdz"]).penalty(c["lam"], nlambda=c["nl"]).opts(o["maxit"], o["eps_abs"], o["eps_rel"], rho)
    else:
        m = admm_enet(c["x"], c["y"], c["icpt"], c["stdz"]).penalty(c["lam"], nlambda=c["nl"], alpha=c["alpha"]).opts(o["maxit"], o["eps_abs"], o["eps_rel"], rho)
    fit, tg = traced_fit(m, capacity=c["nl"] * (o["maxit"] + 2) + 8)
    problem = dict(x=c["x"], y=c["y"], lam=c["lam"], nlambda=c["nl"], lmin_ratio=1e-4, standardize=c["stdz"], intercept=c["icpt"],
                   opts=o, alpha=c["alpha"])
    d = {"trace": []}
    if c["alpha"] is None:
        ref = entry.admm_lasso(c["x"], c["y"], c["lam"], c["nl"], 1e-4, c["stdz"], c["icpt"], o, d)
    else:
        ref = entry.admm_enet(c["x"], c["y"], c["lam"], c["nl"], 1e-4, c["stdz"], c["icpt"], c["alpha"], o, d)
    r = compare_traces(tg, np.array(d["trace"], dtype=np.float64))
    print(f"{name:6s} niter gpu {list(map(int, fit.niter))}\n       niter ref {list(map(int, ref['niter']))}  (unfollowed oracle)")
    print(f"       unfollowed: {r['summary']}")
    try:
        assert_tall_parity(fit.beta_dense, fit.niter, tg, problem, label=name)
    except AssertionError as e:
        print("       FOLLOW-MODE PARITY FAILED:", str(e)[:600])
    sys.stdout.flush()
