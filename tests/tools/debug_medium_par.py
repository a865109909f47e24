"""Dev tool (GPU box): one consensus case of tests/fuzz_cases.py medium_cases under the variants of the consensus solver.
   python tests/tools/debug_medium_par.py <seed> <case>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import numpy as np
import admm_amd
from admm_amd import admm_lasso
from admm_amd._lib import check
from oracle import entry
from fuzz_cases import medium_cases

seed, case = int(sys.argv[1]), int(sys.argv[2])
cs = [c for c in medium_cases(case + 1, seed)][case]
print({k: v for k, v in cs.items() if k not in ("x", "y")})
x, y, n, p, icpt, stdz = (cs[k] for k in ("x", "y", "n", "p", "icpt", "stdz"))
opts = dict(maxit=cs["maxit"], eps_abs=cs["eps"], eps_rel=cs["eps"], rho=cs["rho"])
lam = None
if cs["user_lam"]:
    ref0 = entry.admm_lasso(x, y, None, 3, 0.1, stdz, icpt, dict(entry.LASSO_OPTS, maxit=1), {})
    lam = np.sort(ref0["lambda"][0] * cs["ulam"])[::-1]
rho = None if cs["rho"] <= 0 else cs["rho"]
lmr = 0.01 if n < p else 1e-4
ref = entry.admm_parlasso(x, y, lam, cs["nl"], lmr, stdz, icpt, cs["K"], opts)
floor = 1e-3 * float(np.abs(ref["beta"]).max())
for name, o in (("default", {}), ("two-pass", dict(PAR_ONEPASS="0")), ("unfused pack/z", dict(PAR_FUSE_PZ="0")), ("unbatched", dict(PAR_BATCH="0")), ("guard always", dict(PAR_ONEPASS_TAU="1e30")), ("guard never", dict(PAR_ONEPASS_TAU="0"))):
    with admm_amd.options(PAR_ONEPASS_STATS="1", **o):
        m = admm_lasso(x, y, icpt, stdz).penalty(lam, nlambda=cs["nl"]).opts(cs["maxit"], cs["eps"], cs["eps"], rho)
        m.nthread = cs["K"]
        lib, head, tail, lam_out, bg, ng, stats, keep = m._common()
        check(lib.admm_hip_parlasso(*head, cs["K"], *tail))
    errs = [float(np.abs(bg[:, j].astype(np.float64) - ref["beta"][:, j]).max()) / max(float(np.abs(ref["beta"][:, j]).max()), floor, 1e-300) for j in range(ref["beta"].shape[1])]
    print(f"{name:16s} niter {np.asarray(ng).tolist()} ref {ref['niter'].tolist()} errs {[f'{e:.1e}' for e in errs]} |beta|max gpu {np.abs(bg).max():.3e} ref {np.abs(ref['beta']).max():.3e}", flush=True)
