import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
import torch, numpy as np
import fuzz_parity as fz
from admm_amd import admm_lasso, admm_enet
want = [int(v) for v in sys.argv[1].split(",")]
css = [cs for cs in fz.cases(80, 7) if cs["c"] in want]
junk = []
for rep in range(4):
    for cs in css:
        m = admm_lasso(cs["x"], cs["y"], cs["icpt"], cs["stdz"]).penalty(None, nlambda=cs["nl"])
        fit = m.fit()
        print(rep, cs["c"], "niter", list(map(int, fit.niter)), "rho", fit.stats["rho"], "b[:3,-1]", fit.beta_dense[:3, -1], flush=True)
    # perturb the allocator state between repetitions
    junk.append(torch.full((1 << (18 + rep),), float("nan"), device="cuda"))
    if rep == 2:
        junk.clear(); torch.cuda.empty_cache()
