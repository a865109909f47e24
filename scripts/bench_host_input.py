#!/usr/bin/env python
"""C2 with HOST input (what R hands over): whole-fit wall time including the PCIe crossing of the 8 GB fp64 x."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from admm_amd import admm_lasso
n, p = int(sys.argv[1]) if len(sys.argv) > 1 else 100000, int(sys.argv[2]) if len(sys.argv) > 2 else 10000
rng = np.random.default_rng(123)
t0 = time.time()
x = np.empty((n, p), order="F")
for j0 in range(0, p, 500):
    x[:, j0:j0 + 500] = rng.standard_normal((n, min(500, p - j0))) * 2
b = np.zeros(p); b[:p // 10] = rng.uniform(size=p // 10)
y = x @ b + rng.standard_normal(n)
tg = time.time() - t0
for rep in range(2):
    t0 = time.time()
    fit = admm_lasso(x, y).penalty(nlambda=100).fit()
    t = time.time() - t0
    st = fit.stats
    print(json.dumps({"rep": rep, "n": n, "p": p, "fit_wall_s": t, "t_h2d": st["t_h2d"], "h2d_GBps": 8.0 * n * p / st["t_h2d"] / 1e9,
                      "t_standardize": st["t_standardize"], "t_gram": st["t_gram"], "t_factor": st["t_factor"], "t_loop": st["t_loop"],
                      "iters": int(st["total_iter"]), "iters_per_s_pcie_inclusive": int(st["total_iter"]) / t, "datagen_s": tg}), flush=True)
