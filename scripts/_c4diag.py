import os, sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "tests/golden")
import torch
from admm_amd import admm_lasso
from helpers import traced_fit, col_err
from make_fullsize import lasso_data
g = np.load("tests/golden/c4_fixed_maxit.npz")
x, y = lasso_data(int(g["seed"]), int(g["n"]), int(g["p"]), int(g["m"]))
maxit, K = int(g["maxit"]), int(g["K"])
fit, trace = traced_fit(admm_lasso(x, y).penalty(g["lam"]).parallel(K).opts(maxit=maxit), capacity=2 * (maxit + 2) + 8)
t = np.asarray(trace); t = t[1:] if t[0, 8] == -1 else t
o = g["trace"]
for c, nm in ((2, "eps_p"), (3, "eps_d"), (4, "r_p"), (5, "r_d")):
    den = np.maximum(np.abs(o[:, c]), o[:, 2 if c in (2, 4) else 3])
    d = np.abs(t[:, c] - o[:, c]) / den
    k = int(np.argmax(d))
    print(os.environ.get("ADMM_HIP_PAR_ONEPASS", "1"), nm, "max dev %.3e at record %d (lam %d iter %d): gpu %.9e oracle %.9e thr %.3e" % (d[k], k, o[k, 0], o[k, 1], t[k, c], o[k, c], den[k]),
          "| median %.2e" % np.median(d))
floor = 1e-2 * float(np.abs(g["beta"]).max())
print("col errs", [col_err(fit.beta_dense[:, j], g["beta"][:, j], floor) for j in range(2)], "nnz", [int(np.count_nonzero(fit.beta_dense[1:, j])) for j in range(2)])
