"""Fixture for the README's second LAD performance case (/root/reference/README.md:336-364: set.seed(123); n = 5000; p = 1000;
b = runif(p); x = rnorm(n p, sd = 2); y = x b + rnorm(n); `range(rq.fit(x, y, method = "fn")$coefficients - admm_lad(...)$beta[-1])`
= -0.003577610 0.004135838).

quantreg is not in this image; what `rq.fit` approximates is the optimum of the LAD linear programme
    min sum(u + v)  s.t.  X beta + u - v = y,  u, v >= 0,
solved here exactly with SciPy's HiGHS (9 minutes for this size: hence a fixture).  The inputs are regenerated from the R
snippet with oracle/rrng.py (data, not source); the output is the LP's beta (1000 doubles) and its objective.

    python tests/golden/make_readme_perf.py        ->  tests/golden/readme_lad_n5000_lp.npz
"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
from scipy.optimize import linprog

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))


def lad_lp(x, y):
    n, p = x.shape
    A = sp.hstack([sp.csr_matrix(x), sp.identity(n), -sp.identity(n)]).tocsc()
    c = np.concatenate([np.zeros(p), np.ones(2 * n)])
    res = linprog(c, A_eq=A, b_eq=y, bounds=[(None, None)] * p + [(0, None)] * (2 * n), method="highs")
    assert res.status == 0, res.message
    return res.x[:p], float(res.fun)


if __name__ == "__main__":
    from readme_perf_cases import lad_data
    x, y = lad_data(5000, 1000)
    t0 = time.time()
    beta, obj = lad_lp(x, y)
    print(f"LP solved in {time.time() - t0:.0f} s, objective {obj:.10g}")
    np.savez_compressed(os.path.join(HERE, "readme_lad_n5000_lp.npz"), beta=beta, objective=obj)
