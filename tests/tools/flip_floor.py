"""Dev tool (CPU only): how often does the tall path's iteration count change when ONLY the rounding of the x-update
changes?  Runs the NumPy oracle with four mathematically identical x-updates on the tall parity problems:
   llt32    float Cholesky + two float triangular solves          (the oracle / the reference: ADMMLassoTall.h:70-80)
   inv32    float inverse (from the float factor) x float mat-vec (this build with ADMM_HIP_INVERSE=f32)
   inv64r   inverse formed in double, rounded to float once, float mat-vec   (this build's default)
   exact    double Cholesky solve of the float system, result rounded to float (what all three approximate)
and prints, for each pair, the number of lambdas whose iteration count differs and the first such lambda.
If llt32 flips against `exact` as often as inv32 / inv64r do, the flips are the reference algorithm's own sensitivity to
float rounding at eps = 1e-5, not a defect of the cached-inverse x-update."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import entry
from oracle.variants import MODES, tall_variant
from helpers import synth_lasso


def run(mode, x, y, nl, stdz, icpt, alpha):
    with tall_variant(mode):
        if alpha is None:
            return entry.admm_lasso(x, y, None, nl, 1e-4, stdz, icpt, entry.LASSO_OPTS)
        return entry.admm_enet(x, y, None, nl, 1e-4, stdz, icpt, alpha, entry.LASSO_OPTS)


cases = []
x, y = synth_lasso(2000, 300, 30, seed=7); x += 0.7
for s, i in ((True, True), (True, False), (False, True), (False, False)):
    cases.append((f"std{int(s)}{int(i)}", x, y, 20, s, i, None))
x2, y2 = synth_lasso(1500, 200, 20, seed=11)
cases.append(("enet", x2, y2, 15, True, True, 0.6))
rng = np.random.default_rng(0)
X = rng.standard_normal((3000, 400)) * 2
beta = np.concatenate([rng.uniform(size=40), np.zeros(360)])
cases.append(("smoke", X, X @ beta + rng.standard_normal(3000), 10, True, True, None))
for seed in range(3):
    xs, ys = synth_lasso(1200, 150, 15, seed=100 + seed)
    cases.append((f"rnd{seed}", xs, ys, 20, True, True, None))

modes = list(MODES)
tot = {}
for name, x, y, nl, s, i, alpha in cases:
    res = {m: run(m, x, y, nl, s, i, alpha) for m in modes}
    line = f"{name:6s}"
    for a in range(len(modes)):
        for b in range(a + 1, len(modes)):
            na, nb = res[modes[a]]["niter"].astype(int), res[modes[b]]["niter"].astype(int)
            d = np.nonzero(na != nb)[0]
            tot[(a, b)] = tot.get((a, b), 0) + len(d)
            line += f" | {modes[a]}~{modes[b]}: {len(d):2d}" + (f" (first {d[0]:2d})" if len(d) else "           ")
    print(line, flush=True)
print("total lambdas with different counts:", {f"{modes[a]}~{modes[b]}": v for (a, b), v in tot.items()})
