"""Dev tool (GPU box): the final soak reported, for case 546:23 only, that record 0 of the iterate dump (X'y) differed from the
oracle's beyond summation rounding, while a standalone capture of the same case is clean.  Replays the worker's sequence
(cases 0..c of the seed, then the failing case captured with the dump `reps` times) and prints record 0 against the oracle's
X'y each time.

   python tests/tools/soak_xy_repro.py <seed> <case> [reps]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: F401,E402
from fuzz_cases import cases  # noqa: E402
import test_gpu_fuzz as T  # noqa: E402
from oracle.datastd import DataStd  # noqa: E402

seed, c = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
F = np.float32
target = None
for cs in cases(c + 1, seed):
    cap = T.gpu_capture(cs)
    if cs["c"] == c:
        target = cs
prob = T._lasso_problem(target)
n, p = prob["x"].shape
X = np.array(prob["x"], dtype=F, order="F"); Y = np.array(prob["y"], dtype=F)
std = DataStd(n, p, prob["standardize"], prob["intercept"], F); std.standardize(X, Y)
xy = (X.T @ Y).astype(F)
tol = 64 * np.spacing(np.abs(xy).max())
first = None
for r in range(reps):
    cap = T.gpu_capture(target, state=True)
    S = np.asarray(cap["state"], dtype=F).reshape(len(cap["state"]), 5, p)
    g = S[0, 0]
    if first is None:
        first = g.copy()
    print(f"rep {r}: |X'y_gpu - X'y_oracle| max {np.abs(g - xy).max():.3e} (bound {tol:.3e}), vs first rep {np.abs(g - first).max():.3e}, records {len(S)}, "
          f"niter {cap['niter'].tolist()}", flush=True)
    try:
        T.stepwise_capture(target, cap)
    except AssertionError as e:
        print("   stepwise:", str(e)[:300], flush=True)
        np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"xy_repro_{seed}_{c}_{r}.npz"), **cap)
