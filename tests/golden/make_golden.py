"""Writes tests/golden/readme_vectors.json: the inputs of the reference README's examples (regenerated with the
restatement of R's RNG, oracle/rrng.py) and the numbers the README prints for them.  Data only.
    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import readme  # noqa: E402


def main():
    x, y = readme.lasso_data()
    xb, yb, bt = readme.bp_data()
    out = {
        "source": "yixuan/ADMM README.md: Lasso/Enet/LAD fixture lines 47-53, BP fixture lines 166-174; printed columns as cited in oracle/readme.py",
        "lambda": readme.LAMBDA,
        "lasso": {"n": 100, "p": 20, "x_colmajor": x.flatten(order="F").tolist(), "y": y.tolist(),
                  "glmnet": readme.LASSO_GLMNET.tolist(), "admm": readme.LASSO_ADMM.tolist(), "paradmm": readme.LASSO_PARADMM.tolist()},
        "enet": {"alpha": 0.5, "admm": readme.ENET_ADMM.tolist()},
        "lad": {"admm": readme.LAD_ADMM.tolist()},
        "bp": {"n": 50, "p": 100, "x_colmajor": xb.flatten(order="F").tolist(), "y": yb.tolist(), "beta_true": bt.tolist(),
               "error_range": list(readme.BP_RANGE), "perf_error_range": list(readme.BP_PERF_RANGE)},
    }
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "readme_vectors.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
