"""Dev tool (GPU box): capture the iterate dump (admm_hip_lasso_plan_state_*) together with beta / niter / trace for named
cases of tests/fuzz_cases.py, for the stepwise check of oracle/stepcheck.py on a CPU.

   python tests/tools/soak_state.py <out_dir> <seed>:<case> [<seed>:<case> ...]     -> <out_dir>/state_s<seed>_c<case>.npz
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: F401,E402  (one HIP runtime per process)
from fuzz_cases import cases  # noqa: E402
import test_gpu_fuzz as T  # noqa: E402

out = sys.argv[1]
os.makedirs(out, exist_ok=True)
for spec in sys.argv[2:]:
    seed, c = (int(v) for v in spec.split(":"))
    cs = next(k for k in cases(c + 1, seed) if k["c"] == c)
    cap = T.gpu_capture(cs, state=True)
    path = os.path.join(out, f"state_s{seed}_c{c}.npz")
    np.savez_compressed(path, **cap)
    line = f"{spec} {T.case_label(cs)} records {len(cap['trace'])} state {None if 'state' not in cap else cap['state'].shape} {os.path.getsize(path)} bytes"
    try:
        rep = T.stepwise_capture(cs, cap)
        line += " | stepwise: x_ratio_max %.2f x_vs_ref_max %.2f bit_mismatch %d accum_ties %d norm_rel %.1e" % (
            rep["x_ratio_max"], rep.get("x_vs_ref_max", 0.0), len(rep["bit_mismatch"]), len(rep["accum_ties"]), rep["norm_rel_max"])
        if rep["bit_mismatch"]:
            line += " first " + str(rep["bit_mismatch"][:4])
    except Exception as e:  # noqa: BLE001
        line += " | stepwise FAILED: " + str(e)[:400]
    print(line, flush=True)
