"""Table for DESIGN.md: flops per rank and time of the distributed factorisation against the replicated
one, N rank processes on ONE GPU (this pool's boxes have one): the ranks share the device, so the wall time shows the TOTAL work
(replicated: N x p^3; distributed: ~1.1 p^3 + the panel broadcasts), not a multi-GPU speed-up.
Usage: python scripts/dist_factor_table.py [case ...]   (cases of tests/dist_worker.py; default tallshard2300 tallshard6000)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
from test_gpu_dist2 import _run_ranks  # noqa: E402
from dist_worker import problem  # noqa: E402

cases = sys.argv[1:] or ["tallshard2300", "tallshard6000"]
print("| case | p | ranks | back-end | factorisation | flops per rank / p^3 | t_factor per rank (s) | identical result |")
print("|---|---|---|---|---|---|---|---|")
for case in cases:
    p = problem(case)[0].shape[1]
    for nranks, backend in ((2, "peer"), (4, "peer")):
        env = {"ADMM_HIP_INVERSE": "f32"}
        a = _run_ranks(backend, case, nranks=nranks, timeout=900, extra_env=env)
        b = _run_ranks(backend, case, nranks=nranks, timeout=900, extra_env=dict(env, ADMM_HIP_DIST_FACTOR="0"))
        same = all((a[r]["beta"] == b[r]["beta"]).all() and (a[r]["trace"] == b[r]["trace"]).all() for r in range(nranks))
        for name, res in (("distributed", a), ("replicated", b)):
            fl = [float(res[r]["factor_flops"]) / p ** 3 for r in range(nranks)]
            tf = [float(res[r]["t_factor"]) for r in range(nranks)]
            print(f"| {case} | {p} | {nranks} | {backend} | {name} | {', '.join(f'{v:.3f}' for v in fl)} | {', '.join(f'{v:.3f}' for v in tf)} | {same} |", flush=True)
