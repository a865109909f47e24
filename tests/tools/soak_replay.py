"""Dev tool (CPU only): judge a capture of tests/tools/soak_capture.py again -- the GPU's outputs for one case of
tests/fuzz_cases.py (beta, niter, decision trace) are in the .npz, the inputs are regenerated from (seed, case).

   python tests/tools/soak_replay.py <fail_s<seed>_c<case>.npz> [band]        -> the rule's verdict, the near-ties it took
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402


def load(path):
    from fuzz_cases import cases
    m = re.search(r"s(\d+)_c(\d+)\.npz$", path)
    seed, c = int(m.group(1)), int(m.group(2))
    cs = next(k for k in cases(c + 1, seed) if k["c"] == c)
    cap = dict(np.load(path)) if os.path.exists(path) else None
    return cs, cap


def main():
    import test_gpu_fuzz as T
    cs, cap = load(sys.argv[1])
    band = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
    print(T.case_label(cs), "K", cs.get("K"), "nl", cs.get("nl"), "user_lam", cs.get("user_lam"), "alpha", cs.get("alpha"),
          "| niter", cap["niter"].tolist(), "decisions", len(cap["trace"]))
    try:
        rep = T.judge_capture(cs, cap, band=band, budget=False)
        f = rep["forced"]
        print("PASS with band", band, "forced", len(f), [(q["lam"], q["iter"], q["kind"], round(q["ulps"], 2)) for q in f][:20])
    except AssertionError as e:
        print("FAIL with band", band, type(e).__name__, str(e)[:800])


if __name__ == "__main__":
    main()
