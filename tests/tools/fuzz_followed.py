"""Dev tool (GPU box): the strict trace-based parity rule of tests/test_gpu_fuzz.py on another draw of random small
problems.   python tests/tools/fuzz_followed.py [ncases] [seed] [case,case,...|medium]"""
import os
import sys
import traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402  (one HIP runtime per process)
from fuzz_cases import cases  # noqa: E402
import test_gpu_fuzz as T  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 11
medium = len(sys.argv) > 3 and sys.argv[3] == "medium"
only = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 and not medium else None
bad = 0
if medium:
    from fuzz_cases import medium_cases
    for cs in medium_cases(n, seed):
        if cs["kind"] not in ("tall", "enet_tall"):
            continue
        try:
            T._medium_tall(cs)
        except Exception as e:                              # noqa: BLE001
            bad += 1
            print("FAIL medium case", cs["c"], cs["kind"], "n=%d p=%d" % (cs["n"], cs["p"]), type(e).__name__, str(e)[:300], flush=True)
    print("medium cases", n, "seed", seed, "failures", bad)
    sys.exit(0)
for cs in cases(n, seed):
    if only is not None and cs["c"] not in only:
        continue
    try:
        if cs["kind"] in ("lad", "bp"):
            T._run_dense_case(cs)
        else:
            T._run_lasso_case(cs)
    except Exception as e:                                  # noqa: BLE001
        bad += 1
        print("FAIL case", cs["c"], cs["kind"], "n=%d p=%d" % (cs["n"], cs["p"]), type(e).__name__, str(e)[:300], flush=True)
print("cases", n, "seed", seed, "failures", bad)
