"""Dev tool (CPU only), VERDICT r4 task 8: is the tall family's decision noise -- the only family whose near-ties leave the 8-ulp
band of the follow rule (9-10 near-ties per 1000 decisions, all but one failure of the round-4 soaks) -- caused by THIS build's cached
inverse, or is it the reference's own?

Method: on the soak's population of small tall / elastic-net problems (tests/fuzz_cases.py), run the NumPy oracle with one x-update
rounding, record its decision trace, then run the oracle with ANOTHER rounding in follow mode with an unbounded band on that trace
and collect every decision it had to take from the followed run, with the ulps of rounding it needed (the quantity the follow rule
bounds at 8).  Pairs:
    exact  follows llt32     the reference's float LLT solve against the exact solve of the same float system: the reference's OWN noise
    inv64r follows llt32     this build's default rounding (inverse formed in double, rounded once, float mat-vec) against the reference
    inv64r follows exact     this build against the exact solve
    inv32  follows llt32     the float-built inverse (ADMM_HIP_INVERSE=f32, the default above p = 4096)
If `inv64r~llt32` leaves the band as often as `exact~llt32`, the inverse is not what makes the family noisy: any two correct
executions of the reference's arithmetic part that often.  Usage: tall_noise_study.py [first_seed] [nseeds] [cases_per_seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from fuzz_cases import cases
from oracle import entry
from oracle.solvers import FollowMismatch
from oracle.variants import tall_variant

PAIRS = [("exact", "llt32"), ("inv64r", "llt32"), ("inv64r", "exact"), ("inv32", "llt32")]


def run(cs, mode, detail):
    lam = None if not cs["user_lam"] else np.sort(cs["ulam"])[::-1]
    with tall_variant(mode):
        if cs["alpha"] is None:
            return entry.admm_lasso(cs["x"], cs["y"], lam, cs["nl"], 1e-4, cs["stdz"], cs["icpt"], entry.LASSO_OPTS, detail)
        return entry.admm_enet(cs["x"], cs["y"], lam, cs["nl"], 1e-4, cs["stdz"], cs["icpt"], cs["alpha"], entry.LASSO_OPTS, detail)


def main():
    s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 901
    ns = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    per = int(sys.argv[3]) if len(sys.argv) > 3 else 150
    stat = {p: dict(dec=0, ties=0, over2=0, over8=0, cases=0, cases_over8=0, maxu=0.0) for p in PAIRS}
    for seed in range(s0, s0 + ns):
        for cs in cases(per, seed):
            if cs["kind"] not in ("tall", "enet_tall"):
                continue
            traces = {}
            for a, b in PAIRS:
                if b not in traces:
                    d = {"trace": []}
                    run(cs, b, d)
                    traces[b] = np.asarray(d["trace"], dtype=np.float64)
                d = {"follow": traces[b], "follow_band": 1e9}
                try:
                    run(cs, a, d)
                except (FollowMismatch, StopIteration, AssertionError) as e:          # the trajectories parted for good (counts differ downstream of a forced decision)
                    pass
                forced = [f for f in d.get("forced", []) if f["kind"] != "rho"]
                st = stat[(a, b)]
                st["dec"] += len(traces[b]); st["ties"] += len(forced); st["cases"] += 1
                u = [f["ulps"] for f in forced]
                st["over2"] += sum(1 for v in u if v > 2); st["over8"] += sum(1 for v in u if v > 8)
                st["cases_over8"] += 1 if any(v > 8 for v in u) else 0
                st["maxu"] = max([st["maxu"]] + u)
        print(f"# after seed {seed}", flush=True)
        for (a, b), st in stat.items():
            print(f"  {a:7s} follows {b:6s}: {st['cases']:4d} cases, {st['dec']:7d} decisions, near-ties {1e3 * st['ties'] / max(st['dec'], 1):6.2f} per 1000, "
                  f"> 2 ulps {1e3 * st['over2'] / max(st['dec'], 1):5.2f} per 1000, > 8 ulps {st['over8']:4d} decisions in {st['cases_over8']:3d} cases "
                  f"({100.0 * st['cases_over8'] / max(st['cases'], 1):.1f} %), largest {st['maxu']:.1f} ulps", flush=True)


if __name__ == "__main__":
    main()
