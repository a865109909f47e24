"""GPU: mixed-precision refinement of the tall x-update (option REFINE=1 / admm_hip_options.tall_refine).

x1 = Minv rhs with the cached float inverse, r = rhs - M x1 in DOUBLE from the float system M = X'X + rho I (the system the
reference factorises, ADMMLassoTall.h:191-205), x = x1 + Minv r: one refinement step per x-update, three passes over the
lower triangle instead of one (opt-in).  Evidence that it does what it is for, without comparing executions:

  * the stepwise check (oracle/stepcheck.py) measures, at EVERY iteration, the distance between the library's x and the
    exact solve of the float system on the library's own right-hand side: refined it is a small fraction of the first-order
    float-solve yardstick (about one rounding of x), unrefined it is of the yardstick's order;
  * everything else of the iteration stays bit-identical to the reference's arithmetic (same check);
  * the refined run stays within the parity tolerance of the oracle's rounding variants (oracle/variants.py) on its own
    decisions.  (Measured, one MI355X, n = 2700, p = 2100: x-update error rms 0.003 x the yardstick refined, 0.016 x
    unrefined -- against 0.048 x for the reference's own float Cholesky solve on the same right-hand sides.)"""
import os

import numpy as np
import pytest

from helpers import col_err, oracle_following, synth_lasso, traced_fit

pytestmark = pytest.mark.gpu


def _fit(x, y, nl, refine):
    """(fit, trace, state, system): system = X'X + rho I as the library formed it (kept by the refined plan only)."""
    from admm_amd import admm_lasso
    from admm_amd.api import LassoPlan
    from admm_amd import options
    options.set(REFINE="1" if refine else None)
    try:
        plan = LassoPlan(admm_lasso(x, y).penalty(nlambda=nl))
        plan.enable_trace(1 << 14)
        plan.enable_state(1 << 14)
        fit = plan.run()
        out = (fit, plan.read_trace(), plan.read_state(), plan.read_system() if refine else None)
        plan.close()
        return out
    finally:
        options.set(REFINE=None)


def test_refined_xupdate_is_the_exact_solve_to_one_rounding():
    from oracle import entry, stepcheck
    x, y = synth_lasso(2700, 2100, 40, seed=2100)            # n = 1.3 p: cond(X'X + rho I) in the hundreds
    prob = dict(x=x, y=y, lam=None, nlambda=6, lmin_ratio=1e-4, standardize=True, intercept=True, opts=entry.LASSO_OPTS, alpha=None)
    reps = {}
    fits = {}
    system = None
    for refine in (True, False):           # the refined plan first: it hands out the system matrix both runs are measured against
        fit, trace, state, sysm = _fit(x, y, 6, refine)
        system = sysm if sysm is not None else system
        assert int(fit.stats["refine"]) == int(refine) and int(fit.stats["xupdate_variant"]) == 1
        rep = stepcheck.check_tall(prob, trace, state, label=f"refine={refine}", system=system)
        print(f"[refine={int(refine)}] {rep['records']} iterations, niter {list(map(int, fit.niter))}: x-update error <= {rep['x_ratio_max']:.3f} x yardstick "
              f"(rms {rep['x_rms_vs_yardstick']:.3f} x; {rep['x_rms_vs_ref']:.3f} x the reference float solve's), bit mismatches {len(rep['bit_mismatch'])}")
        stepcheck.assert_stepwise(rep, label=f"refine={refine}", x_factor=4.0)
        reps[refine], fits[refine] = rep, (fit, trace)
    assert reps[True]["x_rms_vs_yardstick"] < 0.35 * reps[False]["x_rms_vs_yardstick"], (reps[True]["x_rms_vs_yardstick"], reps[False]["x_rms_vs_yardstick"])
    assert reps[True]["x_ratio_max"] < 0.1
    # on the refined run's own decisions: closer to the `exact` variant than to the float Cholesky solve
    fit, trace = fits[True]
    errs = {}
    for mode in ("llt32", "exact"):
        ref, _, _ = oracle_following(trace, band=1e9, mode=mode, **prob)
        floor = 1e-2 * float(np.abs(ref["beta"]).max())
        errs[mode] = max(col_err(fit.beta_dense[:, j], ref["beta"][:, j], floor) for j in range(6))
    print(f"[refine] max column error following the refined run's decisions: vs exact variant {errs['exact']:.2e}, vs float Cholesky {errs['llt32']:.2e}")
    # (both distances are dominated by the difference between the library's float Gram and NumPy's, not by the solve: they
    # are reported, and must be inside the parity tolerance)
    assert max(errs.values()) < 1e-4
