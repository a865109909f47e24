import numpy as np

from oracle.tracecmp import compare_traces  # noqa: F401  (re-exported for the tools)


def relerr(a, b):
    """Norm-wise relative error per SURVEY.md section 8c: max|a-b| / max|b| (intercept included)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = max(np.abs(b).max(), 1e-300)
    return np.abs(a - b).max() / den


def synth_lasso(n, p, m, seed=123, sd=2.0):
    """README.md:195-201 recipe scaled: X ~ N(0, sd^2), beta* = U(0,1) on the first m, y = X beta* + N(0,1)."""
    rng = np.random.default_rng(seed)
    b = np.concatenate([rng.uniform(size=m), np.zeros(p - m)])
    x = rng.standard_normal((n, p)) * sd
    y = x @ b + rng.standard_normal(n)
    return x, y


# ---------------------------------------------------------------------------------------------------------------
# Tall path: parity judged on the decision trace, the oracle FOLLOWING the GPU through near-ties.
#
# ADMM's stopping rule (FADMMBase.h:213-217) and Goldstein's restart rule (:243) are threshold tests on float
# residuals.  At the default eps = 1e-5 those residuals are dominated by the rounding of the x-update: the NumPy
# oracle run with mathematically identical x-updates (oracle/variants.py: float LLT solve = the reference, float
# inverse, inverse rounded once from double, exact solve) changes the iteration count of ~40 of 185 lambdas between
# ANY two of them, the reference's float LLT solve against the exact solve included (tests/tools/flip_floor.py,
# tests/test_flip_floor.py), and on ill-conditioned problems one such flip can end a lambda thousands of iterations
# early.  Identical counts against one particular rounding are therefore not a property any implementation can have.
# What is required instead -- with no tolerance for "columns downstream of a count flip":
#   R1  the oracle is run in FOLLOW mode (oracle/solvers.py FADMM._decide): it takes the GPU's outcome of a threshold
#       test only where moving every entry of its OWN x and z by at most `band` (8) float ulps could have produced that
#       outcome (a near-tie rounding decides; the measured need is <= 4.1 ulps over all cases of tests/tools/flip_floor.py);
#       any other disagreement fails the test.  Every decision taken this way is counted and its distance reported.
#   R2  on that common trajectory the iteration counts are identical for every lambda and every beta column is within
#       `tol` (1e-4, the north_star bar) of the oracle's;
#   R3  the only columns that may exceed `tol` are those where the reference's own formula loses the digits: they must be
#       within `factor` (5) x the distance the oracle's rounding variants (oracle/variants.py: the x-update as a float
#       inverse / as an exact solve; the column statistics of DataStd, or X'y, accumulated in double instead of float), following
#       the same decisions, have drifted from it by that lambda
#       (e.g. maxit = 7 with rho five orders below the automatic value: z = (x + y/rho) - lambda/rho cancels 5 digits).
#       The intercept row is held to the variants' largest ABSOLUTE drift along the path instead: its rounding noise (an ulp of a
#       column mean in mean(y) - sum_j mean(x_j) beta_j) is additive, the same in every column whatever the column's own scale.
#       The number of such columns is returned; tests bound and print it.
# R4  (round 3) follow mode is only as strict as the decisions it hands over, so every traced test ASSERTS ceilings on them
#     instead of printing them.  Measured over the passing cases of profiles/r03_soak_summary.md (11.1 M decisions):
#     near-ties are 0.2-1 per 1000 decisions for the wide / consensus / LAD / BP solvers and 9-10 per 1000 for the tall
#     family, where they cluster in unstandardised problems: there the dual residual rho ||z - z_old|| of a finished lambda
#     is a handful of single-ulp flips of z, i.e. quantised right at its threshold, and EVERY lambda ends on a decision
#     inside one rounding (per-case rate: median 0.8 %, 99th percentile 12 %).  What a biased kernel would change is not
#     their number but their SIZE: the ulps needed are 0.03 at the median, 0.5 at the 90th and 4.5 at the 99th percentile.
#     Hence three ceilings:  (a) EPISODES of near-ties needing more than NEAR_TIE_SMALL (2) ulps: at most max(2, 0.3 % of
#     the decisions) -- an episode is a stretch of one lambda in which they recur every iteration or two: a path caught in a
#     limit cycle of single-ulp flips of z (r_d alternating between two quantised values either side of eps_d) repeats the
#     SAME near-tie for dozens of iterations, one event, not dozens;  (b) none beyond NEAR_TIE_ULPS (7.5; the band itself is
#     8);  (c) all of them together: at most max(5, 15 %).
#     What these ceilings cannot see -- a bias that hides inside one rounding -- the stepwise check can: wherever a test has
#     the iterate dump (every tall / elastic-net / consensus case of the random sweep, tests/test_gpu_fuzz.py) it also
#     requires every single iteration to be the reference's, bit for bit (oracle/stepcheck.py).
NEAR_TIE_SMALL = 2.0
NEAR_TIE_LARGE_RATE = 0.003
NEAR_TIE_ULPS = 7.5
NEAR_TIE_TOTAL_RATE = 0.15


def near_tie_stats(forced):
    sized = [f for f in forced if f["kind"] != "rho"]
    large, last = 0, None                          # episodes: same lambda, at most 2 iterations after the previous near-tie
    in_large = False
    for f in sized:
        key = (f["lam"], f["iter"])
        new_episode = last is None or f["lam"] != last[0] or f["iter"] - last[1] > 2
        if new_episode:
            in_large = False
        if f["ulps"] > NEAR_TIE_SMALL and not in_large:
            large += 1
            in_large = True
        last = key
    return dict(total=len(forced), large=large, max_ulps=max((f["ulps"] for f in sized), default=0.0))


def assert_near_tie_budget(forced, ndecisions, label="", enabled=True):
    """R4: the decisions the oracle took from the GPU (`forced`, dicts with kind / ulps / lam / iter) are rounding-sized."""
    if not enabled:
        return
    st = near_tie_stats(forced)
    show = [(f["lam"], f["iter"], f["kind"], round(f["ulps"], 2)) for f in forced if f["kind"] != "rho" and f["ulps"] > NEAR_TIE_SMALL][:10]
    allowed_large = max(2, int(np.ceil(NEAR_TIE_LARGE_RATE * ndecisions)))
    assert st["large"] <= allowed_large, (label, f"{st['large']} episodes (of {ndecisions} decisions) in which the GPU's outcome needed more than "
                                          f"{NEAR_TIE_SMALL} ulps of rounding; at most {allowed_large} allowed", show)
    assert st["max_ulps"] <= NEAR_TIE_ULPS, (label, f"a near-tie needing {st['max_ulps']:.2f} ulps of rounding (ceiling {NEAR_TIE_ULPS})", show)
    allowed = max(5, int(np.ceil(NEAR_TIE_TOTAL_RATE * ndecisions)))
    assert st["total"] <= allowed, (label, f"{st['total']} of {ndecisions} decisions taken from the GPU as near-ties; at most {allowed} allowed")


def traced_fit(model, capacity=1 << 18, state=False, data=False):
    """Run a configured ADMM_Lasso / ADMM_Enet model through the prepared-problem entry points with the decision trace
    (and, with state=True or a record count, the iterate dump of every iteration: returns (fit, trace, state); with data=True
    -- wide solver -- also the standardised (X, Y) as the library holds them: (fit, trace, state, (X, Y)))."""
    from admm_amd.api import LassoPlan
    plan = LassoPlan(model)
    plan.enable_trace(capacity)
    if state:
        plan.enable_state(capacity if state is True else int(state))
    fit = plan.run()
    trace = plan.read_trace()
    st = plan.read_state() if state else None
    xy = plan.read_data() if data else None
    plan.close()
    if data:
        return fit, trace, st, xy
    return (fit, trace, st) if state else (fit, trace)


STEPWISE_MAX_ELEMS = 120_000_000    # the CPU replay holds X and |X| in double: beyond this many entries the stepwise rule is skipped
WIDE_STATE_BYTES = 1 << 30          # iterate dump of the wide solver: at most this much (records x (p + 3 n) floats)


def wide_stepwise(fit, trace, st, xy, problem, label=""):
    """oracle/stepcheck.py on the wide solver's iterate dump (as many records as were captured): every iteration the library made
    is the reference's iteration applied to the library's own previous iterates -- zero pattern, z, y bit for bit, the two
    mat-vecs within the float dot-product yardstick, thresholds / residuals / decisions / rho adaptation / schedule exact."""
    from oracle import stepcheck
    rep = stepcheck.check_wide(problem, trace, st, fit.stats["eig_est"], X=xy[0], Y=xy[1], label=label)
    stepcheck.assert_stepwise_wide(rep, label=label)
    print(f"[stepwise {label}] {rep['decisions_checked']} iterations replayed (zero / regular / active-set {rep['kinds'][0]} / {rep['kinds'][1]} / {rep['kinds'][2]}; "
          f"{rep['rho_changes']} rho changes): z, y, zero pattern bit-exact; X't within {rep['xt_ratio_max']:.2f} (rms {rep['xt_rms']:.2f}), "
          f"A x within {rep['ax_ratio_max']:.2f} (rms {rep['ax_rms']:.2f}) float-dot yardsticks; norms {rep['norm_rel_max']:.1e}; "
          f"{len(rep['accum_ties'])} float-accumulator ties")
    return rep


def wide_state_records(problem, cap):
    n, p = np.asarray(problem["x"]).shape
    return int(max(16, min(cap, WIDE_STATE_BYTES // (4 * (p + 3 * n)))))


def assert_rho_self_consistent(trace, first_iter, label=""):
    """The rho adaptation (FADMMBase.h:109-133 == ADMMBase.h:85-109; from iteration `first_iter` on: i > 3 for ADMMBase,
    i > 5 for FADMMBase) of every non-final record must be the rule applied to the residuals recorded with it: the recorded
    multiplier rho_out / rho_in equals the rule's."""
    from oracle.solvers import _rho_rule
    t = np.asarray(trace, dtype=np.float64)
    if len(t) and t[0, 8] == -1:
        t = t[1:]
    bad = []
    for k, r in enumerate(t):
        if r[8] == 0 or r[9] <= 0:                           # converged: no adaptation; r[9] = rho before, r[10] = rho after
            continue
        class _S:
            pass
        q = _S()
        q.rho, q.eps_primal, q.eps_dual, q.resid_primal, q.resid_dual = 1.0, r[2], r[3], r[4], r[5]
        if int(r[1]) > first_iter - 1:
            _rho_rule(q)
        if abs(q.rho - r[10] / r[9]) > 1e-12 * q.rho:
            bad.append((k, q.rho, r[10] / r[9]))
    assert not bad, (label, "rho adaptation inconsistent with the recorded residuals", bad[:5])


def assert_trace_self_consistent(trace, accelerated, label=""):
    """Every recorded decision must be the one the reference's rule gives for the values recorded WITH it (the GPU's own
    residuals and thresholds): converged iff r_p < eps_p and r_d < eps_d (FADMMBase.h:213-217, ADMMBase.h:196-197,
    PADMMBase.h:230-231); for the accelerated solvers otherwise accelerate iff c < 0.999 c_old, else restart
    (FADMMBase.h:243).  Independent of the oracle: it checks the decision logic, not the iterates."""
    t = np.asarray(trace, dtype=np.float64)
    if len(t) and t[0, 8] == -1:
        t = t[1:]
    if not len(t):
        return 0
    conv = (t[:, 4] < t[:, 2]) & (t[:, 5] < t[:, 3])
    assert np.array_equal(conv, t[:, 8] == 0), (label, "stopping decisions inconsistent with the recorded residuals",
                                                 np.nonzero(conv != (t[:, 8] == 0))[0][:5])
    if accelerated:
        nc = ~conv
        acc = t[nc, 6] < 0.999 * t[nc, 7]
        assert np.array_equal(acc, t[nc, 8] == 1), (label, "restart decisions inconsistent with the recorded combined residuals",
                                                    np.nonzero(acc != (t[nc, 8] == 1))[0][:5])
    return len(t)


def oracle_following(trace, x, y, lam, nlambda, lmin_ratio, standardize, intercept, opts, alpha=None, mode="llt32", band=8.0, nthread=None, follow_x=None):
    """The oracle (x-update rounding `mode`, tall solver only) following the decisions of `trace` (a libadmm_hip trace or
    another oracle's); tall, wide (n <= p) or -- with nthread -- consensus solver, as the reference would dispatch.
    Returns (result dict, forced decisions, number of decisions consumed)."""
    from oracle import entry
    from oracle.variants import tall_variant
    t = np.asarray(trace, dtype=np.float64)
    if len(t) and t[0, 8] == -1:
        t = t[1:]                                       # libadmm_hip's cold-start record
    d = {"follow": t, "follow_band": band}
    if follow_x is not None:                            # wide solver: the followed run's x after every iteration (prox near-ties, oracle/solvers.py LassoWide)
        d["follow_x"] = follow_x
    with tall_variant(mode):
        if nthread is not None:
            ref = entry.admm_parlasso(x, y, lam, nlambda, lmin_ratio, standardize, intercept, nthread, opts, d)
        elif alpha is None:
            ref = entry.admm_lasso(x, y, lam, nlambda, lmin_ratio, standardize, intercept, opts, d)
        else:
            ref = entry.admm_enet(x, y, lam, nlambda, lmin_ratio, standardize, intercept, alpha, opts, d)
    ref["_detail"] = d                                  # solver (rho) and DataStd of this run, for the yardsticks
    return ref, d["forced"], d["solver"].ndecisions


def threshold_quantum(ref, n, alpha=None):
    """Per lambda: one float ulp of the tall z-update's soft-threshold operand, as an absolute error of the returned
    coefficients.  z = soft(v, kappa) with v = x + y_dual / rho and kappa = lambda n / (scaleY rho)
    (ADMMLassoTall.h:55-69,81-85; elastic net: z = (v - alpha kappa) / (1 + kappa (1 - alpha)), ADMMEnet.h:24-45): every
    surviving coordinate is a difference with the threshold, so it is known only to ulp(|v|).  When rho is tiny against
    lambda (a user-fixed rho on unstandardised data) kappa is thousands of times the coefficient and that ulp is
    1e-4 .. 1e-3 of it: the reference's own formula loses those digits, and two correct executions differ by such quanta."""
    d = ref["_detail"]
    rho = float(d["solver"].rho)
    std = d["std"]
    sy = float(std.scaleY)
    sx = np.asarray(std.scaleX, dtype=np.float64) if np.ndim(std.scaleX) else np.full(ref["beta"].shape[0] - 1, float(std.scaleX))
    a = 1.0 if alpha is None else float(alpha)
    out = []
    for j, lam in enumerate(np.asarray(ref["lambda"], dtype=np.float64)):
        kappa = lam * n / sy / rho
        den = 1.0 + kappa * (1.0 - a)
        bstd = np.abs(ref["beta"][1:, j].astype(np.float64)) * sx / sy            # back to the solver's units
        q = float(np.spacing(np.float32(a * kappa + bstd.max() * den))) / den
        out.append(q * sy / float(sx.min()))
    return np.asarray(out)


def assert_followed_parity(beta, niter, trace, problem, tol=1e-4, band=8.0, label="", factor=5.0, budget=True, state=None):
    """Wide / consensus solvers (no rounding variants of the x-update there): the oracle follows the GPU through
    rounding-level near-ties of the stopping test and of the rho adaptation only; iteration counts identical for every
    lambda and every beta column within `tol`."""
    assert_trace_self_consistent(trace, accelerated=False, label=label)
    if problem.get("nthread") is None:
        assert_rho_self_consistent(trace, first_iter=4, label=label)               # ADMMBase::solve: update_rho from i > 3
    # wide solver with the library's iterate dump: the oracle also follows the zero pattern of the x-update through near-ties of the
    # soft threshold (a zero stays out of the active set until the next regular step: a fork like a stopping near-tie) -- counted
    # in `forced` as kind "prox", with the same band in units of the float dot product's error yardstick
    follow_x = None
    t = np.asarray(trace)
    nrec = len(t) - (1 if len(t) and t[0, 8] == -1 else 0)
    if state is not None and problem.get("nthread") is None:
        st = np.asarray(state)
        p_ = np.asarray(problem["x"]).shape[1]
        off = 1 if len(t) and t[0, 8] == -1 else 0       # dump record r belongs to trace record r: row 0 is the cold-start record's (empty)
        if st.ndim == 2 and st.shape[1] >= p_ and len(st) > off:
            follow_x = [row[:p_] for row in st[off:]]    # (a dump shorter than the run: followed while it lasts)
    ref, forced, ndec = oracle_following(trace, band=band, follow_x=follow_x, **problem)
    assert ndec == nrec, (label, "the oracle consumed a different number of decisions than the GPU took", ndec, nrec)
    ng, nr = np.asarray(niter, dtype=int), np.asarray(ref["niter"], dtype=int)
    assert np.array_equal(ng, nr), (label, ng, nr)
    nl = beta.shape[1]
    floor = 1e-2 * max(float(np.abs(ref["beta"]).max()), coef_scale(problem))          # as in assert_tall_parity
    errs = [col_err(beta[:, j], ref["beta"][:, j], floor) for j in range(nl)]
    nstop = sum(1 for f in forced if f["kind"] == "stop")
    nprox = sum(1 for f in forced if f["kind"] == "prox")
    fm = max([f["ulps"] for f in forced if f["kind"] == "stop"], default=0.0)
    print(f"[parity {label}] {nrec} decisions, {nstop} stopping near-ties (largest needs {fm:.2f} ulps), {len(forced) - nstop - nprox} rho near-ties"
          + (f", {nprox} soft-threshold near-ties (largest {max(f['ulps'] for f in forced if f['kind'] == 'prox'):.2f} yardsticks)" if nprox else "") + " "
          f"taken from the GPU ({near_tie_stats(forced)['large']} need > {NEAR_TIE_SMALL:g} ulps); niter identical; max beta err {max(errs):.2e}")
    assert_near_tie_budget(forced, nrec, label, enabled=budget)
    bad = [(j, e) for j, e in enumerate(errs) if e >= tol]
    if bad:
        # like R3 of the tall rule -- a column may exceed `tol` only within `factor` x the distance the oracle's own rounding
        # variants, following the same decisions, have drifted from it by that lambda.  Consensus solver: the workers' solves
        # (float inverse, exact), A_k'b_k and the column statistics (paths that run into maxit accumulate the rounding of
        # hundreds of solves).  Wide solver (no solve, no X'y): the column statistics of DataStd only -- with an intercept on
        # uncentred data the recovered beta_0 = mean(y) - sum_j mean(x_j) beta_j / sd_j cancels, and an ulp of a column mean
        # or scale shows up 1e-4 of beta_0 later (soak case 539:48: row 0 of one column at 3.1e-4, every other row at 1.6e-6,
        # the stats64 variant of the oracle at 1.5e-4 from the oracle proper in the same row)
        drift = np.zeros(nl)
        for mode in (("inv32", "exact", "stats64", "xy64") if problem.get("nthread") is not None else ("stats64",)):
            v, _, _ = oracle_following(trace, band=1e9, mode=mode, **problem)
            drift = np.maximum(drift, [col_err(v["beta"][:, j], ref["beta"][:, j], floor) for j in range(nl)])
        drift = np.maximum.accumulate(drift)
        print(f"[parity {label}] columns beyond {tol:g}: {bad}; the oracle's rounding variants differ by up to {drift.max():.2e}")
        bad = [(j, e) for j, e in bad if e > factor * drift[j]]
    assert not bad, (label, bad)
    return dict(forced=forced, max_err=max(errs), errs=errs, ref=ref)


def coef_scale(problem):
    """Natural size of a coefficient of this problem: the largest univariate least-squares coefficient
    |x_j'y| / (x_j'x_j) (centred when an intercept is fitted).  The yardstick for columns of a path that are (nearly) null:
    at lambda_max the coordinate attaining max|X'y| sits exactly on the soft threshold and survives with a value that is
    the difference of two nearly equal numbers -- its RELATIVE error is meaningless, its error relative to the size
    coefficients have in this problem is not.  (A path of a single lambda at lambda_max has no other scale at all.)"""
    x = np.asarray(problem["x"], dtype=np.float64)
    y = np.asarray(problem["y"], dtype=np.float64)
    if problem.get("intercept", True):
        x = x - x.mean(axis=0)
        y = y - y.mean()
    den = (x * x).sum(axis=0)
    den[den == 0] = np.inf
    return float(np.abs(x.T @ y / den).max())


def col_err(a, b, floor, icpt_row=True):
    """SURVEY.md section 8c's per-column metric max|a - b| / max|b| (intercept row included).  `floor` (1 % of the path's scale,
    callers) replaces the column's own scale ONLY for a column with at most one surviving slope coordinate: at lambda_max the
    coordinate attaining max|X'y| sits exactly on the soft threshold and survives as the difference of two nearly equal numbers
    (or not at all on one side) -- its relative error is meaningless, and a column that is all zeros has no scale at all.
    Every other column is held to ITS OWN largest coefficient (round 3 applied the floor to every column, which accepted a
    column 1e-3 the size of the path's at 1e-3 relative to itself: 10 x looser than the stated metric)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    nnz = int(np.count_nonzero(b[1:] if icpt_row else b))          # icpt_row=False: the caller passes the slope rows only
    own = float(np.abs(b).max())
    scale = max(own, floor) if nnz <= 1 else own
    return np.abs(a - b).max() / max(scale, 1e-300)


def assert_tall_parity(beta, niter, trace, problem, tol=1e-4, factor=5.0, band=8.0, label="", budget=True):
    """problem: dict(x, y, lam, nlambda, lmin_ratio, standardize, intercept, opts, alpha) -- the oracle's arguments."""
    assert_trace_self_consistent(trace, accelerated=True, label=label)
    ref, forced, ndec = oracle_following(trace, band=band, **problem)          # R1 (raises FollowMismatch)
    t = np.asarray(trace)
    nrec = len(t) - (1 if len(t) and t[0, 8] == -1 else 0)
    assert ndec == nrec, (label, "the oracle consumed a different number of decisions than the GPU took", ndec, nrec)
    ng, nr = np.asarray(niter, dtype=int), np.asarray(ref["niter"], dtype=int)
    assert np.array_equal(ng, nr), (label, ng, nr)                               # R2
    nl = beta.shape[1]
    # null / tiny columns (lambda_max: the coordinate attaining max|X'y| sits exactly on the soft-threshold and may
    # survive with a value of 1e-7 of the path's scale on one side) are measured against 1 % of the path's largest coefficient
    floor = 1e-2 * max(float(np.abs(ref["beta"]).max()), coef_scale(problem))
    errs = [col_err(beta[:, j], ref["beta"][:, j], floor) for j in range(nl)]
    loose, yard = [], 0.0
    # two ulps of the soft-threshold operand (threshold_quantum): where the reference's formula itself cannot resolve the
    # coefficient any finer, an error of that size is not a disagreement
    quanta = threshold_quantum(ref, np.asarray(problem["x"]).shape[0], problem.get("alpha"))
    scales = np.array([max(float(np.abs(ref["beta"][:, j]).max()), floor) for j in range(nl)])
    errs_eff = [0.0 if errs[j] * scales[j] <= 2.0 * quanta[j] else errs[j] for j in range(nl)]
    if max(errs_eff) >= tol:                                                     # R3
        drift = np.zeros(nl)
        drift0 = np.zeros(nl)                       # the intercept row's drift in ABSOLUTE terms
        for mode in ("inv32", "exact", "stats64", "xy64"):
            v, _, _ = oracle_following(trace, band=1e9, mode=mode, **problem)
            drift = np.maximum(drift, [col_err(v["beta"][:, j], ref["beta"][:, j], floor) for j in range(nl)])
            drift0 = np.maximum(drift0, np.abs(v["beta"][0] - ref["beta"][0]))
        drift = np.maximum.accumulate(drift)        # along a warm-started path the drift of a lambda carries into the next ones
        drift0 = np.maximum.accumulate(drift0)
        for j in range(nl):
            if errs_eff[j] >= tol:
                yard = max(yard, float(drift[j]))
                ok = errs[j] <= factor * drift[j]
                if not ok and problem.get("intercept"):
                    # The recovered intercept mean(y) - sum_j mean(x_j) beta_j cancels on uncentred data, and its rounding noise --
                    # an ulp of a column mean -- is ADDITIVE: the same absolute amount in every column of the path, whatever the
                    # column's own scale (soak case 705:134, columns of scale 23.5 / 0.49 / 7.9: the variants' intercepts differ by
                    # 6e-5, 1e-6, 7e-5; the GPU's by 7e-5, 1.2e-4, 6e-5 -- relative to the small middle column that is 2.3e-4).
                    # Row 0 is therefore held to the variants' largest ABSOLUTE intercept drift so far, the slope rows to the
                    # relative yardstick as before.
                    e_slopes = col_err(beta[1:, j], ref["beta"][1:, j], floor, icpt_row=False)
                    ok = (e_slopes < tol or e_slopes <= factor * drift[j]) and abs(beta[0, j] - ref["beta"][0, j]) <= factor * drift0[j]
                assert ok, (label, f"lambda {j}: error {errs[j]:.2e} on a common trajectory; the oracle's own "
                                   f"rounding variants differ by up to {drift[j]:.2e} by then (intercept: {drift0[j]:.2e} absolute)")
                loose.append(j)
    fm = max([f["ulps"] for f in forced], default=0.0)
    first = forced[0]["lam"] if forced else None
    print(f"[parity {label}] {nrec} decisions, {len(forced)} near-ties taken from the GPU ({near_tie_stats(forced)['large']} need > {NEAR_TIE_SMALL:g} ulps; "
          f"largest needs {fm:.2f} ulps of rounding, first at lambda {first}); niter identical; max beta err {max(errs):.2e}; columns beyond {tol:g}: {len(loose)} of {nl}"
          + (f" {loose} (oracle rounding variants differ by {yard:.2e})" if loose else ""))
    assert_near_tie_budget(forced, nrec, label, enabled=budget)
    return dict(forced=forced, max_ulps=fm, first_forced_lambda=first, loose=loose, max_err=max(errs), errs=errs, ref=ref)


def assert_dense_followed(kind, fit_beta, fit_niter, trace, x, y, opts, intercept=True, tol=1e-4, band=8.0, label="", budget=True):
    """LAD / BP (FADMMBase::solve with the rho adaptation of FADMMBase.h:109-133, float64): the oracle follows the GPU's
    decision trace through rounding-level near-ties of the stopping / restart tests and of the rho adaptation only;
    iteration counts identical, beta within `tol`."""
    from oracle import entry
    assert_trace_self_consistent(trace, accelerated=True, label=label)
    assert_rho_self_consistent(trace, first_iter=6, label=label)                   # FADMMBase::solve: update_rho from i > 5
    t = np.asarray(trace, dtype=np.float64)
    if len(t) and t[0, 8] == -1:
        t = t[1:]
    d = {"follow": t, "follow_band": band}
    ref = entry.admm_lad(x, y, intercept, opts, d) if kind == "lad" else entry.admm_bp(x, y, opts, d)
    assert d["solver"].ndecisions == len(t), (label, d["solver"].ndecisions, len(t))
    assert int(fit_niter) == int(ref["niter"]), (label, fit_niter, ref["niter"])
    err = relerr(fit_beta, ref["beta"])
    print(f"[parity {label}] {len(t)} decisions, {len(d['forced'])} near-ties taken from the GPU ({near_tie_stats(d['forced'])['large']} need > "
          f"{NEAR_TIE_SMALL:g} ulps); niter identical ({int(fit_niter)}); beta err {err:.2e}")
    assert err < tol, (label, err)
    assert_near_tie_budget(d["forced"], len(t), label, enabled=budget)
    return dict(forced=d["forced"], err=err, ref=ref)


def traced_parity(model, problem, tol=1e-4, label="", capacity=None, **kw):
    """Fit a configured ADMM_Lasso / ADMM_Enet model with the decision trace and judge it by the rule above (R1-R4): the tall
    rule for n > p without $parallel(), the followed rule for the wide and consensus solvers -- the wide solver ALSO by the
    stepwise rule on its iterate dump (wide_stepwise).  `problem` = the oracle's
    arguments (dict x, y, lam, nlambda, lmin_ratio, standardize, intercept, opts, alpha[, nthread]).  Returns (fit, report)."""
    nl = len(problem["lam"]) if problem.get("lam") is not None else int(problem["nlambda"])
    cap = capacity or max(nl, 1) * (int(problem["opts"]["maxit"]) + 2) + 8
    n, p = np.asarray(problem["x"]).shape
    if n <= p and problem.get("nthread") is None and n * p <= STEPWISE_MAX_ELEMS:      # wide solver: also the stepwise rule on its iterate dump
        fit, trace, st, xy = traced_fit(model, capacity=min(cap, 1 << 22), state=wide_state_records(problem, min(cap, 1 << 22)), data=True)
        wide_stepwise(fit, trace, st, xy, problem, label)
        kw = dict(kw, state=st)                                  # the oracle follows the x-update's zero pattern through threshold near-ties
    else:
        fit, trace = traced_fit(model, capacity=min(cap, 1 << 22))
    if n > p and problem.get("nthread") is None:
        rep = assert_tall_parity(fit.beta_dense, fit.niter, trace, problem, tol, label=label, **kw)
    else:
        rep = assert_followed_parity(fit.beta_dense, fit.niter, trace, problem, tol, label=label, **kw)
    return fit, rep


DENSE_STATE_BYTES = 1 << 30


def dense_state_records(dim, maxit):
    return int(max(16, min(maxit + 8, DENSE_STATE_BYTES // (8 * 5 * dim))))


def dense_stepwise(kind, fit, x, y, opts, intercept=True, label=""):
    """oracle/stepcheck.py on the LAD / BP iterate dump (fit.trace, fit.state of `.fit(trace=True, state=N)`): adj, z, y bit for
    bit, the projection against Householder QR, decisions and the rho adaptation exact."""
    from oracle import stepcheck
    rep = stepcheck.check_dense(kind, x, y, opts, fit.trace, fit.state, intercept=intercept, label=label)
    stepcheck.assert_stepwise_dense(rep, label=label)
    print(f"[stepwise {label}] {rep['decisions_checked']} iterations replayed ({rep['rho_changes']} rho changes): adj / z / y bit-exact; projection within "
          f"{rep['x_vs_ref_max']:.2f} x the reference route's error (rms {rep['x_rms_vs_ref']:.2f}; relative {rep['x_rel_max']:.1e}); norms {rep['norm_rel_max']:.1e}")
    return rep
