"""GPU: admm_hip_parbp -- basis pursuit with the columns in blocks (the "sharing" ADMM of the reference's unbuilt
src/TODO/PADMMBP.h, restated on the current PADMMBase_Master loop) -- against
oracle/solvers.py SharingBP iteration by iteration (decision trace), against the serial solver, and against the linear
programme it solves."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fit(x, y, N, **opts):
    import admm_amd
    m = admm_amd.admm_bp(x, y).parallel(N)
    if opts:
        m.opts(**opts)
    return m.fit(trace=True)


def _oracle(x, y, N, maxit=10000, eps=1e-4, rho=1.0):
    from oracle import entry
    d = {"trace": []}
    o = entry.admm_parbp(x, y, N, dict(maxit=maxit, eps_abs=eps, eps_rel=eps, rho_ratio=rho), d)
    return o, np.asarray(d["trace"], dtype=np.float64), d["solver"]


def _compare(x, y, N, label, **opts):
    fit = _fit(x, y, N, **opts)
    o, tr, sol = _oracle(x, y, N, maxit=opts.get("maxit", 10000), eps=opts.get("eps_abs", 1e-4), rho=opts.get("rho", 1.0))
    t = fit.trace
    assert t[0, 8] == -1                                        # the cold-start record
    t = t[1:]
    nrec = min(len(t), len(tr))
    assert abs(fit.stats["rho"] / sol.rho - 1) < 1e-10, (label, fit.stats["rho"], sol.rho)
    # every iteration: thresholds and residuals as the oracle has them, the same regular / active-set schedule, the same decision
    rel = lambda a, b: np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
    errs = dict(eps_p=rel(t[:nrec, 2], tr[:nrec, 1]), eps_d=rel(t[:nrec, 3], tr[:nrec, 2]), rp=rel(t[:nrec, 4], tr[:nrec, 3]),
                rd=np.abs(t[:nrec, 5] - tr[:nrec, 4]).max() / tr[:nrec, 4].max())
    assert np.array_equal(t[:nrec, 1], tr[:nrec, 0]) and np.array_equal(t[:nrec, 11], tr[:nrec, 5]), label
    assert np.array_equal(t[:nrec, 8] == 0, tr[:nrec, 6] == 1), label
    b = fit.beta.toarray().ravel()
    eb = np.abs(b - o["beta"]).max() / np.abs(o["beta"]).max()
    print(f"[parbp {label}] N={N}: {fit.niter} iterations (oracle {o['niter']}), rho {fit.stats['rho']:.4g}; trace vs oracle {({k: float(f'{v:.1e}') for k, v in errs.items()})}, "
          f"beta {eb:.1e}, support {np.count_nonzero(b)} / {np.count_nonzero(o['beta'])}")
    assert fit.niter == o["niter"], (label, fit.niter, o["niter"])
    assert len(t) == len(tr)
    assert max(errs.values()) < 1e-8 and eb < 1e-9, (label, errs, eb)
    assert np.array_equal(b != 0, o["beta"] != 0)               # the active set itself
    return fit, o


def test_readme_problem_block_counts():
    from oracle import readme
    x, y, bt = readme.bp_data()                                  # README.md:217-246: n = 50, p = 100
    import admm_amd
    ser = admm_amd.admm_bp(x, y).fit()
    for N in (2, 3, 4, 7):                                       # 3 and 7: the last block takes the remainder (PADMMBP.h:150-167)
        fit, o = _compare(x, y, N, "README n=50 p=100")
        b = fit.beta.toarray().ravel()
        assert np.abs(b - bt).max() < 2e-3                       # recovers the sparse truth like the serial solver (README range 1e-3)
        assert np.abs(b - ser.beta.toarray().ravel()).max() < 3e-3


def test_readme_perf_problem_and_ragged_rows():
    from oracle import readme
    x, y, bt = readme.bp_data(1000, 2000, 100)                   # README.md:369-393
    fit, o = _compare(x, y, 4, "README n=1000 p=2000")
    assert np.abs(fit.beta.toarray().ravel() - bt).max() < 5e-3
    rng = np.random.default_rng(3)
    n, p = 301, 1203                                             # rows not a multiple of anything, 5 blocks of 240 + 243
    x = rng.standard_normal((n, p))
    b0 = np.zeros(p); b0[rng.choice(p, 25, replace=False)] = rng.standard_normal(25) * 3
    _compare(np.asfortranarray(x), x @ b0, 5, "n=301 p=1203")


def test_against_the_linear_programme():
    """Basis pursuit is an LP: min 1'(u + w) s.t. A (u - w) = b, u, w >= 0.  The sharing solver's ||x||_1 against HiGHS."""
    from scipy.optimize import linprog
    rng = np.random.default_rng(11)
    n, p = 40, 120
    A = rng.standard_normal((n, p))
    b = A @ (rng.standard_normal(p) * (rng.uniform(size=p) < 0.08))
    lp = linprog(np.ones(2 * p), A_eq=np.hstack([A, -A]), b_eq=b, bounds=[(0, None)] * (2 * p), method="highs")
    assert lp.status == 0
    fit = _fit(np.asfortranarray(A), b, 3, eps_abs=1e-6, eps_rel=1e-6, maxit=20000)
    xg = fit.beta.toarray().ravel()
    print(f"[parbp vs LP] ||x||_1 {np.abs(xg).sum():.6f} (LP {lp.fun:.6f}), ||Ax - b|| {np.linalg.norm(A @ xg - b):.2e}, {fit.niter} iterations")
    assert fit.niter <= 20000
    assert np.linalg.norm(A @ xg - b) < 1e-3 * np.linalg.norm(b)
    assert abs(np.abs(xg).sum() / lp.fun - 1) < 1e-3


def test_maxit_exit_and_arguments():
    import admm_amd
    from oracle import readme
    x, y, _ = readme.bp_data()
    fit = _fit(x, y, 2, maxit=37)
    assert fit.niter == 38                                       # `return i + 1` after the loop (PADMMBase.h:236)
    o, tr, _ = _oracle(x, y, 2, maxit=37)
    assert o["niter"] == 38 and np.abs(fit.beta.toarray().ravel() - o["beta"]).max() < 1e-10
    with pytest.raises(RuntimeError):
        admm_amd.admm_bp(x, y).parallel(101).fit()               # more blocks than columns
    one = _fit(x, y, 1)                                          # nthread = 1 through $parallel is the serial solver, as in R
    assert one.stats["branch"] != 6


def test_exactly_8192_rows_needs_the_lds_opt_in_too():
    """n in 8065 .. 8192 pads to 8192 rows: v takes exactly 64 KB of dynamic LDS NEXT TO the kernels' static arrays, so static +
    dynamic exceeds the default per-workgroup limit although the dynamic part alone does not (round-3 advisor finding: those
    launches were refused without a trace and the solve ended 'without a decision')."""
    rng = np.random.default_rng(6)
    for n, p in ((8192, 8300), (8100, 8260)):
        x = np.asfortranarray(rng.standard_normal((n, p)))
        b0 = np.zeros(p); b0[rng.choice(p, 30, replace=False)] = rng.standard_normal(30) * 2
        _compare(x, x @ b0, 2, f"n={n} p={p}", maxit=12)       # (a regular iteration, nine active-set ones, a second regular one and its first active-set one)


def test_more_than_8192_rows_takes_the_large_lds_variant():
    """n > 8192: v needs more than 64 KB of LDS (opt-in per kernel) and the partial 32 double2 per thread.  Held to the oracle like
    the small cases (a short run: the comparison is per iteration)."""
    rng = np.random.default_rng(5)
    n, p = 8400, 8600
    x = np.asfortranarray(rng.standard_normal((n, p)))
    b0 = np.zeros(p); b0[rng.choice(p, 30, replace=False)] = rng.standard_normal(30) * 2
    _compare(x, x @ b0, 2, "n=8400 p=8600", maxit=14)
    import admm_amd
    with pytest.raises(RuntimeError):
        admm_amd.admm_bp(np.zeros((16390, 16400), order="F"), np.zeros(16390)).parallel(2).fit()      # beyond the row limit: a clear error


def test_gram_space_and_its_ways_out(monkeypatch):
    """Round 5: the active-set iterations run in Gram space (one |U| x |U| mat-vec per iteration instead of two passes over the
    non-zero columns; sharing_bp.hip "Gram space").  Every way through that code is held to the oracle like the direct launches:
    the default (a stretch after a Gram-space stretch carries everything over: no n-vector is touched at the regular iteration), every
    stretch started from the direct launches' n-vectors (option SBP_GRAM_CARRY=0), the direct launches alone (option SBP_GRAM=0), a Gram matrix too small for the support (the merge launch halts
    the stream, the host resumes with the direct launches and returns to Gram space 100, 200, ... iterations later), and one that
    overflows with columns that have come and gone first (U is rebuilt from the current lists)."""
    import admm_amd
    from oracle import readme
    x, y, _ = readme.bp_data(1000, 2000, 100)                    # README.md:369-393: ~100 non-zeros at the end, more on the way
    seen = {}
    for label, env in (("default", {}), ("direct", {"ADMM_HIP_SBP_GRAM": "0"}), ("no carry-over", {"ADMM_HIP_SBP_GRAM_CARRY": "0"}),
                       ("cap 64: halt + resume", {"ADMM_HIP_SBP_GRAM_CAP": "64"}),
                       ("cap 96: a rebuild, then halt + resume", {"ADMM_HIP_SBP_GRAM_CAP": "96"})):
        with admm_amd.options(**{k.replace("ADMM_HIP_", ""): v for k, v in env.items()}):
            fit, o = _compare(x, y, 4, f"README n=1000 p=2000, {label}")
        st = fit.stats
        seen[label] = (st["xupdate_variant"], st["xupdate_launches"], st["persist_iter"])
        print(f"[parbp gram] {label}: variant {st['xupdate_variant']}, {st['xupdate_launches']} stretches in Gram space, {st['persist_iter']} rebuilds of U")
    assert seen["default"][0] == 1 and seen["default"][1] >= fit.niter // 10 - 1
    assert seen["direct"] == (0, 0, 0) and seen["no carry-over"][0] == 1
    assert seen["cap 64: halt + resume"][0] == 2 and seen["cap 64: halt + resume"][1] >= 1
    assert seen["cap 96: a rebuild, then halt + resume"][0] == 2 and seen["cap 96: a rebuild, then halt + resume"][2] >= 1
    # a small problem whose support comes and goes: different block counts, ragged
    rng = np.random.default_rng(21)
    n, p = 120, 700
    xs = np.asfortranarray(rng.standard_normal((n, p)))
    b0 = np.zeros(p); b0[rng.choice(p, 12, replace=False)] = rng.standard_normal(12) * 4
    for N, cap in ((3, "1024"), (6, "16"), (6, "24")):
        with admm_amd.options(SBP_GRAM_CAP=cap):
            _compare(xs, xs @ b0, N, f"n=120 p=700 cap {cap}")


def test_gram_space_launches_do_not_depend_on_their_workgroups_starting_together(monkeypatch):
    """Regression (round 5, found by tests/tools/parbp_gram_soak.py: 4 of 1200 cases with one wrong recorded residual, never the same
    ones): a Gram-space launch reads, in every workgroup, the partial sums that ALL workgroups of the previous launch left, and
    every workgroup leaves its own for the next launch -- in the same array at the time, so a workgroup that started a few
    microseconds late read numbers of the launch it belonged to.  The test option SBP_TEST_DELAY_US holds every even workgroup back at the
    start of each such launch (30 us, three launches' worth; workgroup 0, whose decision is the recorded one, among them), which
    turns a dependence of that kind into a certain failure (checked against the single-buffered build when the test was written)."""
    import admm_amd
    from oracle import readme
    admm_amd.options.set(SBP_TEST_DELAY_US="30")
    x, y, _ = readme.bp_data()
    _compare(x, y, 3, "README n=50 p=100, even workgroups 30 us late")
    rng = np.random.default_rng(21)
    n, p = 120, 700
    xs = np.asfortranarray(rng.standard_normal((n, p)))
    b0 = np.zeros(p); b0[rng.choice(p, 12, replace=False)] = rng.standard_normal(12) * 4
    _compare(xs, xs @ b0, 4, "n=120 p=700, even workgroups 30 us late")
    admm_amd.options.set(SBP_GRAM_CARRY="0")
    _compare(xs, xs @ b0, 4, "n=120 p=700, no carry-over, even workgroups 30 us late")


@pytest.mark.parametrize("n,p,N,scale", [(50, 100, 3, "plain"), (300, 2400, 4, "plain"), (700, 3000, 8, "mixed"), (1030, 4100, 2, "huge"), (8400, 8600, 2, "plain")])
def test_screened_regular_iterations_are_bit_identical(n, p, N, scale):
    """The regular iterations' screen (sbp_xreg_screen_kernel: a product with the fp16-rounded column proves "stays zero" for nearly
    every column; the others take the exact double step in the arithmetic of the unscreened launch): the same solve with
    SBP_SCREEN=1 and =0 -- coefficients, iteration count and every record of the decision trace identical, also with columns whose
    entries overflow or underflow fp16.  The screened run is then held to the oracle like every other variant."""
    import admm_amd
    rng = np.random.default_rng(17 + n)
    x = rng.standard_normal((n, p))
    if scale == "mixed":
        x = x * 10.0 ** rng.uniform(-7, 5, p)
    elif scale == "huge":
        x[:, ::3] *= 2e5
        x[rng.random((n, p)) < 0.005] *= 9e4
    x = np.asfortranarray(x)
    b0 = np.zeros(p)
    idx = rng.choice(p, max(3, n // 25), replace=False)
    b0[idx] = rng.standard_normal(len(idx)) * 2 / np.abs(x[:, idx]).mean(axis=0)
    y = x @ b0
    maxit = 10000 if n <= 1100 else 24
    fits = {}
    for scr in ("1", "0"):
        with admm_amd.options(SBP_SCREEN=scr):
            fits[scr] = _fit(x, y, N, maxit=maxit)
    on, off = fits["1"], fits["0"]
    assert on.stats["xupdate_variant"] >= 4 and off.stats["xupdate_variant"] < 4
    assert on.niter == off.niter and np.array_equal(on.trace, off.trace)
    assert np.array_equal(on.beta.toarray(), off.beta.toarray())
    if scale == "plain":
        with admm_amd.options(SBP_SCREEN="1"):
            _compare(x, y, N, f"screened n={n} p={p}", maxit=maxit)
