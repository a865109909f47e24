"""CPU: the stepwise checker (oracle/stepcheck.py) itself.

(1) Applied to the NumPy oracle's OWN run (decision trace + iterate log of oracle.solvers) it must come out clean -- every
    elementwise step bit-identical, the x-update's error equal to the reference float solve's own -- for the tall Lasso, the
    elastic net and the consensus solver (Cholesky and Woodbury blocks).
(2) It must CATCH what it exists for: a fused multiply-add in the dual update (what hipcc had made of `adj_y + rho * r`
    until round 3), an x-update that is off by more than a float solve can be, a decision that is not the rule's.
(3) A capture of libadmm_hip itself (tests/golden/stepwise_capture_*.npz, written on an MI355X by
    tests/tools/soak_state.py: the iterate dump of one of the soak's hard cases) is judged clean on the CPU."""
import os

import numpy as np
import pytest

from helpers import synth_lasso

HERE = os.path.dirname(os.path.abspath(__file__))


def _oracle_run(kind, x, y, nl=6, maxit=300, K=3, alpha=0.6):
    from oracle import entry
    d = {"trace": [], "state": []}
    prob = dict(x=x, y=y, lam=None, nlambda=nl, lmin_ratio=1e-4 if x.shape[0] > x.shape[1] else 0.01, standardize=True, intercept=True,
                opts=dict(entry.LASSO_OPTS, maxit=maxit), alpha=None)
    if kind == "tall":
        entry.admm_lasso(x, y, None, nl, prob["lmin_ratio"], True, True, prob["opts"], d)
    elif kind == "enet":
        prob["alpha"] = alpha
        entry.admm_enet(x, y, None, nl, prob["lmin_ratio"], True, True, alpha, prob["opts"], d)
    else:
        prob["nthread"] = K
        entry.admm_parlasso(x, y, None, nl, prob["lmin_ratio"], True, True, K, prob["opts"], d)
    p = x.shape[1]
    sol = d["solver"]
    tr = np.asarray(d["trace"], dtype=np.float64)
    tr[:, 11] = [float(np.float32(v)) if kind != "par" else v for v in _lams(sol, tr, d, prob, x, y)]
    cold = np.zeros((1, tr.shape[1])); cold[0, 8] = -1
    tr = np.vstack([cold, tr])
    r0 = np.zeros((1, len(d["state"][0])), np.float32)           # record 0: X'y / A_k'b_k, as libadmm_hip's dump holds them
    if kind == "par":
        for w in range(K):
            r0[0, (1 + w) * p:(2 + w) * p] = sol.Ab[w]
    else:
        r0[0, :p] = sol.XY
    return prob, tr, np.vstack([r0, np.asarray(d["state"])])


def _lams(sol, tr, d, prob, x, y):
    """internal lambda per trace record (what libadmm_hip writes into field 11)"""
    from oracle.entry import _lambda_grid
    n = x.shape[0]
    lam = _lambda_grid(sol.lambda0, n, d["std"].scaleY, prob["nlambda"], prob["lmin_ratio"])
    li = lam * n / np.float64(d["std"].scaleY)
    return [li[int(r[0])] for r in tr]


@pytest.mark.parametrize("kind,n,p", [("tall", 300, 40), ("enet", 300, 40), ("par", 300, 40), ("par", 90, 40)])
def test_stepcheck_is_clean_on_the_oracles_own_run(kind, n, p):
    from oracle import stepcheck
    x, y = synth_lasso(n, p, 5, seed=4)
    prob, tr, st = _oracle_run(kind, x, y)
    fn = stepcheck.check_consensus if kind == "par" else stepcheck.check_tall
    rep = fn(prob, tr, st, label=kind)
    assert rep["records"] > 100 and not rep["bit_mismatch"] and not rep["accum_ties"]
    assert rep.get("x_vs_ref_max", 0.0) <= 1.0 + 1e-12          # the oracle's x IS the reference float solve
    # (the oracle records its norms with float accumulators, libadmm_hip with double ones: 1e-6 here, 1e-15 there)
    stepcheck.assert_stepwise(rep, label=kind, x_factor=4.0, norm_tol=1e-5)


def test_stepcheck_catches_a_fused_dual_update_a_bad_solve_and_a_wrong_decision():
    from oracle import stepcheck
    x, y = synth_lasso(300, 40, 5, seed=4)
    prob, tr, st = _oracle_run("tall", x, y)
    p = 40
    S = st.reshape(len(st), 5, p).copy()
    # (a) y = fma(rho, r, adj_y): one rounding instead of two, from iteration 20 on
    rho = np.float32(tr[1, 9])
    bad = S.copy()
    for k in range(20, len(S)):
        r = (bad[k, 0] - bad[k, 1]).astype(np.float32)
        bad[k, 2] = (bad[k, 4].astype(np.float64) + np.float64(rho) * r.astype(np.float64)).astype(np.float32)
    rep = stepcheck.check_tall(prob, tr, bad.reshape(len(S), -1), label="fma")
    assert any(m[3] == "y" for m in rep["bit_mismatch"])
    with pytest.raises(AssertionError, match="elementwise"):
        stepcheck.assert_stepwise(rep, norm_tol=1e-5)
    # (b) an x-update with a relative error of 1e-4 at one iteration (z, y recomputed from it, as a real solver would)
    bad = S.copy()
    bad[30, 0] = (bad[30, 0] * np.float32(1.0001)).astype(np.float32)
    rep = stepcheck.check_tall(prob, tr, bad.reshape(len(S), -1), label="solve")
    assert rep["x_ratio_max"] > 50 and rep["x_worst"][0] == 30
    # (c) a decision that is not the rule's: flip one accelerate into a restart in the trace
    k = int(np.nonzero(tr[:, 8] == 1)[0][10])
    t2 = tr.copy(); t2[k, 8] = 2
    with pytest.raises(AssertionError, match="the library decided"):
        stepcheck.check_tall(prob, t2, st, label="decision")


@pytest.mark.parametrize("name", sorted(f for f in os.listdir(os.path.join(HERE, "golden")) if f.startswith("stepwise_capture_")))
def test_stepcheck_on_a_capture_of_the_library(name):
    """The iterate dump of libadmm_hip for one of the soak's hard cases, captured on an MI355X: every iteration replayed on
    the CPU -- bit-identical elementwise steps, x-update within the float-solve yardstick, decisions = the rule."""
    import re
    import sys
    sys.path.insert(0, HERE)
    from fuzz_cases import cases
    import test_gpu_fuzz as T
    from oracle import stepcheck
    m = re.search(r"s(\d+)_c(\d+)\.npz$", name)
    seed, c = int(m.group(1)), int(m.group(2))
    cs = next(k for k in cases(c + 1, seed) if k["c"] == c)
    cap = dict(np.load(os.path.join(HERE, "golden", name)))
    rep = T.stepwise_capture(cs, cap)
    assert rep["records"] == len(cap["trace"]) - 1 and rep["decisions_checked"] == rep["records"]
    stepcheck.assert_stepwise(rep, label=name, x_factor=4.0)


# ---------------------------------------------------------------------------------------------------- round 4: wide, LAD, BP
def _oracle_wide_run(x, y, nl=6, maxit=400, alpha=None, lam=None):
    """The oracle's wide solver with its decision trace, iterate log and x-update kinds, in libadmm_hip's layouts."""
    from oracle import entry
    from oracle.entry import _lambda_grid
    d = {"trace": [], "state": [], "types": []}
    prob = dict(x=x, y=y, lam=lam, nlambda=nl, lmin_ratio=0.01, standardize=True, intercept=True, opts=dict(entry.LASSO_OPTS, maxit=maxit), alpha=alpha)
    if alpha is None:
        entry.admm_lasso(x, y, lam, nl, 0.01, True, True, prob["opts"], d)
    else:
        entry.admm_enet(x, y, lam, nl, 0.01, True, True, alpha, prob["opts"], d)
    sol, n = d["solver"], x.shape[0]
    grid = np.asarray(lam, dtype=np.float64) if lam is not None else _lambda_grid(sol.lambda0, n, d["std"].scaleY, nl, 0.01)
    li = grid * n / np.float64(d["std"].scaleY)
    tr = np.asarray(d["trace"], dtype=np.float64)
    tr[:, 11] = [float(np.float32(li[int(r[0])])) for r in tr]
    types = d["types"]
    tr[:, 7] = [types[k + 1] if k + 1 < len(types) else 0 for k in range(len(tr))]       # kind of the x-update that FOLLOWS the decision
    cold = np.zeros((1, tr.shape[1])); cold[0, 8] = -1; cold[0, 7] = types[0]
    tr = np.vstack([cold, tr])
    st = np.vstack([np.zeros((1, len(d["state"][0])), np.float32), np.asarray(d["state"])])
    return prob, tr, st, float(sol.sprad)


@pytest.mark.parametrize("alpha", [None, 0.5])
def test_stepcheck_wide_is_clean_on_the_oracles_own_run(alpha):
    from oracle import stepcheck
    x, y = synth_lasso(60, 200, 6, seed=9)
    prob, tr, st, gamma = _oracle_wide_run(x, y, alpha=alpha)
    rep = stepcheck.check_wide(prob, tr, st, gamma, label="wide oracle")
    assert rep["records"] > 150 and rep["kinds"][1] > 10 and rep["kinds"][2] > 100 and rep["rho_changes"] > 0
    stepcheck.assert_stepwise_wide(rep, label="wide oracle", norm_tol=1e-5)          # the oracle's norms accumulate in float


def test_stepcheck_wide_catches_planted_defects():
    from oracle import stepcheck
    x, y = synth_lasso(60, 200, 6, seed=9)
    n, p = x.shape
    prob, tr, st, gamma = _oracle_wide_run(x, y)
    k = int(np.nonzero((tr[:, 7] == 2) & (np.arange(len(tr)) > 40))[0][0]) + 1       # an active-set iteration
    # (a) a fused multiply-add in the dual update y + rho r
    bad = st.copy()
    rho = np.float32(tr[k, 9])
    r = (bad[k, p:p + n] + bad[k, p + n:p + 2 * n]).astype(np.float32)
    bad[k, p + 2 * n:] = (bad[k - 1, p + 2 * n:].astype(np.float64) + np.float64(rho) * r.astype(np.float64)).astype(np.float32)
    rep = stepcheck.check_wide(prob, tr, bad, gamma, label="fma")
    assert rep["bit_mismatch"]
    # (b) a zero coordinate resurrected on an active-set step
    bad = st.copy()
    j = int(np.nonzero(bad[k - 1, :p] == 0)[0][0])
    bad[k, j] = 1e-3
    rep = stepcheck.check_wide(prob, tr, bad, gamma, label="zero")
    assert any("zero coordinate" in m[3] for m in rep["bit_mismatch"])
    # (c) a mat-vec that is off by 1e-4 relative: A x
    bad = st.copy()
    bad[k, p:p + n] *= np.float32(1.0001)
    rep = stepcheck.check_wide(prob, tr, bad, gamma, label="ax")
    assert rep["ax_ratio_max"] > 50
    with pytest.raises(AssertionError):
        stepcheck.assert_stepwise_wide(rep, norm_tol=1e-5)
    # (d) a rho adaptation that is not the rule's
    t2 = tr.copy()
    kk = int(np.nonzero(t2[1:, 10] != t2[1:, 9])[0][0]) + 1
    t2[kk, 10] = t2[kk, 9]
    with pytest.raises(AssertionError, match="rho after the decision"):
        stepcheck.check_wide(prob, t2, st, gamma, label="rho")
    # (e) a schedule that skips a regular step
    t3 = tr.copy()
    kk = int(np.nonzero(t3[1:, 7] == 1)[0][2]) + 1
    t3[kk, 7] = 2
    with pytest.raises(AssertionError, match="schedule"):
        stepcheck.check_wide(prob, t3, st, gamma, label="schedule")


def _oracle_dense_run(kind, x, y, opts, intercept=True):
    from oracle import entry
    d = {"trace": [], "state": []}
    if kind == "lad":
        entry.admm_lad(x, y, intercept, opts, d)
        dvec = d["solver"].Y
    else:
        entry.admm_bp(x, y, opts, d)
        dvec = d["solver"].cache_AAAb
    tr = np.asarray(d["trace"], dtype=np.float64)
    cold = np.zeros((1, tr.shape[1])); cold[0, 8] = -1
    dim = len(dvec)
    st = np.asarray(d["state"]).reshape(len(d["state"]), 5, dim)
    r0 = np.zeros((1, 5, dim)); r0[0, 0] = dvec
    return np.vstack([cold, tr]), np.vstack([r0, st])


@pytest.mark.parametrize("kind", ["lad", "bp"])
def test_stepcheck_dense_is_clean_on_the_oracles_own_run_and_catches_a_fused_update(kind):
    from oracle import entry, readme, stepcheck
    if kind == "lad":
        x, y = readme.lasso_data()
        opts, icpt = entry.LAD_OPTS, False
    else:
        x, y, _ = readme.bp_data()
        opts, icpt = entry.BP_OPTS, True
    tr, st = _oracle_dense_run(kind, x, y, opts, icpt)
    rep = stepcheck.check_dense(kind, x, y, opts, tr, st, intercept=icpt, label=kind)
    assert rep["records"] == len(tr) - 1 and rep["decisions_checked"] == rep["records"] and rep["records"] > 50
    stepcheck.assert_stepwise_dense(rep, label=kind)
    assert rep["x_vs_ref_max"] <= 1.0 + 1e-9                   # the oracle's x IS the reference route
    # a fused multiply-add in y = adj_y + rho r at one iteration (emulated in extended precision: one rounding instead of two)
    bad = st.copy()
    k = 20
    rho = tr[k, 9]
    xg, zg, ajy = bad[k, 0], bad[k, 1], bad[k, 4]
    r = xg - bad[0, 0] - zg if kind == "lad" else xg - zg
    bad[k, 2] = (np.longdouble(rho) * r.astype(np.longdouble) + ajy.astype(np.longdouble)).astype(np.float64)
    if not np.any(bad[k, 2] != st[k, 2]):                      # rho is still a power of two there (the product is exact): one entry off by one ulp
        bad[k, 2, 0] = np.nextafter(bad[k, 2, 0], np.inf)
    rep = stepcheck.check_dense(kind, x, y, opts, tr, bad, intercept=icpt, label=kind)
    assert any(m[2] == "y" for m in rep["bit_mismatch"])
