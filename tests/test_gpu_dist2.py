"""GPU, two PROCESSES on one device: the multi-rank code of the consensus solver (global-moment standardisation,
all-reduced X'y, one exchange per ADMM iteration between par_pack and par_z) running for real -- HIP kernels in both
ranks -- over the SHM and the PEER (hipIpc one-shot all-reduce) backends of admm_amd/csrc/comm.hip.

Replaces the OpenMP master/worker loop of /root/reference/src/PADMMBase.h:174-237, PADMMLasso.h:99-108 across
processes.  Checked against the single-process solver with the same K (same kernels, blocks in one process) and the
NumPy oracle."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from helpers import relerr

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run_ranks(backend, case, nranks=2, timeout=300, extra_env=None):
    with tempfile.TemporaryDirectory(prefix="admmdist") as wd:
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
        procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), backend, str(r), str(nranks), wd, case],
                                  env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(nranks)]
        outs = []
        try:
            for pr in procs:
                o, _ = pr.communicate(timeout=timeout)
                outs.append(o)
        finally:
            for pr in procs:
                if pr.poll() is None:
                    pr.kill()
        for r, pr in enumerate(procs):
            assert pr.returncode == 0, f"rank {r} failed:\n{outs[r][-3000:]}"
        res = [dict(np.load(os.path.join(wd, f"result.{r}.npz"))) for r in range(nranks)]
        for r in range(nranks):                                      # the ranks' iterate dumps, where a case wrote them
            sp = os.path.join(wd, f"state.{r}.npz")
            if os.path.exists(sp):
                res[r]["dump"] = dict(np.load(sp))
        return res


@pytest.mark.parametrize("backend", ["shm", "peer"])
@pytest.mark.parametrize("case", ["tallblocks", "wideblocks", "wideblocks_k2"])
def test_two_process_consensus_matches_single_process(backend, case):
    from admm_amd import admm_lasso
    from admm_amd._lib import check
    from oracle import entry
    sys.path.insert(0, HERE)
    from dist_worker import problem
    res = _run_ranks(backend, case)
    # every rank returns the full result, and the ranks agree bit for bit (identical summation order everywhere)
    assert np.array_equal(res[0]["beta"], res[1]["beta"]) and np.array_equal(res[0]["niter"], res[1]["niter"])
    x, y, K, kw = problem(case)
    m = admm_lasso(x, y).penalty(nlambda=kw["nlambda"]).opts(maxit=kw["maxit"])
    m.nthread = K
    lib, head, tail, lam_out, beta1, niter1, stats, keep = m._common()
    check(lib.admm_hip_parlasso(*head, K, *tail))
    # single process, same K: the only difference is where the column moments / X'y partial sums are added up
    assert np.allclose(res[0]["lam"], lam_out, rtol=1e-6)
    # (two executions with their own roundings: each is held to the oracle by the trace rule -- the single process in
    # tests/test_gpu_parlasso.py, the two processes below -- so no slack on iteration counts is needed between them)
    for j in range(kw["nlambda"]):
        assert relerr(res[0]["beta"][:, j], beta1[:, j]) < 2e-4, j
    # against the oracle on the decision trace (identical counts, every column 1e-4): the global moments are summed in a
    # different order than in one process, so this is a separate execution with its own near-ties
    from helpers import assert_followed_parity
    assert np.array_equal(res[0]["trace"], res[1]["trace"])
    prob = dict(x=x, y=y, lam=None, nlambda=kw["nlambda"], lmin_ratio=0.01 if x.shape[0] < x.shape[1] else 1e-4, standardize=True,
                intercept=True, opts=dict(entry.LASSO_OPTS, maxit=kw["maxit"]), alpha=None, nthread=K)
    assert_followed_parity(res[0]["beta"], res[0]["niter"], res[0]["trace"], prob, 1e-4, label=f"2-process consensus {case} over {backend}")


@pytest.mark.parametrize("backend,case", [("shm", "tallshard300"), ("peer", "tallshard300"), ("peer", "tallshard2300")])
def test_two_process_row_sharded_tall_solver(backend, case):
    """The serial tall solver with its x-update dealt out to two ranks (admm_hip_lasso_plan_create_dist, nthread = 0):
    split-K Gram / X'y over the ranks' row slices, replicated factorisation, per iteration each rank's share of the
    lower-triangle tiles + one all-reduce of 2p floats.  Judged like the single-GPU tall path: the oracle follows the
    decision trace through rounding-level near-ties only, counts identical, every column within 1e-4."""
    from oracle import entry
    from helpers import assert_tall_parity
    sys.path.insert(0, HERE)
    from dist_worker import problem
    res = _run_ranks(backend, case, timeout=600)
    assert np.array_equal(res[0]["beta"], res[1]["beta"]) and np.array_equal(res[0]["niter"], res[1]["niter"])
    assert np.array_equal(res[0]["trace"], res[1]["trace"])              # replicated decisions from identical numbers
    x, y, K, kw = problem(case)
    prob = dict(x=x, y=y, lam=None, nlambda=kw["nlambda"], lmin_ratio=1e-4, standardize=True, intercept=True,
                opts=entry.LASSO_OPTS, alpha=None)
    rep = assert_tall_parity(res[0]["beta"], res[0]["niter"], res[0]["trace"], prob, 1e-4, label=f"{case} over {backend}")
    assert len(rep["loose"]) == 0


@pytest.mark.parametrize("nranks,backend", [(2, "peer"), (4, "shm"), (4, "peer")])
def test_distributed_factorisation_is_bit_identical_to_the_replicated_one(nranks, backend):
    """SURVEY.md section 8f row n1, the multi-GPU half: in the row-sharded tall solver the blocked Cholesky + inverse runs with its
    128-column blocks dealt out to the ranks -- the owner factorises the diagonal block and the panel and broadcasts them, every
    rank updates the block columns it owns, and of the inverse U U' a rank forms only the tiles its share of the x-update reads.
    Every tile sees the same updates in the same order from bit-identical operands as in one process, so the WHOLE solve must be
    bit-identical to the run in which every rank factorises the full matrix (ADMM_HIP_DIST_FACTOR=0): coefficients, iteration
    counts, decision trace.  And the work must have been shared: the flops a rank reports are ~1 / nranks of one process's
    p^3 (2/3 for the factor + U, 1/3 for U U'); the ranks' shares add up to it.  (ADMM_HIP_INVERSE=f32: below p = 4096 the inverse is
    otherwise built in double, replicated.)"""
    env = {"ADMM_HIP_INVERSE": "f32"}
    a = _run_ranks(backend, "tallshard2300", nranks=nranks, timeout=600, extra_env=env)
    b = _run_ranks(backend, "tallshard2300", nranks=nranks, timeout=600, extra_env=dict(env, ADMM_HIP_DIST_FACTOR="0"))
    for r in range(nranks):
        assert np.array_equal(a[r]["beta"], b[r]["beta"]) and np.array_equal(a[r]["niter"], b[r]["niter"]) and np.array_equal(a[r]["trace"], b[r]["trace"]), r
        assert float(b[r]["factor_flops"]) == 0.0 and float(a[r]["factor_flops"]) > 0.0
    p = 2300
    one = float(p) ** 3                                      # 2 p^3 / 3 (factor + U) + p^3 / 3 (U U'), whole 128-blocks make it a little more
    shares = [float(a[r]["factor_flops"]) for r in range(nranks)]
    print(f"[dist factor] {nranks} ranks over {backend}: flops per rank / p^3 = {[round(s / one, 3) for s in shares]}, sum {sum(shares) / one:.3f}")
    assert max(shares) < 1.6 * one / nranks, shares          # shared out (block granularity and the overlapping tiles of U U' cost a little)
    assert 0.9 * one < sum(shares) < 1.5 * one, shares


@pytest.mark.parametrize("backend,case", [("shm", "widecols"), ("peer", "widecols"), ("peer", "widecols_enet")])
def test_two_process_column_sharded_wide_solver(backend, case):
    """The serial wide solver (ADMMLassoWide / ADMMEnetWide) with its COLUMNS dealt out to two ranks
    (admm_hip_lasso_dist_cols): local X_i't / prox / active set, replicated z / dual / decisions, one all-reduce of
    A x = sum_i X_i x_i per iteration.  Same algorithm, so it is held to the single-process solver and to the oracle."""
    from admm_amd import admm_enet, admm_lasso
    from oracle import entry
    sys.path.insert(0, HERE)
    from dist_worker import problem
    res = _run_ranks(backend, case)
    assert np.array_equal(res[0]["beta"], res[1]["beta"]) and np.array_equal(res[0]["niter"], res[1]["niter"])
    x, y, _, kw = problem(case)
    alpha = kw.get("alpha")
    nl = kw["nlambda"]
    if alpha is None:
        one = admm_lasso(x, y).penalty(nlambda=nl, lambda_min_ratio=0.01).fit()
        ref = entry.admm_lasso(x, y, None, nl, 0.01, True, True, entry.LASSO_OPTS)
    else:
        one = admm_enet(x, y).penalty(nlambda=nl, lambda_min_ratio=0.01, alpha=alpha).fit()
        ref = entry.admm_enet(x, y, None, nl, 0.01, True, True, alpha, entry.LASSO_OPTS)
    assert np.allclose(res[0]["lam"], one.lambda_, rtol=1e-6)
    # the sum over ranks of A x rounds differently from the single-process sum over workgroups, so this is a separate
    # execution with its own near-ties: judged against the oracle on its decision trace (identical counts, every column 1e-4)
    from helpers import assert_followed_parity
    assert np.array_equal(res[0]["trace"], res[1]["trace"])
    prob = dict(x=x, y=y, lam=None, nlambda=nl, lmin_ratio=0.01, standardize=True, intercept=True, opts=entry.LASSO_OPTS, alpha=alpha)
    assert_followed_parity(res[0]["beta"], res[0]["niter"], res[0]["trace"], prob, 1e-4, label=f"2-process {case} over {backend}")
    h = nl // 2
    for j in range(h):
        assert relerr(res[0]["beta"][:, j], one.beta_dense[:, j]) < 1e-4, j


@pytest.mark.parametrize("case,screen", [("widecols", ""), ("widecols_enet", ""), ("widecols", "16"), ("widecols_enet", "8")])
def test_two_process_column_sharded_stretch_is_stepwise_clean(case, screen):
    """Round 6: the column-sharded wide solver runs its active-set iterations inside the persistent stretch, A x summed over the ranks
    INSIDE the launch (wide_rows_persist_kernel<true>, AUX region of the PEER exchange).  Held by the stepwise instrument as the
    single-process stretch is: both ranks dump their iterates (x of their own columns | A x | z | y) and their standardised column
    blocks; put together they are one dump of the whole problem, and every iteration in it must be the reference's iteration applied to
    the library's own previous iterates -- zero pattern, z, y bit for bit, the two mat-vecs within the float dot-product yardstick,
    thresholds / residuals / decisions / rho adaptation exact (oracle/stepcheck.py check_wide).  And the replicated vectors must be
    bit-identical on the two ranks in every record.  screen = "1": with the regular steps screened on every rank's column block
    (wide_x_kernel's fp16 bound, forced on at this size): the regular steps in the dump are still the reference's, bit for bit in
    their zero pattern."""
    from oracle import entry, stepcheck
    sys.path.insert(0, HERE)
    from dist_worker import problem
    res = _run_ranks("peer", case, extra_env=dict(ADMM_TEST_WIDECOLS_STATE="1", **({"ADMM_HIP_WIDE_SCREEN": screen} if screen else {})))
    x, y, _, kw = problem(case)
    n, p = x.shape
    d0, d1 = res[0]["dump"], res[1]["dump"]
    assert int(d0["persist_iter"]) > 0 and int(d0["persist_iter"]) == int(d1["persist_iter"]), "the stretch ran, on both ranks alike"
    p0 = int(d0["hi"]) - int(d0["lo"])
    s0, s1 = d0["state"], d1["state"]
    assert s0.shape[0] == s1.shape[0] and s0.shape[1] == p0 + 3 * n
    assert np.array_equal(s0[:, p0:], s1[:, s1.shape[1] - 3 * n:]), "A x, z, y replicated bit for bit in every record"
    state = np.concatenate([s0[:, :p0], s1[:, :s1.shape[1] - 3 * n], s0[:, p0:]], axis=1)
    X = np.asfortranarray(np.concatenate([d0["X"], d1["X"]], axis=1))
    assert np.array_equal(d0["Y"], d1["Y"])
    assert np.array_equal(res[0]["trace"], res[1]["trace"])
    prob = dict(x=x, y=y, lam=None, nlambda=kw["nlambda"], lmin_ratio=0.01, standardize=True, intercept=True, opts=entry.LASSO_OPTS, alpha=kw.get("alpha"))
    rep = stepcheck.check_wide(prob, res[0]["trace"], state, float(d0["eig_est"]), X=X, Y=d0["Y"], label=f"2-process {case}")
    stepcheck.assert_stepwise_wide(rep, label=f"2-process {case}")
    print(f"[stepwise 2-process {case}] {rep['decisions_checked']} iterations replayed ({int(d0['persist_iter'])} of them inside the stretch; zero / regular / active-set "
          f"{rep['kinds'][0]} / {rep['kinds'][1]} / {rep['kinds'][2]}): z, y, zero pattern bit-exact; X't within {rep['xt_ratio_max']:.2f}, A x within {rep['ax_ratio_max']:.2f} "
          f"float-dot yardsticks")


@pytest.mark.parametrize("backend", ["shm", "peer"])
@pytest.mark.parametrize("downdate", ["0", "1"])
def test_two_process_cross_validation_folds_as_replicas(backend, downdate, monkeypatch):
    """admm_hip_lasso_cv with a communicator: fold f runs on rank f mod 2 (independent replicas, no exchange on the data
    path), the score tables are summed over the ranks at the end.  Every rank must return exactly what one process
    computing all five folds returns -- with the folds as direct fits and with the folds as down-dates of the full-data Gram
    (every rank forms the same base from the same data: still bit for bit)."""
    import admm_amd
    sys.path.insert(0, HERE)
    from dist_worker import problem
    if backend == "peer" and downdate == "1":
        pytest.skip("covered over shm (the modes differ in the setup only, not in the exchange)")
    res = _run_ranks(backend, "cv", extra_env={"ADMM_HIP_CV_DOWNDATE": downdate})
    x, y, _, kw = problem("cv")
    admm_amd.options.set(CV_DOWNDATE=downdate)       # this process: through the options of the calling thread (the ranks: the overlay of their environment)
    one = admm_amd.admm_lasso(np.asfortranarray(x), y).penalty(nlambda=kw["nlambda"]).cv(nfolds=5, keep_fold_beta=True)
    for r in res:
        assert np.array_equal(r["fold_mse"], one.fold_mse) and np.array_equal(r["fold_niter"], one.fold_niter)
        assert np.array_equal(r["fold_beta"], one.fold_beta)
        assert np.array_equal(r["cvm"], one.cvm) and np.array_equal(r["cvse"], one.cvse)
        assert list(r["idx"]) == [one.idx_min, one.idx_1se]
        assert np.array_equal(r["beta"], one.fit.beta_dense) and np.array_equal(r["lam"], one.lambda_)


def test_two_process_multi_response_as_replicas():
    """admm_hip_lasso_multi with a communicator: response j runs on rank j mod 2, outputs summed over the ranks at the end;
    every rank returns what separate single-process fits return."""
    import admm_amd
    sys.path.insert(0, HERE)
    from dist_worker import problem
    res = _run_ranks("shm", "multi")
    x, y, _, kw = problem("multi")
    rng = np.random.default_rng(9)
    Y = np.stack([y, y[::-1].copy(), rng.standard_normal(x.shape[0])], axis=1)
    for j in range(3):
        one = admm_amd.admm_lasso(np.asfortranarray(x), np.ascontiguousarray(Y[:, j])).penalty(nlambda=kw["nlambda"]).fit()
        for r in res:
            assert np.array_equal(r["beta"][j], one.beta_dense) and np.array_equal(r["niter"][j], one.niter)
            assert np.array_equal(r["lam"][j], one.lambda_)


def test_shm_bootstrap_ignores_a_stale_segment_of_the_same_name():
    """A segment of the job's name left behind by a crashed run -- sized right, `attached` already at nranks, every
    posted[] sequence number far ahead, slots full of garbage -- used to be attached to by any rank that opened the name
    before rank 0 had replaced it: that rank skipped every wait and summed stale slots (comm.hip, ShmHeader::token).  Now
    the header must carry the job's token.  The stale segment is planted BEFORE the ranks start, rank 0 is held back a
    little so that rank 1 certainly sees the stale one first, and the run must still produce the single-process result."""
    import mmap
    name = "/admm_hip_stale_%d" % os.getpid()
    path = "/dev/shm" + name
    total = 4096 + 2 * 2 * (4 << 20)
    with open(path, "wb") as f:
        f.truncate(total)
    with open(path, "r+b") as f:
        mm = mmap.mmap(f.fileno(), total)
        hdr = np.frombuffer(mm, dtype=np.uint64, count=64)
        hdr[:] = 1 << 40                                       # posted[r]: every exchange "already posted"
        np.frombuffer(mm, dtype=np.uint32, count=2, offset=512)[:] = (2, 0)      # attached = nranks, failed = 0
        np.frombuffer(mm, dtype=np.uint64, count=1, offset=520)[:] = 4 << 20     # slot
        np.frombuffer(mm, dtype=np.uint32, count=1, offset=528)[:] = 2           # nranks
        np.frombuffer(mm, dtype=np.float32, count=(total - 4096) // 4, offset=4096)[:] = 1e30
        del hdr
        mm.flush()
    try:
        res = _run_ranks("shm", "tallblocks", extra_env=dict(ADMM_TEST_SHM_NAME=name, ADMM_TEST_RANK0_DELAY_S="1.5"))
        ref = _run_ranks("shm", "tallblocks")
        assert np.array_equal(res[0]["beta"], res[1]["beta"])
        assert np.array_equal(res[0]["beta"], ref[0]["beta"]) and np.array_equal(res[0]["niter"], ref[0]["niter"])
        assert np.all(np.isfinite(res[0]["beta"]))
    finally:
        if os.path.exists(path):
            os.unlink(path)


@pytest.mark.parametrize("case", ["tallshard2300", "widecols", "wideblocks"])
def test_single_launch_exchange_falls_back_when_its_grid_would_not_be_resident(case):
    """The PEER exchange of the two sharded solvers normally runs producer and consumer in ONE launch (tall_tail_kernel
    <TAIL_PEER1>, wide_tail_kernel<2>): every workgroup publishes its share, counts itself in and then WAITS for the flags,
    which need all workgroups of that launch on every rank -- correct only if the whole grid is resident at once, so the
    host chooses it only when the grid fits into half of what the device holds (lasso_tall.hip / lasso_wide.hip,
    resident_workgroups).  Here that condition is VIOLATED on purpose (ADMM_HIP_TEST_RESIDENT_WGS=4: a device that holds four
    workgroups): the solvers must take the two-launch form (exchange_variant 2, not 3) -- no hang -- and return the very
    same bits."""
    one = _run_ranks("peer", case)
    # (wideblocks, round 6: the consensus solver's `pack` + `z` as one launch, par_z_kernel<1, true>; p = 300 is two workgroups: a device that holds one)
    two = _run_ranks("peer", case, extra_env=dict(ADMM_HIP_TEST_RESIDENT_WGS="1" if case == "wideblocks" else "4"))
    assert int(one[0]["exchange_variant"]) == 3 and int(two[0]["exchange_variant"]) == 2, (one[0]["exchange_variant"], two[0]["exchange_variant"])
    for r in range(2):
        assert np.array_equal(one[r]["beta"], two[r]["beta"]) and np.array_equal(one[r]["niter"], two[r]["niter"])
        assert np.array_equal(one[r]["trace"], two[r]["trace"])


@pytest.mark.parametrize("backend,nranks", [("shm", 2), ("peer", 2), ("shm", 4), ("peer", 4)])
def test_two_process_column_block_sharing_basis_pursuit(backend, nranks):
    """admm_hip_parbp_dist: the column blocks of the sharing solver (admm_amd/csrc/sharing_bp.hip; the reference's unbuilt
    PADMMBP.h) dealt out to two ranks -- local x-updates and block sums, ONE all-reduce of S = sum_i A_i x_i and two block norms per
    iteration, replicated r / y / decisions.  Held to the single-process solver (the same blocks in one process: only the order
    in which the blocks' A_i x_i are added differs) and to the oracle."""
    import admm_amd
    from oracle import entry
    sys.path.insert(0, HERE)
    from dist_worker import problem
    res = _run_ranks(backend, "parbp", nranks=nranks)
    x, y, _, kw = problem("parbp")
    N = kw["nthread"]
    assert all(r["niter"][0] == res[0]["niter"][0] and r["rho"][0] == res[0]["rho"][0] for r in res)
    assert int(res[0]["exchange_variant"]) == 1
    if nranks == 2:
        assert list(res[0]["lo"]) == [0, 400] and list(res[1]["lo"]) == [400, 803]      # two blocks of 200 | 200 + 203
    else:
        assert [list(r["lo"]) for r in res] == [[0, 200], [200, 400], [400, 600], [600, 803]]     # one block per rank, the last takes the remainder
    beta = np.concatenate([r["beta"] for r in res])
    one = admm_amd.admm_bp(x, y).parallel(N).fit()
    ref = entry.admm_parbp(x, y, N, dict(entry.BP_OPTS, rho_ratio=1.0))
    b1 = one.beta.toarray().ravel()
    print(f"[{nranks}-process parbp over {backend}] {int(res[0]['niter'][0])} iterations (one process {one.niter}, oracle {ref['niter']}); "
          f"beta vs one process {np.abs(beta - b1).max():.1e}, vs oracle {np.abs(beta - ref['beta']).max():.1e}")
    assert int(res[0]["niter"][0]) == one.niter == ref["niter"]
    assert np.abs(beta - b1).max() < 1e-10 and np.abs(beta - ref["beta"]).max() < 1e-10
    assert np.array_equal(beta != 0, ref["beta"] != 0)
