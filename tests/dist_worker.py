"""One rank of a multi-process run on ONE GPU (spawned by tests/test_gpu_dist2.py): attaches the SHM or PEER exchange
backend, checks a raw all-reduce, then runs the distributed consensus Lasso on its row slice and stores the result.

    python tests/dist_worker.py <backend shm|peer> <rank> <nranks> <workdir> <case>

No torch here: a plain ctypes caller of libadmm_hip.so, like the R shim would be.  Handles / barriers go through files."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def file_allgather(workdir, tag, rank, nranks, payload, timeout=60.0):
    path = os.path.join(workdir, f"{tag}.{rank}")
    with open(path + ".tmp", "wb") as f:
        f.write(payload.tobytes())
    os.rename(path + ".tmp", path)
    out, t0 = [], time.time()
    for r in range(nranks):
        pr = os.path.join(workdir, f"{tag}.{r}")
        while not os.path.exists(pr):
            if time.time() - t0 > timeout:
                raise TimeoutError(pr)
            time.sleep(0.005)
        out.append(np.fromfile(pr, dtype=payload.dtype))
    return np.concatenate(out)


def barrier(workdir, tag, rank, nranks):
    file_allgather(workdir, "bar_" + tag, rank, nranks, np.zeros(1, np.uint8))


def problem(case):
    """(x, y, K, kwargs) -- the same arrays on every rank (seeded)."""
    from helpers import synth_lasso
    if case == "tallblocks":          # both row blocks tall: Cholesky branch (PADMMLasso.h:23-24)
        x, y = synth_lasso(900, 120, 10, seed=61)
        return x, y, 2, dict(nlambda=6, maxit=400)
    if case == "wideblocks":          # 403 rows in 4 blocks over 2 ranks: Woodbury branch (:25-30), remainder block on the last rank
        x, y = synth_lasso(403, 300, 12, seed=62)
        return x, y, 4, dict(nlambda=4, maxit=300)
    if case == "wideblocks_k2":       # ONE Woodbury block per rank (what K = N GPUs runs: the unbatched launches of the one-pass form)
        x, y = synth_lasso(301, 420, 12, seed=63)
        return x, y, 2, dict(nlambda=4, maxit=300)
    if case == "tallshard300":        # the serial tall solver with its x-update spread over the ranks (small p: one tile row)
        x, y = synth_lasso(2000, 300, 30, seed=7)
        return x, y, 0, dict(nlambda=12)
    if case == "tallshard2300":       # ~90 lower-triangle tiles dealt out to the ranks
        x, y = synth_lasso(4700, 2300, 40, seed=2300)
        return x, y, 0, dict(nlambda=6)
    if case == "tallshard6000":       # scripts/dist_factor_table.py: a size at which the factorisation dominates the setup
        x, y = synth_lasso(6400, 6000, 40, seed=6000)
        return x, y, 0, dict(nlambda=3)
    if case == "widecols":            # the serial wide solver with its columns spread over the ranks (fused x-update, n <= 4096)
        x, y = synth_lasso(300, 2000, 20, seed=29)
        return x, y, -1, dict(nlambda=10)
    if case == "widecols_enet":
        x, y = synth_lasso(250, 900, 12, seed=31)
        return x, y, -1, dict(nlambda=8, alpha=0.5)
    if case == "cv":                  # K-fold cross-validation with the folds dealt out to the ranks (fold f on rank f mod nranks)
        x, y = synth_lasso(500, 60, 8, seed=77)
        return x, y, -2, dict(nlambda=8)
    if case == "multi":               # three responses of one design matrix dealt out to the ranks (response j on rank j mod nranks)
        x, y = synth_lasso(400, 50, 8, seed=78)
        return x, y, -3, dict(nlambda=7)
    if case == "parbp":               # column-block sharing basis pursuit, 4 blocks over 2 ranks (the last block takes the remainder)
        rng = np.random.default_rng(81)
        n, p = 200, 803
        x = np.asfortranarray(rng.standard_normal((n, p)))
        b = np.zeros(p); b[rng.choice(p, 18, replace=False)] = rng.standard_normal(18) * 2
        return x, x @ b, -4, dict(nthread=4)
    raise SystemExit("unknown case " + case)


def main():
    backend, rank, nranks, workdir, case = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    from admm_amd import _lib, dist as adist
    lib = _lib.load()
    assert lib.admm_hip_set_device(0) == 0
    if backend == "shm":
        # the job token travels over the caller's channel (here: files), drawn by rank 0
        mine = np.frombuffer(os.urandom(8), dtype=np.uint64) | np.uint64(1)
        token = int(file_allgather(workdir, "tok", rank, nranks, mine)[0])
        if rank == 0 and os.environ.get("ADMM_TEST_RANK0_DELAY_S"):      # let the others meet a planted stale segment first
            time.sleep(float(os.environ["ADMM_TEST_RANK0_DELAY_S"]))
        adist.init_comm_shm(nranks, rank, os.environ.get("ADMM_TEST_SHM_NAME", "/admm_hip_test_" + os.path.basename(workdir)), token)
    else:
        adist.init_comm_peer(nranks, rank, lambda mine: file_allgather(workdir, "ipc", rank, nranks, mine))
    # ---- raw exchange: p floats + 3 doubles (the consensus payload), then a long message that needs chunking
    rng = np.random.default_rng(1000 + rank)
    f = rng.standard_normal(100003).astype(np.float32)
    d = rng.standard_normal(3)
    f0, d0 = f.copy(), d.copy()
    adist.allreduce_host(f, d)
    allf = file_allgather(workdir, "f", rank, nranks, f0).reshape(nranks, -1)
    alld = file_allgather(workdir, "d", rank, nranks, d0).reshape(nranks, -1)
    ef, ed = allf[0].copy(), alld[0].copy()
    for r in range(1, nranks):
        ef += allf[r]; ed += alld[r]
    assert np.array_equal(f, ef) and np.array_equal(d, ed), "all-reduce differs from the rank-ordered sum"
    for rep in range(5):                                   # several in a row: both parities of the slots, flag reuse
        g = np.full(2500000, float(rank + 1 + rep), dtype=np.float32)     # 10 MB > one 4 MB slot
        adist.allreduce_host(g, None)
        assert np.all(g == sum(range(1, nranks + 1)) + nranks * rep), (rep, g[:3])
    # ---- raw reduce-scatter (the split-K Gram of the row-sharded tall solver): chunk q of every rank's buffer summed at rank q,
    # in rank order -- two sizes, the second one longer than a slot (chunked)
    for cnt in (1000, 1500000):
        snd = [np.random.default_rng(7000 + 10 * r + cnt % 7).standard_normal(nranks * cnt).astype(np.float32) for r in range(nranks)]
        got = adist.reduce_scatter_host(snd[rank], cnt)
        want = snd[0][rank * cnt:(rank + 1) * cnt].copy()
        for r in range(1, nranks):
            want += snd[r][rank * cnt:(rank + 1) * cnt]
        assert np.array_equal(got, want), ("reduce-scatter differs from the rank-ordered sum", cnt, float(np.abs(got - want).max()))
    # ---- the distributed consensus solver on this rank's row slice
    x, y, K, kw = problem(case)
    n, p = x.shape
    if K == -4:
        lo, hi = adist.parbp_partition(p, kw["nthread"], nranks, rank)
        beta, niter, st = adist.parbp_dist(np.asfortranarray(x[:, lo:hi]), y, p, lo, kw["nthread"])
        np.savez(os.path.join(workdir, f"result.{rank}.npz"), beta=beta, niter=np.array([niter]), lo=np.array([lo, hi]), rho=np.array([st["rho"]]),
                 exchange_variant=int(st["exchange_variant"]))
        barrier(workdir, "end", rank, nranks)
        adist.finalize_comm()
        print("rank", rank, "ok", flush=True)
        return
    if K == -3:
        import admm_amd
        rng = np.random.default_rng(9)
        Y = np.stack([y, y[::-1].copy(), rng.standard_normal(n)], axis=1)
        fits = admm_amd.admm_lasso(np.asfortranarray(x), y).penalty(nlambda=kw["nlambda"]).fit_responses(Y)
        np.savez(os.path.join(workdir, f"result.{rank}.npz"), beta=np.stack([f.beta_dense for f in fits]),
                 niter=np.stack([f.niter for f in fits]), lam=np.stack([f.lambda_ for f in fits]))
        barrier(workdir, "end", rank, nranks)
        adist.finalize_comm()
        print("rank", rank, "ok", flush=True)
        return
    if K == -2:
        import admm_amd
        cv = admm_amd.admm_lasso(np.asfortranarray(x), y).penalty(nlambda=kw["nlambda"]).cv(nfolds=5, keep_fold_beta=True)
        np.savez(os.path.join(workdir, f"result.{rank}.npz"), cvm=cv.cvm, cvse=cv.cvse, fold_mse=cv.fold_mse, fold_niter=cv.fold_niter,
                 fold_beta=cv.fold_beta, idx=np.array([cv.idx_min, cv.idx_1se]), beta=cv.fit.beta_dense, niter=cv.fit.niter, lam=cv.lambda_)
        barrier(workdir, "end", rank, nranks)
        adist.finalize_comm()
        print("rank", rank, "ok", flush=True)
        return
    if K < 0:
        lo, hi = adist.col_partition(p, nranks, rank)
        plan = adist.DistColsPlan(np.asfortranarray(x[:, lo:hi]), y, p, lo, lambda_min_ratio=0.01, **kw)
        plan.enable_trace(1 << 16)
        dump = os.environ.get("ADMM_TEST_WIDECOLS_STATE") == "1"       # the iterate dump of this rank (x of its columns | A x | z | y) + its standardised data
        if dump:
            plan.enable_state(1 << 13)
        fit = plan.run()
        trace = plan.read_trace()
        if dump:
            st = plan.read_state()
            Xl, Yl = plan.read_data(n, hi - lo)
            np.savez(os.path.join(workdir, f"state.{rank}.npz"), state=st, X=Xl, Y=Yl, lo=lo, hi=hi, persist_iter=int(fit.stats.get("persist_iter", 0)),
                     eig_est=float(fit.stats["eig_est"]))
        plan.close()
        assert fit.stats["branch"] == 1
        one = adist.lasso_dist_cols(np.asfortranarray(x[:, lo:hi]), y, p, lo, lambda_min_ratio=0.01, **kw)       # the one-shot entry point
        assert np.array_equal(one.beta_dense, fit.beta_dense) and list(one.niter) == list(fit.niter)
    elif K > 0:
        lo, hi = adist.row_partition(n, K, nranks, rank)
        plan = adist.DistLassoPlan(np.asfortranarray(x[lo:hi]), y[lo:hi], n, p, K, n_local=hi - lo, **kw)
        plan.enable_trace(1 << 16)
        fit = plan.run()
        trace = plan.read_trace()
        plan.close()
        one = adist.parlasso_dist(np.asfortranarray(x[lo:hi]), y[lo:hi], n, p, K, n_local=hi - lo, **kw)     # the one-shot entry point
        assert np.array_equal(one.beta_dense, fit.beta_dense) and list(one.niter) == list(fit.niter)
    else:
        cut = [0] + [int(n * (r + 1) / nranks) + (17 if r < nranks - 1 else 0) for r in range(nranks)]     # uneven slices on purpose
        lo, hi = cut[rank], cut[rank + 1]
        plan = adist.DistLassoPlan(np.asfortranarray(x[lo:hi]), y[lo:hi], n, p, 0, lambda_min_ratio=1e-4, n_local=hi - lo, **kw)
        plan.enable_trace(1 << 16)
        fit = plan.run()
        trace = plan.read_trace()
        assert fit.stats["xupdate_variant"] == 2
        plan.close()
    np.savez(os.path.join(workdir, f"result.{rank}.npz"), beta=fit.beta_dense, niter=fit.niter, lam=fit.lambda_, trace=trace,
             exchange_variant=int(fit.stats.get("exchange_variant", 0)), factor_flops=float(fit.stats.get("factor_flops", 0.0)),
             t_factor=float(fit.stats.get("t_factor", 0.0)))
    barrier(workdir, "end", rank, nranks)                  # nobody unmaps while a peer may still push
    adist.finalize_comm()
    print("rank", rank, "ok", flush=True)


if __name__ == "__main__":
    main()
