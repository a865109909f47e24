"""world_size-2 (gloo, CPU) check of the multi-process consensus protocol.

No GPU here, so the HIP kernels cannot run; what this covers is everything around them that only
exists when N > 1: the reference row partition dealt out over ranks (`admm_amd.dist.row_partition`),
standardisation from all-reduced global column moments, lambda_0 from an all-reduced X'y, and the
per-iteration exchange used by padmm_lasso.hip -- ONE all-reduce of [consensus sum (p floats), three
worker-summed squared norms] between `pack` and `z`, with the convergence decision of iteration g-1
taken after the all-reduce of iteration g.  Each rank runs a NumPy model of its device-side steps
(built from the oracle's pieces); the result must equal the serial oracle of
PADMMLasso_Master::solve (oracle/solvers.py) in coefficients and (to +-2) iteration counts.
"""
import os
import socket

import numpy as np
import pytest
import scipy.linalg as sla
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import relerr, synth_lasso

F = np.float32


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _allreduce(a):
    t = torch.from_numpy(np.ascontiguousarray(a))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.numpy()


def _rank_main(rank, world, port, x, y, K, lam_user, maxit, out_path):
    from admm_amd.dist import row_partition
    from oracle.solvers import _soft_d, _sqnorm
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, p = x.shape
    lo, hi = row_partition(n, K, world, rank)
    Xl = np.array(x[lo:hi], dtype=F, order="F")
    yl = np.array(y[lo:hi], dtype=F)
    # --- global standardisation (flag 3) from all-reduced moments (prep.hip: colstat passes + all-reduce)
    s = _allreduce(np.concatenate([Xl.sum(axis=0, dtype=np.float64), [yl.sum(dtype=np.float64)]]))
    mean = (s / n).astype(F)
    Xl -= mean[None, :p]
    yl -= mean[p]
    ss = _allreduce(np.concatenate([(Xl.astype(np.float64) ** 2).sum(axis=0), [(yl.astype(np.float64) ** 2).sum()]]))
    n_invsqrt = F(1.0 / np.sqrt(F(n)))
    scale = (np.sqrt(ss).astype(F) * n_invsqrt).astype(F)
    Xl *= (F(1.0) / scale[:p])[None, :]
    yl /= scale[p]
    scaleY, meanY = scale[p], mean[p]
    # --- lambda_0 is not needed for a user lambda; internal lambda (ParLasso.cpp:91) and rho (PADMMLasso.h:199-200)
    lam_int = [lu * n / np.float64(scaleY) for lu in lam_user]
    rho = lam_int[0] / K
    # --- local workers (reference row blocks owned by this rank)
    Kl = K // world
    chunk = n // K
    A, Ab, chol = [], [], []
    for k in range(Kl):
        a0 = k * chunk
        a1 = (k + 1) * chunk if (k < Kl - 1 or rank < world - 1) else Xl.shape[0]
        Ak = np.ascontiguousarray(Xl[a0:a1])
        A.append(Ak)
        Ab.append((Ak.T @ yl[a0:a1]).astype(F))
        AA = (Ak.T @ Ak if Ak.shape[0] >= p else Ak @ Ak.T).astype(F)
        AA[np.arange(AA.shape[0]), np.arange(AA.shape[0])] += F(rho)
        chol.append(sla.cho_factor(AA, lower=True, check_finite=False))
    xs = [np.zeros(p, F) for _ in range(Kl)]
    ys = [np.zeros(p, F) for _ in range(Kl)]
    z = np.zeros(p, F)
    nsum = np.zeros(5)                 # local: sum_k|x_k|^2, sum_k|y_k|^2, sum_k|x_k - z|^2 ; global: |z|^2, |dz|^2
    first, it, li, lam = True, 0, 0, lam_int[0]
    eps_p = eps_d = 0.0
    betas, niters = [], []
    eps_abs = eps_rel = 1e-5
    for g in range(100000):
        # head + x-update + pack
        wsum = np.zeros(p, F)
        for k in range(Kl):
            rhs = (Ab[k] - ys[k]).astype(F)
            rhs = (rhs.astype(np.float64) + rho * z.astype(np.float64)).astype(F)
            if A[k].shape[0] >= p:
                xs[k] = sla.cho_solve(chol[k], rhs, check_finite=False).astype(F)
            else:
                t = (A[k] @ rhs).astype(F)
                sv = sla.cho_solve(chol[k], t, check_finite=False).astype(F)
                xs[k] = ((rhs - (A[k].T @ sv).astype(F)) / F(rho)).astype(F)
            wsum = (wsum + (xs[k] + ys[k] / F(rho))).astype(F)
        # the exchange: p floats + 3 doubles, one all-reduce per iteration
        payload = _allreduce(np.concatenate([wsum.astype(np.float64), nsum[:3]]))
        wsum_g = payload[:p].astype(F)
        x2, y2, r2 = payload[p:]
        z2, dz2 = nsum[3], nsum[4]
        # z: decision for iteration g-1, identical on every rank
        fin = False
        if not first:
            rp, rd = np.sqrt(r2), rho * np.sqrt(K * dz2)
            if rp < eps_p and rd < eps_d:
                fin, nit = True, it + 1
            else:
                it += 1
                if it >= maxit:
                    fin, nit = True, maxit + 1
            if fin:
                betas.append(z.copy())
                niters.append(nit)
                li += 1
                it = 0
                if li >= len(lam_int):
                    break
                lam = lam_int[li]
        first = False
        eps_p = max(np.sqrt(x2), np.sqrt(z2) * np.sqrt(K)) * eps_rel + np.sqrt(float(p * K)) * eps_abs
        eps_d = np.sqrt(y2) * eps_rel + np.sqrt(float(p * K)) * eps_abs
        zn = _soft_d((wsum_g / F(K)).astype(F), lam / (rho * K), F)
        acc = np.zeros(5)
        for k in range(Kl):
            r = (xs[k] - zn).astype(F)
            ys[k] = (ys[k] + F(rho) * r).astype(F)
            acc[0] += np.float64(_sqnorm(xs[k], F)); acc[1] += np.float64(_sqnorm(ys[k], F)); acc[2] += np.float64(_sqnorm(r, F))
        acc[3] = np.float64(_sqnorm(zn, F)); acc[4] = np.float64(_sqnorm(zn - z, F))
        z, nsum = zn, acc
    # recover (DataStd flag 3) with the GLOBAL moments
    out = []
    for b in betas:
        coef = (b / scale[:p] * scaleY).astype(F)
        b0 = F(meanY - F((coef * mean[:p]).sum(dtype=F)))
        out.append(np.concatenate([[b0], coef]))
    if rank == 0:
        np.savez(out_path, beta=np.array(out).T, niter=np.array(niters))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,p,K", [(400, 30, 2), (403, 40, 4), (120, 200, 2)])
def test_two_rank_consensus_protocol_matches_serial_oracle(tmp_path, n, p, K):
    from oracle import entry
    x, y = synth_lasso(n, p, max(3, p // 8), seed=53)
    lam = [0.5, 0.15]
    maxit = 4000
    out_path = str(tmp_path / "dist.npz")
    mp.spawn(_rank_main, args=(2, _free_port(), x, y, K, lam, maxit, out_path), nprocs=2, join=True)
    got = np.load(out_path)
    ref = entry.admm_parlasso(x, y, lam, 100, 1e-4, True, True, K, dict(entry.LASSO_OPTS, maxit=maxit))
    # the global moments are all-reduced in double while the serial oracle averages in float32: trajectories
    # agree to rounding, so the counts may differ by an iteration out of several hundred
    assert np.abs(got["niter"].astype(int) - ref["niter"].astype(int)).max() <= 2
    for j in range(len(lam)):
        assert relerr(got["beta"][:, j], ref["beta"][:, j]) < 5e-5, j


def test_row_partition_covers_all_rows():
    from admm_amd.dist import row_partition
    for n, K, W in [(403, 4, 2), (1000, 8, 8), (17, 2, 1), (1201, 6, 3)]:
        cover = []
        for r in range(W):
            lo, hi = row_partition(n, K, W, r)
            cover.extend(range(lo, hi))
        assert cover == list(range(n))
    with pytest.raises(ValueError):
        row_partition(100, 3, 2, 0)


# ------------------------------------------------------------------------------------------------------------------
# Row-sharded serial tall solver (admm_hip_lasso_dist): world_size-2 NumPy model of what lasso_tall.hip does per rank.
def _tall_rank_main(rank, world, port, x, y, nl, out_path):
    from admm_amd.dist import symv_tiles
    from oracle.entry import _lambda_grid
    from oracle.solvers import LassoTall
    from oracle.spectra import sym_eigs_largest
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, p = x.shape
    cut = [0] + [n * (r + 1) // world + (13 if r < world - 1 else 0) for r in range(world)]      # uneven row slices
    lo, hi = cut[rank], cut[rank + 1]
    Xl = np.array(x[lo:hi], dtype=F, order="F")
    yl = np.array(y[lo:hi], dtype=F)
    # global-moment standardisation (prep.hip), X'y and the Gram matrix as split-K sums over the ranks (lasso_tall.hip)
    s = _allreduce(np.concatenate([Xl.sum(axis=0, dtype=np.float64), [yl.sum(dtype=np.float64)]]))
    mean = (s / n).astype(F)
    Xl -= mean[None, :p]
    yl -= mean[p]
    ss = _allreduce(np.concatenate([(Xl.astype(np.float64) ** 2).sum(axis=0), [(yl.astype(np.float64) ** 2).sum()]]))
    scale = (np.sqrt(ss).astype(F) * F(1.0 / np.sqrt(F(n)))).astype(F)
    Xl *= (F(1.0) / scale[:p])[None, :]
    yl /= scale[p]
    XY = _allreduce((Xl.T @ yl).astype(F))
    G = _allreduce((Xl.T @ Xl).astype(F))
    tiles = symv_tiles(p, rank, world, 64, 32)            # small tiles so that a p of a few hundred has many of them

    class Sharded(LassoTall):
        def __init__(self):
            self.p, self.eps_abs, self.eps_rel, self.alpha, self.info = p, 1e-5, 1e-5, None, {}
            self.XY = XY
            self.lambda0 = F(np.abs(XY).max())

        def init(self, lam, rho):
            self.main_x = np.zeros(p, F); self.aux_z = np.zeros(p, F); self.dual_y = np.zeros(p, F)
            self.adj_z = np.zeros(p, F); self.adj_y = np.zeros(p, F)
            self.lam = F(lam)
            ev = sym_eigs_largest(lambda v: G @ v, p, 3, 10, 0.1, F, self.info)          # replicated: identical on every rank
            self.rho = float(np.float64(ev) ** (1.0 / 3) * np.float64(self.lam) ** (2.0 / 3))
            A = G.copy()
            A[np.arange(p), np.arange(p)] += F(self.rho)
            self.Minv = np.linalg.inv(A.astype(np.float64)).astype(F)
            self.eps_primal = self.eps_dual = 0.0
            self.resid_primal = self.resid_dual = 9999.0
            self._init_accel()

        def next_x(self):
            rhs = (self.XY - self.adj_y).astype(F)
            rhs = (rhs.astype(np.float64) + self.rho * self.adj_z.astype(np.float64)).astype(F)
            part = np.zeros(p, F)
            for rb, cb in tiles:                            # this rank's tiles: both halves of the symmetric product
                r0, r1, c0, c1 = rb * 64, min(p, rb * 64 + 64), cb * 32, min(p, cb * 32 + 32)
                blk = self.Minv[r0:r1, c0:c1].copy()
                ii, jj = np.meshgrid(np.arange(r0, r1), np.arange(c0, c1), indexing="ij")
                blk[ii < jj] = 0                            # only entries on / below the diagonal are read
                part[c0:c1] += (blk.T @ rhs[r0:r1]).astype(F)                            # dot half (diagonal included)
                blk[ii == jj] = 0
                part[r0:r1] += (blk @ rhs[c0:c1]).astype(F)                              # axpy half
            return _allreduce(part)                         # ONE all-reduce of p floats (2p with both candidates on the device)

    sol = Sharded()
    lam = _lambda_grid(sol.lambda0, n, scale[p], nl, 1e-4)
    trace, betas, niters = [], [], []
    sol.trace = trace
    for i in range(nl):
        sol.lam_idx = i
        li = lam[i] * n / np.float64(scale[p])
        sol.init(li, -1.0) if i == 0 else sol.init_warm(li)
        niters.append(sol.solve(10000))
        coef = (sol.aux_z / scale[:p] * scale[p]).astype(F)
        betas.append(np.concatenate([[F(mean[p] - F((coef * mean[:p]).sum(dtype=F)))], coef]))
    np.savez(out_path + f".{rank}.npz", beta=np.array(betas).T, niter=np.array(niters), trace=np.array(trace), lam=lam)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_row_sharded_tall_protocol_matches_serial_oracle(tmp_path):
    """Tile shares cover the triangle exactly once, the split-K sums reproduce the global Gram / X'y, both ranks take
    identical decisions, and the result is the serial oracle's (judged like the GPU tall path: follow mode, 1e-4)."""
    from admm_amd.dist import symv_tiles
    from helpers import assert_tall_parity
    from oracle import entry
    for pp in (100, 300, 1000):
        full = symv_tiles(pp)
        shares = [symv_tiles(pp, r, 3) for r in range(3)]
        assert sorted(sum(shares, [])) == sorted(full) and len(set(full)) == len(full)
        assert max(len(s) for s in shares) - min(len(s) for s in shares) <= 1
    x, y = synth_lasso(1200, 150, 12, seed=71)
    out = str(tmp_path / "tall")
    mp.spawn(_tall_rank_main, args=(2, _free_port(), x, y, 10, out), nprocs=2, join=True)
    r0, r1 = np.load(out + ".0.npz"), np.load(out + ".1.npz")
    assert np.array_equal(r0["beta"], r1["beta"]) and np.array_equal(r0["trace"], r1["trace"])
    prob = dict(x=x, y=y, lam=None, nlambda=10, lmin_ratio=1e-4, standardize=True, intercept=True, opts=entry.LASSO_OPTS, alpha=None)
    rep = assert_tall_parity(r0["beta"], r0["niter"], r0["trace"], prob, 1e-4, label="gloo model of the row-sharded tall solver")
    assert len(rep["loose"]) == 0


# ------------------------------------------------------------------------------------------------------------------
# Column-sharded serial wide solver (admm_hip_lasso_dist_cols): world_size-2 NumPy model of what lasso_wide.hip does.
def _wide_rank_main(rank, world, port, x, y, nl, out_path):
    from admm_amd.dist import col_partition
    from oracle.datastd import DataStd
    from oracle.entry import _lambda_grid
    from oracle.solvers import LassoWide
    from oracle.spectra import sym_eigs_largest
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, p = x.shape
    lo, hi = col_partition(p, world, rank)
    Xl = np.array(x[:, lo:hi], dtype=F, order="F")
    yl = np.array(y, dtype=F)
    std = DataStd(n, hi - lo, True, True, F)                 # column moments are local, y is replicated
    std.standardize(Xl, yl)

    class Cols(LassoWide):
        def __init__(self):
            self.X, self.Y, self.n, self.p = Xl, yl, n, hi - lo
            self.eps_abs = self.eps_rel = 1e-5
            self.alpha, self.info, self.trace_nnz = None, {}, []
            lam0 = np.zeros(world, F)
            lam0[rank] = np.abs((Xl.T @ yl).astype(F)).max()
            self.lambda0 = F(_allreduce(lam0).max())                         # max over ranks through a sum of one-hot vectors
            XXt = _allreduce((Xl @ Xl.T).astype(F))                          # X X' = sum over the column blocks
            self.sprad = F(sym_eigs_largest(lambda v: XXt @ v, n, 3, 10, 0.1, F, self.info))

        def compute_eps_dual(self):                                         # sqrt(p) counts ALL columns
            return (np.float64(F(np.sqrt(self.sprad))) * np.float64(F(np.linalg.norm(self.dual_y))) * self.eps_rel
                    + np.sqrt(float(p)) * self.eps_abs)

        def next_z(self):                                                   # the one exchange: A x summed over the ranks
            idx = np.nonzero(self.main_x)[0]
            part = (self.X[:, idx] @ self.main_x[idx]).astype(F) if idx.size else np.zeros(n, F)
            self.cache_Ax = _allreduce(part)
            return ((self.Y + self.dual_y + F(self.rho) * self.cache_Ax) / F(-1 - self.rho)).astype(F)

    sol = Cols()
    lam = _lambda_grid(sol.lambda0, n, std.scaleY, nl, 0.01)
    beta = np.zeros((p + 1, nl), F)
    niters = []
    for i in range(nl):
        li = lam[i] * n / np.float64(std.scaleY)
        sol.init(li, -1.0) if i == 0 else sol.init_warm(li)
        niters.append(sol.solve(10000))
        b0, coef = std.recover(sol.get_coef())
        full = np.zeros(p + 1, np.float64)
        full[1 + lo:1 + hi] = coef
        full[0] = np.float64(std.meanY) - np.float64(b0)                    # this block's share of sum_j beta_j meanX_j
        full = _allreduce(full)
        full[0] = np.float64(std.meanY) - full[0]
        beta[:, i] = full.astype(F)
    np.savez(out_path + f".{rank}.npz", beta=beta, niter=np.array(niters), lam=lam)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_column_sharded_wide_protocol_matches_serial_oracle(tmp_path):
    from oracle import entry
    x, y = synth_lasso(120, 500, 10, seed=83)
    out = str(tmp_path / "wide")
    mp.spawn(_wide_rank_main, args=(2, _free_port(), x, y, 8, out), nprocs=2, join=True)
    r0, r1 = np.load(out + ".0.npz"), np.load(out + ".1.npz")
    assert np.array_equal(r0["beta"], r1["beta"]) and np.array_equal(r0["niter"], r1["niter"])
    ref = entry.admm_lasso(x, y, None, 8, 0.01, True, True, entry.LASSO_OPTS)
    assert np.allclose(r0["lam"], ref["lambda"], rtol=1e-6)
    # identical algorithm; the two-term sum of A x rounds differently from the one-piece product, so late lambdas may
    # stop an iteration apart
    assert np.abs(r0["niter"][:4].astype(int) - ref["niter"][:4].astype(int)).max() <= 2, (r0["niter"], ref["niter"])
    for j in range(8):
        assert relerr(r0["beta"][:, j], ref["beta"][:, j]) < (1e-4 if j < 4 else 5e-3), j


# ------------------------------------------------------------------------------------------------------------------
# Replicas with a final exchange (admm_hip_lasso_cv, admm_hip_lasso_multi): world_size-2 model of the dealing rule of
# api.hip -- unit u (fold / response) runs on rank u mod world, every rank starts from zeroed tables, ONE sum all-reduce at
# the end -- with the oracle as the fit.  Every rank must end with the tables a single process computes.
def _replica_rank_main(rank, world, port, x, y, nfolds, lam, out_path):
    from oracle import entry
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, p = x.shape
    nl = len(lam)
    fid = np.arange(n) % nfolds                                 # fold_id NULL: i mod nfolds
    mse = np.zeros((nfolds, nl))
    nit = np.zeros((nfolds, nl))
    for f in range(nfolds):
        if f % world != rank:
            continue
        tr, te = fid != f, fid == f
        fit = entry.admm_lasso(x[tr], y[tr], lam, nl, 1e-4, True, True, entry.LASSO_OPTS)
        b = fit["beta"].astype(np.float64)
        pred = b[0][None, :] + x[te] @ b[1:]
        mse[f] = ((y[te][:, None] - pred) ** 2).mean(axis=0)
        nit[f] = fit["niter"]
    both = _allreduce(np.concatenate([mse.ravel(), nit.ravel()]))
    np.savez(out_path + f".{rank}.npz", mse=both[:nfolds * nl].reshape(nfolds, nl), nit=both[nfolds * nl:].reshape(nfolds, nl))
    dist.destroy_process_group()


def test_two_rank_replica_dealing_and_final_exchange(tmp_path):
    from oracle import entry
    x, y = synth_lasso(240, 12, 4, seed=71)
    lam = [0.4, 0.1, 0.02]
    nfolds = 5
    out = str(tmp_path / "cv")
    mp.spawn(_replica_rank_main, args=(2, _free_port(), x, y, nfolds, lam, out), nprocs=2, join=True)
    fid = np.arange(240) % nfolds
    ref_mse = np.zeros((nfolds, 3))
    ref_nit = np.zeros((nfolds, 3))
    for f in range(nfolds):
        tr, te = fid != f, fid == f
        fit = entry.admm_lasso(x[tr], y[tr], lam, 3, 1e-4, True, True, entry.LASSO_OPTS)
        b = fit["beta"].astype(np.float64)
        ref_mse[f] = ((y[te][:, None] - (b[0][None, :] + x[te] @ b[1:])) ** 2).mean(axis=0)
        ref_nit[f] = fit["niter"]
    for r in range(2):
        got = np.load(out + f".{r}.npz")
        assert np.array_equal(got["mse"], ref_mse) and np.array_equal(got["nit"], ref_nit)


# ------------------------------------------------------------------------------------------------------------------
# Column-block sharing basis pursuit (admm_hip_parbp_dist, sharing_bp.hip): world_size-2 model of its exchange -- the blocks of
# PADMMBP's partition dealt out to the ranks (admm_amd.dist.parbp_partition), the spectral radii gathered through a sum of
# one-hot vectors, and per iteration ONE sum all-reduce of [S = sum_i A_i x_i (n), sum ||A_i x_i||^2, sum ||A_i dx_i||^2]
# between the block sums and the replicated r / y / decision step.  Every rank runs the oracle's own worker arithmetic on
# its blocks; the joint result must be the serial oracle's.
def _parbp_rank_main(rank, world, port, A, b, N, out_path):
    from admm_amd.dist import parbp_partition
    from oracle.solvers import SharingBP
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, p = A.shape
    lo, hi = parbp_partition(p, N, world, rank)
    chunk, per = p // N, N // world
    b_first = rank * per
    ser = SharingBP(A, b, N, 1e-4, 1e-4)                       # only its partition / block views are used below
    mine = list(range(b_first, b_first + per))
    assert ser.off[mine[0]] == lo and ser.off[mine[-1] + 1] == hi
    sprad = np.zeros(N)
    for i in mine:
        sprad[i] = ser.sprad[i]
    sprad = _allreduce(sprad)
    rho = 1.0 / float(np.mean(sprad))
    x = {i: np.zeros(ser.A[i].shape[1]) for i in mine}
    Ax = {i: np.zeros(n) for i in mine}
    y, r, S = np.zeros(n), np.zeros(n), np.zeros(n)
    zbar = b / N
    sax = abar_r = 0.0
    niter = 10001
    for it in range(10000):
        r2, y2 = float(r @ r), float(y @ y)
        sz = sax - 2.0 * N * abar_r + N * r2
        eps_p = 1e-4 * np.sqrt(max(sax, sz, 0.0)) + np.sqrt(float(n * N)) * 1e-4
        eps_d = 1e-4 * np.sqrt(float(N)) * np.sqrt(y2) + np.sqrt(float(n * N)) * 1e-4
        v = y / rho + r
        payload = np.zeros(n + 2)
        for i in mine:
            Ai = ser.A[i]
            gamma = 2.0 * rho + sprad[i]
            pen = 1.0 / (rho * gamma)
            xi = x[i]
            if it % 10 == 0:
                vec = xi - (Ai.T @ v) / gamma
                xi = np.sign(vec) * np.maximum(np.abs(vec) - pen, 0.0)
            else:
                nz = np.nonzero(xi)[0]
                xn = np.zeros_like(xi)
                if nz.size:
                    val = xi[nz] - (Ai[:, nz].T @ v) / gamma
                    xn[nz] = np.sign(val) * np.maximum(np.abs(val) - pen, 0.0)
                xi = xn
            x[i] = xi
            nz = np.nonzero(xi)[0]
            new = Ai[:, nz] @ xi[nz] if nz.size else np.zeros(n)
            d = new - Ax[i]
            Ax[i] = new
            payload[:n] += new
            payload[n] += float(new @ new)
            payload[n + 1] += float(d @ d)
        payload = _allreduce(payload)                           # THE exchange of the iteration
        Snew, sax, q = payload[:n], float(payload[n]), float(payload[n + 1])
        rnew = Snew / N - zbar
        dr, dS = rnew - r, Snew - S
        sd = q - 2.0 * float(dr @ dS) + N * float(dr @ dr)
        S, r = Snew, rnew
        y = y + rho * rnew
        abar_r = float((Snew / N) @ rnew)
        if np.sqrt(N * float(rnew @ rnew)) < eps_p and rho * np.sqrt(max(sd, 0.0)) < eps_d:
            niter = it + 1
            break
    np.savez(out_path + f".{rank}.npz", beta=np.concatenate([x[i] for i in mine]), niter=np.array([niter]), lo=np.array([lo, hi]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_column_block_sharing_bp_protocol_matches_serial_oracle(tmp_path):
    from oracle import entry
    rng = np.random.default_rng(91)
    n, p, N = 60, 243, 4                                        # 3 blocks of 60 + one of 63
    A = rng.standard_normal((n, p))
    b0 = np.zeros(p)
    b0[rng.choice(p, 9, replace=False)] = rng.standard_normal(9) * 2
    b = A @ b0
    out = str(tmp_path / "parbp")
    mp.spawn(_parbp_rank_main, args=(2, _free_port(), A, b, N, out), nprocs=2, join=True)
    r0, r1 = np.load(out + ".0.npz"), np.load(out + ".1.npz")
    assert list(r0["lo"]) == [0, 120] and list(r1["lo"]) == [120, 243]
    ref = entry.admm_parbp(A, b, N, dict(entry.BP_OPTS, rho_ratio=1.0))
    assert r0["niter"][0] == r1["niter"][0] == ref["niter"]      # double arithmetic: the two-piece sum of S moves nothing visible
    beta = np.concatenate([r0["beta"], r1["beta"]])
    assert np.abs(beta - ref["beta"]).max() < 1e-12 and np.abs(beta - b0).max() < 5e-3


# ------------------------------------------------------------------------------------------------------------------
# Distributed factorisation of the row-sharded tall solver (chol_inverse.h cholesky_linvt_blocked_dist,
# lasso_tall.hip): world_size-2 model of its protocol -- block column k of the lower triangle and of U = L^-T belongs to rank
# k mod N; the owner factorises the diagonal block, finishes its column of U and the panel and BROADCASTS them as one message;
# every rank applies the two rank-B updates to the block columns it owns.  The claim the GPU test
# (tests/test_gpu_dist2.py::test_distributed_factorisation_is_bit_identical_to_the_replicated_one) rests on is modelled here in
# float32 NumPy: every tile sees the same updates in the same order from bit-identical operands as in one process, so the ranks'
# pieces ARE the single-process result, bit for bit, and each rank performs ~1 / N of the update flops.
def _blocked_factor_model(A, B, nranks, rank, bcast):
    """Right-looking blocked Cholesky of the lower triangle of A (float32, in place) + U = L^-T, block size B, the block columns
    dealt out to `nranks`; bcast(buf, root) returns root's buffer on every rank.  Returns (A, U, update flops of this rank)."""
    p = A.shape[0]
    nb = (p + B - 1) // B
    U = np.eye(p, dtype=F)
    flops = 0
    for k in range(nb):
        r0, r1 = k * B, min((k + 1) * B, p)
        owner = k % nranks
        if rank == owner:
            Lkk = np.linalg.cholesky(A[r0:r1, r0:r1].astype(np.float64)).astype(F)      # (the model's diagonal-block kernel)
            Dinv = sla.solve_triangular(Lkk.astype(np.float64), np.eye(r1 - r0), lower=True).astype(F)
            A[r0:r1, r0:r1] = Lkk
            U[:r1, r0:r1] = U[:r1, r0:r1] @ Dinv.T                                       # this block column of U is final
            if r1 < p:
                A[r1:, r0:r1] = A[r1:, r0:r1] @ Dinv.T                                   # the panel L_ik
        msg = np.concatenate([A[r0:, r0:r1].ravel(), U[:r1, r0:r1].ravel()]) if rank == owner else None
        msg = bcast(msg, owner, (p - r0) * (r1 - r0) + r1 * (r1 - r0))
        nl = (p - r0) * (r1 - r0)
        A[r0:, r0:r1] = msg[:nl].reshape(p - r0, r1 - r0)
        U[:r1, r0:r1] = msg[nl:].reshape(r1, r1 - r0)
        for j in range(k + 1, nb):
            if j % nranks != rank:
                continue
            c0, c1 = j * B, min((j + 1) * B, p)
            Lj = A[c0:, r0:r1]
            A[c0:, c0:c1] -= Lj @ A[c0:c1, r0:r1].T
            U[:r1, c0:c1] -= U[:r1, r0:r1] @ A[c0:c1, r0:r1].T
            flops += 2 * (r1 - r0) * (c1 - c0) * ((p - c0) + r1)
    return A, U, flops


def _factor_rank_main(rank, world, port, M, B, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def bcast(buf, root, n):
        t = torch.from_numpy(np.ascontiguousarray(buf)) if rank == root else torch.empty(n, dtype=torch.float32)
        dist.broadcast(t, src=root)
        return t.numpy()

    A, U, flops = _blocked_factor_model(M.copy(), B, world, rank, bcast)
    np.savez(out_path + f".{rank}.npz", A=A, U=U, flops=np.array([flops]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_distributed_factorisation_protocol_is_bit_identical_to_one_process(tmp_path):
    rng = np.random.default_rng(123)
    p, B = 150, 16                                               # 10 block columns, the last one ragged (6 columns)
    X = rng.standard_normal((220, p)).astype(F)
    M = (X.T @ X + F(3.0) * np.eye(p, dtype=F)).astype(F)
    A1, U1, fl1 = _blocked_factor_model(M.copy(), B, 1, 0, lambda buf, root, n: buf)
    out = str(tmp_path / "factor")
    mp.spawn(_factor_rank_main, args=(2, _free_port(), M, B, out), nprocs=2, join=True)
    r = [np.load(out + f".{k}.npz") for k in range(2)]
    nb = (p + B - 1) // B
    for k in range(2):
        # U: every block column was broadcast when it became final -- the whole of it on every rank, bit-identical
        assert np.array_equal(np.triu(r[k]["U"]), np.triu(U1)), k
        # L: the panels (broadcast) on every rank; the trailing matrix is only kept up to date in the columns a rank owns
        assert np.array_equal(np.tril(r[k]["A"]), np.tril(A1)), k
    # the factor is a factor: L L' = M and U = L^-T, to float accuracy
    L = np.tril(A1).astype(np.float64)
    assert np.abs(L @ L.T - M).max() < 1e-3 * np.abs(M).max()
    assert np.abs(np.triu(U1).astype(np.float64).T @ L - np.eye(p)).max() < 1e-4
    # the update flops are shared out: the two ranks' add up to one process's, neither does more than 60 % (block columns k mod 2)
    f0, f1 = int(r[0]["flops"][0]), int(r[1]["flops"][0])
    assert f0 + f1 == fl1 and max(f0, f1) < 0.6 * fl1, (f0, f1, fl1)
    assert nb == 10


# ------------------------------------------------------------------------------------------------------------------
# Split-K Gram of the row-sharded tall solver as a REDUCE-SCATTER (lasso_tall.hip): world_size-2 model.
# Every rank forms the Gram of its row slice; the Lanczos products are sum_r (G_r v) with one all-reduce of p floats each (the
# whole matrix never exists on a rank); the block columns are packed owner by owner (block k -> rank k mod N, slot k div N, zero
# padded to equal counts) and reduce-scattered, so a rank receives exactly the block columns the distributed factorisation lets it
# read.  Checked: the received block columns equal those of the all-reduced Gram bit for bit (two ranks: one addition per entry,
# the same one), the distributed product equals the product with the summed matrix to float rounding.
def _gram_rs_rank_main(rank, world, port, X, B, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, p = X.shape
    lo, hi = rank * (n // world), (n if rank == world - 1 else (rank + 1) * (n // world))
    Xl = np.asarray(X[lo:hi], dtype=F)
    G = (Xl.T @ Xl).astype(F)                                    # this rank's term of the split-K sum
    v = np.random.default_rng(5).standard_normal(p).astype(F)
    w = _allreduce((G @ v).astype(F))                            # one Lanczos product: local product, p floats all-reduced
    nb = (p + B - 1) // B
    pp = nb * B
    Gp = np.zeros((pp, pp), dtype=F)
    Gp[:p, :p] = G
    nown = (nb + world - 1) // world
    send = np.zeros((world, nown, B, pp), dtype=F)               # [owner][slot][column of the block][row]
    for k in range(nb):
        send[k % world, k // world] = Gp[:, k * B:(k + 1) * B].T
    recv = torch.empty(nown * B * pp, dtype=torch.float32)
    dist.reduce_scatter(recv, [torch.from_numpy(np.ascontiguousarray(send[r]).ravel()) for r in range(world)])
    full = _allreduce(Gp.copy())
    np.savez(out_path + f".{rank}.npz", recv=recv.numpy().reshape(nown, B, pp), full=full, w=w, nown=np.array([nown]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gram_reduce_scatter_hands_every_rank_its_block_columns(tmp_path):
    rng = np.random.default_rng(321)
    n, p, B = 160, 70, 16                                        # 5 block columns (the last one ragged): rank 0 owns 0, 2, 4; rank 1 owns 1, 3
    X = rng.standard_normal((n, p)).astype(F)
    out = str(tmp_path / "gramrs")
    mp.spawn(_gram_rs_rank_main, args=(2, _free_port(), X, B, out), nprocs=2, join=True)
    r = [np.load(out + f".{k}.npz") for k in range(2)]
    nb = (p + B - 1) // B
    assert np.array_equal(r[0]["full"], r[1]["full"])
    for rank in range(2):
        for k in range(rank, nb, 2):
            assert np.array_equal(r[rank]["recv"][k // 2], r[rank]["full"][:, k * B:(k + 1) * B].T), (rank, k)
    assert np.array_equal(r[0]["recv"][2], r[0]["full"][:, 4 * B:5 * B].T) and not r[1]["recv"][2].any()     # rank 1's third slot is padding
    G = r[0]["full"][:p, :p].astype(np.float64)
    v = np.random.default_rng(5).standard_normal(p)
    assert np.array_equal(r[0]["w"], r[1]["w"])
    assert np.abs(r[0]["w"] - G @ v.astype(F)).max() < 1e-4 * np.abs(G @ v).max()


# ------------------------------------------------------------------------------------------------------------------
# The exchange INSIDE a launch (round 6: the column-sharded wide solver's persistent stretch; peer_device.h "AUX region",
# lasso_wide.hip wide_rows_persist_kernel<true>): world_size-2 model of the PROTOCOL -- a sequence number kept by every rank and
# advanced identically, two parities of slots, one flag per (source rank, row group), a once-per-launch agreement word, row groups
# that run concurrently and are only coupled by the rank's own hand-overs.  What the model checks is the claim the kernel's comment
# makes: a rank that runs ahead can never overwrite a slot a slower rank (or a slower row group's reader) has yet to read, whatever
# the timing -- every sum a reader forms is the sum of the values that belong to ITS exchange number.
def _aux_model_rank(rank, world, port, data, flags, nlaunch, seed, out):
    import threading
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    G, W = 3, 4                                             # row groups, floats per group
    rng = np.random.default_rng(seed + 17 * rank)          # the TIMING differs per rank; the values are functions of (exchange, group, rank)
    seq = 0                                                 # the device word: exchanges made so far (replicated)
    errors = []

    def value(e, r, src):
        return float(1000 * e + 10 * r + src)

    def push(e, r, payload):                                # this rank's slot of EVERY rank's buffer, then the flag there
        for dst in range(world):
            data[dst, e & 1, rank, r] = payload
        for dst in range(world):
            flags[dst, e & 1, rank, r] = e

    def wait_and_sum(e, r):
        t0 = time.time()
        while any(int(flags[rank, e & 1, src, r]) < e for src in range(world)):
            if time.time() - t0 > 20:
                raise RuntimeError(f"rank {rank}: exchange {e} group {r} never arrived")
            time.sleep(0)
        return sum(data[rank, e & 1, src, r].clone() for src in range(world))          # rank order

    for launch in range(nlaunch):
        # ---- agreement: every rank's count (group index G of the flag array is the agreement word's flag)
        e = seq + 1
        count = int((launch * 7 + rank * 3) % 5)             # rank-dependent; the verdict is a function of ALL counts
        push(e, G, torch.full((W,), float(count)))
        s = wait_and_sum(e, G)
        total = int(round(float(s[0])))
        if total == 0:                                       # nobody has anything: all ranks leave, one exchange made
            seq += 1
            continue
        niter = 1 + total % 4                                # replicated: the "decisions" end the stretch after the same iteration everywhere
        bar = threading.Barrier(G)                           # the rank's own hand-over couples its row groups once per iteration

        def group(r):
            try:
                g_rng = np.random.default_rng(seed + 1000 * launch + 31 * rank + r)
                for k in range(niter):
                    ek = seq + 2 + k
                    time.sleep(float(g_rng.uniform(0, 2e-3)) if g_rng.random() < 0.5 else 0)
                    push(ek, r, torch.full((W,), value(ek, r, rank)))
                    got = wait_and_sum(ek, r)
                    want = sum(value(ek, r, src) for src in range(world))
                    if not torch.all(got == want):
                        errors.append((launch, k, r, got.tolist(), want))
                    time.sleep(float(g_rng.uniform(0, 2e-3)) if g_rng.random() < 0.3 else 0)
                    bar.wait()                               # (hand-over B: every row group of the rank has read before any starts the next iteration)
            except Exception as ex:                         # noqa: BLE001
                errors.append(("exception", repr(ex)))
                bar.abort()

        th = [threading.Thread(target=group, args=(r,)) for r in range(G)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        seq += 1 + niter
        if rng.random() < 0.5:
            time.sleep(float(rng.uniform(0, 3e-3)))          # the ranks' hosts enqueue at different paces
    torch.save({"errors": errors, "seq": seq}, out + f".{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_in_launch_exchange_protocol_never_reads_another_exchanges_slot(tmp_path):
    world, G, W = 2, 3, 4
    data = torch.zeros((world, 2, world, G + 1, W), dtype=torch.float64).share_memory_()          # [owner][parity][source rank][group | agreement][floats]
    flags = torch.zeros((world, 2, world, G + 1), dtype=torch.int64).share_memory_()
    out = str(tmp_path / "aux")
    mp.spawn(_aux_model_rank, args=(world, _free_port(), data, flags, 60, 5, out), nprocs=world, join=True)
    r = [torch.load(out + f".{k}.pt") for k in range(world)]
    assert r[0]["errors"] == [] and r[1]["errors"] == [], (r[0]["errors"][:3], r[1]["errors"][:3])
    assert r[0]["seq"] == r[1]["seq"] > 60                    # both ranks advanced the sequence identically
