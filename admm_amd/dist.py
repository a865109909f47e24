"""One-process-per-GPU consensus Lasso: communicator bootstrap and the distributed entry points.

The RCCL communicator lives inside libadmm_hip.so; this module only moves the 128-byte unique id
between ranks (over torch.distributed, any backend) and marshals arguments.  Each rank passes its
contiguous ROW SLICE of the global problem in the reference's partition (PADMMLasso.h:163-179);
`row_partition` computes that slice.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import AdmmOpts, AdmmStats, as_input, check
from .api import ADMM_Lasso_fit

UNIQUE_ID_BYTES = 128


def row_partition(n_total, nblocks, nranks, rank):
    """Rows [lo, hi) owned by `rank` when `nblocks` reference row blocks (chunk = n_total // nblocks, the
    last block takes the remainder) are dealt out contiguously, nblocks // nranks per rank."""
    if nblocks % nranks:
        raise ValueError("the number of row blocks must be a multiple of the number of ranks")
    chunk = n_total // nblocks
    per = nblocks // nranks
    lo = rank * per * chunk
    hi = (rank + 1) * per * chunk if rank < nranks - 1 else n_total
    return lo, hi


def symv_tiles(p, part=0, nparts=1, rows_per_tile=256, cols_per_tile=128):
    """Host mirror of SymvPlan::init (csrc/symv_kernels.h): the (row block, column block) tiles on or below the diagonal
    of a p x p symmetric matrix, longest row strips first, and the share of them rank `part` of `nparts` streams in the
    row-sharded tall x-update (tile i of the full list goes to rank i % nparts)."""
    nrb = (p + rows_per_tile - 1) // rows_per_tile
    ncb = (p + cols_per_tile - 1) // cols_per_tile
    tiles = [(rb, cb) for rb in range(nrb - 1, -1, -1) for cb in range(ncb)
             if cb * cols_per_tile <= rb * rows_per_tile + (rows_per_tile - 1)]
    return tiles[part::nparts]


def init_comm(nranks=1, rank=0, broadcast=None):
    """Attach the process-wide RCCL communicator.  `broadcast(buf: np.ndarray[uint8]) -> np.ndarray` must return
    rank 0's buffer on every rank (not needed for nranks == 1)."""
    lib = _lib.load()
    buf = np.zeros(UNIQUE_ID_BYTES, dtype=np.uint8)
    if rank == 0:
        check(lib.admm_hip_comm_unique_id(buf.ctypes.data))
    if nranks > 1:
        if broadcast is None:
            raise ValueError("a broadcast function is required for nranks > 1")
        buf = np.ascontiguousarray(broadcast(buf), dtype=np.uint8)
    check(lib.admm_hip_comm_init(int(nranks), int(rank), buf.ctypes.data))


def comm_info():
    """(nranks, rank, backend name) of the communicator the library REALLY holds (admm_hip_comm_info): RCCL's own count of the live
    communicator, the ranks attached to the SHM segment, the PEER buffers mapped -- (1, 0, "none") when nothing is attached."""
    import ctypes
    lib = _lib.load()
    n, r, b = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    check(lib.admm_hip_comm_info(ctypes.byref(n), ctypes.byref(r), ctypes.byref(b)))
    return n.value, r.value, {0: "none", 1: "rccl", 2: "shm", 3: "peer"}[b.value]


def init_comm_from_torch(device=None):
    """Bootstrap over an initialised torch.distributed process group (gloo or nccl)."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()

    def bcast(buf):
        t = torch.from_numpy(buf.copy())
        if dist.get_backend() == "nccl":
            t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
        dist.broadcast(t, src=0)
        return t.cpu().numpy()

    init_comm(world, rank, bcast if world > 1 else None)


PEER_HANDLE_BYTES = 64


def init_comm_shm(nranks, rank, name, token):
    """Attach the host shared-memory backend (all ranks on one host; `name` like "/admm_job42"; `token`: a non-zero
    job-unique integer all ranks agree on -- a stale or foreign segment of the same name is never attached to)."""
    check(_lib.load().admm_hip_comm_init_shm(int(nranks), int(rank), name.encode(), int(token) & 0xFFFFFFFFFFFFFFFF))


def job_token_from_torch():
    """A fresh non-zero 63-bit token drawn by rank 0 and broadcast over the initialised torch.distributed group."""
    import os
    import torch
    import torch.distributed as dist
    t = torch.zeros(1, dtype=torch.int64)
    if dist.get_rank() == 0:
        t[0] = (int.from_bytes(os.urandom(8), "little") >> 1) | 1
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.broadcast(t, src=0)
    return int(t.item())


def init_comm_peer(nranks, rank, allgather):
    """Attach the one-shot peer-mapped all-reduce.  `allgather(buf: np.ndarray[uint8, 64]) -> np.ndarray[uint8, nranks * 64]`
    must return the buffers of all ranks concatenated in rank order."""
    lib = _lib.load()
    mine = np.zeros(PEER_HANDLE_BYTES, dtype=np.uint8)
    check(lib.admm_hip_comm_peer_prepare(int(nranks), mine.ctypes.data))
    allh = np.ascontiguousarray(allgather(mine), dtype=np.uint8)
    if allh.size != nranks * PEER_HANDLE_BYTES:
        raise ValueError("allgather returned %d bytes, expected %d" % (allh.size, nranks * PEER_HANDLE_BYTES))
    check(lib.admm_hip_comm_init_peer(int(nranks), int(rank), allh.ctypes.data))


def init_comm_backend_from_torch(backend, device=None, name=None):
    """Bootstrap `backend` in {"rccl", "peer", "shm"} over an initialised torch.distributed process group."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    if backend == "rccl":
        return init_comm_from_torch(device)
    if backend == "shm":
        token = job_token_from_torch()                      # also makes the default name unique per job
        return init_comm_shm(world, rank, name or "/admm_hip_%d_%x" % (world, token), token)
    if backend != "peer":
        raise ValueError(backend)

    def allgather(buf):
        t = torch.from_numpy(buf.copy())
        on_gpu = dist.get_backend() == "nccl"
        if on_gpu:
            t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return torch.cat(out).cpu().numpy()

    init_comm_peer(world, rank, allgather)


def reduce_scatter_host(send, count):
    """Sum reduce-scatter of a host float32 array of nranks * count entries through the attached backend (test hook): returns this
    rank's chunk summed over the ranks."""
    lib = _lib.load()
    send = np.ascontiguousarray(send, dtype=np.float32)
    recv = np.zeros(int(count), dtype=np.float32)
    check(lib.admm_hip_comm_test_reduce_scatter(send.ctypes.data, int(count), recv.ctypes.data))
    return recv


def allreduce_host(fbuf=None, dbuf=None):
    """In-place sum all-reduce of host float32 / float64 arrays through the attached backend (test hook)."""
    lib = _lib.load()
    nf = 0 if fbuf is None else fbuf.size
    nd = 0 if dbuf is None else dbuf.size
    check(lib.admm_hip_comm_test_allreduce(fbuf.ctypes.data if nf else None, nf, dbuf.ctypes.data if nd else None, nd, 0))


def finalize_comm():
    check(_lib.load().admm_hip_comm_finalize())


def _marshal(x_local, y_local, n_local, n_total, p, lam, nlambda, lambda_min_ratio, standardize, intercept, nthread, opts):
    xp, xmem, xk = as_input(x_local)
    yp, ymem, yk = as_input(y_local)
    lam_in = np.ascontiguousarray(np.sort(np.atleast_1d(np.asarray(lam, dtype=np.float64)))[::-1]) if lam is not None else np.zeros(0)
    o = AdmmOpts(int(opts.get("maxit", 10000)), float(opts.get("eps_abs", 1e-5)), float(opts.get("eps_rel", 1e-5)),
                 float(opts.get("rho", -1.0) if opts.get("rho") is not None else -1.0))
    head = (xp, yp, int(n_local), int(n_total), int(p), xmem,
            ctypes.c_void_p(lam_in.ctypes.data if lam_in.size else 0), int(lam_in.size), int(nlambda), float(lambda_min_ratio),
            int(bool(standardize)), int(bool(intercept)), int(nthread), ctypes.byref(o))
    return head, (xk, yk, lam_in, o), (lam_in.size if lam_in.size else int(nlambda))


def parlasso_dist(x_local, y_local, n_total, p, nthread, lam=None, nlambda=100, lambda_min_ratio=None,
                  standardize=True, intercept=True, n_local=None, **opts):
    """Distributed `admm_lasso(x, y)$penalty(...)$parallel(nthread)$fit()`: every rank calls this with its row slice."""
    lib = _lib.load()
    if n_local is None:
        n_local = np.asarray(x_local).shape[0]
    if lambda_min_ratio is None:
        lambda_min_ratio = 0.01 if n_total < p else 1e-4
    head, keep, nl = _marshal(x_local, y_local, n_local, n_total, p, lam, nlambda, lambda_min_ratio, standardize, intercept, nthread, opts)
    lam_out = np.zeros(nl)
    beta = np.zeros((p + 1, nl), dtype=np.float32, order="F")
    niter = np.zeros(nl, dtype=np.int32)
    stats = AdmmStats()
    check(lib.admm_hip_parlasso_dist(*head, lam_out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                     beta.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                     niter.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), ctypes.byref(stats)))
    return ADMM_Lasso_fit(lam_out, beta, niter, stats.as_dict())


def lasso_dist(x_local, y_local, n_total, p, lam=None, nlambda=100, lambda_min_ratio=1e-4, standardize=True, intercept=True,
               alpha=None, n_local=None, **opts):
    """Row-sharded serial tall Lasso / elastic net (`admm_lasso(x, y)$fit()` over several GPUs): every rank calls this
    with a contiguous row slice."""
    lib = _lib.load()
    if n_local is None:
        n_local = np.asarray(x_local).shape[0]
    head, keep, nl = _marshal(x_local, y_local, n_local, n_total, p, lam, nlambda, lambda_min_ratio, standardize, intercept, 0, opts)
    head = head[:-2] + (float(-1.0 if alpha is None else alpha), head[-1])        # (..., intercept, alpha, opts) instead of (..., nthread, opts)
    lam_out = np.zeros(nl)
    beta = np.zeros((p + 1, nl), dtype=np.float32, order="F")
    niter = np.zeros(nl, dtype=np.int32)
    stats = AdmmStats()
    check(lib.admm_hip_lasso_dist(*head, lam_out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                  beta.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                  niter.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), ctypes.byref(stats)))
    return ADMM_Lasso_fit(lam_out, beta, niter, stats.as_dict())


def col_partition(p_total, nranks, rank):
    """Columns [lo, hi) of rank `rank` when p_total columns are dealt out in contiguous, nearly equal blocks."""
    return p_total * rank // nranks, p_total * (rank + 1) // nranks


def lasso_dist_cols(x_cols, y, p_total, col_offset, lam=None, nlambda=100, lambda_min_ratio=0.01, standardize=True, intercept=True,
                    alpha=None, **opts):
    """Column-sharded serial wide Lasso / elastic net: every rank calls this with its column block (n x p_local) and the full y."""
    lib = _lib.load()
    xp, xmem, xk = as_input(x_cols)
    yp, ymem, yk = as_input(y)
    n, p_local = np.asarray(x_cols).shape
    lam_in = np.ascontiguousarray(np.sort(np.atleast_1d(np.asarray(lam, dtype=np.float64)))[::-1]) if lam is not None else np.zeros(0)
    o = AdmmOpts(int(opts.get("maxit", 10000)), float(opts.get("eps_abs", 1e-5)), float(opts.get("eps_rel", 1e-5)),
                 float(opts.get("rho", -1.0) if opts.get("rho") is not None else -1.0))
    nl = lam_in.size if lam_in.size else int(nlambda)
    lam_out = np.zeros(nl)
    beta = np.zeros((p_total + 1, nl), dtype=np.float32, order="F")
    niter = np.zeros(nl, dtype=np.int32)
    stats = AdmmStats()
    check(lib.admm_hip_lasso_dist_cols(xp, yp, int(n), int(p_local), int(p_total), int(col_offset), xmem,
                                       ctypes.c_void_p(lam_in.ctypes.data if lam_in.size else 0), int(lam_in.size), int(nlambda),
                                       float(lambda_min_ratio), int(bool(standardize)), int(bool(intercept)),
                                       float(-1.0 if alpha is None else alpha), ctypes.byref(o),
                                       lam_out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), beta.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                       niter.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), ctypes.byref(stats)))
    return ADMM_Lasso_fit(lam_out, beta, niter, stats.as_dict())


def parbp_partition(p_total, nthread, nranks, rank):
    """Columns [lo, hi) of rank `rank` for admm_hip_parbp_dist: whole blocks of PADMMBP's partition (nthread - 1 blocks of
    p div nthread columns, the last takes the remainder; PADMMBP.h:150-167), nthread / nranks consecutive blocks per rank."""
    if nthread % nranks:
        raise ValueError("nthread must be a multiple of the number of ranks")
    chunk = p_total // nthread
    per = nthread // nranks
    lo = rank * per * chunk
    hi = p_total if rank == nranks - 1 else (rank + 1) * per * chunk
    return lo, hi


def parbp_dist(x_cols, y, p_total, col_offset, nthread, maxit=10000, eps_abs=1e-4, eps_rel=1e-4, rho=1.0):
    """Column-block sharing basis pursuit (admm_hip_parbp_dist) with the blocks spread over the ranks: every rank calls this with
    its columns (n x p_local, whole blocks -- parbp_partition) and the full y; returns (this rank's coefficients, niter, stats)."""
    lib = _lib.load()
    xp, xmem, xk = as_input(x_cols)
    yp, ymem, yk = as_input(y)
    n, p_local = np.asarray(x_cols).shape
    o = AdmmOpts(int(maxit), float(eps_abs), float(eps_rel), float(rho))
    beta = np.zeros(p_local)
    niter = np.zeros(1, dtype=np.int32)
    stats = AdmmStats()
    check(lib.admm_hip_parbp_dist(xp, yp, int(n), int(p_local), int(p_total), int(col_offset), xmem, int(nthread), ctypes.byref(o),
                                  beta.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), niter.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                                  ctypes.byref(stats)))
    return beta, int(niter[0]), stats.as_dict()


class DistColsPlan:
    """Prepared column-sharded wide problem (admm_hip_lasso_plan_create_dist_cols): setup once, run the path repeatedly."""

    def __init__(self, x_cols, y, p_total, col_offset, lam=None, nlambda=100, lambda_min_ratio=0.01, standardize=True, intercept=True,
                 alpha=None, **opts):
        lib = _lib.load()
        self._lib = lib
        self.p = int(p_total)
        xp, xmem, xk = as_input(x_cols)
        yp, ymem, yk = as_input(y)
        n, p_local = np.asarray(x_cols).shape
        lam_in = np.ascontiguousarray(np.sort(np.atleast_1d(np.asarray(lam, dtype=np.float64)))[::-1]) if lam is not None else np.zeros(0)
        o = AdmmOpts(int(opts.get("maxit", 10000)), float(opts.get("eps_abs", 1e-5)), float(opts.get("eps_rel", 1e-5)),
                     float(opts.get("rho", -1.0) if opts.get("rho") is not None else -1.0))
        h = ctypes.c_void_p()
        nlo = ctypes.c_int()
        check(lib.admm_hip_lasso_plan_create_dist_cols(xp, yp, int(n), int(p_local), int(p_total), int(col_offset), xmem,
                                                       ctypes.c_void_p(lam_in.ctypes.data if lam_in.size else 0), int(lam_in.size), int(nlambda),
                                                       float(lambda_min_ratio), int(bool(standardize)), int(bool(intercept)),
                                                       float(-1.0 if alpha is None else alpha), ctypes.byref(o), ctypes.byref(h), ctypes.byref(nlo)))
        self._h = h
        self.nlambda = nlo.value

    run = None            # bound below (shared with DistLassoPlan)


class DistLassoPlan:
    """Prepared distributed problem (setup once, run the lambda path repeatedly): the consensus solver for nthread >= 1,
    the row-sharded serial tall solver for nthread == 0."""

    def __init__(self, x_local, y_local, n_total, p, nthread, lam=None, nlambda=100, lambda_min_ratio=None,
                 standardize=True, intercept=True, n_local=None, **opts):
        lib = _lib.load()
        self._lib = lib
        self.p = p
        if n_local is None:
            n_local = np.asarray(x_local).shape[0]
        if lambda_min_ratio is None:
            lambda_min_ratio = 0.01 if n_total < p else 1e-4
        head, keep, nl = _marshal(x_local, y_local, n_local, n_total, p, lam, nlambda, lambda_min_ratio, standardize, intercept, nthread, opts)
        h = ctypes.c_void_p()
        nlo = ctypes.c_int()
        check(lib.admm_hip_lasso_plan_create_dist(*head, ctypes.byref(h), ctypes.byref(nlo)))
        self._h = h
        self.nlambda = nlo.value

    def run(self):
        lam_out = np.zeros(self.nlambda)
        beta = np.zeros((self.p + 1, self.nlambda), dtype=np.float32, order="F")
        niter = np.zeros(self.nlambda, dtype=np.int32)
        stats = AdmmStats()
        check(self._lib.admm_hip_lasso_plan_run(self._h, lam_out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                                beta.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                                niter.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), ctypes.byref(stats)))
        return ADMM_Lasso_fit(lam_out, beta, niter, stats.as_dict())

    def enable_trace(self, capacity=1 << 18):
        check(self._lib.admm_hip_lasso_plan_trace_enable(self._h, int(capacity)))
        self._trace_cap = int(capacity)

    def read_trace(self):
        buf = np.zeros((self._trace_cap, _lib.TRACE_FIELDS), dtype=np.float64)
        n = ctypes.c_longlong()
        check(self._lib.admm_hip_lasso_plan_trace_read(self._h, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                                       self._trace_cap, ctypes.byref(n)))
        return buf[:n.value].copy()

    def enable_state(self, capacity):
        """Iterate dump of every iteration of the following run() calls (admm_hip_lasso_plan_state_*): record s = the iterates trace
        record s judged.  Column-sharded wide solver: x of THIS rank's columns | A x | z | y (p_local + 3 n floats)."""
        check(self._lib.admm_hip_lasso_plan_state_enable(self._h, int(capacity)))

    def read_state(self):
        n, rf = ctypes.c_longlong(), ctypes.c_longlong()
        check(self._lib.admm_hip_lasso_plan_state_read(self._h, None, 0, ctypes.byref(n), ctypes.byref(rf)))     # size query
        nrec = max(int(n.value), 1)
        buf = np.zeros((nrec, rf.value), dtype=np.float32)
        check(self._lib.admm_hip_lasso_plan_state_read(self._h, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), nrec, ctypes.byref(n), ctypes.byref(rf)))
        return buf[:n.value]

    def read_data(self, n, p_local):
        """(X_local, Y): the standardised float32 data as the solver holds them (admm_hip_lasso_plan_data_read)."""
        X = np.zeros((n, p_local), dtype=np.float32, order="F")
        Y = np.zeros(n, dtype=np.float32)
        check(self._lib.admm_hip_lasso_plan_data_read(self._h, X.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), n, Y.ctypes.data_as(ctypes.POINTER(ctypes.c_float))))
        return X, Y

    def close(self):
        if self._h:
            check(self._lib.admm_hip_lasso_plan_destroy(self._h))
            self._h = None


for _name in ("run", "enable_trace", "read_trace", "close", "enable_state", "read_state", "read_data"):
    setattr(DistColsPlan, _name, getattr(DistLassoPlan, _name))
