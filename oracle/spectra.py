"""Restatement of the one Spectra call the hot path makes (TEST INFRASTRUCTURE).

    SymEigsSolver<float, LARGEST_ALGE, DenseSymMatProd<float>> eigs(&op, 1, 3);
    eigs.init(); eigs.compute(10, 0.1); evals[0]
      -- /root/reference/src/ADMMLassoTall.h:196-201, ADMMLassoWide.h:202-207

i.e. implicitly restarted Lanczos with nev=1, ncv=3, at most 10 restarts and a
very loose tolerance 0.1: the returned Ritz value is a deliberate *under*-estimate
of lambda_max (it sets rho and the wide step size) and must be reproduced, not
improved.  Follows src/Spectra/SymEigsSolver.h (factorize_from :201-280, restart
:283-323, num_converged :326-335, nev_adjusted :338-353, retrieve_ritzpair
:356-397, init :494-544, compute :564-587), SimpleRandom.h:38-76,
LinAlg/TridiagEigen.h:47-157, LinAlg/UpperHessenbergQR.h TridiagQR :460-600.
All arithmetic is carried in `dtype` (float32 for the Lasso solvers).
"""
import numpy as np


def simple_random_vec(n, seed, dtype=np.float32):
    """SimpleRandom<Scalar>(seed).random_vec(n): minstd LCG, U(-0.5, 0.5). SimpleRandom.h:38-76."""
    a, mx = 16807, 2147483647
    r = (seed & mx) if seed else 1
    out = np.empty(n, dtype=dtype)
    fmx = dtype(mx)
    for i in range(n):
        lo = a * (r & 0xFFFF)
        hi = a * (r >> 16)
        lo += (hi & 0x7FFF) << 16
        if lo > mx:
            lo &= mx
            lo += 1
        lo += hi >> 15
        if lo > mx:
            lo &= mx
            lo += 1
        r = lo
        out[i] = dtype(r) / fmx - dtype(0.5)
    return out


def _make_givens(p, q, T):
    """Eigen::JacobiRotation<real>::makeGivens (Eigen/src/Jacobi/Jacobi.h), returns (c, s)."""
    if q == 0:
        return (T(-1) if p < 0 else T(1)), T(0)
    if p == 0:
        return T(0), (T(1) if q < 0 else T(-1))
    if abs(p) > abs(q):
        t = T(q / p)
        u = T(np.sqrt(T(1) + t * t))
        if p < 0:
            u = -u
        c = T(T(1) / u)
        return c, T(-t * c)
    t = T(p / q)
    u = T(np.sqrt(T(1) + t * t))
    if q < 0:
        u = -u
    s = T(-T(1) / u)
    return T(-t * s), s


def tridiag_eigen(H, T=np.float32):
    """TridiagEigen<Scalar>(H): symmetric tridiagonal QR iteration (TridiagEigen.h:47-157).

    Returns (eigenvalues in deflation order, eigenvector matrix)."""
    n = H.shape[0]
    d = np.array(np.diag(H), dtype=T)
    e = np.array(np.diag(H, -1), dtype=T)
    Q = np.eye(n, dtype=T)
    prec = T(1e-5) if T == np.float32 else T(1e-12)   # NumTraits::dummy_precision()
    end, start, it = n - 1, 0, 0
    while end > 0:
        for i in range(start, end):
            # is_much_smaller_than(|e_i|, |d_i|+|d_{i+1}|)
            x = abs(e[i])
            y = abs(d[i]) + abs(d[i + 1])
            if T(x * x) <= T(T(y * y) * prec * prec):
                e[i] = 0
        while end > 0 and e[end - 1] == 0:
            end -= 1
        if end <= 0:
            break
        it += 1
        if it > 30 * n:
            raise RuntimeError("TridiagEigen: failed to compute all the eigenvalues")
        start = end - 1
        while start > 0 and e[start - 1] != 0:
            start -= 1
        # tridiagonal_qr_step (TridiagEigen.h:47-103)
        td = T((d[end - 1] - d[end]) * T(0.5))
        ee = e[end - 1]
        mu = d[end]
        if td == 0:
            mu = T(mu - abs(ee))
        else:
            e2 = T(ee * ee)
            h = T(np.hypot(td, ee))
            if e2 == 0:
                mu = T(mu - T(ee / T(td + (T(1) if td > 0 else T(-1)))) * T(ee / h))
            else:
                mu = T(mu - e2 / T(td + (h if td > 0 else -h)))
        x = T(d[start] - mu)
        z = e[start]
        for k in range(start, end):
            c, s = _make_givens(x, z, T)
            sdk = T(s * d[k] + c * e[k])
            dkp1 = T(s * e[k] + c * d[k + 1])
            d[k] = T(c * T(c * d[k] - s * e[k]) - s * T(c * e[k] - s * d[k + 1]))
            d[k + 1] = T(s * sdk + c * dkp1)
            e[k] = T(c * sdk - s * dkp1)
            if k > start:
                e[k - 1] = T(c * e[k - 1] - s * z)
            x = e[k]
            if k < end - 1:
                z = T(-s * e[k + 1])
                e[k + 1] = T(c * e[k + 1])
            # Q = Q * G  (applyOnTheRight(k, k+1, rot))
            qk = Q[:, k].copy()
            qk1 = Q[:, k + 1].copy()
            Q[:, k] = (c * qk - s * qk1).astype(T)
            Q[:, k + 1] = (s * qk + c * qk1).astype(T)
    return d, Q


class _TridiagQR:
    """TridiagQR<Scalar> (UpperHessenbergQR.h:460-600): Givens QR of a symmetric tridiagonal."""

    def __init__(self, mat, T):
        n = mat.shape[0]
        self.n, self.T = n, T
        Tm = np.zeros((n, n), dtype=T)
        idx = np.arange(n)
        Tm[idx, idx] = np.diag(mat)
        Tm[idx[:-1], idx[:-1] + 1] = np.diag(mat, -1)
        Tm[idx[:-1] + 1, idx[:-1]] = np.diag(mat, -1)
        self.c = np.empty(n - 1, dtype=T)
        self.s = np.empty(n - 1, dtype=T)
        eps = np.finfo(T).eps
        for i in range(n - 1):
            a, b = Tm[i, i], Tm[i + 1, i]
            r = T(np.sqrt(T(a * a) + T(b * b)))
            if r <= eps:
                r, c, s = T(0), T(1), T(0)
            else:
                c, s = T(a / r), T(-b / r)
            self.c[i], self.s[i] = c, s
            Tm[i, i], Tm[i + 1, i] = r, 0
            tmp = Tm[i, i + 1]
            Tm[i, i + 1] = T(c * tmp - s * Tm[i + 1, i + 1])
            Tm[i + 1, i + 1] = T(s * tmp + c * Tm[i + 1, i + 1])
            if i < n - 2:
                Tm[i, i + 2] = T(-s * Tm[i + 1, i + 2])
                Tm[i + 1, i + 2] = T(Tm[i + 1, i + 2] * c)
        self.R = Tm

    def apply_YQ(self, Y):
        """Y -> Y * G1 * G2 ... (UpperHessenbergQR.h apply_YQ)."""
        T = self.T
        for i in range(self.n - 1):
            c, s = self.c[i], self.s[i]
            yi = Y[:, i].copy()
            yi1 = Y[:, i + 1].copy()
            Y[:, i] = (c * yi - s * yi1).astype(T)
            Y[:, i + 1] = (s * yi + c * yi1).astype(T)

    def matrix_RQ(self):
        n, T = self.n, self.T
        RQ = np.zeros((n, n), dtype=T)
        idx = np.arange(n)
        RQ[idx, idx] = np.diag(self.R)
        RQ[idx[:-1], idx[:-1] + 1] = np.diag(self.R, 1)
        for i in range(n - 1):
            c, s = self.c[i], self.s[i]
            m11, m12 = RQ[i, i], RQ[i, i + 1]
            m21, m22 = RQ[i + 1, i], RQ[i + 1, i + 1]
            RQ[i, i] = T(c * m11 - s * m12)
            RQ[i + 1, i] = T(c * m21 - s * m22)
            RQ[i + 1, i + 1] = T(s * m21 + c * m22)
        RQ[idx[:-1], idx[:-1] + 1] = np.diag(RQ, -1)
        return RQ


def sym_eigs_largest(op, n, ncv=3, maxit=10, tol=0.1, dtype=np.float32, info=None):
    """Largest-algebraic Ritz value after `compute(maxit, tol)` with nev=1.

    `op(v)` returns A v for a length-n `dtype` vector.  Returns the value the
    reference reads as `evals[0]`.  If the Ritz value never passes the loose
    test the reference reads an empty vector (undefined behaviour,
    SymEigsSolver.h:612-618 vs ADMMLassoTall.h:200-201); this restatement then
    returns the last Ritz value and sets info['converged'] = False.
    """
    T = dtype
    nev = 1
    ncv = min(ncv, n)
    prec = T(np.finfo(T).eps ** (2.0 / 3.0))
    V = np.zeros((n, ncv), dtype=T)
    H = np.zeros((ncv, ncv), dtype=T)
    nmatop = 0

    # init(): SymEigsSolver.h:494-544
    v = simple_random_vec(n, 0, T)
    v = (v / T(np.linalg.norm(v))).astype(T)
    w = op(v).astype(T)
    nmatop += 1
    H[0, 0] = T(v @ w)
    f = (w - v * H[0, 0]).astype(T)
    V[:, 0] = v

    def factorize_from(from_k, to_m, fk):
        nonlocal f, nmatop
        if to_m <= from_k:
            return
        f = fk.astype(T)
        beta = T(np.linalg.norm(f))
        H[:, from_k:] = 0
        H[from_k:, :from_k] = 0
        for i in range(from_k, to_m):
            restart = False
            if beta < prec:
                f = simple_random_vec(n, 2 * i, T)
                Vi = V[:, :i]
                f = (f - Vi @ (Vi.T @ f)).astype(T)
                beta = T(np.linalg.norm(f))
                restart = True
            vi = (f / beta).astype(T)
            V[:, i] = vi
            H[i, i - 1] = T(0) if restart else beta
            w = op(vi).astype(T)
            nmatop += 1
            Hii = T(vi @ w)
            H[i - 1, i] = H[i, i - 1]
            H[i, i] = Hii
            if restart:
                f = (w - Hii * vi).astype(T)
            else:
                f = (w - H[i, i - 1] * V[:, i - 1] - Hii * vi).astype(T)
            beta = T(np.linalg.norm(f))
            Vi = V[:, :i + 1]
            Vf = (Vi.T @ f).astype(T)
            count = 0
            while count < 5 and np.abs(Vf).max() > prec * beta:
                f = (f - Vi @ Vf).astype(T)
                H[i - 1, i] = T(H[i - 1, i] + Vf[i - 1])
                H[i, i - 1] = H[i - 1, i]
                H[i, i] = T(H[i, i] + Vf[i])
                beta = T(np.linalg.norm(f))
                Vf = (Vi.T @ f).astype(T)
                count += 1

    ritz_val = np.zeros(ncv, dtype=T)
    ritz_est = np.zeros(ncv, dtype=T)

    def retrieve_ritzpair():
        evals, evecs = tridiag_eigen(H, T)
        # SortEigenvalue<LARGEST_ALGE>: std::sort on -val (ties: unspecified)
        ind = sorted(range(ncv), key=lambda k: -evals[k])
        for i in range(ncv):
            ritz_val[i] = evals[ind[i]]
            ritz_est[i] = evecs[ncv - 1, ind[i]]

    factorize_from(1, ncv, f)
    retrieve_ritzpair()
    nconv = 0
    nrestart = 0
    for _ in range(maxit):
        # num_converged(tol): :326-335
        thresh = T(tol) * max(abs(ritz_val[0]), prec)
        resid = abs(ritz_est[0]) * T(np.linalg.norm(f))
        nconv = int(resid < thresh)
        if nconv >= nev:
            break
        # nev_adjusted: :338-353
        nev_new = nev
        for i in range(nev, ncv):
            if abs(ritz_est[i]) < prec:
                nev_new += 1
        nev_new += min(nconv, (ncv - nev_new) // 2)
        if nev_new == 1 and ncv >= 6:
            nev_new = ncv // 2
        elif nev_new == 1 and ncv > 2:
            nev_new = 2
        # restart(k): :283-323
        k = nev_new
        nrestart += 1
        if k >= ncv:
            continue
        Q = np.eye(ncv, dtype=T)
        for i in range(k, ncv):
            H[np.arange(ncv), np.arange(ncv)] -= ritz_val[i]
            dec = _TridiagQR(H, T)
            dec.apply_YQ(Q)
            H[:, :] = dec.matrix_RQ()
            H[np.arange(ncv), np.arange(ncv)] += ritz_val[i]
        Vs = np.empty((n, k + 1), dtype=T)
        for i in range(k):
            nnz = ncv - k + i + 1
            Vs[:, i] = (V[:, :nnz] @ Q[:nnz, i]).astype(T)
        Vs[:, k] = (V @ Q[:, k]).astype(T)
        V[:, :k + 1] = Vs
        fk = (f * Q[ncv - 1, k - 1] + V[:, k] * H[k, k - 1]).astype(T)
        factorize_from(k, ncv, fk)
        retrieve_ritzpair()
    if info is not None:
        info.update(converged=bool(nconv >= nev), nmatop=nmatop, nrestart=nrestart)
    return T(ritz_val[0])
