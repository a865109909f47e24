"""CPU: the algebra behind admm_hip_parbp's Gram-space active-set iterations (admm_amd/csrc/sharing_bp.hip, "Gram space"), restated
in NumPy and held to the oracle (oracle/solvers.py SharingBP) decision by decision -- no GPU, no product code.

Between two regular iterations only the columns U that have been non-zero take part, and everything an iteration and its
stopping test need of A are inner products of those columns:
    A_U'r = G x_U / N - gz          G = A_U'A_U, gz = A_U'zbar
    A_U'y   by the recurrence  gy <- gy + rho A_U'r
    ||r+||^2 = ||r||^2 + 2 (A_U'r)'dx / N + dx'G dx / N^2,      ||y+||^2 = ||y||^2 + 2 rho y'r+ + rho^2 ||r+||^2,   y'r+ = gy'x+ / N - y'zbar
    dr'dS = dx'G dx / N,  ||dr||^2 = dx'G dx / N^2,  sum_i ||A_i dx_i||^2 = dx'G_blk dx,  sum_i ||A_i x_i||^2 = x'G_blk x,  abar'r = x'(A_U'r) / N
and the n-vectors follow once per stretch:  y <- y + rho (m r_last + A sum_t (x_t - x_last) / N).  A stretch that follows a
stretch carries x_U, A_U'r, A_U'y across the regular iteration (the regular iteration's x comes from the full transposed mat-vec);
the carried norms are re-anchored on the exact ||r||^2, ||y||^2, y'zbar of the materialised vectors.  This file is that scheme, with
NumPy's own summation orders: what the kernels compute up to rounding."""
import numpy as np
import pytest


class GramSpaceSharingBP:
    def __init__(self, A, b, N, eps_abs, eps_rel, sprad, rho_ratio, cap=10 ** 9):
        self.A = np.asarray(A, dtype=np.float64)
        self.n, self.p = self.A.shape
        self.N, self.eps_abs, self.eps_rel = int(N), eps_abs, eps_rel
        chunk = self.p // self.N
        self.off = [i * chunk for i in range(self.N)] + [self.p]
        self.blk = np.concatenate([np.full(self.off[i + 1] - self.off[i], i) for i in range(self.N)])
        self.sprad = np.asarray(sprad, dtype=np.float64)
        self.rho = 1.0 / (rho_ratio * float(np.mean(self.sprad)))
        self.gamma = 2.0 * self.rho + self.sprad[self.blk]                 # per column
        self.pen = 1.0 / (self.rho * self.gamma)
        self.zbar = np.asarray(b, dtype=np.float64) / self.N
        self.zz = float(self.zbar @ self.zbar)
        self.x = np.zeros(self.p)
        self.y = np.zeros(self.n); self.r = np.zeros(self.n)
        self.U = np.zeros(0, dtype=int)                                     # columns, in the order they appeared
        self.G = np.zeros((0, 0)); self.gz = np.zeros(0)
        self.trace = []

    # ---- the decision: thresholds of the coming iteration from the seven sums, and the verdict on the one just finished
    def _decide(self, it, sums, eps):
        drdS, dr2, r2, y2, abar_r, sax, qq = sums
        N, n = self.N, self.n
        conv = False
        if it > 0:
            sd = qq - 2.0 * drdS + N * dr2
            rp, rd = np.sqrt(N * r2), self.rho * np.sqrt(max(sd, 0.0))
            conv = rp < eps[0] and rd < eps[1]
            self.trace.append((it - 1, eps[0], eps[1], rp, rd, int((it - 1) % 10 == 0), int(conv)))
        sz = sax - 2.0 * N * abar_r + N * r2
        eps_p = self.eps_rel * np.sqrt(max(sax, sz, 0.0)) + np.sqrt(float(n * N)) * self.eps_abs
        eps_d = self.eps_rel * np.sqrt(float(N)) * np.sqrt(y2) + np.sqrt(float(n * N)) * self.eps_abs
        return conv, (eps_p, eps_d)

    def _soft(self, v, pen):
        return np.sign(v) * np.maximum(np.abs(v) - pen, 0.0)

    def _grow(self, cols):
        """Append the columns that are not in U yet: new rows / columns of G, gz; exact A_c'y, A_c'r for them (returned)."""
        new = np.array([c for c in cols if c not in set(self.U.tolist())], dtype=int)
        if new.size == 0:
            return new
        U2 = np.concatenate([self.U, new])
        G2 = np.zeros((U2.size, U2.size))
        G2[:self.U.size, :self.U.size] = self.G
        G2[:, self.U.size:] = self.A[:, U2].T @ self.A[:, new]
        G2[self.U.size:, :] = G2[:, self.U.size:].T
        self.G, self.U = G2, U2
        self.gz = np.concatenate([self.gz, self.A[:, new].T @ self.zbar])
        return new

    def _forms(self, xU, dx, hr_old, hr_new, gy_old):
        """The seven sums of the stopping test from Gram-space quantities (R, Y, yz are carried by the caller)."""
        N = self.N
        same = self.blk[self.U][:, None] == self.blk[self.U][None, :]
        Gb = np.where(same, self.G, 0.0)
        D = float(dx @ (self.G @ dx))
        return dict(D=D, hdx=float(hr_old @ dx), xh=float(xU @ hr_new), sax=float(xU @ (Gb @ xU)), qq=float(dx @ (Gb @ dx)),
                    gyx=float(gy_old @ xU), gzx=float(self.gz @ xU))

    def solve(self, maxit):
        N, rho = self.N, self.rho
        # iteration 0 (regular) .. : state of the "previous stretch" is all zero
        xU = np.zeros(0); hr = np.zeros(0); gy = np.zeros(0)
        R = Y = yz = 0.0                                                     # carried ||r||^2, ||y||^2, y'zbar
        eps = (0.0, 0.0)
        sums = (0.0,) * 7
        sx = None
        it = 0
        while True:
            conv, eps = self._decide(it, sums, eps)
            self.xU = xU
            if conv:
                return it
            if it >= maxit:
                return maxit + 1
            if it % 10 == 0:
                # ---- regular iteration: x of EVERY column from the n-vectors (materialised at the end of the previous stretch)
                if it > 0:
                    xl = self.x.copy(); xl[self.U] = xU                      # x_last of the stretch, dense
                    S = self.A[:, self.U] @ xU
                    r_last = S / N - self.zbar
                    m = 10 if it > 10 else 9                                 # the first stretch starts from the direct tail's vectors (9 iterates), carried ones sum 10
                    T = self.A[:, self.U] @ (sx - m * xU)
                    self.y = self.y + rho * (m * r_last + T / N)
                    self.r = r_last
                    self.x = xl
                    # re-anchor the carried norms on the exact values
                    R, Y, yz = float(self.r @ self.r), float(self.y @ self.y), float(self.y @ self.zbar)
                v = self.y / rho + self.r
                xnew = self._soft(self.x - (self.A.T @ v) / self.gamma, self.pen)
                if it == 0:
                    # the first regular iteration through n-space, as the direct launches do it: S, r, y exactly; then U, G, gy
                    self.x = xnew
                    nz = np.nonzero(self.x)[0]
                    self._grow(nz)
                    xU = self.x[self.U]
                    S = self.A[:, self.U] @ xU
                    Sb = [self.A[:, self.U[self.blk[self.U] == i]] @ xU[self.blk[self.U] == i] for i in range(N)]
                    rn = S / N - self.zbar
                    dr, dS = rn - self.r, S
                    sums = (float(dr @ dS), float(dr @ dr), float(rn @ rn), 0.0, float((S / N) @ rn), sum(float(s @ s) for s in Sb), sum(float(s @ s) for s in Sb))
                    self.r = rn
                    self.y = self.y + rho * rn
                    sums = sums[:3] + (float(self.y @ self.y),) + sums[4:]
                    R, Y, yz = float(self.r @ self.r), float(self.y @ self.y), float(self.y @ self.zbar)
                    hr = self.G @ xU / N - self.gz
                    gy = self.A[:, self.U].T @ self.y
                    sx = np.zeros(self.U.size)
                    it += 1
                    continue
                # ---- carried: the regular iteration's x enters Gram space (exact dots for the columns that are new to U)
                nz = np.nonzero(xnew)[0]
                nU = self.U.size
                new = self._grow(nz)
                x_old = np.concatenate([xU, np.zeros(new.size)])
                hr_old = np.concatenate([hr, self.A[:, new].T @ self.r]) if new.size else hr
                gy_old = np.concatenate([gy, self.A[:, new].T @ self.y]) if new.size else gy
                xU = xnew[self.U]
                assert np.count_nonzero(xnew) == np.count_nonzero(xU)
                sx = xU.copy()
                assert nU + new.size == self.U.size
            else:
                # ---- active-set iteration: only the current non-zeros move
                x_old, hr_old, gy_old = xU, hr, gy
                d = gy / rho + hr
                upd = self._soft(xU - d / self.gamma[self.U], self.pen[self.U])
                xU = np.where(x_old != 0.0, upd, 0.0)
                sx = sx + xU
            dx = xU - x_old
            hr = self.G @ xU / N - self.gz
            gy = gy_old + rho * hr
            f = self._forms(xU, dx, hr_old, hr, gy_old)
            Rn = max(R + 2.0 * f["hdx"] / N + f["D"] / (N * N), 0.0)
            yr = f["gyx"] / N - yz
            rz = f["gzx"] / N - self.zz
            Yn = max(Y + 2.0 * rho * yr + rho * rho * Rn, 0.0)
            R, Y, yz = Rn, Yn, yz + rho * rz
            sums = (f["D"] / N, f["D"] / (N * N), R, Y, f["xh"] / N, f["sax"], f["qq"])
            it += 1

    def get_x(self):
        x = self.x.copy()
        x[self.U] = self.xU
        return x


def _case(seed, n, p, k, scale=1.0):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((n, p)) * scale
    b0 = np.zeros(p); b0[rng.choice(p, k, replace=False)] = rng.standard_normal(k) * 3
    return A, A @ b0


@pytest.mark.parametrize("seed,n,p,k,N,eps", [(1, 40, 160, 5, 3, 1e-4), (2, 64, 301, 9, 4, 1e-4), (3, 30, 90, 4, 2, 1e-6), (4, 120, 700, 12, 6, 1e-4)])
def test_gram_space_recurrences_follow_the_oracle(seed, n, p, k, N, eps):
    from oracle.solvers import SharingBP
    A, b = _case(seed, n, p, k)
    ref = SharingBP(A, b, N, eps, eps)
    ref.init(1.0)
    ref.trace = []
    nref = ref.solve(10000)
    g = GramSpaceSharingBP(A, b, N, eps, eps, ref.sprad, 1.0)
    ngs = g.solve(10000)
    assert ngs == nref, (ngs, nref)
    a, r = np.asarray(g.trace, dtype=np.float64), np.asarray(ref.trace, dtype=np.float64)
    assert a.shape == r.shape
    assert np.array_equal(a[:, [0, 5, 6]], r[:, [0, 5, 6]])                  # iteration, regular / active-set schedule, verdict
    for col in range(1, 5):
        assert np.abs(a[:, col] - r[:, col]).max() < 1e-9 * np.abs(r[:, col]).max(), col
    xr = ref.get_x()
    assert np.abs(g.get_x() - xr).max() < 1e-10 * np.abs(xr).max() and np.array_equal(g.get_x() != 0, xr != 0)
