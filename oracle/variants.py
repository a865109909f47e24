"""Rounding variants of the tall x-update for the NumPy oracle -- TEST INFRASTRUCTURE.

The reference's x-update is `solver.solve(rhs)` with a float Eigen LLT (ADMMLassoTall.h:70-80,204-205).  The classes
below run the SAME algorithm (everything else inherited from oracle.solvers.LassoTall) with mathematically identical
x-updates that round differently:
   llt32    float LAPACK Cholesky + two float triangular solves   (= oracle.solvers.LassoTall, the oracle proper)
   inv32    float inverse formed from the float factor, float mat-vec        (libadmm_hip with ADMM_HIP_INVERSE=f32)
   inv64r   inverse formed in double, rounded to float once, float mat-vec   (libadmm_hip with ADMM_HIP_INVERSE=f64)
   exact    double Cholesky solve of the float system, result rounded to float (what the three above approximate)
They calibrate how far two correct implementations of the reference's arithmetic may drift apart: ADMM's stopping rule
and the restart rule are threshold tests on float residuals that, at eps = 1e-5, are mostly rounding noise, so the
iteration counts (and with them the stopping iterate) change when only the rounding of the solve changes.
Used by tests/test_flip_floor.py, tests/helpers.py and tests/tools/flip_floor.py.
"""
import contextlib

import numpy as np
import scipy.linalg as sla

from . import entry, solvers

F = np.float32
MODES = ("llt32", "inv32", "inv64r", "exact", "stats64", "xy64", "mv64")
#  mv64     wide solver: the oracle proper with its mat-vecs (X't of the x-update, A x) accumulated in double and rounded once -- "any other
#           order of a float dot product"; an unstandardised n = 21, p = 269 problem at scale 0.01 moves a 252-coefficient column by 2 % on it
#           within 20 iterations (out-of-sample soak 824:130)
#  xy64     the oracle proper with X'y (consensus: every A_k'b_k) accumulated in double and rounded once: stands for "any
#           other summation order" of that product (the reference's is Eigen's, NumPy's and libadmm_hip's are their own);
#           an unconverged, ill-conditioned consensus path moved a coefficient column by 3e-4 on it (soak case 505:139)
#  stats64  the oracle proper (llt32) with the column statistics of DataStd accumulated in double and rounded once
#           (libadmm_hip's prep kernels) instead of in float (the reference, in Eigen's order): oracle/datastd.py


class LassoTallVariant(solvers.LassoTall):
    mode = "llt32"

    def init(self, lam, rho):
        super().init(lam, rho)
        if self.mode == "llt32":
            return
        p = self.p
        XX = (self.X.T @ self.X).astype(F)                          # the float system the reference factorises
        XX[np.arange(p), np.arange(p)] += F(self.rho)
        A64 = XX.astype(np.float64)
        if self.mode == "inv32":
            L = np.tril(self.chol[0])
            Li = sla.solve_triangular(L, np.eye(p, dtype=F), lower=True, check_finite=False).astype(F)
            self.Minv = (Li.T @ Li).astype(F)
        elif self.mode == "inv64r":
            self.Minv = np.linalg.inv(A64).astype(F)
        elif self.mode == "exact":
            self.chol64 = sla.cho_factor(A64, lower=True)
        else:
            raise ValueError(self.mode)

    def next_x(self):                                               # ADMMLassoTall.h:70-80
        if self.mode == "llt32":
            return super().next_x()
        rhs = (self.XY - self.adj_y).astype(F)
        rhs = (rhs.astype(np.float64) + self.rho * self.adj_z.astype(np.float64)).astype(F)
        if self.mode == "exact":
            return sla.cho_solve(self.chol64, rhs.astype(np.float64), check_finite=False).astype(F)
        return (self.Minv @ rhs).astype(F)


@contextlib.contextmanager
def tall_variant(mode):
    """Within the block oracle.entry's tall solver -- and the consensus solver's workers (llt32 / inv32 / exact; inv64r
    runs as inv32 there) -- use the given x-update rounding."""
    assert mode in MODES
    from .datastd import DataStd
    xmode = "llt32" if mode in ("stats64", "xy64", "mv64") else mode
    cls = type("LassoTall_" + xmode, (LassoTallVariant,), {"mode": xmode})
    orig = entry.LassoTall
    orig_par = solvers.PADMMLasso.xmode
    orig_acc = DataStd.acc
    entry.LassoTall = cls
    solvers.PADMMLasso.xmode = {"llt32": "llt32", "exact": "exact"}.get(xmode, "inv32")
    if mode == "stats64":
        DataStd.acc = np.float64
    if mode == "xy64":
        solvers.LassoTall.xy_acc = np.float64
        solvers.PADMMLasso.xy_acc = np.float64
    if mode == "mv64":                                   # wide solver: X't and A x accumulated in double (the float dot's order is the implementation's)
        solvers.LassoWide.mv_acc = np.float64
    try:
        yield
    finally:
        entry.LassoTall = orig
        solvers.PADMMLasso.xmode = orig_par
        DataStd.acc = orig_acc
        solvers.LassoTall.xy_acc = None
        solvers.PADMMLasso.xy_acc = None
        solvers.LassoWide.mv_acc = None
