"""DataStd<Scalar> restated (TEST INFRASTRUCTURE). /root/reference/src/DataStd.h:76-209."""
import numpy as np


class DataStd:
    """flag = standardize + 2*intercept (DataStd.h:21-29,77-78).

    `acc`: the type the column sums / sums of squares ACCUMULATE in.  The reference accumulates in Scalar (float for the
    Lasso family: Eigen's vectorised `mean()` / `norm()`, DataStd.h:39-53,104,133), in an order Eigen chooses; NumPy's float
    pairwise sums stand in for it here.  libadmm_hip accumulates in double and rounds the statistic to float once
    (prep.hip, colstat_kernel).  `acc = np.float64` is that second ROUNDING VARIANT of the same arithmetic
    (oracle/variants.py "stats64"): on unstandardised data with large column means the two differ by a few 1e-5 of a mean,
    and the intercept meanY - sum_j beta_j meanX_j turns that into 1e-4 of the largest coefficient (the soak's case 522:126)."""
    acc = None      # None: accumulate in the data type T

    def __init__(self, n, p, standardize, intercept, dtype=np.float32):
        self.flag = int(bool(standardize)) + 2 * int(bool(intercept))
        self.n, self.p, self.T = n, p, dtype
        self.meanY = dtype(0.0)
        self.scaleY = dtype(1.0)
        self.meanX = np.zeros(p, dtype=dtype)
        self.scaleX = np.ones(p, dtype=dtype)

    def _sd_n(self, v, T):
        # sd_n (:39-53, non-AVX branch): ||v - mean|| / sqrt(n)
        A = self.acc or T
        mean = T(v.mean(dtype=A))
        vc = (v - mean).astype(T)
        return T(T(np.sqrt((vc.astype(A) ** 2).sum(dtype=A))) / T(np.sqrt(T(v.size))))

    def standardize(self, X, Y):
        """In place on X (n x p, dtype) and Y (n). DataStd.h:89-155."""
        T, n, flag = self.T, self.n, self.flag
        A = self.acc or T
        n_invsqrt = T(1.0 / np.sqrt(T(n)))
        if flag == 1:
            self.scaleY = self._sd_n(Y, T)
            Y /= self.scaleY
        elif flag in (2, 3):
            self.meanY = T(Y.mean(dtype=A))
            Y -= self.meanY
            self.scaleY = T(T(np.linalg.norm(Y) if A is T else np.sqrt((Y.astype(A) ** 2).sum(dtype=A))) * n_invsqrt)
            Y /= self.scaleY
        if flag == 1:
            for i in range(self.p):
                self.scaleX[i] = self._sd_n(X[:, i], T)
                X[:, i] *= T(1.0 / self.scaleX[i])
        elif flag == 2:
            self.meanX[:] = X.mean(axis=0, dtype=A)
            X -= self.meanX[None, :]
        elif flag == 3:
            # column loop of :130-150, vectorised over columns (same per-column arithmetic)
            self.meanX[:] = X.mean(axis=0, dtype=A)
            X -= self.meanX[None, :]
            self.scaleX[:] = (np.sqrt((X.astype(A) * X.astype(A)).sum(axis=0, dtype=A)).astype(T) * n_invsqrt).astype(T)
            X *= (T(1.0) / self.scaleX)[None, :]

    def recover(self, coef):
        """Return (beta0, coef on the original scale). DataStd.h:157-207."""
        T, flag = self.T, self.flag
        coef = coef.astype(T).copy()
        beta0 = T(0)
        if flag == 1:
            coef = (coef / self.scaleX * self.scaleY).astype(T)
        elif flag == 2:
            coef = (coef * self.scaleY).astype(T)
            beta0 = T(self.meanY - T((coef * self.meanX).sum(dtype=T)))
        elif flag == 3:
            coef = (coef / self.scaleX * self.scaleY).astype(T)
            beta0 = T(self.meanY - T((coef * self.meanX).sum(dtype=T)))
        return beta0, coef
