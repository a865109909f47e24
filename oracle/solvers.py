"""ADMM drivers and problem classes of the reference, restated (TEST INFRASTRUCTURE).

Follows, line by line where arithmetic order matters:
  FADMMBase::solve            /root/reference/src/FADMMBase.h:185-265
  ADMMBase::solve/update_rho  /root/reference/src/ADMMBase.h:85-109,158-216
  ADMMLassoTall               /root/reference/src/ADMMLassoTall.h:55-231
  ADMMEnetTall / ADMMEnetWide /root/reference/src/ADMMEnet.h:19-154
  ADMMLassoWide               /root/reference/src/ADMMLassoWide.h:70-251
  PADMMBase / PADMMLasso      /root/reference/src/PADMMBase.h:57-237, PADMMLasso.h:17-223
  ADMMLAD                     /root/reference/src/ADMMLAD.h:62-225
  ADMMBP                      /root/reference/src/ADMMBP.h:48-197
Sparse vectors of the reference are kept dense here (they are a storage detail:
every entry the reference leaves out is an exact zero).  float32 where the
reference uses `float`, float64 scalars where it uses `double`.
"""
import numpy as np
import scipy.linalg as sla

from .spectra import sym_eigs_largest

F = np.float32


def _rho_rule(s):
    """update_rho(): FADMMBase.h:109-133 == ADMMBase.h:85-109."""
    if s.resid_primal / s.eps_primal > 10 * s.resid_dual / s.eps_dual:
        s.rho *= 2
    elif s.resid_dual / s.eps_dual > 10 * s.resid_primal / s.eps_primal:
        s.rho /= 2
    if s.resid_primal < s.eps_primal:
        s.rho /= 1.2
    if s.resid_dual < s.eps_dual:
        s.rho *= 1.2


def _soft_d(vec, penalty, T):
    """soft_threshold with a double penalty compared against T entries
    (ADMMLassoTall.h:55-69, ADMMLassoWide.h:70-84, PADMMLasso.h:83-97, ADMMLAD.h:79-93)."""
    v = vec.astype(np.float64)
    out = np.where(v > penalty, v - penalty, np.where(v < -penalty, v + penalty, 0.0))
    return out.astype(T)


def _enet_f(vec, penalty, alpha):
    """enet(): ADMMEnet.h:24-40 / :67-83 -- thresh and denom are floats."""
    penalty = np.float64(penalty)           # (a Python float would make NumPy 2 form the products in float32)
    thresh = F(np.float64(F(alpha)) * penalty)          # Scalar thresh = alpha * penalty (double product -> float)
    denom = F(1.0 + penalty * (1.0 - np.float64(F(alpha))))
    v = vec
    out = np.where(v > thresh, (v - thresh) / denom, np.where(v < -thresh, (v + thresh) / denom, F(0)))
    return out.astype(F)


def _sqnorm(v, T):
    return T((v * v).sum(dtype=T))


class FollowMismatch(AssertionError):
    """Follow mode: the followed execution took a decision this run cannot explain as a near-tie."""


def _ulp_norm(v, T, extra=0.0):
    """|| ulp(|v| + extra) ||_2: what a vector's norm-type functionals move by when every entry moves by one ulp."""
    return float(np.linalg.norm(np.spacing((np.abs(v) + T(extra)).astype(T)).astype(np.float64)))


def _stop_ulps(rp, eps_p, n_p, rd, eps_d, n_d, own_converged):
    """How many ulps of rounding (of the iterates) separate this run's stopping test from the opposite outcome."""
    tiny = 1e-300
    gp, gd = rp - eps_p, rd - eps_d                      # >= 0 means "not converged"
    if not own_converged:                                 # they stopped, this run did not: both residuals must come under
        return max(gp / max(n_p, tiny), gd / max(n_d, tiny), 0.0)
    return min(-gp / max(n_p, tiny), -gd / max(n_d, tiny))    # this run stopped, they did not: one residual must reach its threshold


def _rho_candidates(rho, rp, eps_p, n_p, rd, eps_d, n_d, band):
    """Every rho the adaptation rule (FADMMBase.h:109-133 == ADMMBase.h:85-109) can produce when r_p and r_d move by up to
    `band` ulps of rounding each (the rule is monotone in both, so the corners suffice)."""
    out = set()

    class _S:
        pass
    # the box [rp -+ band n_p] x [rd -+ band n_d] (clamped at 0): its corners, and -- the rule's regions are not nested, a large box
    # can contain a region none of its corners lies in (e.g. "balanced and r_d < eps_d", x1.2, between "r_p far too large", x2, and
    # "r_d far too large", x0.5) -- a grid over it that is geometric around both thresholds
    lo_p, hi_p = max(rp - band * n_p, 0.0), rp + band * n_p
    lo_d, hi_d = max(rd - band * n_d, 0.0), rd + band * n_d
    def axis(lo, hi, centre, thr):
        pts = {lo, hi, min(max(centre, lo), hi)}
        for f in (0.05, 0.2, 0.5, 0.9, 0.999, 1.001, 1.1, 2.0, 5.0, 20.0):
            for base in (thr, centre):
                v = base * f
                if lo <= v <= hi:
                    pts.add(v)
        return sorted(pts)
    for vp in axis(lo_p, hi_p, rp, eps_p):
        for vd in axis(lo_d, hi_d, rd, eps_d):
            t = _S()
            t.rho, t.eps_primal, t.eps_dual = rho, eps_p, eps_d
            t.resid_primal, t.resid_dual = vp, vd
            _rho_rule(t)
            out.add(t.rho)
    return out


class FADMM:
    """Goldstein fast ADMM with restart; subclasses define next_x/next_z/residual."""

    update_rho_active = True
    trace = None     # set to a list to collect one record per iteration, in the layout of include/admm_hip.h (ADMM_TRACE_*)
    lam_idx = 0
    # Follow mode (tests only): `follow` = iterator over the decision records of ANOTHER execution of this algorithm
    # (libadmm_hip's trace, cold-start record removed).  Wherever this run's own threshold test disagrees with the
    # followed one AND the disagreement is a near-tie that rounding decides, the followed outcome is taken and logged in
    # `forced`; any other disagreement raises FollowMismatch.  The two executions then stay on one trajectory and can be
    # compared column by column.  "Near-tie" is measured against the rounding noise of the deciding quantity itself, in
    # units of float ulps of the iterates (`_noise`): a decision may be taken from the followed run only if moving every
    # entry of x and z by at most `follow_band` ulps could have produced it here.
    follow = None
    follow_band = 8.0
    forced = None
    ndecisions = 0
    _followed = None
    state_log = None  # set to a list to collect (x, z, y, adj_z, adj_y) of every iteration, like admm_hip_lasso_plan_state_* (oracle/stepcheck.py)

    def _trace(self, i, c, c_old, code, rho_in):
        if self.trace is not None:
            self.trace.append((self.lam_idx, i, self.eps_primal, self.eps_dual, self.resid_primal, self.resid_dual,
                               c, c_old, code, rho_in, self.rho, 0.0))

    def _noise(self):
        """One-ulp noise floors of (resid_primal, resid_dual, c): what the three quantities move by when every entry of
        main_x and aux_z moves by one float ulp.  aux_z comes out of a soft-threshold, so its entries carry the ulp of
        the thresholded value |z| + penalty (an entry may sit at 0 in one execution and just beyond in another)."""
        T = self.T
        pen = abs(float(getattr(self, "lam", 0.0))) / self.rho if hasattr(self, "lam") else 1.0 / self.rho
        uz = float(np.linalg.norm(np.spacing((np.abs(self.aux_z) + T(pen)).astype(T)).astype(np.float64)))
        ux = float(np.linalg.norm(np.spacing(np.abs(self.main_x).astype(T)).astype(np.float64)))
        # the x-update is a linear solve: its rounding error is cond(M) ulps of x, not one (`_solve_noise`, measured); z is a
        # soft-threshold of x + y/rho, so it carries the same error on its support
        ex = self._solve_noise() / self.follow_band if self.follow_band > 0 else 0.0
        ux, uz = np.hypot(ux, ex), np.hypot(uz, ex)
        n_p = np.hypot(ux, uz)
        n_d = self.rho * 2.0 * uz                                   # z - old_z: both move
        daz = float(np.sqrt(np.float64(_sqnorm(self.aux_z - self.adj_z, T))))
        n_c = 2.0 * self.rho * (self.resid_primal * n_p + daz * 3.0 * uz)     # adj_z = (1+t) z - t z_old: up to 3 ulps of z
        # c = rho (||r||^2 + ||z - adj_z||^2) is QUADRATIC in the perturbation: where r and z - adj_z vanish exactly (x = z in
        # float: every |y / rho| and lambda / rho below half an ulp of x) the first-order term is zero and b ulps move c by
        # rho (n_p^2 + 9 uz^2) b^2
        self._n_c2 = self.rho * (n_p * n_p + 9.0 * uz * uz)
        return n_p, n_d, n_c

    def _solve_noise(self):
        """|| x - x_exact ||_2 of the x-update just made: the distance between this run's float solve and the same system
        solved in double (rounded to float) -- ONE realisation of the rounding error any float implementation of the
        reference's x-update makes on this right-hand side (another correct implementation makes an error of the same
        size in another direction).  Counted ONCE, not scaled by the band (`_noise` divides it out): a decision may be taken
        from the followed run if moving x by its own measured solve error, plus `follow_band` ulps of every entry, reaches
        it.  0 where the x-update is not a solve whose error is measured (LAD / BP: float64, never needed)."""
        return 0.0

    def _decide(self, i, own, c, c_old):
        """own: this run's outcome (0 converged, 1 accelerate, 2 restart).  Returns the outcome to act on."""
        self.ndecisions += 1
        self._followed = None
        if self.follow is None:
            return own
        g = next(self.follow)
        self._followed = g
        if int(g[0]) != self.lam_idx or int(g[1]) != i:
            raise FollowMismatch(f"followed trace is at (lambda {int(g[0])}, iteration {int(g[1])}), this run at ({self.lam_idx}, {i})")
        theirs = int(g[8])
        if theirs == own:
            return own
        n_p, n_d, n_c = self._noise()
        tiny = 1e-300
        if (theirs == 0) != (own == 0):
            kind = "stop"
            gp, gd = self.resid_primal - self.eps_primal, self.resid_dual - self.eps_dual      # >= 0 means "not converged"
            if own != 0:      # they stopped, this run did not: both residuals must come under their thresholds within noise
                ulps = max(gp / max(n_p, tiny), gd / max(n_d, tiny), 0.0)
            else:             # this run stopped, they did not: one residual must reach its threshold within noise
                ulps = min(-gp / max(n_p, tiny), -gd / max(n_d, tiny))
        else:
            kind = "restart"
            gap, a2 = abs(c - 0.999 * c_old), self._n_c2             # c_old carries the same noise from the previous iteration:
            if a2 * gap > 1e-6 * n_c * n_c:                          # 2 (n_c b + a2 b^2) = gap
                ulps = (-n_c + np.sqrt(n_c * n_c + 2.0 * a2 * gap)) / (2.0 * a2)
            else:
                ulps = gap / max(2.0 * n_c, tiny)
        rec = dict(record=self.ndecisions - 1, lam=self.lam_idx, iter=i, kind=kind, ulps=float(ulps), own=own, theirs=theirs)
        if ulps > self.follow_band:
            raise FollowMismatch(f"decision differs beyond rounding noise: {rec}")
        self.forced.append(rec)
        return theirs

    def _init_accel(self):
        self.adj_a = 1.0
        self.adj_c = 9999.0

    def solve(self, maxit):
        T = self.T
        i = 0
        for i in range(maxit):
            old_z = self.aux_z.copy()
            old_y = self.dual_y.copy()
            # update_x (:185-194): eps from the *current* iterate, then x
            self.eps_primal = self.compute_eps_primal()
            self.eps_dual = self.compute_eps_dual()
            self.main_x = self.next_x()
            # update_z (:195-202)
            self.aux_z = self.next_z()
            self.resid_dual = self.rho * np.sqrt(np.float64(_sqnorm(self.aux_z - old_z, T)))
            # update_y (:203-211)
            r = self.next_residual()
            self.resid_primal = np.float64(T(np.linalg.norm(r)))
            self.dual_y = (self.adj_y + T(self.rho) * r).astype(T)
            if self.state_log is not None:
                self.state_log.append(np.concatenate([self.main_x, self.aux_z, self.dual_y, self.adj_z, self.adj_y]).astype(T))
            converged = self.resid_primal < self.eps_primal and self.resid_dual < self.eps_dual     # :213-217
            old_c = self.adj_c
            c = (self.rho * self.resid_primal * self.resid_primal
                 + self.rho * np.float64(_sqnorm(self.aux_z - self.adj_z, T)))                       # :100-107, evaluated at :241
            own = 0 if converged else (1 if c < 0.999 * old_c else 2)
            rho_in = self.rho
            code = self._decide(i, own, c, old_c)
            if code == 0:
                self._trace(i, 0.0, old_c, own, rho_in)
                return i + 1                                                                        # :237-238
            self.adj_c = c
            if code == 1:                                                                           # :243-249
                old_a = self.adj_a
                self.adj_a = 0.5 + 0.5 * np.sqrt(1 + 4.0 * old_a * old_a)
                ratio = (old_a - 1.0) / self.adj_a
                self.adj_z = (T(1 + ratio) * self.aux_z - T(ratio) * old_z).astype(T)
                self.adj_y = (T(1 + ratio) * self.dual_y - T(ratio) * old_y).astype(T)
            else:                                                                                   # :250-256
                self.adj_a = 1.0
                self.adj_z = old_z
                self.adj_y = old_y
                self.adj_c = old_c / 0.999
            if i > 5 and self.update_rho_active:
                _rho_rule(self)
                g = self._followed
                if g is not None and abs(self.rho / rho_in - g[10] / g[9]) > 1e-9:        # compare the multiplier applied (ADMMPlain.solve)
                    n_p, n_d, _ = self._noise()
                    cands = _rho_candidates(rho_in, self.resid_primal, self.eps_primal, n_p, self.resid_dual, self.eps_dual, n_d, self.follow_band)
                    if not any(abs(cd / rho_in - g[10] / g[9]) <= 1e-9 for cd in cands):
                        raise FollowMismatch(f"rho adaptation differs beyond rounding noise at iteration {i}: x{self.rho / rho_in} here, "
                                             f"x{g[10] / g[9]} followed, reachable {sorted(cd / rho_in for cd in cands)}")
                    self.forced.append(dict(record=self.ndecisions - 1, lam=self.lam_idx, iter=i, kind="rho", ulps=float(self.follow_band)))
                    self.rho = rho_in * float(g[10] / g[9])
            self._trace(i, c, old_c, own, rho_in)
        return maxit + 1          # `return i + 1` after the loop ran out (FADMMBase.h:264)


class LassoTall(FADMM):
    T = F
    update_rho_active = False    # ADMMLassoTall.h:97  void update_rho() {}
    xy_acc = None

    def __init__(self, X, Y, eps_abs, eps_rel, alpha=None):
        self.X, self.Y = X, Y
        self.p = X.shape[1]
        self.eps_abs, self.eps_rel = eps_abs, eps_rel
        # :172.  `xy_acc` (class attribute, oracle/variants.py "xy64"): the type X'y ACCUMULATES in -- float in the reference, in
        # Eigen's order; np.float64 is the rounding variant "any other summation order" (libadmm_hip sums in its own)
        self.XY = (X.T @ Y).astype(F) if self.xy_acc is None else (X.astype(self.xy_acc).T @ Y.astype(self.xy_acc)).astype(F)
        self.lambda0 = F(np.abs(self.XY).max())             # :173
        self.alpha = alpha
        if alpha is not None:
            self.alpha = F(alpha)
            self.lambda0 = F(self.lambda0 / (np.float64(self.alpha) + 0.0001))   # ADMMEnet.h:56
        self.info = {}

    def init(self, lam, rho):
        p = self.p
        self.main_x = np.zeros(p, F)
        self.aux_z = np.zeros(p, F)
        self.dual_y = np.zeros(p, F)
        self.adj_z = np.zeros(p, F)
        self.adj_y = np.zeros(p, F)
        self.lam = F(lam)
        self.rho = float(rho)
        XX = (self.X.T @ self.X).astype(F)                  # cross_prod_lower, :191-192
        if self.rho <= 0:                                   # :194-202
            ev = sym_eigs_largest(lambda v: XX @ v, p, 3, 10, 0.1, F, self.info)
            self.lmax_est = ev
            self.rho = float(np.float64(ev) ** (1.0 / 3) * np.float64(self.lam) ** (2.0 / 3))
        XX[np.arange(p), np.arange(p)] += F(self.rho)
        self.chol = sla.cho_factor(XX, lower=True, check_finite=False)   # :204-205
        self.eps_primal = self.eps_dual = 0.0
        self.resid_primal = self.resid_dual = 9999.0
        self._init_accel()

    def init_warm(self, lam):                               # :219-230 (a, c deliberately kept)
        self.lam = F(lam)
        self.eps_primal = self.eps_dual = 0.0
        self.resid_primal = self.resid_dual = 9999.0

    def compute_eps_primal(self):                           # :141-145
        r = max(np.float64(F(np.linalg.norm(self.main_x))), np.float64(F(np.linalg.norm(self.aux_z))))
        return r * self.eps_rel + np.sqrt(float(self.p)) * self.eps_abs

    def compute_eps_dual(self):                             # :146-149
        return np.float64(F(np.linalg.norm(self.dual_y))) * self.eps_rel + np.sqrt(float(self.p)) * self.eps_abs

    def next_x(self):                                       # :70-80
        rhs = (self.XY - self.adj_y).astype(F)
        rhs = (rhs.astype(np.float64) + self.rho * self.adj_z.astype(np.float64)).astype(F)
        return sla.cho_solve(self.chol, rhs, check_finite=False).astype(F)

    def _solve_noise(self):
        """FADMM._solve_noise for x = (X'X + rho I)^-1 rhs: the float system (as the reference factorises it) solved in
        double on the float right-hand side of this iteration, against the x this run computed."""
        if getattr(self, "_chol64_rho", None) != self.rho:
            XX = (self.X.T @ self.X).astype(F).astype(np.float64)
            XX[np.arange(self.p), np.arange(self.p)] += np.float64(F(self.rho))
            self._chol64 = sla.cho_factor(XX, lower=True, check_finite=False)
            self._chol64_rho = self.rho
        rhs = (self.XY - self.adj_y).astype(F)
        rhs = (rhs.astype(np.float64) + self.rho * self.adj_z.astype(np.float64)).astype(F)
        # adj_z / adj_y still hold the values this iteration's x-update used (they are replaced after the decision)
        xe = sla.cho_solve(self._chol64, rhs.astype(np.float64), check_finite=False)
        return float(np.linalg.norm(self.main_x.astype(np.float64) - xe))

    def next_z(self):                                       # :81-85 / ADMMEnet.h:41-45
        vec = (self.main_x + self.adj_y / F(self.rho)).astype(F)
        pen = np.float64(self.lam) / self.rho
        if self.alpha is None:
            return _soft_d(vec, pen, F)
        return _enet_f(vec, pen, self.alpha)

    def next_residual(self):                                # :86-95
        return (self.main_x - self.aux_z).astype(F)

    def get_coef(self):
        return self.aux_z                                   # Lasso.cpp:108 get_z()


class ADMMPlain:
    """ADMMBase::solve, ADMMBase.h:192-216 (used by the wide solver).  `trace` / `follow` as in FADMM: records in the
    layout of include/admm_hip.h (wide flavour: [6] rho after the adaptation, [8] 0 converged / 1 continue, [9] rho before);
    in follow mode the followed run's stopping outcome and adapted rho are taken only where this run's own residuals are
    within `follow_band` ulps of rounding of producing them."""
    trace = None
    follow = None
    follow_band = 8.0
    forced = None
    ndecisions = 0
    lam_idx = 0
    state_log = None  # set to a list to collect (x, Ax, z, y) of every iteration, like admm_hip_lasso_plan_state_* of the wide solver (oracle/stepcheck.py)
    type_log = None   # set to a list to collect the kind of every x-update (0 zero, 1 regular, 2 active set: trace field 7 of libadmm_hip)

    def _noise(self):
        """One-ulp noise floors of r_p = ||Ax + z|| and r_d = rho sqrt(gamma) ||z_new - z|| (ADMMLassoWide.h:166-186)."""
        uax = _ulp_norm(self.cache_Ax, F)
        uz = _ulp_norm(self.aux_z, F)
        return np.hypot(uax, uz), self.rho * np.float64(F(np.sqrt(self.sprad))) * 2.0 * uz

    def solve(self, maxit):
        for i in range(maxit):
            self.eps_primal = self.compute_eps_primal()
            self.eps_dual = self.compute_eps_dual()
            self.main_x = self.next_x()
            newz = self.next_z()
            self.resid_dual = self.compute_resid_dual(newz)     # before the swap (:167-175)
            self.aux_z = newz
            r = self.next_residual()
            self.resid_primal = np.float64(F(np.linalg.norm(r)))
            self.dual_y = (self.dual_y + F(self.rho) * r).astype(F)
            if self.state_log is not None:
                self.state_log.append(np.concatenate([self.main_x, self.cache_Ax, self.aux_z, self.dual_y]).astype(F))
            converged = self.resid_primal < self.eps_primal and self.resid_dual < self.eps_dual
            rho_in = self.rho
            self.ndecisions += 1
            g = None
            if self.follow is not None:
                g = next(self.follow)
                if int(g[0]) != self.lam_idx or int(g[1]) != i:
                    raise FollowMismatch(f"followed trace is at (lambda {int(g[0])}, iteration {int(g[1])}), this run at ({self.lam_idx}, {i})")
                n_p, n_d = self._noise()
                if (int(g[8]) == 0) != converged:
                    ulps = _stop_ulps(self.resid_primal, self.eps_primal, n_p, self.resid_dual, self.eps_dual, n_d, converged)
                    rec = dict(record=self.ndecisions - 1, lam=self.lam_idx, iter=i, kind="stop", ulps=float(ulps))
                    if ulps > self.follow_band:
                        raise FollowMismatch(f"decision differs beyond rounding noise: {rec}")
                    self.forced.append(rec)
                    converged = int(g[8]) == 0
            if converged:
                if self.trace is not None:
                    self.trace.append((self.lam_idx, i, self.eps_primal, self.eps_dual, self.resid_primal, self.resid_dual, self.rho, 0, 0, rho_in, self.rho, 0.0))
                return i + 1
            if i > 3:
                _rho_rule(self)
                # the two executions' rho differ in the last digits from the start (their Lanczos values do): compare the
                # MULTIPLIER this decision applied (1, 2, 1/2, 1.2, 1/1.2 and their products), not rho itself
                if g is not None and abs(self.rho / rho_in - g[6] / g[9]) > 1e-9:
                    cands = _rho_candidates(rho_in, self.resid_primal, self.eps_primal, n_p, self.resid_dual, self.eps_dual, n_d, self.follow_band)
                    if not any(abs(c / rho_in - g[6] / g[9]) <= 1e-9 for c in cands):
                        raise FollowMismatch(f"rho adaptation differs beyond rounding noise at (lambda {self.lam_idx}, iteration {i}): "
                                             f"x{self.rho / rho_in} here, x{g[6] / g[9]} followed, reachable {sorted(c / rho_in for c in cands)}")
                    self.forced.append(dict(record=self.ndecisions - 1, lam=self.lam_idx, iter=i, kind="rho", ulps=float(self.follow_band)))
                    self.rho = rho_in * float(g[6] / g[9])
            if self.trace is not None:
                self.trace.append((self.lam_idx, i, self.eps_primal, self.eps_dual, self.resid_primal, self.resid_dual, self.rho, 0, 1, rho_in, self.rho, 0.0))
        return maxit + 1


def is_regular_update(x):
    """4^k - 1 (ADMMLassoWide.h:121-127)."""
    if x in (0, 3, 15, 63):
        return True
    x += 1
    if x & (x - 1):
        return False
    return bool(x & 0x55555555)


class LassoWide(ADMMPlain):
    def __init__(self, X, Y, eps_abs, eps_rel, alpha=None):
        self.X, self.Y = X, Y
        self.n, self.p = X.shape
        self.eps_abs, self.eps_rel = eps_abs, eps_rel
        self.lambda0 = F(np.abs((X.T @ Y).astype(F)).max())                 # :197
        XXt = (X @ X.T).astype(F)                                           # tcross_prod_lower :200-201
        self.info = {}
        self.sprad = F(sym_eigs_largest(lambda v: XXt @ v, self.n, 3, 10, 0.1, F, self.info))
        self.alpha = alpha
        if alpha is not None:
            self.alpha = F(alpha)
            self.lambda0 = F(self.lambda0 / (np.float64(self.alpha) + 0.0001))   # ADMMEnet.h:152
        self.trace_nnz = []

    mv_acc = None                                                           # rounding variant "mv64" (oracle/variants.py): the mat-vecs accumulated in double
    # Follow mode for the PROX (tests/helpers.py, with the followed run's iterate dump): the active-set schedule makes the path depend
    # on whether a coordinate is EXACTLY zero after a step -- a zero stays out until the next regular step -- so a soft-threshold
    # operand within rounding of its threshold is a fork just like a stopping near-tie (out-of-sample soak 824:130: one of 269
    # coordinates on the threshold at the first regular step of a lambda, 2 % of the column 20 iterations later).  follow_x: iterator
    # over the followed run's x after every iteration; where its zero pattern differs from this run's and the coordinate is within
    # follow_band yardsticks of the float dot product X_j't (plus an ulp of x_j) of the threshold, this run takes the followed value.
    follow_x = None

    def _mv(self, A, v):
        """A v rounded to float: float accumulation (the reference's Eigen products) unless the mv64 variant is on."""
        if self.mv_acc is None:
            return (A @ v).astype(F)
        return (A.astype(self.mv_acc) @ v.astype(self.mv_acc)).astype(F)

    def init(self, lam, rho):                                               # :215-237
        self.main_x = np.zeros(self.p, F)
        self.cache_Ax = np.zeros(self.n, F)
        self.aux_z = np.zeros(self.n, F)
        self.dual_y = np.zeros(self.n, F)
        self.lam = F(lam)
        self.rho = float(rho)
        if self.rho <= 0:
            self.rho = float((np.float64(self.lam) / np.float64(self.sprad)) ** (1.0 / 3))
        self.eps_primal = self.eps_dual = 0.0
        self.resid_primal = self.resid_dual = 9999.0
        self.iter_counter = 0

    def init_warm(self, lam):                                               # :241-251
        self.lam = F(lam)
        self.eps_primal = self.eps_dual = 0.0
        self.resid_primal = self.resid_dual = 9999.0
        self.iter_counter = 0

    def compute_eps_primal(self):                                           # :174-178
        r = max(np.float64(F(np.linalg.norm(self.cache_Ax))), np.float64(F(np.linalg.norm(self.aux_z))))
        return r * self.eps_rel + np.sqrt(float(self.n)) * self.eps_abs

    def compute_eps_dual(self):                                             # :179-182
        return (np.float64(F(np.sqrt(self.sprad))) * np.float64(F(np.linalg.norm(self.dual_y))) * self.eps_rel
                + np.sqrt(float(self.p)) * self.eps_abs)

    def compute_resid_dual(self, new_z):                                    # :183-186
        return self.rho * np.float64(F(np.sqrt(self.sprad))) * np.float64(F(np.linalg.norm(new_z - self.aux_z)))

    def _prox(self, v, penalty_d):
        if self.alpha is None:
            return _soft_d(v, penalty_d, F)
        return _enet_f(v, penalty_d, self.alpha)

    def _active_set_update(self):                                           # :86-118 / ADMMEnet.h:85-122
        gamma = self.sprad
        penalty = F(np.float64(self.lam) / (self.rho * np.float64(gamma)))  # Scalar penalty (float)
        tmp = ((self.cache_Ax + self.aux_z + self.dual_y / F(self.rho)) / gamma).astype(F)
        res = self.main_x.copy()
        idx = np.nonzero(res)[0]
        if idx.size:
            val = (res[idx] - self._mv(self.X[:, idx].T, tmp)).astype(F)
            if self.alpha is None:
                new = np.where(val > penalty, val - penalty, np.where(val < -penalty, val + penalty, F(0)))
            else:
                thresh = F(self.alpha * penalty)
                denom = F(1.0 + np.float64(penalty) * (1.0 - np.float64(self.alpha)))
                new = np.where(val > thresh, (val - thresh) / denom,
                               np.where(val < -thresh, (val + thresh) / denom, F(0)))
            res[idx] = new.astype(F)
            if self.follow_x is not None:
                self._prox_yard = np.zeros(self.p)
                self._prox_yard[idx] = np.abs(self.X[:, idx].T.astype(np.float64)) @ np.abs(tmp.astype(np.float64)) + np.abs(val.astype(np.float64))
        elif self.follow_x is not None:
            self._prox_yard = np.zeros(self.p)
        return res

    def _regular_update(self):                                              # :141-150
        gamma = self.sprad
        tmp = (self.cache_Ax + self.aux_z + self.dual_y / F(self.rho)).astype(F)
        vec = (-self._mv(self.X.T, tmp) / gamma).astype(F)
        vec = (vec + self.main_x).astype(F)
        if self.follow_x is not None:
            self._prox_yard = (np.abs(self.X.T.astype(np.float64)) @ np.abs(tmp.astype(np.float64))) / np.float64(gamma) + np.abs(vec.astype(np.float64))
        return self._prox(vec, np.float64(self.lam) / (self.rho * np.float64(gamma)))

    def next_x(self):
        if self.alpha is None:                                              # ADMMLassoWide.h:129-155
            if np.float64(self.lam) > np.float64(self.lambda0) - 1e-5:
                if self.type_log is not None:
                    self.type_log.append(0)
                if self.follow_x is not None:
                    next(self.follow_x, None)                               # (keeps the followed dump aligned: one record per iteration)
                return np.zeros(self.p, F)
            reg = is_regular_update(self.iter_counter)
        else:                                                               # ADMMEnet.h:124-141
            reg = is_regular_update(self.iter_counter) and self.lam < self.lambda0
        if self.type_log is not None:
            self.type_log.append(1 if reg else 2)
        res = self._regular_update() if reg else self._active_set_update()
        if self.follow_x is not None:
            gx = next(self.follow_x, None)
            if gx is not None:
                gx = np.asarray(gx, dtype=F)
                for jj in np.nonzero((res != 0) != (gx != 0))[0]:
                    yard = float(np.finfo(F).eps) * float(self._prox_yard[jj])
                    dist = abs(float(res[jj]) - float(gx[jj]))             # one of the two is zero: the other one's distance from the threshold
                    units = dist / yard if yard > 0 else np.inf
                    rec = dict(record=self.ndecisions, lam=self.lam_idx, iter=int(self.iter_counter), kind="prox", ulps=float(units), coord=int(jj))
                    if units > self.follow_band:
                        continue                                            # not a near-tie: this run keeps its own value (the coefficient tolerance judges the outcome)
                    if self.forced is not None:
                        self.forced.append(rec)
                    res[jj] = gx[jj]
        self.iter_counter += 1
        self.trace_nnz.append(int(np.count_nonzero(res)))
        return res

    def next_z(self):                                                       # :156-165
        idx = np.nonzero(self.main_x)[0]
        self.cache_Ax = self._mv(self.X[:, idx], self.main_x[idx]) if idx.size else np.zeros(self.n, F)
        return ((self.Y + self.dual_y + F(self.rho) * self.cache_Ax) / F(-1 - self.rho)).astype(F)

    def next_residual(self):                                                # :166-170
        return (self.cache_Ax + self.aux_z).astype(F)

    def get_coef(self):
        return self.main_x                                                  # Lasso.cpp:119 get_x()


class PADMMLasso:
    """Row-block consensus ADMM: PADMMBase_Master/Worker + PADMMLasso_*."""

    def __init__(self, X, Y, K, eps_abs, eps_rel):
        n, p = X.shape
        self.K, self.p = K, p
        self.eps_abs, self.eps_rel = eps_abs, eps_rel
        self.lambda0 = np.float64(F(np.abs((X.T @ Y).astype(F)).max()))     # PADMMLasso.h:161
        chunk = n // K                                                      # :163-179
        self.A, self.b = [], []
        for i in range(K):
            lo = i * chunk
            hi = (i + 1) * chunk if i < K - 1 else n
            self.A.append(np.ascontiguousarray(X[lo:hi]))
            self.b.append(Y[lo:hi].copy())
        if self.xy_acc is None:
            self.Ab = [(A.T @ b).astype(F) for A, b in zip(self.A, self.b)]    # worker ctor :42
        else:                                                               # rounding variant "xy64" (LassoTall.xy_acc)
            self.Ab = [(A.astype(self.xy_acc).T @ b.astype(self.xy_acc)).astype(F) for A, b in zip(self.A, self.b)]

    def init(self, lam, rho):                                               # :193-212
        p, K = self.p, self.K
        self.aux_z = np.zeros(p, F)
        self.lam = float(lam)
        self.rho = float(rho)
        if self.rho <= 0:
            self.rho = self.lam / K
        self.x = [np.zeros(p, F) for _ in range(K)]
        self.y = [np.zeros(p, F) for _ in range(K)]
        self.chol = []
        for A in self.A:                                                    # worker init :48-63
            AA = (A.T @ A if A.shape[0] >= A.shape[1] else A @ A.T).astype(F)
            m = AA.shape[0]
            AA[np.arange(m), np.arange(m)] += F(self.rho)
            if self.xmode == "llt32":                                       # the reference: float LLT
                self.chol.append(sla.cho_factor(AA, lower=True, check_finite=False))
            elif self.xmode == "exact":                                     # rounding variants (oracle/variants.py): same system,
                self.chol.append(sla.cho_factor(AA.astype(np.float64), lower=True, check_finite=False))
            else:                                                           # "inv32": float inverse from the float factor
                L = np.tril(sla.cho_factor(AA, lower=True, check_finite=False)[0])
                Li = sla.solve_triangular(L, np.eye(m, dtype=F), lower=True, check_finite=False).astype(F)
                self.chol.append((Li.T @ Li).astype(F))
        self.sq_r = [0.0] * K

    def init_warm(self, lam):                                               # :215-223
        self.lam = float(lam)

    def solve(self, maxit):                                                 # PADMMBase.h:222-237
        K, p = self.K, self.p
        for it in range(maxit):
            # update_x (:174-188)
            xn = sum(np.float64(_sqnorm(x, F)) for x in self.x)
            eps_primal = (max(np.sqrt(xn), np.float64(F(np.linalg.norm(self.aux_z))) * np.sqrt(K)) * self.eps_rel
                          + np.sqrt(float(p * K)) * self.eps_abs)
            yn = sum(np.float64(_sqnorm(y, F)) for y in self.y)
            eps_dual = np.sqrt(yn) * self.eps_rel + np.sqrt(float(p * K)) * self.eps_abs
            rhs_used = []
            for k in range(K):                                              # worker next_x PADMMLasso.h:17-31
                A = self.A[k]
                rhs = (self.Ab[k] - self.y[k]).astype(F)
                rhs = (rhs.astype(np.float64) + self.rho * self.aux_z.astype(np.float64)).astype(F)
                rhs_used.append(rhs)
                if A.shape[0] >= A.shape[1]:
                    self.x[k] = self._solve(k, rhs)
                else:
                    t = (A @ rhs).astype(F)
                    s = self._solve(k, t)
                    self.x[k] = ((rhs - (A.T @ s).astype(F)) / F(self.rho)).astype(F)
            # update_z (:190-198), master next_z PADMMLasso.h:99-108
            vec = np.zeros(p, F)
            for k in range(K):
                vec = (vec + (self.x[k] + self.y[k] / F(self.rho))).astype(F)
            vec = (vec / F(K)).astype(F)
            newz = _soft_d(vec, self.lam / (self.rho * K), F)
            resid_dual = self.rho * np.sqrt(K * np.float64(_sqnorm(newz - self.aux_z, F)))   # :149-152
            self.aux_z = newz
            # update_y (:200-214)
            coll = 0.0
            for k in range(K):
                r = (self.x[k] - self.aux_z).astype(F)
                coll += np.float64(_sqnorm(r, F))
                self.y[k] = (self.y[k] + F(self.rho) * r).astype(F)
            resid_primal = np.sqrt(coll)
            if self.state_log is not None:
                self.state_log.append(np.concatenate([self.aux_z] + list(self.x) + list(self.y)).astype(F))
            converged = resid_primal < eps_primal and resid_dual < eps_dual
            self.ndecisions += 1
            if self.follow is not None:
                g = next(self.follow)
                if int(g[0]) != self.lam_idx or int(g[1]) != it:
                    raise FollowMismatch(f"followed trace is at (lambda {int(g[0])}, iteration {int(g[1])}), this run at ({self.lam_idx}, {it})")
                if (int(g[8]) == 0) != converged:
                    pen = abs(float(self.lam)) / (self.rho * K)
                    uz = _ulp_norm(self.aux_z, F, pen)
                    # the workers' x-updates are linear solves (Cholesky, or Woodbury whose last step cancels
                    # (sigma^2 + rho) / rho digits): their rounding error E = sqrt(sum_k ||x_k - x_k exact||^2), measured on
                    # this iteration's right-hand sides (`_solve_noise`), counted once (not scaled by the band); z is a
                    # soft-threshold of the mean of the x_k + y_k / rho and carries at most E / sqrt(K) of it
                    E = self._solve_noise(rhs_used) / self.follow_band if self.follow_band > 0 else 0.0
                    uz = float(np.hypot(uz, E / np.sqrt(K)))
                    n_p = float(np.sqrt(sum(_ulp_norm(x, F) ** 2 for x in self.x) + E * E + K * uz ** 2))
                    n_d = self.rho * np.sqrt(K) * 2.0 * uz
                    ulps = _stop_ulps(resid_primal, eps_primal, n_p, resid_dual, eps_dual, n_d, converged)
                    rec = dict(record=self.ndecisions - 1, lam=self.lam_idx, iter=it, kind="stop", ulps=float(ulps))
                    if ulps > self.follow_band:
                        raise FollowMismatch(f"decision differs beyond rounding noise: {rec}")
                    self.forced.append(rec)
                    converged = int(g[8]) == 0
            if self.trace is not None:
                self.trace.append((self.lam_idx, it, eps_primal, eps_dual, resid_primal, resid_dual, self.rho, 0, 0 if converged else 1, self.rho, self.rho, 0.0))
            if converged:
                return it + 1
        return maxit + 1

    def _solve_noise(self, rhs_used):
        """sqrt(sum_k || x_k - x_k exact ||^2): every worker's system (float Gram + float rho, as the reference builds it)
        solved in double on the float right-hand side it just used, against the x_k this run computed (FADMM._solve_noise)."""
        if getattr(self, "_m64_rho", None) != self.rho:
            self._m64 = []
            for A in self.A:
                A64 = A.astype(np.float64)
                M = A64.T @ A64
                M[np.arange(self.p), np.arange(self.p)] += np.float64(F(self.rho))
                self._m64.append(sla.cho_factor(M, lower=True, check_finite=False))
            self._m64_rho = self.rho
        e2 = 0.0
        for k in range(self.K):
            xe = sla.cho_solve(self._m64[k], rhs_used[k].astype(np.float64), check_finite=False)
            e2 += float(np.sum((self.x[k].astype(np.float64) - xe) ** 2))
        return float(np.sqrt(e2))

    def _solve(self, k, v):
        """(A'A + rho I)^-1 v or (AA' + rho I)^-1 v with the rounding of `xmode`."""
        if self.xmode == "llt32":
            return sla.cho_solve(self.chol[k], v, check_finite=False).astype(F)
        if self.xmode == "exact":
            return sla.cho_solve(self.chol[k], v.astype(np.float64), check_finite=False).astype(F)
        return (self.chol[k] @ v).astype(F)

    xmode = "llt32"
    xy_acc = None
    trace = None
    state_log = None
    follow = None
    follow_band = 8.0
    forced = None
    ndecisions = 0
    lam_idx = 0

    def get_coef(self):
        return self.aux_z


class LAD(FADMM):
    T = np.float64

    def __init__(self, X, Y, rho, eps_abs, eps_rel):
        self.X, self.Y = X, Y
        self.n, self.p = X.shape
        self.eps_abs, self.eps_rel = eps_abs, eps_rel
        self.ynorm = float(np.linalg.norm(Y))
        XX = X.T @ X                                                        # ADMMLAD.h:186-189
        self.chol = sla.cho_factor(XX, lower=True, check_finite=False)
        self.H = None
        if self.n <= 2000:                                                  # :191-203
            L = np.tril(self.chol[0])
            Tm = sla.solve_triangular(L, X.T, lower=True, check_finite=False).T   # T L' = X
            self.H = Tm @ Tm.T
        n = self.n
        self.main_x = np.zeros(n)
        self.aux_z = np.zeros(n)
        self.dual_y = np.zeros(n)
        self.adj_z = np.zeros(n)
        self.adj_y = np.zeros(n)
        self.rho = float(rho)
        self.eps_primal = self.eps_dual = 0.0
        self.resid_primal = self.resid_dual = 9999.0
        self._init_accel()

    def compute_eps_primal(self):                                           # :152-157
        r = max(np.linalg.norm(self.main_x), np.linalg.norm(self.aux_z), self.ynorm)
        return r * self.eps_rel + np.sqrt(float(self.n)) * self.eps_abs

    def compute_eps_dual(self):                                             # :158-161
        return np.linalg.norm(self.dual_y) * self.eps_rel + np.sqrt(float(self.n)) * self.eps_abs

    def next_x(self):                                                       # :62-78
        vec = self.Y - self.adj_y / self.rho + self.adj_z
        if self.H is not None:
            return self.H @ vec
        return self.X @ sla.cho_solve(self.chol, self.X.T @ vec, check_finite=False)

    def next_z(self):                                                       # :94-98
        vec = self.main_x - self.Y + self.adj_y / self.rho
        return _soft_d(vec, 1.0 / self.rho, np.float64)

    def next_residual(self):                                                # :99-107
        return self.main_x - self.Y - self.aux_z

    def get_coef(self):                                                     # get_x :220-225
        vec = self.Y - self.adj_y / self.rho + self.adj_z
        return sla.cho_solve(self.chol, self.X.T @ vec, check_finite=False)


class BP(FADMM):
    T = np.float64

    def __init__(self, A, b, rho, eps_abs, eps_rel):
        self.A = A
        self.n, self.p = A.shape
        self.eps_abs, self.eps_rel = eps_abs, eps_rel
        AAt = A @ A.T                                                       # ADMMBP.h:167-170
        chol = sla.cho_factor(AAt, lower=True, check_finite=False)
        self.cache_AAAb = A.T @ sla.cho_solve(chol, b, check_finite=False)
        L = np.tril(chol[0])
        self.LinvA = sla.solve_triangular(L, A, lower=True, check_finite=False)   # :173-182
        p = self.p
        self.main_x = np.zeros(p)
        self.aux_z = np.zeros(p)
        self.dual_y = np.zeros(p)
        self.adj_z = np.zeros(p)
        self.adj_y = np.zeros(p)
        self.rho = float(rho)
        self.eps_primal = self.eps_dual = 0.0
        self.resid_primal = self.resid_dual = 9999.0
        self._init_accel()

    def compute_eps_primal(self):                                           # :138-142
        r = max(np.linalg.norm(self.main_x), np.linalg.norm(self.aux_z))
        return r * self.eps_rel + np.sqrt(float(self.p)) * self.eps_abs

    def compute_eps_dual(self):                                             # :143-146
        return np.linalg.norm(self.dual_y) * self.eps_rel + np.sqrt(float(self.p)) * self.eps_abs

    def next_x(self):                                                       # :48-67
        vec = -self.adj_y / self.rho + self.adj_z
        res = vec + self.cache_AAAb
        return res - self.LinvA.T @ (self.LinvA @ vec)

    def next_z(self):                                                       # :84-88
        return _soft_d(self.main_x + self.adj_y / self.rho, 1.0 / self.rho, np.float64)

    def next_residual(self):                                                # :89-93
        return self.main_x - self.aux_z

    def get_coef(self):
        return self.aux_z                                                   # BP.cpp:40 get_z()


class Dantzig:
    """ADMMDantzig -- the Dantzig selector, min ||beta||_1 s.t. ||X'(X beta - y)||_inf <= lambda -- restated from the
    reference's UNBUILT source /root/reference/src/TODO/ADMMDantzig.h (its R wrapper R/50_admm_dantzig.R:30-46 calls a
    symbol the package never compiles) against the CURRENT driver ADMMBase::solve (ADMMBase.h:158-216, update_rho :85-109):
    linearised ADMM on  x = beta, A = X'X, c = X'y, z = -(A x - c) clipped to [-lambda, lambda]:
        x-update   rhs = (A x + z + y / rho - c) / (-gamma);  x = soft(A rhs + x, 1 / (rho gamma))      (:132-144; the active-set
                   form is commented out there: every step is a regular one)          gamma = sprad = (lambda_max estimate)^2
        z-update   A x;  zz = A x + y / rho - c;  z_i = -min(zz_i, lambda) if zz_i > 0 else min(-zz_i, lambda)     (:170-187)
        residual   r = A x + z - c;  y += rho r                                                             (:188-191)
        eps_p = max(||A x||, ||z||, ||c||) eps_rel + sqrt(p) eps_abs;  eps_d = sqrt(gamma) ||y|| eps_rel + sqrt(p) eps_abs (:196-204)
        r_d = rho sqrt(gamma) ||z_new - z||   (:201-204);   rho = 1 / sqrt(gamma) by default (:256-259)
    float64 throughout (`typedef double Scalar`).  A = X'X explicitly when n > p and p <= 1000 (:222), else X'(X v)."""
    trace = None
    lam_idx = 0

    def __init__(self, X, Y, eps_abs, eps_rel):
        from .spectra import sym_eigs_largest
        n, p = X.shape
        self.X, self.p = X, p
        self.eps_abs, self.eps_rel = eps_abs, eps_rel
        self.use_XX = n > p and p <= 1000
        self.XX = X.T @ X if self.use_XX else None
        self.XY = X.T @ Y
        self.XY_norm = float(np.linalg.norm(self.XY))
        self.lambda0 = float(np.abs(self.XY).max())
        self.info = {}
        ev = sym_eigs_largest(self.A_mult, p, 3, 10, 0.1, np.float64, self.info)          # srand(0); eigs.init(); compute(10, 0.1)  (:226-233)
        self.lmax_est = float(ev)
        self.sprad = float(ev) * float(ev)

    def A_mult(self, v):
        return self.XX @ v if self.use_XX else self.X.T @ (self.X @ v)

    def init(self, lam, rho):
        p = self.p
        self.main_x = np.zeros(p); self.aux_z = np.zeros(p); self.dual_y = np.zeros(p); self.cache_Ax = np.zeros(p)
        self.lam = float(lam)
        self.rho = float(rho) if rho > 0 else 1.0 / np.sqrt(self.sprad)
        self.iter_counter = 0

    def init_warm(self, lam):
        self.lam = float(lam)
        self.iter_counter = 0

    def solve(self, maxit):
        sq = np.sqrt(self.sprad)
        sqp = np.sqrt(float(self.p))
        for i in range(maxit):
            eps_primal = max(np.linalg.norm(self.cache_Ax), np.linalg.norm(self.aux_z), self.XY_norm) * self.eps_rel + sqp * self.eps_abs
            eps_dual = sq * np.linalg.norm(self.dual_y) * self.eps_rel + sqp * self.eps_abs
            if self.lam > self.lambda0 - 1e-5:
                self.main_x = np.zeros(self.p)
            else:
                rhs = (self.cache_Ax + self.aux_z + self.dual_y / self.rho - self.XY) / (-self.sprad)
                vec = self.A_mult(rhs) + self.main_x
                pen = 1.0 / (self.rho * self.sprad)
                self.main_x = np.where(vec > pen, vec - pen, np.where(vec < -pen, vec + pen, 0.0))
                self.iter_counter += 1
            self.cache_Ax = self.A_mult(self.main_x)
            zz = self.cache_Ax + self.dual_y / self.rho - self.XY
            newz = np.where(zz > 0, -np.minimum(zz, self.lam), np.minimum(-zz, self.lam))
            resid_dual = self.rho * sq * np.linalg.norm(newz - self.aux_z)
            self.aux_z = newz
            r = self.cache_Ax + self.aux_z - self.XY
            resid_primal = float(np.linalg.norm(r))
            self.dual_y = self.dual_y + self.rho * r
            self.eps_primal, self.eps_dual, self.resid_primal, self.resid_dual = eps_primal, eps_dual, resid_primal, resid_dual
            converged = resid_primal < eps_primal and resid_dual < eps_dual
            rho_in = self.rho
            if converged:
                if self.trace is not None:
                    self.trace.append((self.lam_idx, i, eps_primal, eps_dual, resid_primal, resid_dual, self.rho, 0, 0, rho_in, self.rho, self.lam))
                return i + 1
            if i > 3:
                _rho_rule(self)
            if self.trace is not None:
                self.trace.append((self.lam_idx, i, eps_primal, eps_dual, resid_primal, resid_dual, self.rho, 0, 1, rho_in, self.rho, self.lam))
        return maxit + 1

    def get_coef(self):
        return self.main_x


class SharingBP:
    """`admm_parbp` -- basis pursuit  min ||x||_1  s.t.  A x = b  with the COLUMNS of A split into N blocks ("sharing" ADMM,
    Boyd et al. 2011 section 7.3) -- restated from the reference's UNBUILT source /root/reference/src/TODO/PADMMBP.h (its R
    wrapper R/10_admm_bp.R:111 calls a symbol the package never compiles; the file is written against a master / worker
    base class that no longer exists) against the loop shape of the CURRENT PADMMBase_Master (PADMMBase.h:174-237:
    update_x -> update_z -> update_y -> converged, thresholds recomputed at the top of update_x, rho fixed).
    TEST INFRASTRUCTURE.

    What PADMMBP.h fixes (kept to the letter):
      * the partition: N - 1 blocks of p div N columns, the last one takes the remainder                  (:150-167)
      * rho = 1 / (rho_ratio * mean_i sprad_i), sprad_i = lambda_max(A_i'A_i)                             (:181-186)
      * z-bar = b / N                                                                                     (:137-140)
      * the worker's x-update, linearised with gamma = 2 rho + sprad_i:  v = y / rho + r,  r = mean_i(A_i x_i) - z-bar,
        x_i <- soft(x_i - A_i'v / gamma, 1 / (rho gamma))  on iterations 0, 10, 20, ... of the worker     (:47-61)
        and on the others only on the current non-zeros of x_i (active set), followed by prune            (:19-44)
    What it does not contain (the old base class held it) and is therefore OURS, modelled on the current base class with
    the identity of the consensus constraint replaced by A_i (constraint  A_i x_i - z_i = 0,  sum_i z_i = b, so
    z_i = A_i x_i - r after the z-update):
      * y <- y + rho r                                              (Boyd's u <- u + x-bar - z-bar, y = rho u)
      * resid_primal = sqrt(N) ||r||                                (the N stacked copies of r; PADMMBase.h:209-221)
      * resid_dual   = rho sqrt(sum_i ||dz_i||^2), dz_i = A_i dx_i - dr                                   (:137-142)
                       evaluated as  sum_i ||A_i dx_i||^2 - 2 dr'dS + N ||dr||^2,  S = sum_i A_i x_i  (clamped at 0): the column
                       blocks may live on different ranks and this form needs S and two scalars summed over them, nothing else
      * eps_primal = eps_rel max(sqrt(sum_i ||A_i x_i||^2), sqrt(sum_i ||z_i||^2)) + sqrt(n N) eps_abs    (:118-127)
      * eps_dual   = eps_rel sqrt(N) ||y|| + sqrt(n N) eps_abs                                            (:129-136)
    sprad_i: the reference asks an R function that does not exist (`.spectral_radius_xx`, :64-71); the exact largest
    eigenvalue here (the library: Lanczos with full re-orthogonalisation run to 1e-13)."""

    def __init__(self, A, b, nblocks, eps_abs, eps_rel):
        A = np.asarray(A, dtype=np.float64)
        self.n, self.p = A.shape
        self.N = int(nblocks)
        chunk = self.p // self.N
        self.off = [i * chunk for i in range(self.N)] + [self.p]
        self.A = [A[:, self.off[i]:self.off[i + 1]] for i in range(self.N)]
        self.b = np.asarray(b, dtype=np.float64)
        self.eps_abs, self.eps_rel = float(eps_abs), float(eps_rel)
        self.sprad = [float(np.linalg.norm(Ai, 2) ** 2) for Ai in self.A]
        self.trace = None

    def init(self, rho_ratio):
        self.rho = 1.0 / (float(rho_ratio) * float(np.mean(self.sprad)))
        self.x = [np.zeros(Ai.shape[1]) for Ai in self.A]
        self.Ax = [np.zeros(self.n) for _ in self.A]
        self.y = np.zeros(self.n)
        self.r = np.zeros(self.n)
        self.S = np.zeros(self.n)
        self.counter = 0
        self.eps_primal = self.eps_dual = 0.0
        self.resid_primal = self.resid_dual = 9999.0

    def _eps(self):
        N, n = self.N, self.n
        sax = sum(float(a @ a) for a in self.Ax)
        abar = sum(self.Ax) / N
        sz = sax - 2.0 * N * float(abar @ self.r) + N * float(self.r @ self.r)      # sum_i ||A_i x_i - r||^2
        self.eps_primal = self.eps_rel * np.sqrt(max(sax, sz, 0.0)) + np.sqrt(float(n * N)) * self.eps_abs
        self.eps_dual = self.eps_rel * np.sqrt(float(N)) * float(np.linalg.norm(self.y)) + np.sqrt(float(n * N)) * self.eps_abs

    def solve(self, maxit):
        N = self.N
        zbar = self.b / N
        for it in range(int(maxit)):
            self._eps()
            v = self.y / self.rho + self.r
            regular = self.counter % 10 == 0
            newAx = []
            for i, Ai in enumerate(self.A):
                gamma = 2.0 * self.rho + self.sprad[i]
                pen = 1.0 / (self.rho * gamma)
                xi = self.x[i]
                if regular:
                    vec = xi - (Ai.T @ v) / gamma
                    xi = np.sign(vec) * np.maximum(np.abs(vec) - pen, 0.0)
                else:
                    nz = np.nonzero(xi)[0]
                    xn = np.zeros_like(xi)
                    if nz.size:
                        val = xi[nz] - (Ai[:, nz].T @ v) / gamma
                        xn[nz] = np.sign(val) * np.maximum(np.abs(val) - pen, 0.0)
                    xi = xn
                self.x[i] = xi
                nz = np.nonzero(xi)[0]
                newAx.append(Ai[:, nz] @ xi[nz] if nz.size else np.zeros(self.n))
            self.counter += 1
            Snew = sum(newAx)
            rnew = Snew / N - zbar
            dr = rnew - self.r
            dS = Snew - self.S
            q = sum(float((newAx[i] - self.Ax[i]) @ (newAx[i] - self.Ax[i])) for i in range(N))
            sd = q - 2.0 * float(dr @ dS) + N * float(dr @ dr)
            self.resid_dual = self.rho * np.sqrt(max(sd, 0.0))
            self.Ax = newAx
            self.S = Snew
            self.r = rnew
            self.y = self.y + self.rho * rnew
            self.resid_primal = np.sqrt(N * float(rnew @ rnew))
            conv = self.resid_primal < self.eps_primal and self.resid_dual < self.eps_dual
            if self.trace is not None:
                self.trace.append((it, self.eps_primal, self.eps_dual, self.resid_primal, self.resid_dual, int(regular), int(conv)))
            if conv:
                return it + 1
        return int(maxit) + 1                                       # `return i + 1` after the loop, PADMMBase.h:236

    def get_x(self):
        return np.concatenate(self.x)
