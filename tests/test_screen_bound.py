"""CPU: the inequality behind the safe screening of the regular steps (admm_amd/csrc/lasso_wide.hip wide_x_kernel "screen",
wide_screen_prep_kernel / wide_screen_prep8_kernel; sharing_bp.hip sbp_xreg_screen_kernel), restated in NumPy and attacked.

The kernels leave a column with x_j = 0 alone when   |fl32(Xc_j't)| (x scale_j) + s_j ||t||_2 <= gamma * threshold * (1 - 2^-21),
Xc_j the fp16 rounding of the column or its 8-bit linear code, s_j = (||X_j - Xc_j||_2 + 2 * 1.001 n 2^-24 max(||X_j||_2, ||Xc_j||_2)) * (1 + 1e-6)
rounded up to float.  Bit-identity with the unscreened step rests on two facts checked here for float sums in several orders, random and
ADVERSARIAL t (aligned with the column's rounding error, with the column, with alternating signs), and columns over fourteen decades:
  (1) |fl32(X_j't)| <= |fl32(Xc_j't)| + s_j ||t||_2                         (whatever order either sum is formed in),
  (2) a <= gamma * thr * (1 - 2^-21)  ==>  fl32(fl32(a) / gamma) <= thr      (the exact step divides by gamma in float and compares)."""
import numpy as np

U = 2.0 ** -24


def _copy_fp16(col):
    with np.errstate(over="ignore"):
        h = col.astype(np.float16).astype(np.float64)
    h[~np.isfinite(h)] = 0.0                                      # a rounding that is not finite is stored as zero
    return h


def _copy_int8(col):
    mx = np.float32(np.abs(col).max())
    sc = np.float32(mx / np.float32(127.0))
    if not sc > 0:
        return np.zeros_like(col, dtype=np.float64)
    qv = np.clip(np.rint(col / sc), -127, 127)
    return np.float64(sc) * qv.astype(np.float64)


def _bound(col, cp):
    n = len(col)
    x = col.astype(np.float64)
    e = np.sqrt(((x - cp) ** 2).sum())
    gn = 1.001 * n * U
    sd = (e + 2.0 * gn * np.sqrt(max((x * x).sum(), (cp * cp).sum()))) * (1.0 + 1e-6)
    s = np.float32(sd)
    if float(s) < sd:
        s = np.nextafter(s, np.float32(np.inf))                  # __double2float_ru
    return float(s)


def _sums32(a, t):
    """The float inner product in several orders: pairwise (np.dot), strictly sequential with one rounding per step (FMA), eight
    interleaved accumulators then a tree (the kernels' shape), reversed."""
    a = a.astype(np.float32)
    t = t.astype(np.float32)
    out = [float(np.dot(a, t))]
    prod = a.astype(np.float64) * t.astype(np.float64)            # exact products (24 x 24 bits)
    acc = np.float32(0)
    for v in prod:
        acc = np.float32(np.float64(acc) + v)                     # one rounding per step = fmaf
    out.append(float(acc))
    lanes = np.zeros(8, np.float32)
    for i, v in enumerate(prod):
        lanes[i % 8] = np.float32(np.float64(lanes[i % 8]) + v)
    while len(lanes) > 1:
        lanes = (lanes[0::2] + lanes[1::2]).astype(np.float32)
    out.append(float(lanes[0]))
    acc = np.float32(0)
    for v in prod[::-1]:
        acc = np.float32(np.float64(acc) + v)
    out.append(float(acc))
    return out


def test_the_bound_covers_every_order_of_either_sum():
    rng = np.random.default_rng(11)
    worst = 0.0
    for n in (37, 300, 2000):
        for trial in range(24):
            scale = 10.0 ** rng.uniform(-7, 7)
            col = (rng.standard_normal(n) * scale).astype(np.float32)
            if trial % 4 == 1:
                col[rng.integers(0, n, 3)] *= np.float32(300.0)   # outliers: the 8-bit code loses the rest of the column
            if trial % 4 == 2 and scale < 1e3:
                col[rng.integers(0, n)] = np.float32(7e4)          # beyond fp16's range
            for make in (_copy_fp16, _copy_int8):
                cp = make(col)
                s = _bound(col, cp)
                delta = col.astype(np.float64) - cp
                ts = [rng.standard_normal(n), delta / (np.abs(delta).max() or 1.0), np.sign(delta), col.astype(np.float64) / scale,
                      np.where(np.arange(n) % 2 == 0, 1.0, -1.0), rng.standard_normal(n) * 10.0 ** rng.uniform(-20, 15)]
                for t in ts:
                    t = t.astype(np.float32)
                    if not np.all(np.isfinite(t)):
                        continue
                    tn = float(np.sqrt((t.astype(np.float64) ** 2).sum())) * (1.0 + 1e-12)
                    exact = _sums32(col, t)
                    screen = _sums32(cp.astype(np.float32) if make is _copy_fp16 else cp, t) if make is _copy_fp16 else None
                    if make is _copy_int8:                         # the kernel sums q_j't in float and multiplies by scale_j in double
                        mx = np.float32(np.abs(col).max())
                        sc = np.float32(mx / np.float32(127.0))
                        qv = np.clip(np.rint(col / sc), -127, 127) if sc > 0 else np.zeros(n)
                        screen = [float(sc) * v for v in _sums32(qv, t)]
                    for de in exact:
                        for ds in screen:
                            lhs, rhs = abs(de), abs(ds) + s * tn
                            assert lhs <= rhs, (n, trial, make.__name__, lhs, rhs)
                            if rhs > 0:
                                worst = max(worst, lhs / rhs)
    assert 0.5 < worst <= 1.0                                      # ... and it is not vacuous: the adversarial t get close


def test_below_the_margin_the_float_division_cannot_cross_the_threshold():
    rng = np.random.default_rng(5)
    for _ in range(20000):
        gamma = np.float32(10.0 ** rng.uniform(-3, 6))
        thr = float(np.float32(10.0 ** rng.uniform(-8, 2)))        # the enet threshold is a float; the lasso's a double (same argument)
        G = float(gamma) * thr * (1.0 - 4.8e-7)
        a = G * (1.0 - 10.0 ** rng.uniform(-12, -1)) if rng.random() < 0.5 else G
        d = np.float32(a)
        if float(d) > a:                                           # the computed |d| is a float at or below the bound
            d = np.nextafter(d, np.float32(0))
        vec = np.float32(d / gamma)                                # (-d) / gamma + 0
        assert float(vec) <= thr, (a, float(d), float(gamma), thr)
