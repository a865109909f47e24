"""Deterministic stream of small random problems for the five entry points (used by test_gpu_fuzz.py and
tests/tools/fuzz_parity.py).  All random draws happen here, so case c of (ncases, seed) is reproducible on its own."""
import numpy as np


def cases(ncases, seed):
    """Deterministic stream of random problems (all random draws happen here)."""
    rng = np.random.default_rng(seed)
    for c in range(ncases):
        kind = rng.choice(["tall", "wide", "enet_tall", "enet_wide", "par", "lad", "bp"])
        icpt, stdz = bool(rng.integers(2)), bool(rng.integers(2))
        m = int(rng.integers(1, 6))
        scale = float(rng.choice([0.01, 1.0, 2.0, 50.0]))
        if kind in ("tall", "enet_tall", "lad"):
            p = int(rng.integers(3, 70)); n = p + int(rng.integers(1, 300))
        elif kind in ("wide", "enet_wide"):
            n = int(rng.integers(4, 60)); p = n + int(rng.integers(0, 300))
        elif kind == "par":
            p = int(rng.integers(12, 60)); n = int(rng.integers(40, 400))
        else:
            n = int(rng.integers(5, 50)); p = n + int(rng.integers(5, 200))
        m = min(m, p)
        x = rng.standard_normal((n, p)) * scale
        if rng.random() < 0.3:
            x += rng.standard_normal(p) * 3 * scale          # non-zero column means
        b = np.zeros(p); b[:m] = rng.uniform(size=m)
        y = x @ b + (0 if kind == "bp" else 1) * rng.standard_normal(n) * scale
        cs = dict(c=c, kind=kind, icpt=icpt, stdz=stdz, scale=scale, n=n, p=p, x=x, y=y)
        if kind in ("tall", "wide", "enet_tall", "enet_wide", "par"):
            cs["user_lam"] = rng.random() < 0.5
            cs["nl"] = int(rng.integers(1, 8))
            cs["alpha"] = float(rng.choice([0.1, 0.5, 0.9, 1.0])) if kind.startswith("enet") else None
            cs["ulam"] = rng.uniform(0.02, 0.9, size=cs["nl"]) if cs["user_lam"] else None
            cs["K"] = int(rng.integers(1, max(2, min(6, (p - 1) // 5)))) if kind == "par" else 0
        yield cs


def medium_cases(ncases, seed):
    """Lasso-family problems large enough for the matrix-core setup (p >= 256) and the lower-triangle x-update
    (p >= 2048), with random solver options.  Still seconds per case for the oracle."""
    rng = np.random.default_rng(seed)
    for c in range(ncases):
        kind = rng.choice(["tall", "enet_tall", "wide", "par"])
        icpt, stdz = bool(rng.integers(2)), bool(rng.integers(2))
        scale = float(rng.choice([0.1, 1.0, 2.0, 20.0]))
        if kind in ("tall", "enet_tall"):
            p = int(rng.choice([257, 300, 640, 1100, 2048, 2300]))
            n = p + int(rng.integers(1, 2 * p))
        elif kind == "wide":
            n = int(rng.integers(100, 600)); p = n + int(rng.integers(0, 3000))
        else:
            p = int(rng.integers(200, 900)); n = int(rng.integers(300, 3000))
        m = int(rng.integers(1, 40))
        x = rng.standard_normal((n, p)) * scale
        if rng.random() < 0.3:
            x += rng.standard_normal(p) * 3 * scale
        b = np.zeros(p); b[:m] = rng.uniform(size=m)
        y = x @ b + rng.standard_normal(n) * scale
        cs = dict(c=c, kind=kind, icpt=icpt, stdz=stdz, scale=scale, n=n, p=p, x=x, y=y)
        cs["user_lam"] = rng.random() < 0.3
        cs["nl"] = int(rng.integers(2, 12))
        cs["alpha"] = float(rng.choice([0.2, 0.7, 1.0])) if kind == "enet_tall" else None
        cs["ulam"] = rng.uniform(0.02, 0.9, size=cs["nl"]) if cs["user_lam"] else None
        cs["K"] = int(rng.integers(2, 9)) if kind == "par" else 0
        cs["maxit"] = int(rng.choice([7, 300, 10000])) if kind != "par" else int(rng.choice([7, 300]))
        cs["eps"] = float(rng.choice([1e-5, 1e-3]))
        cs["rho"] = float(rng.choice([-1.0, -1.0, 5.0, 200.0]))
        yield cs
