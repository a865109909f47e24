"""The kernel the headline number comes from, under NumPy: symv2_lower_kernel (tall x-update for p >= 2048) plus the
tail kernel's ordered partial reduction, through the C-ABI test hook admm_hip_test_symv, against a float64 mat-vec.

Replaces, for p >= 2048, the solve of ADMMLassoTall::next_x (/root/reference/src/ADMMLassoTall.h:70-80) with a
product by the cached inverse; this test pins the product itself (the solver-level tests pin what is done with it)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

_fp = ctypes.POINTER(ctypes.c_float)


def _symv(A, v0, v1):
    from admm_amd import _lib
    lib = _lib.load()
    p = A.shape[0]
    A = np.asfortranarray(A, dtype=np.float32)
    v0 = np.ascontiguousarray(v0, dtype=np.float32)
    v1 = np.ascontiguousarray(v1, dtype=np.float32)
    y0 = np.empty(p, np.float32)
    y1 = np.empty(p, np.float32)
    _lib.check(lib.admm_hip_test_symv(A.ctypes.data_as(_fp), p, v0.ctypes.data_as(_fp), v1.ctypes.data_as(_fp),
                                      y0.ctypes.data_as(_fp), y1.ctypes.data_as(_fp)))
    return y0, y1


def _sym(p, seed, kind):
    rng = np.random.default_rng(seed)
    if kind == "gauss":                       # dense, sign-alternating entries: worst case for cancellation
        B = rng.standard_normal((p, p)).astype(np.float32)
        A = np.tril(B) + np.tril(B, -1).T
    else:                                     # what the solver multiplies by: the inverse of a shifted Gram matrix
        n = 2 * p
        X = rng.standard_normal((n, p))
        G = X.T @ X + 0.05 * n * np.eye(p)
        A = np.linalg.inv(G)
        A = ((A + A.T) / 2).astype(np.float32)
    return np.asfortranarray(A, dtype=np.float32)


@pytest.mark.parametrize("p", [2048, 2049, 2300, 4200, 10000, 12800])
def test_symv_lower_vs_float64(p):
    """<= 1e-6 norm-wise against float64 (the float products / sums of 10^4 terms themselves round at ~1e-7).
    p = 12800: the triangle (328 MB) exceeds the 310 MB crossover, so the launch takes the non-temporal variant."""
    A = _sym(p, p, "gauss")
    rng = np.random.default_rng(p + 1)
    v0 = rng.standard_normal(p).astype(np.float32)
    v1 = (rng.uniform(size=p) * (rng.uniform(size=p) < 0.1)).astype(np.float32)       # sparse, like rho * adj_z
    y0, y1 = _symv(A, v0, v1)
    A64 = A.astype(np.float64)
    for y, v in ((y0, v0), (y1, v1)):
        ref = A64 @ v.astype(np.float64)
        scale = np.abs(A64) @ np.abs(v.astype(np.float64))            # norm-wise: |A||v| bounds the rounding of any summation order
        err = np.abs(y.astype(np.float64) - ref).max() / scale.max()
        assert err <= 1e-6, (p, err)
        assert np.linalg.norm(y - ref) / np.linalg.norm(ref) <= 1e-6


def test_symv_reads_only_the_lower_triangle():
    """Poison the strict upper triangle: the result must not change (the kernel may only read tiles on / below the
    diagonal and must mask the upper part of diagonal tiles)."""
    p = 2300
    A = _sym(p, 5, "gauss")
    rng = np.random.default_rng(6)
    v0, v1 = rng.standard_normal(p).astype(np.float32), rng.standard_normal(p).astype(np.float32)
    y0, y1 = _symv(A, v0, v1)
    Ap = A.copy()
    iu = np.triu_indices(p, 1)
    Ap[iu] = np.nan
    z0, z1 = _symv(Ap, v0, v1)
    assert np.array_equal(y0, z0) and np.array_equal(y1, z1)


def test_symv_inverse_gram_and_determinism():
    """On an inverse Gram matrix (the solver's operand) and bit-identical run to run."""
    p = 3000
    A = _sym(p, 9, "invgram")
    rng = np.random.default_rng(10)
    v0, v1 = rng.standard_normal(p).astype(np.float32) * 100, rng.standard_normal(p).astype(np.float32)
    y0, y1 = _symv(A, v0, v1)
    ref = A.astype(np.float64) @ v0.astype(np.float64)
    assert np.linalg.norm(y0 - ref) / np.linalg.norm(ref) <= 1e-6
    for _ in range(3):
        z0, z1 = _symv(A, v0, v1)
        assert np.array_equal(y0, z0) and np.array_equal(y1, z1)
