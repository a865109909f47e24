"""One-time matrix-core kernels under NumPy, through the C-ABI test hooks: the Gram products (fp32 / fp64 SYRK on
v_mfma_f32_32x32x2_f32 / v_mfma_f64_16x16x4_f64, split-K for small orders) and the blocked Cholesky + inverse.
Replaces Linalg::cross_prod_lower / tcross_prod_lower (/root/reference/src/Linalg/BlasWrapper.h:73-154) and the LLT of
ADMMLassoTall.h:204-205, PADMMLasso.h:55-60, ADMMLAD.h:187-189, ADMMBP.h:168-169."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _gram(A, atA):
    from admm_amd import _lib
    lib = _lib.load()
    A = np.asfortranarray(A)
    dbl = A.dtype == np.float64
    k = A.shape[1] if atA else A.shape[0]
    G = np.zeros((k, k), dtype=A.dtype, order="F")
    _lib.check(lib.admm_hip_test_gram(A.ctypes.data, A.shape[0], A.shape[1], int(atA), int(dbl), G.ctypes.data))
    return G


def _inverse(A, precision):
    from admm_amd import _lib
    lib = _lib.load()
    A = np.asfortranarray(A)
    out = np.zeros_like(A, order="F")
    _lib.check(lib.admm_hip_test_spd_inverse(A.ctypes.data, A.shape[0], precision, out.ctypes.data))
    return out


@pytest.mark.parametrize("rows,cols,atA", [(3000, 700, True), (1037, 513, True), (5000, 2300, True), (300, 4000, False),
                                           (129, 129, True), (2000, 64, True)])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_gram_vs_numpy(rows, cols, atA, dtype):
    rng = np.random.default_rng(rows + cols)
    A = (rng.standard_normal((rows, cols)) * 2 + 0.3).astype(dtype)
    G = _gram(A, atA)
    A64 = A.astype(np.float64)
    ref = A64.T @ A64 if atA else A64 @ A64.T
    assert np.array_equal(G, G.T)                                    # both triangles, mirrored exactly
    tol = 2e-5 if dtype == np.float32 else 1e-13                      # float sums of thousands of products, no compensation
    assert np.abs(G - ref).max() / np.abs(ref).max() < tol


@pytest.mark.parametrize("rows,cols", [(300, 4136), (200, 5700), (200, 5800)])
def test_gram_tail_of_quarter_tiles(rows, cols):
    """Deep-K Gram launches end with QUARTER work items instead of a straggling round of full tiles (syrk_mfma.hip,
    gram_work_lists / gemm_nt_quarter): the ragged last block row when at most 64 of its rows are real (4136 = 32 x 128 + 40;
    5800 = 45 x 128 + 40) and the tiles of a last round that would be mostly empty (5700: 45 block rows = 1035 tiles on 1024
    resident workgroups -> 11 tiles as 41 quarters).  (Round 4 held them bit-identical to the launch without them and to the
    row-major tile order; those two forms lost their A/B and are gone.)  Against float64, mirrored exactly."""
    import admm_amd
    rng = np.random.default_rng(cols)
    A = (rng.standard_normal((rows, cols)) * 2 + 0.3).astype(np.float32)
    with admm_amd.options(GRAM_SPLIT="0"):                         # the exact-fp32 kernel: orders >= ~4000 take the 16-bit split by default (below)
        G = _gram(A, True)
    assert np.array_equal(G, G.T)
    A64 = A.astype(np.float64)
    ref = A64.T @ A64
    assert np.abs(G - ref).max() / np.abs(ref).max() < 2e-5


@pytest.mark.parametrize("split", ["f16x2", "bf16x3"])
@pytest.mark.parametrize("rows,cols,kind", [(9000, 4136, "gauss"), (3000, 5700, "gauss"), (20000, 4200, "standardised"), (2500, 4100, "wild"), (2100, 4000, "scales")])
def test_gram_16bit_split_is_fp32_accurate(rows, cols, kind, split):
    """Tall Grams of order >= ~4000 run on the 16-bit matrix cores (gram_bf16x3.hip): every fp32 entry as two fp16 terms with an exact
    per-column power-of-two scale and the three significant cross products (default), or as three bf16 terms and six products.
    Against float64: no worse than 2 x the error of the exact-fp32 matrix-core kernel on the same input (all accumulate K products
    in fp32), mirrored exactly -- also on entries spanning 12 orders of magnitude ("wild") and on columns whose scales span 1e-12 ..
    1e12 ("scales": without the column scaling fp16 would overflow / flush)."""
    import admm_amd
    rng = np.random.default_rng(rows + cols)
    if kind == "gauss":
        A = rng.standard_normal((rows, cols)) * 2 + 0.3
    elif kind == "standardised":
        A = rng.standard_normal((rows, cols)) * rng.uniform(0.1, 30.0, size=cols)[None, :] + rng.uniform(-5, 5, size=cols)[None, :]
        A = (A - A.mean(0)) / A.std(0)
    elif kind == "wild":
        A = rng.standard_normal((rows, cols)) * 10.0 ** rng.uniform(-6, 6, size=(rows, cols))
    else:
        A = rng.standard_normal((rows, cols)) * 10.0 ** rng.uniform(-12, 12, size=cols)[None, :]
        A[:, 7] = 0.0                                                  # a null column: scale 1
    A = A.astype(np.float32)
    with admm_amd.options(GRAM_SPLIT=split):
        G3 = _gram(A, True)
    with admm_amd.options(GRAM_SPLIT="0"):
        G1 = _gram(A, True)
    A64 = A.astype(np.float64)
    ref = A64.T @ A64
    assert np.array_equal(G3, G3.T)
    assert not np.array_equal(G3, G1), "the 16-bit path was not taken"
    d = np.where(np.diag(ref) > 0, np.diag(ref), 1.0)
    scale = np.sqrt(np.outer(d, d))                                   # entrywise: |g_ij| <= sqrt(g_ii g_jj)
    e3, e1 = np.abs(G3 - ref) / scale, np.abs(G1 - ref) / scale
    print(f"[gram {split} {kind} {rows}x{cols}] max error / sqrt(g_ii g_jj): split {e3.max():.2e} (rms {np.sqrt((e3 ** 2).mean()):.2e}), fp32 kernel {e1.max():.2e} (rms {np.sqrt((e1 ** 2).mean()):.2e})")
    assert e3.max() <= 2.0 * e1.max() + 1e-7
    assert np.sqrt((e3 ** 2).mean()) <= 2.0 * np.sqrt((e1 ** 2).mean()) + 1e-8


def test_gram_16bit_split_at_the_headline_row_count():
    """The DEFAULT Gram of BASELINE configs[1] at its real K range: 100 000 rows x 10 000 standardised columns through the fp16 x 2 split
    (13 launches of 8192-row slabs accumulated in fp32), against float64 X'X formed on the device by torch.  Same normalised metric and
    same bound as test_gram_16bit_split_is_fp32_accurate (which stops at 20 000 rows): no worse than 2 x the exact-fp32 matrix-core
    kernel on the same input.  (BlasWrapper.h:89-112: the reference forms this matrix in float with Eigen.)"""
    import torch
    rows, cols = 100000, 10000
    rng = np.random.default_rng(100000 + 10000)
    A = np.empty((rows, cols), dtype=np.float32, order="F")
    for j0 in range(0, cols, 500):
        blk = rng.standard_normal((rows, 500), dtype=np.float32) * rng.uniform(0.1, 30.0, size=500).astype(np.float32)[None, :] \
            + rng.uniform(-5, 5, size=500).astype(np.float32)[None, :]
        blk64 = blk.astype(np.float64)
        A[:, j0:j0 + 500] = ((blk64 - blk64.mean(0)) / blk64.std(0)).astype(np.float32)
    import admm_amd
    with admm_amd.options(GRAM_SPLIT="f16x2"):
        G3 = _gram(A, True)
    with admm_amd.options(GRAM_SPLIT="0"):
        G1 = _gram(A, True)
    assert np.array_equal(G3, G3.T) and not np.array_equal(G3, G1), "the 16-bit path was not taken"
    dev = torch.device("cuda", 0)
    ref = torch.zeros((cols, cols), dtype=torch.float64, device=dev)
    for r0 in range(0, rows, 10000):                                   # float64 X'X on the device, 10 000 rows at a time
        blk = torch.from_numpy(np.ascontiguousarray(A[r0:r0 + 10000])).to(dev).double()
        ref += blk.T @ blk
    del blk
    d = torch.diagonal(ref).clamp_min(1e-300).sqrt()
    out = {}
    for name, G in (("split", G3), ("fp32", G1)):
        e = (torch.from_numpy(G).to(dev).double() - ref).abs() / (d[:, None] * d[None, :])
        out[name] = (float(e.max()), float((e * e).mean().sqrt()))
        del e
    print(f"[gram f16x2 at n = 10^5, p = 10^4] max error / sqrt(g_ii g_jj): split {out['split'][0]:.2e} (rms {out['split'][1]:.2e}), "
          f"fp32 kernel {out['fp32'][0]:.2e} (rms {out['fp32'][1]:.2e})")
    assert out["split"][0] <= 2.0 * out["fp32"][0] + 1e-7
    assert out["split"][1] <= 2.0 * out["fp32"][1] + 1e-8


@pytest.mark.parametrize("n", [256, 300, 1100, 2048, 2300])
@pytest.mark.parametrize("precision", [0, 1, 2])
def test_spd_inverse_vs_numpy(n, precision):
    """Residual ||A Ainv - I|| against what LAPACK achieves in the same precision (cond(A) ~ 30)."""
    rng = np.random.default_rng(n)
    X = rng.standard_normal((2 * n, n))
    A64 = X.T @ X + 0.05 * n * np.eye(n)
    dtype = np.float64 if precision == 1 else np.float32
    A = A64.astype(dtype)
    Ainv = _inverse(A, precision)
    assert np.abs(Ainv - Ainv.T).max() <= 1e-6 * np.abs(Ainv).max()
    exact = np.linalg.inv(A.astype(np.float64))
    err = np.abs(Ainv - exact).max() / np.abs(exact).max()
    lapack = np.linalg.inv(A).astype(np.float64)
    err_lapack = np.abs(lapack - exact).max() / np.abs(exact).max()
    bound = {0: max(5 * err_lapack, 2e-5), 1: 1e-12, 2: 1.2e-7}[precision]      # precision 2: one float rounding of the double inverse
    print(f"[inverse n={n} precision={precision}] max error {err:.2e} (LAPACK in the same precision: {err_lapack:.2e})")
    assert err < bound, (n, precision, err, err_lapack)


def test_inverse_rejects_an_indefinite_matrix():
    from admm_amd._lib import AdmmHipError
    rng = np.random.default_rng(1)
    X = rng.standard_normal((400, 300))
    A = (X.T @ X).astype(np.float32)
    A[200, 200] = -5.0
    with pytest.raises(AdmmHipError) as e:
        _inverse(A, 0)
    assert e.value.code == 5                                          # ADMM_ERR_NOT_SPD (the reference never checks LLT::info())


@pytest.mark.parametrize("rows,cols", [(100, 20), (2000, 300), (50000, 700), (1250, 20000), (33, 4097), (100001, 64)])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_gemv_t_vs_numpy(rows, cols, dtype):
    """The streaming mat-vec (gemv_t_kernel) at tall, wide and ragged shapes: y = A'v against float64."""
    from admm_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(rows * 7 + cols)
    A = np.asfortranarray(rng.standard_normal((rows, cols)).astype(dtype))
    v = rng.standard_normal(rows).astype(dtype)
    y = np.zeros(cols, dtype=dtype)
    _lib.check(lib.admm_hip_test_gemv_t(A.ctypes.data, rows, cols, int(dtype == np.float64), v.ctypes.data, y.ctypes.data))
    ref = A.astype(np.float64).T @ v.astype(np.float64)
    scale = (np.abs(A.astype(np.float64)).T @ np.abs(v.astype(np.float64))).max()
    assert np.abs(y - ref).max() / scale < (2e-6 if dtype == np.float32 else 1e-14)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("rows,cols,density", [(1250, 20000, 0.003), (5000, 6000, 0.02), (130, 777, 1.0), (2049, 300, 0.0), (37, 5000, 0.5), (1, 1, 1.0)])
def test_gather_matvec_vs_numpy(rows, cols, density, dtype):
    """The gather mat-vec of the one-pass forms (gather_kernels.h): y = A v over the non-zeros of v, double accumulation, fixed
    summation order -- sparse, dense, empty and ragged right-hand sides (row tiles of 1024 / 512 rows, column chunks of 256, groups of
    whole chunks), NaN in columns whose coefficient is zero (they must never be read), bit reproducibility."""
    from admm_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(rows * 7 + cols)
    A = np.asfortranarray(rng.standard_normal((rows, cols)).astype(dtype))
    v = np.zeros(cols, dtype=dtype)
    nz = rng.random(cols) < density
    v[nz] = rng.standard_normal(int(nz.sum())).astype(dtype)
    if 0.0 < density < 1.0 and (~nz).any():
        A[:, np.nonzero(~nz)[0][:5]] = np.nan                          # unlisted columns are not touched
    y = np.zeros(rows)
    _lib.check(lib.admm_hip_test_gather(A.ctypes.data, rows, cols, int(dtype == np.float64), v.ctypes.data, y.ctypes.data_as(ctypes.POINTER(ctypes.c_double))))
    ref = np.nan_to_num(A.astype(np.float64), nan=0.0) @ v.astype(np.float64)
    scale = np.abs(np.nan_to_num(A.astype(np.float64), nan=0.0)) @ np.abs(v.astype(np.float64)) + 1e-300
    assert np.all(np.isfinite(y))
    assert (np.abs(y - ref) / scale).max() < 1e-14, float((np.abs(y - ref) / scale).max())     # products exact in double, sums in double
    y2 = np.zeros(rows)
    _lib.check(lib.admm_hip_test_gather(A.ctypes.data, rows, cols, int(dtype == np.float64), v.ctypes.data, y2.ctypes.data_as(ctypes.POINTER(ctypes.c_double))))
    assert np.array_equal(y, y2)


@pytest.mark.parametrize("kind,n,p", [("tall", 3000, 700), ("wide", 300, 2100), ("lad", 2600, 300), ("bp", 200, 900)])
@pytest.mark.parametrize("flags", [(True, True), (True, False), (False, True), (False, False)])
def test_fused_convert_and_standardise_is_bit_identical(kind, n, p, flags):
    """Round 6: convert + column sums + means + centred sums of squares + scales + apply as ONE launch per block of columns
    (prep.hip convert_standardize_kernel: every thread re-reads only what it wrote) against the separate sweeps (PREP_FUSED=0): the
    statistics are formed from the same elements in the same order, so coefficients, iteration counts and the recovered intercepts are
    identical to the bit -- float (tall / wide) and double (LAD / BP) data, every standardize / intercept combination, ragged n."""
    from admm_amd import admm_bp, admm_lad, admm_lasso, options
    standardize, intercept = flags
    rng = np.random.default_rng(7 + n)
    x = rng.standard_normal((n + 3, p)) * rng.uniform(0.1, 30.0, p) + rng.uniform(-5.0, 5.0, p)
    b = np.zeros(p); b[:8] = rng.standard_normal(8)
    y = x @ b + rng.standard_normal(n + 3) + 2.0
    yb = x @ np.concatenate([rng.standard_normal(5), np.zeros(p - 5)])
    out = {}
    for fused in ("1", "0"):
        with options(PREP_FUSED=fused):
            if kind in ("tall", "wide"):
                f = admm_lasso(x, y, intercept=intercept, standardize=standardize).penalty(nlambda=5, lambda_min_ratio=0.05).opts(maxit=300).fit()
                out[fused] = (f.beta_dense, list(f.niter))
            elif kind == "lad":
                f = admm_lad(x, y, intercept=intercept).opts(maxit=60).fit()
                out[fused] = (np.asarray(f.beta), [f.niter])
            else:
                f = admm_bp(x, yb).opts(maxit=60).fit()
                out[fused] = (f.beta.toarray(), [f.niter])
    assert out["1"][1] == out["0"][1]
    assert np.array_equal(out["1"][0], out["0"][0])
