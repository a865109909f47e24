"""Pin the CPU oracle against every known-answer vector the reference holds (README only)."""
import numpy as np
import pytest

from oracle import entry, readme
from oracle.rrng import RRandom


def test_r_rng_known_values():
    # R: set.seed(123); runif(3) / rnorm(3)
    assert np.allclose(RRandom(123).runif(3), [0.2875775, 0.7883051, 0.4089769], atol=5e-8)
    assert np.allclose(RRandom(123).rnorm(3), [-0.56047565, -0.23017749, 1.55870831], atol=5e-9)


def test_readme_lasso_admm_column(readme_lasso_xy):
    x, y = readme_lasso_xy
    d = {}
    out = entry.admm_lasso(x, y, [readme.LAMBDA], 100, 1e-4, True, True, entry.LASSO_OPTS, d)
    beta = out["beta"][:, 0]
    # README.md:66-88; 1.2e-5 abs (2e-6 of max|beta|) -- SURVEY.md section 8c
    assert np.abs(beta - readme.LASSO_ADMM).max() < 2e-5
    assert np.array_equal(beta != 0, readme.LASSO_ADMM != 0)
    assert out["niter"][0] == 31
    assert abs(d["solver"].rho - 13.678) < 1e-2
    # the Lanczos value is the loose under-estimate, not lambda_max
    assert abs(float(d["solver"].lmax_est) - 178.95) < 0.05
    # secondary reference: glmnet column
    assert np.abs(beta - readme.LASSO_GLMNET).max() < 1e-4


def test_readme_lasso_paradmm_column(readme_lasso_xy):
    x, y = readme_lasso_xy
    out = entry.admm_parlasso(x, y, [readme.LAMBDA], 100, 1e-4, True, True, 2, entry.LASSO_OPTS)
    assert np.abs(out["beta"][:, 0] - readme.LASSO_PARADMM).max() < 5e-6
    assert out["niter"][0] == 339


def test_readme_enet_column(readme_lasso_xy):
    x, y = readme_lasso_xy
    out = entry.admm_enet(x, y, [readme.LAMBDA], 100, 1e-4, True, True, 0.5, entry.LASSO_OPTS)
    assert np.abs(out["beta"][:, 0] - readme.ENET_ADMM).max() < 2e-6
    assert out["niter"][0] == 22


def test_readme_lad_column(readme_lasso_xy):
    x, y = readme_lasso_xy
    d = {}
    out = entry.admm_lad(x, y, False, entry.LAD_OPTS, d)
    assert out["beta"][0] == 0.0
    assert np.abs(out["beta"][1:] - readme.LAD_ADMM).max() < 1e-9
    assert out["niter"] == 443
    assert abs(d["solver"].rho - 0.5787037) < 1e-6


def test_readme_bp_range():
    x, y, bt = readme.bp_data()
    out = entry.admm_bp(x, y, entry.BP_OPTS)
    e = bt - out["beta"]
    assert abs(e.min() - readme.BP_RANGE[0]) < 5e-10
    assert abs(e.max() - readme.BP_RANGE[1]) < 5e-10
    assert out["niter"] == 72


def test_readme_bp_perf_range():
    x, y, bt = readme.bp_data(1000, 2000, 100)       # README.md:369-393
    out = entry.admm_bp(x, y, entry.BP_OPTS)
    e = bt - out["beta"]
    assert abs(e.min() - readme.BP_PERF_RANGE[0]) < 5e-9
    assert abs(e.max() - readme.BP_PERF_RANGE[1]) < 5e-9


def test_golden_file_matches_the_regenerated_fixtures_and_pins_the_oracle():
    """tests/golden/readme_vectors.json (committed data): inputs equal what the R-RNG restatement regenerates,
    and the oracle reproduces the README's printed Lasso column from the STORED inputs."""
    import json
    import os
    from oracle import entry, readme
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "readme_vectors.json")
    g = json.load(open(path))
    x, y = readme.lasso_data()
    xs = np.array(g["lasso"]["x_colmajor"]).reshape((100, 20), order="F")
    assert np.array_equal(xs, x) and np.array_equal(np.array(g["lasso"]["y"]), y)
    xb, yb, _ = readme.bp_data()
    assert np.array_equal(np.array(g["bp"]["x_colmajor"]).reshape((50, 100), order="F"), xb)
    ref = entry.admm_lasso(xs, np.array(g["lasso"]["y"]), [g["lambda"]], 100, 1e-4, True, True, entry.LASSO_OPTS)
    want = np.array(g["lasso"]["admm"])
    assert np.abs(ref["beta"][:, 0] - want).max() / np.abs(want).max() < 1e-4
