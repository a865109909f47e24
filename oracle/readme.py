"""The reference README's known-answer vectors (TEST INFRASTRUCTURE).

Inputs are regenerated from the R snippets with oracle.rrng (data, not source:
the snippets are `set.seed(123)` + `runif/rnorm/sample` calls); expected outputs
are the printed columns of /root/reference/README.md, copied as literal numbers
with their line provenance.
"""
import numpy as np

from .rrng import RRandom


def lasso_data():
    """README.md:47-53: set.seed(123); n=100; p=20; m=5; b; x ~ N(1.2, 2^2); y = 5 + x b + N(0,1)."""
    r = RRandom(123)
    n, p, m = 100, 20, 5
    b = np.concatenate([r.runif(m), np.zeros(p - m)])
    x = r.rnorm(n * p, mean=1.2, sd=2.0).reshape((n, p), order="F")
    y = 5 + x @ b + r.rnorm(n)
    return x, y


def bp_data(n=50, p=100, nsig=15):
    """README.md:166-174 (and :383-389 with n=1000,p=2000,nsig=100 -- see perf_bp_data)."""
    r = RRandom(123)
    beta_true = np.concatenate([r.runif(nsig), np.zeros(p - nsig)])
    beta_true = beta_true[r.sample_perm(p)]
    x = r.rnorm(n * p).reshape((n, p), order="F")
    y = x @ beta_true
    return x, y, beta_true


LAMBDA = float(np.exp(-2.0))

# README.md:66-88  (rows 1..21: intercept then 20 coefficients)
LASSO_GLMNET = np.array([
    5.357410774, 0.178916019, 0.683606818, 0.310518550, 0.861034415, 0.879797912, 0.007854581,
    0.0, 0.0, 0.023462980, 0.010952896, 0.0, -0.003800159, 0.0, 0.094591923,
    0.0, 0.0, 0.0, 0.0, -0.002916255, 0.0])
LASSO_ADMM = np.array([
    5.357455254, 0.178915471, 0.683609307, 0.310507625, 0.861029863, 0.879794598, 0.007850002,
    0.0, 0.0, 0.023467677, 0.010957017, 0.0, -0.003811116, 0.0, 0.094586611,
    0.0, 0.0, 0.0, 0.0, -0.002929136, 0.0])
LASSO_PARADMM = np.array([
    5.357429504, 0.178917870, 0.683610320, 0.310525119, 0.861012816, 0.879801810, 0.007853498,
    0.0, 0.0, 0.023452930, 0.010950469, 0.0, -0.003801103, 0.0, 0.094600648,
    0.0, 0.0, 0.0, 0.0, -0.002919935, 0.0])
# README.md:100-123  admm_enet(x, y)$penalty(exp(-2), alpha = 0.5)
ENET_GLMNET = np.array([
    5.150556538, 0.204543779, 0.705652674, 0.330650192, 0.872594728, 0.884433876, 0.048044107,
    0.025072878, 0.0, 0.057804317, 0.041853068, -0.004476248, -0.035255637, 0.0, 0.110919341,
    0.0, 0.0, 0.0, 0.0, -0.021003756, 0.0])
ENET_ADMM = np.array([
    5.150497437, 0.204526767, 0.705665767, 0.330640256, 0.872611761, 0.884422064, 0.048055928,
    0.025097074, 0.0, 0.057830613, 0.041876025, -0.004499977, -0.035279647, 0.0, 0.110915266,
    0.0, 0.0, 0.0, 0.0, -0.020984368, 0.0])
# README.md:139-161  admm_lad(x, y, intercept = FALSE)$fit()$beta[-1]
LAD_RQ = np.array([
    0.463871497, 0.829243353, 0.151432833, 1.074107564, 0.958979798, 0.502539859, 0.337640338,
    0.209127703, 0.361765382, 0.323168985, -0.002009264, -0.036099511, 0.328007777, 0.296038071,
    0.310187867, 0.071713681, 0.166827429, 0.260366502, 0.324487629, 0.209758565])
LAD_ADMM = np.array([
    0.4630289961, 0.8324149339, 0.1493799430, 1.0707590072, 0.9569585188, 0.5028832829, 0.3360263689,
    0.2120946512, 0.3630356485, 0.3217875563, 0.0007319653, -0.0370447075, 0.3290499302, 0.2992857234,
    0.3117528782, 0.0711670377, 0.1622600454, 0.2580854533, 0.3251952295, 0.2131039214])
# README.md:180-182  range(beta_true - out_admm$beta), n=50 p=100
BP_RANGE = (-0.0006052779, 0.0004780069)
# README.md:389-393  same with n=1000, p=2000, nsig=100
BP_PERF_RANGE = (-0.001267782, 0.002108828)
