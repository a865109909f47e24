"""Child of tests/test_gpu_sanitizers.py: every solver family once, small, through the host-sanitized build (ASan + UBSan on the
host half: plan construction, the batch-enqueue loop drivers, result marshalling, the trace / state read-back, teardown)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import admm_amd  # noqa: E402
from admm_amd.api import LassoPlan  # noqa: E402
from helpers import synth_lasso  # noqa: E402

lib = admm_amd.load()
assert lib.admm_hip_device_count() >= 1
x, y = synth_lasso(700, 60, 8, seed=1)
f = admm_amd.admm_lasso(x, y).penalty(nlambda=8).fit(); assert f.niter.min() > 0
f = admm_amd.admm_enet(x, y).penalty(nlambda=6, alpha=0.5).fit()
plan = LassoPlan(admm_amd.admm_lasso(x, y).penalty(nlambda=5)); plan.enable_trace(4096); plan.enable_state(4096)
plan.run(); tr = plan.read_trace(); stt = plan.read_state(); plan.run(); plan.close()
assert len(tr) > 5 and len(stt) > 5
xt, yt = synth_lasso(2600, 2100, 20, seed=2)                       # p >= 2048: the symmetric mat-vec path and the blocked inverse
f = admm_amd.admm_lasso(xt, yt).penalty(nlambda=3).fit()
xw, yw = synth_lasso(90, 700, 8, seed=3)
f = admm_amd.admm_lasso(xw, yw).penalty(nlambda=8).fit()            # wide: fused x-update + persistent active-set stretches
f = admm_amd.admm_enet(xw, yw).penalty(nlambda=5, alpha=0.3).fit()
m = admm_amd.admm_lasso(x, y).penalty(nlambda=4); m.parallel(3); f = m.fit()         # consensus, Cholesky blocks
m = admm_amd.admm_lasso(xw, yw).penalty(nlambda=3).opts(maxit=200); m.nthread = 4; f = m.fit()   # Woodbury blocks
cv = admm_amd.admm_lasso(x, y).penalty(nlambda=6).cv(nfolds=3, keep_fold_beta=True)
os.environ["ADMM_HIP_CV_DOWNDATE"] = "1"
cv = admm_amd.admm_lasso(x, y).penalty(nlambda=6).cv(nfolds=3)
os.environ.pop("ADMM_HIP_CV_DOWNDATE")
fits = admm_amd.admm_lasso(x, y).penalty(nlambda=4).fit_responses(np.stack([y, y[::-1]], axis=1)); assert len(fits) == 2
rng = np.random.default_rng(4)
xl = rng.standard_normal((400, 30)); yl = xl @ rng.standard_normal(30) + rng.standard_cauchy(400) * 0.1
f = admm_amd.admm_lad(xl, yl).fit(trace=True); assert f.niter > 0
A = rng.standard_normal((60, 150)); b0 = np.zeros(150); b0[:6] = 1.0
f = admm_amd.admm_bp(A, A @ b0).fit(trace=True)
f = admm_amd.admm_bp(A, A @ b0).parallel(3).fit(trace=True); assert np.abs(f.beta.toarray().ravel() - b0).max() < 5e-3
f = admm_amd.admm_dantzig(x, y).penalty(nlambda=4, lambda_min_ratio=0.1).opts(maxit=500).fit(trace=True)
try:                                                                # an error path that unwinds through device buffers
    admm_amd.admm_bp(A, A @ b0).parallel(1000).fit()
except RuntimeError:
    pass
print("gpu paths ok", flush=True)
# leave without the interpreter's / the HSA runtime's exit handlers: ASan's ROCm allocator shim aborts there on its own
# bookkeeping ("dev_runtime_unloaded_" CHECK inside libhsa-runtime64 teardown), after every call of ours has returned
sys.stdout.flush(); sys.stderr.flush()
os._exit(0)
