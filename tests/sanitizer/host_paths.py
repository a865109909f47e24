"""Child of tests/test_sanitizers.py: the library's HOST-only paths under ASan + UBSan (no GPU needed).
Exports, argument validation and its error strings, struct round trips, the host Lanczos driver (float), the no-device path."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from admm_amd import _lib  # noqa: E402
from admm_amd._lib import AdmmOpts, AdmmStats  # noqa: E402

lib = _lib.load()
assert b"gfx950" in lib.admm_hip_version()
for sym in _lib.EXPORTS:
    getattr(lib, sym)
# ---- argument validation: every early return writes the error string
x = np.asfortranarray(np.ones((6, 3))); y = np.ones(6)
lo = np.zeros(4); beta = np.zeros(16, dtype=np.float32); betad = np.zeros(16); nit = np.zeros(4, dtype=np.int32)
dp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
ip = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
bad = AdmmOpts(0, 1e-5, 1e-5, -1.0)
ok = AdmmOpts(10, 1e-5, 1e-5, -1.0)
st = AdmmStats()
assert lib.admm_hip_lasso(x.ctypes.data, y.ctypes.data, 6, 3, 0, None, 0, 4, 1e-4, 1, 1, ctypes.byref(bad), dp(lo), fp(beta), ip(nit), ctypes.byref(st)) == 1
assert b"maxit" in lib.admm_hip_last_error()
assert lib.admm_hip_enet(x.ctypes.data, y.ctypes.data, 6, 3, 0, None, 0, 4, 1e-4, 1, 1, 2.0, ctypes.byref(ok), dp(lo), fp(beta), ip(nit), None) == 1
assert lib.admm_hip_lasso(x.ctypes.data, y.ctypes.data, 6, 3, 0, None, 0, 4, 2.0, 1, 1, ctypes.byref(ok), dp(lo), fp(beta), ip(nit), None) == 1
neg = np.array([0.5, -1.0])
assert lib.admm_hip_lasso(x.ctypes.data, y.ctypes.data, 6, 3, 0, neg.ctypes.data, 2, 0, 1e-4, 1, 1, ctypes.byref(ok), dp(lo), fp(beta), ip(nit), None) == 1
assert lib.admm_hip_bp(x.ctypes.data, y.ctypes.data, 6, 3, 0, ctypes.byref(AdmmOpts(10, 1e-4, 1e-4, 1.0)), dp(betad), ip(nit), None) == 1      # p <= n
assert lib.admm_hip_parbp(x.ctypes.data, y.ctypes.data, 6, 3, 0, 2, ctypes.byref(AdmmOpts(10, 1e-4, 1e-4, 1.0)), dp(betad), ip(nit), None) == 1
assert lib.admm_hip_dantzig(x.ctypes.data, y.ctypes.data, 6, 3, 0, None, 0, 0, 1e-4, 1, 1, ctypes.byref(ok), dp(lo), dp(betad), ip(nit), None) == 1
assert lib.admm_hip_lad(x.ctypes.data, y.ctypes.data, 3, 6, 0, 1, ctypes.byref(AdmmOpts(10, 1e-4, 1e-4, 1.0)), dp(betad), ip(nit), None) != 0
# ---- no device in this process: every solver entry point must fail with ADMM_ERR_NO_DEVICE, not touch memory
if not os.path.exists("/dev/kfd"):
    rc = lib.admm_hip_lasso(x.ctypes.data, y.ctypes.data, 6, 3, 0, None, 0, 4, 1e-4, 1, 1, ctypes.byref(ok), dp(lo), fp(beta), ip(nit), ctypes.byref(st))
    assert rc == 2, (rc, lib.admm_hip_last_error())
# ---- the host Lanczos driver: restart branch included (matrices of tests/test_cabi_host.py), plus degenerate inputs
rng = np.random.default_rng(0)
for (n, p) in [(100, 20), (400, 60), (50, 200), (1000, 150), (40, 3)]:
    X = (rng.standard_normal((n, p)) * 2).astype(np.float32)
    G = np.asfortranarray((X.T @ X).astype(np.float32))
    out, nm = ctypes.c_float(), ctypes.c_int()
    assert lib.admm_hip_host_lanczos(G.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), p, ctypes.byref(out), ctypes.byref(nm)) == 0
    assert out.value > 0 and nm.value >= 3
Z = np.zeros((5, 5), dtype=np.float32, order="F")
out, nm = ctypes.c_float(), ctypes.c_int()
lib.admm_hip_host_lanczos(Z.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), 5, ctypes.byref(out), ctypes.byref(nm))      # zero matrix: any rc, no UB
rc = lib.admm_hip_host_lanczos(Z.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), 2, ctypes.byref(out), ctypes.byref(nm))  # too small for ncv = 3
assert rc != 0
print("host paths ok", flush=True)
# leave without the interpreter's / the HSA runtime's exit handlers: ASan's ROCm allocator shim aborts there on its own
# bookkeeping ("dev_runtime_unloaded_" CHECK inside libhsa-runtime64 teardown), after every call of ours has returned
sys.stdout.flush(); sys.stderr.flush()
os._exit(0)
