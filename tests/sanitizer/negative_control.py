"""Child of tests/test_sanitizers.py: proof that the sanitized build is the one loaded and that it reports -- the host Lanczos
driver is handed a 30 x 30 float matrix and told it is 40 x 40: the mat-vec reads past the end of the heap block."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from admm_amd import _lib  # noqa: E402

lib = _lib.load()
buf = (ctypes.c_float * 900)(*([1.0] * 900))                      # 3600 bytes: a malloc block of its own
out, nm = ctypes.c_float(), ctypes.c_int()
lib.admm_hip_host_lanczos(ctypes.cast(buf, ctypes.POINTER(ctypes.c_float)), 40, ctypes.byref(out), ctypes.byref(nm))
print("not reported", flush=True)
