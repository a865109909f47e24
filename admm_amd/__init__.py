"""admm_amd -- MI355X (gfx950) ADMM solvers behind the R interface of yixuan/ADMM.

Only what the hot path needs: `csrc/` (HIP kernels + the C ABI of include/admm_hip.h, built into
lib/libadmm_hip.so by `python -m admm_amd.build`) and `api.py`, the host-side mirror of the
reference's `admm_lasso()/admm_enet()/admm_lad()/admm_bp()` builder chain.
"""
from .api import (LassoPlan, ADMM_BP, ADMM_Dantzig, ADMM_Enet, ADMM_LAD, ADMM_Lasso, admm_bp, admm_dantzig, admm_enet, admm_lad, admm_lasso)
from ._lib import AdmmHipError, DevicePtr, load, options

__all__ = ["admm_lasso", "admm_enet", "admm_lad", "admm_bp", "admm_dantzig", "ADMM_Dantzig", "ADMM_Lasso", "ADMM_Enet", "ADMM_LAD", "ADMM_BP",
           "LassoPlan", "DevicePtr", "AdmmHipError", "load", "options"]
