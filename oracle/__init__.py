"""CPU oracle for the yixuan/ADMM hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

A NumPy restatement of the reference's solvers (reference = /root/reference,
R package ADMM 1.0): every function cites the reference file:line it follows.
Beside it:
  c/admm_tall_cpu.c + ctall.py  the tall loop (FADMMBase::solve + ADMMLassoTall / ADMMEnetTall) restated in C
                                (gcc, built by __graft_entry__.build()): the compiled CPU baseline of bench.py and a
                                second, independent restatement pinned to the same README vectors;
  variants.py                   the tall x-update with mathematically identical roundings (float LLT = the reference,
                                float inverse, inverse rounded from double, exact): how far two correct executions of
                                the reference's arithmetic drift apart (tests/test_flip_floor.py);
  solvers.FADMM follow mode     the oracle taking another execution's outcome of a threshold test ONLY where its own
                                deciding quantity is within 8 float ulps of rounding noise of the threshold
                                (FollowMismatch otherwise) -- the GPU parity tests' rule (tests/helpers.py);
  tracecmp.py                   lock-step comparison of two decision traces (dev tools).
Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import this package, and only as the checker / reported CPU
baseline.  Nothing under `admm_amd/` imports it; the product path fails
loudly when the HIP library is missing.

Pinning.  The reference cannot be compiled here (needs R,
Rcpp, RcppEigen/Eigen -- none present, no network) and has no test-suite; its
only known-answer vectors are the README snippets.  This oracle is pinned
against all of them (tests/test_oracle_readme.py):
  * Lasso `admm` column         README.md:66-88
  * Lasso `paradmm` column      README.md:66-88  (2 row blocks)
  * Elastic net alpha=0.5       README.md:100-123
  * LAD                         README.md:139-161
  * Basis pursuit error range   README.md:180-182 and :389-393 (n=1000, p=2000)
Third-party arithmetic the reference delegates to and that is not vendored:
Eigen (dense products, LLT, SparseVector; RcppEigen, version unpinned in
DESCRIPTION:13-15) and R's BLAS.  These are standard dense linear algebra;
NumPy/LAPACK float32/float64 calls stand in for them here.
NOT pinned by any reference vector ("parity unpinned" rows): the wide solver
(ADMMLassoWide/ADMMEnetWide), multi-lambda warm starts, flags 0/2 of DataStd, the
Woodbury branch of PADMMLasso, the n>2000 branch of ADMMLAD.  For those this
restatement (sharing all code with the pinned cases) is the oracle, cross-checked
against scikit-learn's optimum within the solver's tolerance.
"""
