/*
 * admm_hip.h -- C ABI of libadmm_hip.so, the MI355X (gfx950) replacement for the
 * numeric part of the reference's five `.Call` entry points (yixuan/ADMM 1.0).
 *
 * Boundary: R calls `.Call("<sym>", ..., PACKAGE = "ADMM")`
 * by name; each RcppExport function unpacks SEXPs, runs the solver and builds an R
 * list.  The functions below take exactly the unpacked values of those calls
 * (plain pointers and sizes, column-major double like R memory), run the whole
 * solve on the GPU and fill caller-owned output arrays.  An Rcpp shim that keeps
 * the reference's symbol names and forwards here is shown in INTEGRATION.md.
 *
 *   admm_hip_lasso     replaces  admm_lasso     /root/reference/src/Lasso.cpp:32-137
 *   admm_hip_enet      replaces  admm_enet      /root/reference/src/Enet.cpp:31-137
 *   admm_hip_parlasso  replaces  admm_parlasso  /root/reference/src/ParLasso.cpp:33-110
 *   admm_hip_lad       replaces  admm_lad       /root/reference/src/LAD.cpp:16-47
 *   admm_hip_bp        replaces  admm_bp        /root/reference/src/BP.cpp:20-45
 *
 * Conventions
 *   - x: n x p column-major (leading dimension n), y: length n.  Borrowed and
 *     read-only for the duration of the call (the reference copies or maps them:
 *     Lasso.cpp:42-50, LAD.cpp:20-21, BP.cpp:24-27).  `mem` says where they live:
 *     ADMM_MEM_HOST (what R hands over) or ADMM_MEM_DEVICE (already resident in
 *     HBM on the current HIP device; used by bench.py so that PCIe is not in the
 *     timed region).
 *   - Return value: 0 on success, an ADMM_ERR_* code otherwise; the message is
 *     available from admm_hip_last_error().  Nothing throws across the ABI and
 *     nothing calls the R API.  There is no CPU fallback: without a usable HIP
 *     device every solver returns ADMM_ERR_NO_DEVICE.
 *   - No state survives a call (the reference creates and destroys its solver
 *     inside the .Call: Lasso.cpp:74-76,126-129).
 */
#ifndef ADMM_HIP_H
#define ADMM_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* the library is built with -fvisibility=hidden; only the functions below are exported */
#if defined(__GNUC__)
#define ADMM_HIP_API __attribute__((visibility("default")))
#else
#define ADMM_HIP_API
#endif

#define ADMM_OK 0
#define ADMM_ERR_INVALID_ARG 1
#define ADMM_ERR_NO_DEVICE 2
#define ADMM_ERR_HIP 3
#define ADMM_ERR_BLAS 4
#define ADMM_ERR_NOT_SPD 5      /* Cholesky of the Gram (+rho I) failed; the reference never checks LLT::info() */
#define ADMM_ERR_EIGS 6         /* Lanczos produced no converged Ritz value (the reference then reads evals[0] of an empty vector) */
#define ADMM_ERR_COMM 7
#define ADMM_ERR_INTERNAL 8
#define ADMM_ERR_MEMORY 9       /* the problem does not fit the device: the tall solver caches a p x p inverse (4 p^2 bytes, 160 GB at p = 2e5) --
                                   use the consensus solver ($parallel(): row blocks, Woodbury form) or the wide solver's shape instead */

#define ADMM_MEM_HOST 0
#define ADMM_MEM_DEVICE 1

/* The R `opts` list: list(maxit=, eps_abs=, eps_rel=, rho=)  (R/30_admm_lasso.R:139-146).
 * rho <= 0 means "choose automatically" (ADMMLassoTall.h:194-202, ADMMLassoWide.h:227-228,
 * PADMMLasso.h:199-200); LAD/BP use it as given (R default 1.0). */
typedef struct admm_opts {
    int maxit;
    double eps_abs;
    double eps_rel;
    double rho;
} admm_opts;

/* Optional per-call measurements (may be NULL). Times are seconds. */
typedef struct admm_stats {
    double t_h2d;          /* host -> device copies (0 for ADMM_MEM_DEVICE inputs) */
    double t_standardize;  /* convert + DataStd */
    double t_gram;         /* X'X or XX' */
    double t_eigs;         /* Lanczos lambda_max estimate */
    double t_factor;       /* Cholesky + cached inverse / triangular solves */
    double t_loop;         /* the ADMM iterations over all lambdas (wall, host clock around a device sync) */
    double t_total;
    double loop_ms_events; /* same loop measured with HIP events on the solver's stream */
    double xupdate_ms_avg; /* average duration of the x-update kernel, HIP events on the solver's stream (sampled launches) */
    long long xupdate_samples;
    long long total_iter;  /* sum of niter */
    long long xupdate_launches;
    double rho;            /* rho actually used (first lambda) */
    double eig_est;        /* the loose Lanczos value (lambda_max or spectral-radius estimate) */
    int branch;            /* 0 tall (Cholesky), 1 wide (linearised), 2 consensus; 6 admm_parbp (column-block sharing); 7 admm_dantzig */
    int xupdate_variant;   /* admm_hip_lad: 1 = one pass over the rows of X per iteration (general branch, p <= 6144), 0 = two products / hat matrix;
                              wide path: 1 / 2 = the regular steps ran screened through the fp16 copy / the 8-bit code of X (+ the exact step on the
                              few columns the bound does not settle: bit-identical iterates, 2np / np instead of 4np bytes), 0 = unscreened;
                              tall path: 0 = full-matrix mat-vec (4p^2 B), 1 = lower-triangle symmetric mat-vec (2p^2 B),
                              2 = the same with the tiles dealt out to the ranks + one all-reduce of 2p floats (admm_hip_lasso_dist),
                              (3, a single-launch iteration, existed in rounds 3 - 5: measured slower, removed in round 6);
                              admm_hip_parbp: + 4 = the regular iterations ran screened (as the wide path's);
                              0 = every active-set iteration streams the non-zero columns twice, 1 = active-set iterations
                              in Gram space (one |U| x |U| mat-vec each; xupdate_launches = stretches of nine that ran so, persist_iter =
                              times the column set U was rebuilt), 2 = 1 except for stretches of 100, 200, 400, ... iterations after the non-zeros
                              outgrew the Gram matrix, which ran as 0 */
    int exchange_variant;  /* sharded solvers: how the per-iteration exchange ran -- 0 none, 1 the exchange layer's all-reduce, 2 PEER slots written /
                              read by the solver's own kernels (producer and consumer launches), 3 the same with producer and consumer in ONE
                              launch (chosen only when that launch is resident as a whole: its workgroups wait for one another) */
    int refine;            /* tall path: 1 = every x-update refined once with a double-precision residual (ADMM_HIP_REFINE=1) */
    long long persist_iter;/* wide path: iterations that ran inside persistent active-set launches (of total_iter); -1: a hand-over of a stretch timed out and the whole path was run again without stretches */
    double factor_flops;   /* row-sharded tall solver: flops of the Cholesky + inverse THIS rank performed when the factorisation is
                              distributed over the ranks (block columns dealt out, panels broadcast); 0 when every
                              rank factorises the whole matrix */
} admm_stats;

/* lambda_in: user grid of length nlambda_in (sorted decreasing by the R wrapper), or NULL/0
 * for the automatic log-spaced grid of nlambda_auto values from lambda_0 * scaleY / n down to
 * lmin_ratio times that (Lasso.cpp:78-89).
 * Outputs (nl = nlambda_in > 0 ? nlambda_in : nlambda_auto):
 *   lambda_out[nl]            the grid used
 *   beta_out[(p+1) * nl]      column-major dense float, row 0 = intercept, on the ORIGINAL scale
 *                             (DataStd::recover; the reference returns this as a dgCMatrix whose
 *                             structural zeros are exactly the zeros here, Lasso.cpp:22-30)
 *   niter_out[nl]             ADMM iterations per lambda
 */
ADMM_HIP_API int admm_hip_lasso(const double* x, const double* y, int n, int p, int mem,
                   const double* lambda_in, int nlambda_in, int nlambda_auto, double lmin_ratio,
                   int standardize, int intercept, const admm_opts* opts,
                   double* lambda_out, float* beta_out, int* niter_out, admm_stats* stats);

/* As admm_hip_lasso with the elastic-net mixing weight alpha (Enet.cpp:63, ADMMEnet.h:24-57). */
ADMM_HIP_API int admm_hip_enet(const double* x, const double* y, int n, int p, int mem,
                  const double* lambda_in, int nlambda_in, int nlambda_auto, double lmin_ratio,
                  int standardize, int intercept, double alpha, const admm_opts* opts,
                  double* lambda_out, float* beta_out, int* niter_out, admm_stats* stats);

/* K-fold cross-validation of the lambda path of admm_hip_lasso (alpha < 0) / admm_hip_enet (alpha in [0, 1]).
 * SURVEY.md section 8(f) row n4 ("cross-validation folds as independent replicas across GPUs"); the reference package has
 * no CV driver, so there is no reference interface to cite -- the fold fits are the admm_lasso / admm_enet path
 * (/root/reference/src/Lasso.cpp:32-135, Enet.cpp:31-34) on row subsets.
 *   grid      the full data's (lambda_in, or the automatic grid of the full-data fit, Lasso.cpp:78-89), used by every fold;
 *   fold f    fitted on the rows with fold_id[i] != f by the SAME plan a direct call on that row subset gets (coefficients
 *             bit-identical to it), scored on the rows with fold_id[i] == f: mean squared prediction error per lambda, on
 *             the original scale, computed in double on the device.  fold_id NULL: i mod nfolds;
 *   ranks     with a communicator attached, fold f runs on rank f mod nranks (every rank is handed the full x, y;
 *             nothing is exchanged on the data path) and the tables are summed over the ranks at the end: all ranks
 *             return identical outputs.
 * Outputs: lambda_out[nlam]; beta_out (p+1) x nlam and niter_out[nlam] of the full-data fit (either may be NULL);
 * cv_mean[nlam]; cv_se[nlam] = sample sd over the folds / sqrt(nfolds); fold_mse[nfolds x nlam] (fold-major),
 * fold_niter[nfolds x nlam], fold_beta[nfolds x (p+1) x nlam] (each may be NULL); idx_min = argmin cv_mean;
 * idx_1se = index of the largest lambda with cv_mean <= cv_mean[idx_min] + cv_se[idx_min].  stats: the full-data fit's,
 * with t_total = the whole call. */
ADMM_HIP_API int admm_hip_lasso_cv(const double* x, const double* y, int n, int p, int mem, const int* fold_id, int nfolds,
                      const double* lambda_in, int nlambda_in, int nlambda_auto, double lmin_ratio,
                      int standardize, int intercept, double alpha, const admm_opts* opts,
                      double* lambda_out, float* beta_out, int* niter_out,
                      double* cv_mean, double* cv_se, double* fold_mse, int* fold_niter, float* fold_beta,
                      int* idx_min, int* idx_1se, admm_stats* stats);

/* Several responses of one design matrix: Y is n x m column-major, response j is the ordinary admm_hip_lasso (alpha < 0) /
 * admm_hip_enet (alpha in [0, 1]) fit of (x, Y[:, j]) -- coefficients, grids and iteration counts bit-identical to that call
 * -- but x is uploaded, converted and standardised once and, for the tall solver (n > p), X'X is formed once (it does not
 * depend on y; the cached inverse does, through rho, and is rebuilt per response).  SURVEY.md section 8(f) row n4; the
 * reference has no multi-response entry (its fits are one .Call each, /root/reference/src/Lasso.cpp:32-135).  With a
 * communicator attached response j runs on rank j mod nranks (every rank is handed the full x, Y; nothing is exchanged on
 * the data path) and the outputs are summed over the ranks at the end: all ranks return identical outputs.
 * Outputs (response-major): lambda_out[m x nlam], beta_out[m x (p+1) x nlam], niter_out[m x nlam], stats[m] (may be NULL;
 * the one-time shared work is charged to the first response of each rank; entries of responses fitted by other ranks stay zero). */
ADMM_HIP_API int admm_hip_lasso_multi(const double* x, const double* Y, int n, int p, int m, int mem,
                         const double* lambda_in, int nlambda_in, int nlambda_auto, double lmin_ratio,
                         int standardize, int intercept, double alpha, const admm_opts* opts,
                         double* lambda_out, float* beta_out, int* niter_out, admm_stats* stats);

/* Row-block consensus ADMM with `nthread` blocks (ParLasso.cpp:33-36,71-72: nthread only sets the
 * number of blocks K).  All K blocks run on the current device. */
ADMM_HIP_API int admm_hip_parlasso(const double* x, const double* y, int n, int p, int mem,
                      const double* lambda_in, int nlambda_in, int nlambda_auto, double lmin_ratio,
                      int standardize, int intercept, int nthread, const admm_opts* opts,
                      double* lambda_out, float* beta_out, int* niter_out, admm_stats* stats);

/* beta_out[p+1] (row 0 = intercept), niter_out[1]. Requires n > p (R/20_admm_lad.R:21-22). */
ADMM_HIP_API int admm_hip_lad(const double* x, const double* y, int n, int p, int mem, int intercept,
                 const admm_opts* opts, double* beta_out, int* niter_out, admm_stats* stats);

/* beta_out[p], niter_out[1]. Requires p > n (R/10_admm_bp.R:30-31). */
ADMM_HIP_API int admm_hip_bp(const double* x, const double* y, int n, int p, int mem,
                const admm_opts* opts, double* beta_out, int* niter_out, admm_stats* stats);

/* The Dantzig selector path, min ||beta||_1 s.t. ||X'(X beta - y)||_inf <= lambda -- what R's admm_dantzig(x, y)$fit() asks for,
 * .Call("admm_dantzig", x, y, lambda, nlambda, lambda_min_ratio, standardize, intercept, opts) (R/50_admm_dantzig.R:30-46), a
 * symbol the reference never builds (src/TODO/Dantzig.cpp:32-99, src/TODO/ADMMDantzig.h against an older ADMMBase).  Restated
 * on the current ADMMBase::solve (admm_amd/csrc/dantzig.hip, oracle/solvers.py Dantzig).  Arguments as admm_hip_lasso; opts->rho
 * <= 0: automatic (1 / loose Lanczos value of X'X).  Everything is double: beta_out[(p + 1) * nl] column-major, row 0 = intercept.
 * niter_out[l] = maxit + 1 when lambda l did not converge.  NOTE (tests/test_oracle_dantzig.py): the iteration converges on
 * comfortably tall problems (n >= 5 p) and does not for p > n -- the algorithm as the reference holds it.
 * _traced: decision records, layout ADMM_TRACE_* ([0] lambda index, [11] internal lambda; CONVERGED / CONTINUE). */
ADMM_HIP_API int admm_hip_dantzig(const double* x, const double* y, int n, int p, int mem,
                                  const double* lambda_in, int nlambda_in, int nlambda_auto, double lmin_ratio,
                                  int standardize, int intercept, const admm_opts* opts,
                                  double* lambda_out, double* beta_out, int* niter_out, admm_stats* stats);
ADMM_HIP_API int admm_hip_dantzig_traced(const double* x, const double* y, int n, int p, int mem,
                                         const double* lambda_in, int nlambda_in, int nlambda_auto, double lmin_ratio,
                                         int standardize, int intercept, const admm_opts* opts,
                                         double* lambda_out, double* beta_out, int* niter_out, admm_stats* stats,
                                         double* trace_out, long long trace_cap, long long* ntrace_out);

/* Basis pursuit with the COLUMNS of x in `nthread` blocks -- what R's admm_bp(x, y)$parallel(nthread)$fit() asks for:
 * .Call("admm_parbp", x, y, nthread, list(maxit, eps_abs, eps_rel, rho_ratio = rho)) (R/10_admm_bp.R:111-116), a symbol the
 * reference never builds (src/TODO/ParBP.cppp:26-71, src/TODO/PADMMBP.h against a base class that no longer exists).  The
 * feature-split "sharing" ADMM of that source, restated on the current PADMMBase_Master loop (admm_amd/csrc/sharing_bp.hip,
 * oracle/solvers.py SharingBP).  opts->rho carries rho_ratio (R default 1): rho = 1 / (rho_ratio * mean_i lambda_max(A_i'A_i)).
 * Partition as PADMMBP.h:150-167: nthread - 1 blocks of p div nthread columns, the last takes the remainder.  n <= 16384.
 * beta_out[p] dense doubles (the reference returns a one-column dgCMatrix), niter_out[1] (maxit + 1 when not converged, as
 * PADMMBase_Master::solve returns).  _traced: decision records as for admm_hip_bp_traced ([11] = 1 on a regular iteration,
 * outcome ADMM_TRACE_CONVERGED / ADMM_TRACE_CONTINUE).
 * _dist: the blocks spread over the ranks of the attached communicator -- this rank passes columns
 * [col_offset, col_offset + p_local) of the p_total, whole blocks of the partition above, and gets their coefficients back;
 * one sum all-reduce of n + O(n / 32) doubles per iteration (the "all-reduce of X_i beta_i" of SURVEY.md section 8f row n2). */
ADMM_HIP_API int admm_hip_parbp(const double* x, const double* y, int n, int p, int mem, int nthread, const admm_opts* opts,
                                double* beta_out, int* niter_out, admm_stats* stats);
ADMM_HIP_API int admm_hip_parbp_traced(const double* x, const double* y, int n, int p, int mem, int nthread, const admm_opts* opts,
                                       double* beta_out, int* niter_out, admm_stats* stats,
                                       double* trace_out, long long trace_cap, long long* ntrace_out);
ADMM_HIP_API int admm_hip_parbp_dist(const double* x_cols, const double* y, int n, int p_local, long long p_total, long long col_offset,
                                     int mem, int nthread, const admm_opts* opts, double* beta_local_out, int* niter_out,
                                     admm_stats* stats);

/* admm_hip_lad / admm_hip_bp that also return the decision trace (layout below, ADMM_TRACE_*; lambda index 0):
 * trace_out receives min(decisions taken, trace_cap) records of ADMM_TRACE_FIELDS doubles, *ntrace_out their number. */
ADMM_HIP_API int admm_hip_lad_traced(const double* x, const double* y, int n, int p, int mem, int intercept, const admm_opts* opts,
                        double* beta_out, int* niter_out, admm_stats* stats, double* trace_out, long long trace_cap, long long* ntrace_out);
ADMM_HIP_API int admm_hip_bp_traced(const double* x, const double* y, int n, int p, int mem, const admm_opts* opts,
                       double* beta_out, int* niter_out, admm_stats* stats, double* trace_out, long long trace_cap, long long* ntrace_out);

/* admm_hip_lad_traced / admm_hip_bp_traced that also return the ITERATE DUMP (test / diagnosis facility, oracle/stepcheck.py):
 * state_out receives min(decisions taken, state_cap) records of 5 * dim doubles  x | z | y | adj_z | adj_y  (main_x, aux_z, dual_y
 * and the extrapolated pair the iteration started from, FADMMBase.h:185-211; dim = n for LAD, ADMMLAD.h:62-107, p for BP,
 * ADMMBP.h:48-93), record s = the iterates trace record s judged; record 0 (the cold start) carries in its x slot the data
 * vector as the library holds it (LAD: the standardised y; BP: A'(AA')^-1 b, ADMMBP.h:170).  Needs trace_cap > 0. */
ADMM_HIP_API int admm_hip_lad_state(const double* x, const double* y, int n, int p, int mem, int intercept, const admm_opts* opts,
                       double* beta_out, int* niter_out, admm_stats* stats, double* trace_out, long long trace_cap, long long* ntrace_out,
                       double* state_out, long long state_cap, long long* nstate_out);
ADMM_HIP_API int admm_hip_bp_state(const double* x, const double* y, int n, int p, int mem, const admm_opts* opts,
                      double* beta_out, int* niter_out, admm_stats* stats, double* trace_out, long long trace_cap, long long* ntrace_out,
                      double* state_out, long long state_cap, long long* nstate_out);

/* Prepared-problem variant of the Lasso family (the "persistent context" anticipated for a
 * re-fitting caller; the R shim does not need it).  create = everything the reference does
 * before its lambda loop (copy/convert, DataStd, X'y, Gram, Spectra, factorisation:
 * Lasso.cpp:42-76 + ADMM*::init); run = the warm-started lambda loop of Lasso.cpp:97-124 from a
 * cold start, repeatable; destroy frees all device memory.  alpha < 0 selects the Lasso prox,
 * 0 <= alpha <= 1 the elastic net; nthread > 1 selects the consensus solver (Lasso only: alpha >= 0 with nthread > 1 is
 * ADMM_ERR_INVALID_ARG -- the reference has no parallel elastic net).
 * bench.py times run() only (ADMM iterations/s excludes the one-time setup). */
typedef struct admm_hip_plan admm_hip_plan;
ADMM_HIP_API int admm_hip_lasso_plan_create(const double* x, const double* y, int n, int p, int mem,
                               const double* lambda_in, int nlambda_in, int nlambda_auto, double lmin_ratio,
                               int standardize, int intercept, double alpha, int nthread, const admm_opts* opts,
                               admm_hip_plan** plan_out, int* nlambda_out);
ADMM_HIP_API int admm_hip_lasso_plan_run(admm_hip_plan* plan, double* lambda_out, float* beta_out, int* niter_out, admm_stats* stats);
ADMM_HIP_API int admm_hip_lasso_plan_destroy(admm_hip_plan* plan);

/* Decision trace of a prepared Lasso-family problem (tall, wide and consensus solvers): what the reference's commented-out iteration table
 * (print_row, FADMMBase.h:135-170, ADMMBase.h:111-146) would print, recorded on the device by the iteration control itself, one record
 * per decision (the cold-start decision first, then one per ADMM iteration, over all lambdas of a run in order).
 * enable() before run(); read() afterwards returns min(records of the last run, capacity, cap_records) records of
 * ADMM_TRACE_FIELDS doubles:
 *   [0] lambda index  [1] iteration i within the lambda  [2] eps_primal  [3] eps_dual  (the thresholds iteration i was tested against)
 *   [4] resid_primal  [5] resid_dual  [6] c = rho r_p^2 + rho ||z - adj_z||^2 (0 when converged)  [7] c_old (adj_c before)
 *   [8] outcome ADMM_TRACE_*  [9] rho the iteration ran with  [10] rho after this decision (the adaptation of
 *   FADMMBase.h:109-133 / ADMMBase.h:85-109 where the solver adapts: wide, LAD, BP)  [11] the penalty lambda the iteration
 *   ran with, in the solver's internal units (lambda n / scaleY, Lasso.cpp:99; 0 for LAD / BP)
 * Wide solver (ADMMBase::solve, rho adaptation ADMMBase.h:85-109): [6] rho AFTER this decision's adaptation, [7] kind of
 * the x-update that follows (0 zero, 1 regular, 2 active set), [9] rho before; consensus solver: [6] = [9] = rho, [7] = 0.
 * The parity tests use it to show that a lambda whose iteration count differs from the oracle's diverged at a
 * threshold test decided inside rounding noise, instead of excusing count differences wholesale. */
#define ADMM_TRACE_FIELDS 12
#define ADMM_TRACE_COLD (-1)        /* first decision of a run: nothing to test yet */
#define ADMM_TRACE_CONVERGED 0      /* r_p < eps_p and r_d < eps_d  (FADMMBase.h:213-217,237-238) */
#define ADMM_TRACE_ACCELERATE 1     /* c < 0.999 c_old              (FADMMBase.h:243-249) */
#define ADMM_TRACE_RESTART 2        /* otherwise                    (FADMMBase.h:250-256) */
#define ADMM_TRACE_CONTINUE 1       /* wide / consensus solvers (no acceleration): not converged    (ADMMBase.h:206-207, PADMMBase.h:230-231) */
ADMM_HIP_API int admm_hip_lasso_plan_trace_enable(admm_hip_plan* plan, long long capacity_records);
ADMM_HIP_API int admm_hip_lasso_plan_trace_read(admm_hip_plan* plan, double* out, long long cap_records, long long* nrecords_out);
/* Iterate dump of a prepared problem (tall, wide and consensus solvers; test / diagnosis facility): the vectors every ADMM
 * iteration leaves behind, in the solver's own (standardised) units, one record per decision with the SAME numbering as
 * the decision trace -- record s holds the iterates whose residuals trace record s judged (record 0, the cold start, is
 * zero).  Tall solver (FADMMBase.h:185-211): 5 p floats  x | z | y | adj_z | adj_y  (main_x, aux_z, dual_y and the
 * extrapolated pair the iteration started from).  Consensus solver (PADMMBase.h:174-214), K row blocks:
 * (1 + 2 K) p floats  z | x_0 .. x_{K-1} | y_0 .. y_{K-1}.  Wide solver (ADMMBase.h:158-216, ADMMLassoWide.h:129-170; single process):
 * p + 3 n floats  x | A x | z | y  (main_x, cache_Ax, aux_z, dual_y), written by the two-launch path and by the persistent
 * active-set launches alike; record 0 is zero.  With it a test can replay the reference's arithmetic for
 * ONE iteration from the library's own previous iterates (oracle/stepcheck.py): every step of a run is then checked on its
 * own, however far two executions have drifted apart over hundreds of iterations at the rounding floor.
 * enable() before run(); read() afterwards (out may be NULL with cap_records = 0 to query the sizes). */
/* Test hook: the float system matrix X'X + rho I (p x p, column-major, leading dimension ld >= p) the tall solver's x-update solves,
 * as THIS library formed it (its Gram rounds differently from anybody else's).  Only kept with ADMM_HIP_REFINE=1. */
ADMM_HIP_API int admm_hip_lasso_plan_system_read(admm_hip_plan* plan, float* out, long long ld);
/* Test hook: the standardised data as the WIDE solver holds them -- X (n x p floats, column-major, leading dimension ld >= n) and
 * Y (n floats): DataStd's statistics are accumulated in double here, so the library's X differs from any other
 * standardisation in the last bit, and a stepwise replay of its mat-vecs needs ITS matrix.  Either output may be NULL. */
ADMM_HIP_API int admm_hip_lasso_plan_data_read(admm_hip_plan* plan, float* x_out, long long ld, float* y_out);
ADMM_HIP_API int admm_hip_lasso_plan_state_enable(admm_hip_plan* plan, long long capacity_records);
ADMM_HIP_API int admm_hip_lasso_plan_state_read(admm_hip_plan* plan, float* out, long long cap_records, long long* nrecords_out,
                                                long long* record_floats_out);

/* ---- variant selectors and tuning values (round 6: replaces the ADMM_HIP_* environment variables of earlier rounds).
 * Options belong to the CALLING THREAD and are read by the entry points when they run (a prepared problem reads them when it is
 * created): two threads can run two different variants at the same time, and nothing reads the environment per call.  The typed
 * struct carries the selectors that choose between numerically different (all parity-tested) paths; admm_hip_option_set reaches the
 * long tail of tuning / diagnostic values by name (INTEGRATION.md lists them).  Every field 0 = the library's default.
 * Debugging aid: ADMM_HIP_<NAME> environment variables present when the library is FIRST used form a process-wide overlay under
 * the thread's own settings (read once; changing the environment afterwards has no effect). */
typedef struct admm_hip_options {
    int struct_size;          /* sizeof(admm_hip_options): lets the library accept older, shorter structs */
    int gram_backend;         /* 0 hand-written matrix-core kernels, 1 rocBLAS (dlopen) */
    int gram_split;           /* tall Gram of order >= ~4000: 0 default (fp16 x 2 planes), 1 exact fp32 kernel, 2 fp16 x 2, 3 bf16 x 3 */
    int factor_backend;       /* 0 hand-written blocked Cholesky + inverse, 1 rocSOLVER (dlopen) */
    int inverse_precision;    /* cached inverse: 0 default (built in double below order 4096), 1 always float, 2 always double */
    int tall_xupdate;         /* tall x-update: 0 default (symmetric lower-triangle kernel for p >= 2048), 1 full-matrix mat-vec, 2 symmetric */
    int tall_refine;          /* 1: mixed-precision refinement of the tall x-update */
    int consensus_two_pass;   /* 1: the reference's two products per Woodbury worker (default: one-pass form) */
    int consensus_unfused;    /* 1: `pack` and `z` as two launches; 2: also one launch per worker and product */
    int bp_two_pass;          /* 1: basis pursuit with the reference's two products (default: one-pass form) */
    int lad_no_hat;           /* 1: LAD never forms the n x n hat matrix (the reference does for n <= 2000) */
    int wide_no_persist;      /* 1: wide solver without the persistent active-set stretch; 2: only the column-sharded solver without it */
    int wide_unfused;         /* 1: wide solver with three launches per iteration */
    int wide_gram_sprad;      /* 1: spectral radius from the explicit n x n Gram (default single process: Gram-free) */
    int sharing_bp_direct;    /* 1: column-block basis pursuit without Gram-space stretches */
    int cv_downdate;          /* cross-validation folds as down-dates of the full-data Gram: 0 default (when the Gram is what setup costs), 1 always, 2 never */
    int peer_exchange;        /* PEER back-end: 0 default (producer + consumer in one launch when resident), 1 two launches, 2 through the exchange layer */
    int batch_iters;          /* iterations enqueued between two host polls (0: default 16) */
    int profile_stride;       /* time every k-th x-update launch with HIP events (0: off) */
    int pool_mb;              /* cache of released device blocks: -1 off, 0 default (min(16 GB, memory / 8)), > 0 megabytes */
    int screen;               /* wide solver and admm_hip_parbp: regular steps screened through a 1- or 2-byte copy of the matrix (bit-identical
                                 iterates, a quarter to a half of the bytes): 0 default (when the matrix streams from HBM; the wide solver
                                 picks the 8-bit code where its bounds are tight enough, else fp16), 1 always (fp16), 2 never, 3 always, wide
                                 solver with the 8-bit code */
    int lad_two_pass;         /* 1: LAD with the reference's two products per iteration (default: one pass over the rows of X, p <= 6144) */
    int reserved[10];
} admm_hip_options;
ADMM_HIP_API int admm_hip_options_default(admm_hip_options* o);               /* zero-fills and sets struct_size */
ADMM_HIP_API int admm_hip_options_set(const admm_hip_options* o);             /* NULL: back to the defaults (keeps nothing of the thread's earlier settings) */
ADMM_HIP_API int admm_hip_option_set(const char* name, const char* value);    /* one value by name ("GRAM_SPLIT", "f16x2"); value NULL: back to the default */
ADMM_HIP_API int admm_hip_options_reset(void);
ADMM_HIP_API const char* admm_hip_option_get(const char* name);               /* value in force for this thread, or NULL (library default) */

/* ---- one process per GPU: consensus Lasso with its row blocks spread over ranks (RCCL over xGMI).
 * Bootstrap: rank 0 calls admm_hip_comm_unique_id and ships the ADMM_HIP_UNIQUE_ID_BYTES bytes to the
 * other ranks over any channel (bench.py uses torch.distributed); every rank then calls
 * admm_hip_comm_init on its own device.  In the *_dist entry points x/y are this rank's contiguous
 * ROW SLICE of the global n_total x p problem, in the reference's partition (PADMMLasso.h:163-179:
 * nthread blocks of n_total / nthread rows, the last block takes the remainder; rank r owns blocks
 * [r * nthread / nranks, (r+1) * nthread / nranks)).  Standardisation uses the global column moments
 * (the reference standardises before it splits, ParLasso.cpp:68-72); per ADMM iteration the ranks
 * exchange ONE grouped all-reduce: the consensus sum (p floats) and three squared norms.  Every rank
 * returns the full result.  nthread >= 1 in these two entry points. */
#define ADMM_HIP_UNIQUE_ID_BYTES 128
ADMM_HIP_API int admm_hip_comm_unique_id(void* id_out);
ADMM_HIP_API int admm_hip_comm_init(int nranks, int rank, const void* id);
ADMM_HIP_API int admm_hip_comm_finalize(void);
/* What the attached communicator REALLY holds (RCCL: ncclCommCount / ncclCommUserRank of the live communicator; SHM: ranks attached to
 * the segment; PEER: exchange buffers mapped): *nranks_out = 1, *rank_out = 0, *backend_out = 0 when none is attached;
 * backend 1 RCCL, 2 SHM, 3 PEER.  A launcher (bench.py) checks this against the number of GPUs it was asked to use. */
ADMM_HIP_API int admm_hip_comm_info(int* nranks_out, int* rank_out, int* backend_out);
/* Two more exchange backends behind the same entry points (one of the three is attached at a time; comm.h):
 *  - PEER: one-shot all-reduce over peer-mapped device memory.  Every rank calls admm_hip_comm_peer_prepare on its own
 *    device (allocates its exchange buffer, returns its ADMM_HIP_PEER_HANDLE_BYTES-byte hipIpc handle), the caller
 *    gathers the handles of all ranks in rank order over any channel, then every rank calls admm_hip_comm_init_peer.
 *    The ranks may be processes on different GPUs of one node (xGMI) or -- for tests -- on the same GPU.
 *  - SHM: through a POSIX shared-memory segment `name` ("/something", the same on every rank of one host) carrying the
 *    job's `token` (a non-zero number every rank received over the caller's channel, e.g. drawn by rank 0 and broadcast:
 *    a segment of that name left by a crashed run or owned by a concurrent job is never attached to).  Slow; lets the
 *    multi-rank code run as several processes on ONE GPU without RCCL, bit-identical to PEER.
 * Every wait is bounded: the per-iteration exchanges of a solve by ADMM_HIP_COMM_TIMEOUT_S (20 s), setup reductions and
 * the one join of the replica modes (cross-validation folds, several responses -- ranks are unbalanced there by design) by
 * ADMM_HIP_COMM_PATIENT_TIMEOUT_S (one hour); a missing rank yields ADMM_ERR_COMM from the running call, never a hang.
 * RCCL: the host waits of the solvers poll (stream / event queries + ncclCommGetAsyncError) under the same two bounds; on an
 * asynchronous error or a timed-out wait the communicator is aborted (ncclCommAbort) and the call returns ADMM_ERR_COMM.
 * Ranks should synchronise (barrier) before admm_hip_comm_finalize. */
#define ADMM_HIP_PEER_HANDLE_BYTES 64
ADMM_HIP_API int admm_hip_comm_peer_prepare(int nranks, void* handle_out);
ADMM_HIP_API int admm_hip_comm_init_peer(int nranks, int rank, const void* handles_all_ranks);
ADMM_HIP_API int admm_hip_comm_init_shm(int nranks, int rank, const char* name, unsigned long long token);
/* Test hook: in-place sum all-reduce of a device (mem = ADMM_MEM_DEVICE) or host float / double pair through the
 * attached backend, synchronous.  nf / nd may be 0. */
ADMM_HIP_API int admm_hip_comm_test_allreduce(float* fbuf, long long nf, double* dbuf, long long nd, int mem);
/* Test hook: sum reduce-scatter of host floats through the attached backend -- `send` holds nranks chunks of `count` floats (chunk q
 * is what rank q receives), `recv` (count floats) gets the sum over the ranks of this rank's chunk.  count: a positive multiple of 4.
 * Without a communicator: a copy of chunk 0.  (The exchange of the row-sharded tall solver's split-K Gram) */
ADMM_HIP_API int admm_hip_comm_test_reduce_scatter(const float* send, long long count, float* recv);
ADMM_HIP_API int admm_hip_parlasso_dist(const double* x_local, const double* y_local, int n_local, long long n_total, int p, int mem,
                           const double* lambda_in, int nlambda_in, int nlambda_auto, double lmin_ratio,
                           int standardize, int intercept, int nthread, const admm_opts* opts,
                           double* lambda_out, float* beta_out, int* niter_out, admm_stats* stats);
ADMM_HIP_API int admm_hip_lasso_plan_create_dist(const double* x_local, const double* y_local, int n_local, long long n_total, int p, int mem,
                                    const double* lambda_in, int nlambda_in, int nlambda_auto, double lmin_ratio,
                                    int standardize, int intercept, int nthread, const admm_opts* opts,
                                    admm_hip_plan** plan_out, int* nlambda_out);
/* The SERIAL tall solver (admm_lasso / admm_enet with n_total > p) spread over the ranks -- not in the reference, whose
 * serial solvers are single-threaded; it gives the headline configuration a multi-GPU path.
 * x_local / y_local: any contiguous row slice of the global problem (the slices of all ranks tile the rows; unlike the
 * consensus solver the algorithm does not depend on the split).  Setup: global-moment standardisation, X'y and the
 * Gram matrix as split-K sums over the ranks' row blocks (one all-reduce each), Lanczos value / rho replicated; the
 * Cholesky factorisation and the cached inverse DISTRIBUTED for p >= 4096 (block columns dealt out to the ranks, panels
 * broadcast, every rank forms only the tiles of the inverse its x-update share reads -- bit-identical to the replicated
 * factorisation; ADMM_HIP_DIST_FACTOR=0 switches back to it), replicated below.  Per ADMM iteration every rank streams 1/nranks of the lower-triangle tiles of the inverse and the ranks
 * exchange ONE all-reduce of 2p floats; the element-wise tail and all decisions run replicated on identical numbers.
 * The iterates equal the single-GPU ones up to the summation order of that all-reduce.  alpha < 0: Lasso, else
 * elastic net.  admm_hip_lasso_plan_create_dist with nthread == 0 prepares the same solver for repeated runs. */
ADMM_HIP_API int admm_hip_lasso_dist(const double* x_local, const double* y_local, int n_local, long long n_total, int p, int mem,
                        const double* lambda_in, int nlambda_in, int nlambda_auto, double lmin_ratio,
                        int standardize, int intercept, double alpha, const admm_opts* opts,
                        double* lambda_out, float* beta_out, int* niter_out, admm_stats* stats);

/* The serial WIDE solver (admm_lasso / admm_enet with n <= p_total: ADMMLassoWide / ADMMEnetWide, linearised ADMM with
 * active-set iterations) with its COLUMNS spread over the ranks -- the "column-block consensus ... all-reduce of X_i beta_i"
 * of the problem statement.  The compiled reference has no such solver (its only column-block code is the dead
 * src/TODO/PADMMBP.h:137-156); this is the SAME algorithm as ADMMLassoWide.h:86-186 executed in blocks: rank i holds the
 * columns X_i (x_cols: n x p_local column-major, the global columns [col_offset, col_offset + p_local)) and x_i, and the
 * full y.  Everything of length p is local (X_i't, the prox, the active set, DataStd's column moments); everything of
 * length n (z, the dual, t = Ax + z + y/rho, every decision) is replicated; per ADMM iteration the ranks exchange ONE
 * all-reduce of A x = sum_i X_i x_i (n floats).  Setup: lambda_0 is the max over ranks, X X' = sum_i X_i X_i' one
 * all-reduce, the Lanczos value replicated.  The iterates equal the single-GPU ones up to the summation order of A x.
 * Every rank returns the full (p_total + 1) x nl coefficient matrix.  alpha < 0: Lasso, else elastic net. */
ADMM_HIP_API int admm_hip_lasso_dist_cols(const double* x_cols, const double* y, int n, int p_local, long long p_total, long long col_offset, int mem,
                             const double* lambda_in, int nlambda_in, int nlambda_auto, double lmin_ratio,
                             int standardize, int intercept, double alpha, const admm_opts* opts,
                             double* lambda_out, float* beta_out, int* niter_out, admm_stats* stats);

/* prepared-problem form of the above (run with admm_hip_lasso_plan_run; beta_out there is (p_total + 1) x nl) */
ADMM_HIP_API int admm_hip_lasso_plan_create_dist_cols(const double* x_cols, const double* y, int n, int p_local, long long p_total, long long col_offset, int mem,
                                         const double* lambda_in, int nlambda_in, int nlambda_auto, double lmin_ratio,
                                         int standardize, int intercept, double alpha, const admm_opts* opts,
                                         admm_hip_plan** plan_out, int* nlambda_out);

ADMM_HIP_API const char* admm_hip_last_error(void);
ADMM_HIP_API const char* admm_hip_version(void);
/* The library keeps large device blocks (>= 32 MB) it has released for re-use by its next call -- at most ADMM_HIP_POOL_MB megabytes,
 * default 24576, 0 = off -- because hipMalloc / hipFree of multi-GB buffers cost up to a quarter of a second per call on some hosts.
 * No solver STATE survives a call (Lasso.cpp:74-76,126-129: neither does the reference's); only empty memory does.  This returns it
 * all to the driver (an R session would call it from a finalizer or `gc()` hook; a failed allocation does it by itself). */
ADMM_HIP_API int admm_hip_trim_memory(void);
ADMM_HIP_API int admm_hip_device_count(void);
ADMM_HIP_API int admm_hip_set_device(int device);
/* hipDeviceSynchronize on the current device (bench.py brackets its timed region with it). */
ADMM_HIP_API int admm_hip_device_synchronize(void);


/* ---- test hooks: exported so that the test-suite can put single kernels / host routines under the oracle through
 * the C ABI.  Not used by any solver entry point above. ---- */

/* The host logic of the loose Lanczos call (ADMMLassoTall.h:196-201 -> SymEigsSolver.h) against a dense symmetric
 * float matrix in HOST memory (n x n, column-major).  Runs without a GPU (CPU test-suite). */
ADMM_HIP_API int admm_hip_host_lanczos(const float* A, int n, float* eig_out, int* nmatop_out);

/* The tall x-update mat-vec exactly as the solver runs it for p >= 2048: symv2_lower_kernel on the lower triangle of
 * the symmetric p x p float matrix A (HOST, column-major, leading dimension p) against the two right-hand sides
 * v0, v1 (HOST, length p), followed by the tail kernel's ordered partial reduction.  y0 = A v0, y1 = A v1 (HOST). */
ADMM_HIP_API int admm_hip_test_symv(const float* A, int p, const float* v0, const float* v1, float* y0, float* y1);

/* The one-time matrix-core kernels as the solvers call them (Linalg::cross_prod_lower / tcross_prod_lower,
 * BlasWrapper.h:73-154; LLT, ADMMLassoTall.h:204-205), on HOST matrices (column-major, tight leading dimensions):
 * G = A'A (atA != 0, order cols) or AA' (order rows), both triangles, float (is_double == 0) or double;
 * Ainv = inverse of the SPD matrix A of order n: precision 0 = float, 1 = double, 2 = float matrix inverted in double
 * and rounded once (ADMM_HIP_INVERSE=f64). */
ADMM_HIP_API int admm_hip_test_gram(const void* A, int rows, int cols, int atA, int is_double, void* G);
/* y = A' v with the streaming mat-vec every other product of the solvers uses (X'y, the wide regular step, the consensus /
 * LAD / BP products, the tall x-update below p = 2048): A HOST rows x cols column-major (leading dimension rows). */
ADMM_HIP_API int admm_hip_test_gemv_t(const void* A, int rows, int cols, int is_double, const void* v, void* y);
/* y = A v over the NON-ZERO entries of v with the gather mat-vec of the one-pass consensus workers and of basis pursuit (the product
 * that replaces the reference's first streaming pass, PADMMLasso.h:25 / ADMMBP.h:65): A HOST rows x cols column-major (leading
 * dimension rows), v length cols in A's type, y length rows in double (the kernel's accumulation type). */
ADMM_HIP_API int admm_hip_test_gather(const void* A, int rows, int cols, int is_double, const void* v, double* y);
ADMM_HIP_API int admm_hip_test_spd_inverse(const void* A, int n, int precision, void* Ainv);
/* The system admm_hip_lasso_cv hands the tall solver for fold `fold` when it forms the folds as down-dates of the full-data
 * Gram (cv.hip): x, y HOST column-major doubles, fold_id as in admm_hip_lasso_cv (NULL: i mod nfolds).  Out (HOST): gram p x p
 * float (X_T'X_T of the training rows standardised by THEIR statistics), xy p floats, mean_x / scale_x p floats, and
 * mean_scale_y[2]. */
ADMM_HIP_API int admm_hip_test_cv_fold_system(const double* x, const double* y, int n, int p, const int* fold_id, int nfolds, int fold,
                                              int standardize, int intercept, float* gram, float* xy, float* mean_x, float* scale_x,
                                              float* mean_scale_y);

#ifdef __cplusplus
}
#endif
#endif /* ADMM_HIP_H */
