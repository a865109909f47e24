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
    int branch;            /* 0 tall (Cholesky), 1 wide (linearised), 2 consensus */
    int xupdate_variant;   /* tall path: 0 = full-matrix mat-vec (4p^2 B), 1 = lower-triangle symmetric mat-vec (2p^2 B) */
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

/* Row-block consensus ADMM with `nthread` blocks (ParLasso.cpp:71-72: nthread only sets the
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

/* Prepared-problem variant of the Lasso family (the "persistent context" anticipated for a
 * re-fitting caller; the R shim does not need it).  create = everything the reference does
 * before its lambda loop (copy/convert, DataStd, X'y, Gram, Spectra, factorisation:
 * Lasso.cpp:42-76 + ADMM*::init); run = the warm-started lambda loop of Lasso.cpp:97-124 from a
 * cold start, repeatable; destroy frees all device memory.  alpha < 0 selects the Lasso prox,
 * 0 <= alpha <= 1 the elastic net; nthread > 1 selects the consensus solver.
 * bench.py times run() only (ADMM iterations/s excludes the one-time setup). */
typedef struct admm_hip_plan admm_hip_plan;
ADMM_HIP_API int admm_hip_lasso_plan_create(const double* x, const double* y, int n, int p, int mem,
                               const double* lambda_in, int nlambda_in, int nlambda_auto, double lmin_ratio,
                               int standardize, int intercept, double alpha, int nthread, const admm_opts* opts,
                               admm_hip_plan** plan_out, int* nlambda_out);
ADMM_HIP_API int admm_hip_lasso_plan_run(admm_hip_plan* plan, double* lambda_out, float* beta_out, int* niter_out, admm_stats* stats);
ADMM_HIP_API int admm_hip_lasso_plan_destroy(admm_hip_plan* plan);

/* ---- one process per GPU: consensus Lasso with its row blocks spread over ranks (RCCL over xGMI).
 * Bootstrap: rank 0 calls admm_hip_comm_unique_id and ships the ADMM_HIP_UNIQUE_ID_BYTES bytes to the
 * other ranks over any channel (bench.py uses torch.distributed); every rank then calls
 * admm_hip_comm_init on its own device.  In the *_dist entry points x/y are this rank's contiguous
 * ROW SLICE of the global n_total x p problem, in the reference's partition (PADMMLasso.h:163-179:
 * nthread blocks of n_total / nthread rows, the last block takes the remainder; rank r owns blocks
 * [r * nthread / nranks, (r+1) * nthread / nranks)).  Standardisation uses the global column moments
 * (the reference standardises before it splits, ParLasso.cpp:68-72); per ADMM iteration the ranks
 * exchange ONE grouped all-reduce: the consensus sum (p floats) and three squared norms.  Every rank
 * returns the full result. */
#define ADMM_HIP_UNIQUE_ID_BYTES 128
ADMM_HIP_API int admm_hip_comm_unique_id(void* id_out);
ADMM_HIP_API int admm_hip_comm_init(int nranks, int rank, const void* id);
ADMM_HIP_API int admm_hip_comm_finalize(void);
ADMM_HIP_API int admm_hip_parlasso_dist(const double* x_local, const double* y_local, int n_local, long long n_total, int p, int mem,
                           const double* lambda_in, int nlambda_in, int nlambda_auto, double lmin_ratio,
                           int standardize, int intercept, int nthread, const admm_opts* opts,
                           double* lambda_out, float* beta_out, int* niter_out, admm_stats* stats);
ADMM_HIP_API int admm_hip_lasso_plan_create_dist(const double* x_local, const double* y_local, int n_local, long long n_total, int p, int mem,
                                    const double* lambda_in, int nlambda_in, int nlambda_auto, double lmin_ratio,
                                    int standardize, int intercept, int nthread, const admm_opts* opts,
                                    admm_hip_plan** plan_out, int* nlambda_out);

ADMM_HIP_API const char* admm_hip_last_error(void);
ADMM_HIP_API const char* admm_hip_version(void);
ADMM_HIP_API int admm_hip_device_count(void);
ADMM_HIP_API int admm_hip_set_device(int device);
/* hipDeviceSynchronize on the current device (bench.py brackets its timed region with it). */
ADMM_HIP_API int admm_hip_device_synchronize(void);

#ifdef __cplusplus
}
#endif
#endif /* ADMM_HIP_H */
