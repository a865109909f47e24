/*
 * admm_loops_cpu.c -- CPU restatements in C of the reference's OTHER ADMM loops: the wide Lasso solver, the row-block consensus
 * solver, LAD and basis pursuit (the tall loop is admm_tall_cpu.c).
 * TEST INFRASTRUCTURE / CPU BASELINE, NOT PRODUCT: only tests/ and the cpu_baseline legs of bench.py load the library built
 * from this file (oracle/c/Makefile -> oracle/c/liboracle_loops.so, through oracle/cloops.py).  Nothing under admm_amd/ links it.
 *
 * Follows (reference = /root/reference, yixuan/ADMM 1.0):
 *   ADMMBase::solve / update_rho          src/ADMMBase.h:85-109,158-216
 *   ADMMLassoWide                         src/ADMMLassoWide.h:70-186 (soft threshold, active-set / regular x-update, z, residuals, eps)
 *   PADMMBase_Master::solve, workers      src/PADMMBase.h:57-78,117-145,174-237;  src/PADMMLasso.h:17-31,99-108,149-152
 *   FADMMBase::solve / update_rho         src/FADMMBase.h:100-133,185-265
 *   ADMMLAD                               src/ADMMLAD.h:62-107,152-169      (general branch X (X'X)^-1 X')
 *   ADMMBP                                src/ADMMBP.h:48-93,138-153
 *   column-block ("sharing") basis pursuit src/TODO/PADMMBP.h:19-61,137-140 (worker x-update, z-bar) on the loop shape of
 *                                         src/PADMMBase.h:118-142,174-237 -- as restated in oracle/solvers.py SharingBP, line by line
 * One-time quantities (standardised data, spectral radius, Cholesky factors, L^-1 A, ...) are INPUTS: the Python side builds them
 * with NumPy / LAPACK exactly as the NumPy oracle does (oracle/cloops.py); this file is the per-iteration hot loop in the two CPU
 * configurations bench.py reports -- nthreads = 1 (the reference's effective configuration: Eigen products are serial under
 * EIGEN_DONT_PARALLELIZE, Lasso.cpp:1; its only OpenMP loops are the active-set columns, ADMMLassoWide.h:100, and the consensus
 * workers, PADMMBase.h:180,204) and nthreads = all cores, every product spread over the threads ("best effort").
 * Compiled with -ffp-contract=off: float products round like the reference's scalar code.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static float norm_f(const float* v, long n) {            /* VectorXf::norm(): float accumulation */
    float s = 0.f;
    for (long i = 0; i < n; ++i) s += v[i] * v[i];
    return sqrtf(s);
}
static float sqnorm_f(const float* v, long n) {
    float s = 0.f;
    for (long i = 0; i < n; ++i) s += v[i] * v[i];
    return s;
}
static double norm_d(const double* v, long n) {
    double s = 0.0;
    for (long i = 0; i < n; ++i) s += v[i] * v[i];
    return sqrt(s);
}

/* dot products with 16 (float) / 8 (double) independent partial sums: the compiler keeps them in two AVX2 registers without
 * re-associating anything (no -ffast-math); the partial sums are added pairwise at the end */
static float dot_f(const float* a, const float* b, int n) {
    float acc[16];
    for (int u = 0; u < 16; ++u) acc[u] = 0.f;
    int i = 0;
    for (; i + 15 < n; i += 16)
        for (int u = 0; u < 16; ++u) acc[u] += a[i + u] * b[i + u];
    for (; i < n; ++i) acc[i & 15] += a[i] * b[i];
    for (int w = 8; w >= 1; w >>= 1)
        for (int u = 0; u < w; ++u) acc[u] += acc[u + w];
    return acc[0];
}
static double dot_d(const double* a, const double* b, int n) {
    double acc[8];
    for (int u = 0; u < 8; ++u) acc[u] = 0.0;
    int i = 0;
    for (; i + 7 < n; i += 8)
        for (int u = 0; u < 8; ++u) acc[u] += a[i + u] * b[i + u];
    for (; i < n; ++i) acc[i & 7] += a[i] * b[i];
    for (int w = 4; w >= 1; w >>= 1)
        for (int u = 0; u < w; ++u) acc[u] += acc[u + w];
    return acc[0];
}

/* update_rho(): ADMMBase.h:85-109 == FADMMBase.h:109-133 */
static double rho_rule(double rho, double rp, double ep, double rd, double ed) {
    if (rp / ep > 10 * rd / ed) rho *= 2;
    else if (rd / ed > 10 * rp / ep) rho /= 2;
    if (rp < ep) rho /= 1.2;
    if (rd < ed) rho *= 1.2;
    return rho;
}

static int is_regular_update(int x) {                   /* 4^k - 1, ADMMLassoWide.h:121-127 */
    if (x == 0 || x == 3 || x == 15 || x == 63) return 1;
    x += 1;
    if (x & (x - 1)) return 0;
    return (x & 0x55555555) != 0;
}

/* out (rows) = sum_j coef[j] * A[:, cols[j]]  (column-major A, ld lda).  One thread: float accumulation in column order.
 * Several threads ("best effort"): every thread streams a contiguous RANGE of the columns -- the same static partition as the column
 * dot products and as the first-touch copy below, so that a thread reads pages of its own NUMA node, sequentially -- into a private
 * accumulator; the accumulators are summed in thread order. */
static void axpy_cols_f(const float* A, long lda, int rows, const int* cols, const float* coef, int ncols, float* out, int nthreads) {
    if (nthreads <= 1 || ncols < 4 * nthreads) {
        for (int i = 0; i < rows; ++i) out[i] = 0.f;
        for (int j = 0; j < ncols; ++j) {
            const float* c = A + (size_t)(cols ? cols[j] : j) * lda;
            const float xj = coef[j];
            for (int i = 0; i < rows; ++i) out[i] += xj * c[i];
        }
        return;
    }
    float* part = malloc((size_t)nthreads * rows * sizeof(float));
#pragma omp parallel num_threads(nthreads)
    {
        const int t = omp_get_thread_num(), nt = omp_get_num_threads();
        const int chunk = (ncols + nt - 1) / nt, j0 = t * chunk, j1 = j0 + chunk < ncols ? j0 + chunk : ncols;
        float* o = part + (size_t)t * rows;
        for (int i = 0; i < rows; ++i) o[i] = 0.f;
        for (int j = j0; j < j1; ++j) {
            const float* c = A + (size_t)(cols ? cols[j] : j) * lda;
            const float xj = coef[j];
            for (int i = 0; i < rows; ++i) o[i] += xj * c[i];
        }
#pragma omp barrier
#pragma omp for schedule(static)
        for (int i = 0; i < rows; ++i) {
            float v = 0.f;
            for (int q = 0; q < nt; ++q) v += part[(size_t)q * rows + i];
            out[i] = v;
        }
    }
    free(part);
}
static void axpy_cols_d(const double* A, long lda, int rows, const double* coef, int ncols, double* out, int nthreads) {
    if (nthreads <= 1 || ncols < 4 * nthreads) {
        for (int i = 0; i < rows; ++i) out[i] = 0.0;
        for (int j = 0; j < ncols; ++j) {
            const double* c = A + (size_t)j * lda;
            const double xj = coef[j];
            for (int i = 0; i < rows; ++i) out[i] += xj * c[i];
        }
        return;
    }
    double* part = malloc((size_t)nthreads * rows * sizeof(double));
#pragma omp parallel num_threads(nthreads)
    {
        const int t = omp_get_thread_num(), nt = omp_get_num_threads();
        const int chunk = (ncols + nt - 1) / nt, j0 = t * chunk, j1 = j0 + chunk < ncols ? j0 + chunk : ncols;
        double* o = part + (size_t)t * rows;
        for (int i = 0; i < rows; ++i) o[i] = 0.0;
        for (int j = j0; j < j1; ++j) {
            const double* c = A + (size_t)j * lda;
            const double xj = coef[j];
            for (int i = 0; i < rows; ++i) o[i] += xj * c[i];
        }
#pragma omp barrier
#pragma omp for schedule(static)
        for (int i = 0; i < rows; ++i) {
            double v = 0.0;
            for (int q = 0; q < nt; ++q) v += part[(size_t)q * rows + i];
            out[i] = v;
        }
    }
    free(part);
}
/* all-core legs: a copy of the matrix whose pages each thread touched first, with the column partition of the loops above and of
 * `#pragma omp parallel for schedule(static)` over the columns (the caller's array was written by one thread, i.e. sits on one NUMA
 * node).  Not timed (setup).  Returns NULL when there is no memory for it (the caller's array is used then). */
static void* own_copy(const void* A, size_t col_bytes, int ncols, int nthreads) {
    if (nthreads <= 1) return NULL;
    char* B = malloc(col_bytes * (size_t)ncols);
    if (!B) return NULL;
#pragma omp parallel num_threads(nthreads)
    {
        const int t = omp_get_thread_num(), nt = omp_get_num_threads();
        const int chunk = (ncols + nt - 1) / nt, j0 = t * chunk, j1 = j0 + chunk < ncols ? j0 + chunk : ncols;
        if (j1 > j0) memcpy(B + (size_t)j0 * col_bytes, (const char*)A + (size_t)j0 * col_bytes, (size_t)(j1 - j0) * col_bytes);
    }
    return B;
}
/* the column dot products with the same partition (schedule(static) hands out ceil-sized chunks in thread order as well) */

/* ------------------------------------------------------------------------------------------------------------------- wide Lasso
 * One warm-started lambda path of ADMMLassoWide (Lasso.cpp:97-124 with n <= p) from a cold start.
 *   X n x p float column-major (ld ldx), standardised; Y; sprad = the loose Spectra value; lambda0 = max|X'y|
 *   lam[nlam] internal lambdas; rho0 <= 0: (lambda / sprad)^(1/3) at the first lambda (ADMMLassoWide.h:227-228)
 *   beta_out: nlam x p (get_x() after each solve), or NULL;  nnz_sum: sum over iterations of the non-zeros AFTER the x-update
 *   trace: NULL or trace_cap x 12 doubles in the layout of include/admm_hip.h (wide flavour) */
int oracle_wide_path(const float* X, long ldx, int n, int p, const float* Y, float sprad, float lambda0, const double* lam, int nlam,
                     double rho0, double eps_abs, double eps_rel, int maxit, int nthreads, float* beta_out, int* niter_out,
                     double* loop_seconds, long long* nnz_sum, double* trace, int trace_cap, int* ntrace, double budget_s) {
    /* budget_s > 0 (bench.py's bounded CPU sample): the path is cut off once that many seconds have passed; niter_out then holds
     * the iterations actually made (the cut-off lambda's count is partial, later lambdas 0) */
    float* x = calloc((size_t)p, sizeof(float));
    int* idx = malloc((size_t)p * sizeof(int));
    float* xv = malloc((size_t)p * sizeof(float));
    float* Ax = calloc((size_t)n, sizeof(float));
    float* z = calloc((size_t)n, sizeof(float));
    float* y = calloc((size_t)n, sizeof(float));
    float* t = calloc((size_t)n, sizeof(float));
    float* nz = calloc((size_t)n, sizeof(float));
    float* r = calloc((size_t)n, sizeof(float));
    if (!x || !idx || !xv || !Ax || !z || !y || !t || !nz || !r) return 1;
    if (nthreads < 1) nthreads = 1;
    float* Xown = own_copy(X, (size_t)ldx * sizeof(float), p, nthreads);
    if (Xown) X = Xown;
    const float sq_gamma = sqrtf(sprad);
    const double gamma_d = (double)sprad;
    double rho = rho0;
    int nact = 0, ntr = 0, cut = 0;
    long long nnz_total = 0;
    const double t0 = now_s();
    for (int l = 0; l < nlam; ++l) {
        const float lam_f = (float)lam[l];
        if (l == 0 && rho <= 0) rho = pow((double)lam_f / gamma_d, 1.0 / 3.0);
        int counter = 0;                                                    /* init / init_warm: :215-251 */
        int it = maxit + 1;
        for (int i = 0; i < maxit; ++i) {
            const float rho_f = (float)rho;
            const double nAx = norm_f(Ax, n), nzz = norm_f(z, n);
            const double eps_primal = (nAx > nzz ? nAx : nzz) * eps_rel + sqrt((double)n) * eps_abs;      /* :174-178 */
            const double eps_dual = (double)sq_gamma * (double)norm_f(y, n) * eps_rel + sqrt((double)p) * eps_abs;  /* :179-182 */
            int kind = 0;
            if ((double)lam_f > (double)lambda0 - 1e-5) {                   /* :131-135 */
                for (int k = 0; k < nact; ++k) x[idx[k]] = 0.f;
                nact = 0;
            } else if (is_regular_update(counter)) {                        /* :138-150 */
                kind = 1;
                for (int k = 0; k < n; ++k) t[k] = (Ax[k] + z[k]) + y[k] / rho_f;
                const double pen = (double)lam_f / (rho * gamma_d);
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
                for (int j = 0; j < p; ++j) {
                    float v = -dot_f(X + (size_t)j * ldx, t, n) / sprad;
                    v = v + x[j];
                    const double vd = (double)v;
                    x[j] = vd > pen ? (float)(vd - pen) : (vd < -pen ? (float)(vd + pen) : 0.f);
                }
                nact = 0;
                for (int j = 0; j < p; ++j) if (x[j] != 0.f) idx[nact++] = j;
                ++counter;
            } else {                                                        /* active_set_update :86-118 (the reference's OpenMP loop :100) */
                kind = 2;
                const float pen = (float)((double)lam_f / (rho * gamma_d));
                for (int k = 0; k < n; ++k) t[k] = ((Ax[k] + z[k]) + y[k] / rho_f) / sprad;
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
                for (int k = 0; k < nact; ++k) {
                    const int j = idx[k];
                    const float val = x[j] - dot_f(X + (size_t)j * ldx, t, n);
                    x[j] = val > pen ? val - pen : (val < -pen ? val + pen : 0.f);
                }
                int m = 0;
                for (int k = 0; k < nact; ++k) if (x[idx[k]] != 0.f) idx[m++] = idx[k];      /* prune() */
                nact = m;
                ++counter;
            }
            nnz_total += nact;
            /* next_z :156-165 */
            for (int k = 0; k < nact; ++k) xv[k] = x[idx[k]];
            axpy_cols_f(X, ldx, n, idx, xv, nact, Ax, nthreads);
            const float den = (float)(-1.0 - rho);
            for (int k = 0; k < n; ++k) nz[k] = ((Y[k] + y[k]) + rho_f * Ax[k]) / den;
            float d2 = 0.f;
            for (int k = 0; k < n; ++k) { const float d = nz[k] - z[k]; d2 += d * d; }
            const double resid_dual = rho * (double)sq_gamma * (double)sqrtf(d2);           /* :183-186, before the swap (ADMMBase.h:167-175) */
            memcpy(z, nz, (size_t)n * sizeof(float));
            for (int k = 0; k < n; ++k) r[k] = Ax[k] + z[k];                /* :166-170 */
            const double resid_primal = (double)norm_f(r, n);
            for (int k = 0; k < n; ++k) y[k] = y[k] + rho_f * r[k];
            const int conv = resid_primal < eps_primal && resid_dual < eps_dual;
            const double rho_in = rho;
            if (!conv && i > 3) rho = rho_rule(rho, resid_primal, eps_primal, resid_dual, eps_dual);
            if (trace && ntr < trace_cap) {
                double* tr = trace + 12 * (size_t)ntr++;
                tr[0] = l; tr[1] = i; tr[2] = eps_primal; tr[3] = eps_dual; tr[4] = resid_primal; tr[5] = resid_dual; tr[6] = rho; tr[7] = kind;
                tr[8] = conv ? 0 : 1; tr[9] = rho_in; tr[10] = rho; tr[11] = 0.0;
            }
            if (conv) { it = i + 1; break; }
            if (budget_s > 0 && now_s() - t0 > budget_s) { it = i + 1; cut = 1; break; }
        }
        niter_out[l] = it;
        if (beta_out) memcpy(beta_out + (size_t)l * p, x, (size_t)p * sizeof(float));       /* get_x(): Lasso.cpp:119 */
        if (cut) { for (int q = l + 1; q < nlam; ++q) niter_out[q] = 0; break; }
    }
    *loop_seconds = now_s() - t0;
    if (nnz_sum) *nnz_sum = nnz_total;
    if (ntrace) *ntrace = ntr;
    free(x); free(idx); free(xv); free(Ax); free(z); free(y); free(t); free(nz); free(r); free(Xown);
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------------- consensus
 * x <- L^-1 x then x <- L^-T x, L lower triangular column-major (ld m): Eigen's LLT::solve on one right-hand side */
static void llt_solve_f(const float* L, int m, float* x) {
    for (int j = 0; j < m; ++j) {
        const float* c = L + (size_t)j * m;
        const float xj = x[j] / c[j];
        x[j] = xj;
        for (int i = j + 1; i < m; ++i) x[i] -= xj * c[i];
    }
    for (int j = m - 1; j >= 0; --j) {
        const float* c = L + (size_t)j * m;
        float s = 0.f;
        for (int i = j + 1; i < m; ++i) s += c[i] * x[i];
        x[j] = (x[j] - s) / c[j];
    }
}

/* the same in double on a double factor: the rounding VARIANT "exact" of oracle/solvers.py PADMMLasso (the float system the reference
 * builds, solved without the float LLT's rounding; the right-hand side and the result stay the float vectors they are) */
static void llt_solve_fd(const double* L, int m, float* xf, double* x) {
    for (int i = 0; i < m; ++i) x[i] = (double)xf[i];
    for (int j = 0; j < m; ++j) {
        const double* c = L + (size_t)j * m;
        const double xj = x[j] / c[j];
        x[j] = xj;
        for (int i = j + 1; i < m; ++i) x[i] -= xj * c[i];
    }
    for (int j = m - 1; j >= 0; --j) {
        const double* c = L + (size_t)j * m;
        double s = 0.0;
        for (int i = j + 1; i < m; ++i) s += c[i] * x[i];
        x[j] = (x[j] - s) / c[j];
    }
    for (int i = 0; i < m; ++i) xf[i] = (float)x[i];
}

/* PADMMBase_Master::solve with PADMMLasso workers, one warm-started lambda path.
 *   A[k]: rows[k] x p float column-major (ld rows[k]);  Ab[k] = A_k'b_k (p);  L[k]: Cholesky factor of A_k'A_k + rho I (p x p, tall block)
 *   or of A_k A_k' + rho I (rows x rows, wide block: Woodbury, PADMMLasso.h:23-30), lower, column-major, dense.
 *   nthreads: OpenMP threads over the workers (the reference's loop, PADMMBase.h:180) -- and, when nthreads > K, inside the products. */
static int consensus_path_impl(const float* const* A, const int* rows, const float* const* Ab, const float* const* L, const double* const* Ld, int K, int p,
                          const double* lam, int nlam, double rho, double eps_abs, double eps_rel, int maxit, int nthreads,
                          float* beta_out, int* niter_out, double* loop_seconds, double* trace, int trace_cap, int* ntrace, double budget_s);
int oracle_consensus_path(const float* const* A, const int* rows, const float* const* Ab, const float* const* L, int K, int p,
                          const double* lam, int nlam, double rho, double eps_abs, double eps_rel, int maxit, int nthreads,
                          float* beta_out, int* niter_out, double* loop_seconds, double* trace, int trace_cap, int* ntrace, double budget_s) {
    return consensus_path_impl(A, rows, Ab, L, NULL, K, p, lam, nlam, rho, eps_abs, eps_rel, maxit, nthreads, beta_out, niter_out, loop_seconds, trace, trace_cap, ntrace, budget_s);
}
/* Ld[k]: the Cholesky factor of the SAME float system in double -- the workers' small solves then run in double (rounding variant) */
int oracle_consensus_path_exact(const float* const* A, const int* rows, const float* const* Ab, const double* const* Ld, int K, int p,
                          const double* lam, int nlam, double rho, double eps_abs, double eps_rel, int maxit, int nthreads,
                          float* beta_out, int* niter_out, double* loop_seconds, double* trace, int trace_cap, int* ntrace, double budget_s) {
    return consensus_path_impl(A, rows, Ab, NULL, Ld, K, p, lam, nlam, rho, eps_abs, eps_rel, maxit, nthreads, beta_out, niter_out, loop_seconds, trace, trace_cap, ntrace, budget_s);
}
static int consensus_path_impl(const float* const* A, const int* rows, const float* const* Ab, const float* const* L, const double* const* Ld, int K, int p,
                          const double* lam, int nlam, double rho, double eps_abs, double eps_rel, int maxit, int nthreads,
                          float* beta_out, int* niter_out, double* loop_seconds, double* trace, int trace_cap, int* ntrace, double budget_s) {
    if (nthreads < 1) nthreads = 1;
    double** dwork = malloc((size_t)K * sizeof(double*));
    for (int k = 0; k < K; ++k) dwork[k] = Ld ? malloc((size_t)(rows[k] > p ? rows[k] : p) * sizeof(double)) : NULL;
    int cut = 0;
    float** x = malloc((size_t)K * sizeof(float*));
    float** y = malloc((size_t)K * sizeof(float*));
    float** rhs = malloc((size_t)K * sizeof(float*));
    float** g = malloc((size_t)K * sizeof(float*));
    float** tv = malloc((size_t)K * sizeof(float*));
    double* sqr = calloc((size_t)K, sizeof(double));
    float* z = calloc((size_t)p, sizeof(float));
    float* nz = calloc((size_t)p, sizeof(float));
    for (int k = 0; k < K; ++k) {
        x[k] = calloc((size_t)p, sizeof(float)); y[k] = calloc((size_t)p, sizeof(float));
        rhs[k] = calloc((size_t)p, sizeof(float)); g[k] = calloc((size_t)p, sizeof(float)); tv[k] = calloc((size_t)(rows[k] > p ? rows[k] : p), sizeof(float));
    }
    const float rho_f = (float)rho;
    const int outer = nthreads < K ? nthreads : K;                          /* threads over the workers */
    const int inner = nthreads > K ? nthreads / K : 1;                      /* threads inside a worker's products (best effort only) */
    /* all-core leg: every worker streams a copy of its block that its own thread touched first (NUMA placement; not timed) */
    const float** Ause = malloc((size_t)K * sizeof(float*));
    float** Aown = calloc((size_t)K, sizeof(float*));
    for (int k = 0; k < K; ++k) Ause[k] = A[k];
    if (nthreads > 1) {
#pragma omp parallel for schedule(static) num_threads(outer)
        for (int k = 0; k < K; ++k) {
            Aown[k] = malloc((size_t)rows[k] * (size_t)p * sizeof(float));
            if (Aown[k]) { memcpy(Aown[k], A[k], (size_t)rows[k] * (size_t)p * sizeof(float)); Ause[k] = Aown[k]; }
        }
    }
#ifdef _OPENMP
    if (inner > 1) omp_set_max_active_levels(2);
#endif
    int ntr = 0;
    const double spK = sqrt((double)p * (double)K), sK = sqrt((double)K);
    const double t0 = now_s();
    for (int l = 0; l < nlam; ++l) {
        const double lambda = lam[l];
        int it = maxit + 1;
        for (int i = 0; i < maxit; ++i) {
            double xn = 0.0, yn = 0.0;                                      /* PADMMBase.h:117-139 */
            for (int k = 0; k < K; ++k) { xn += (double)sqnorm_f(x[k], p); yn += (double)sqnorm_f(y[k], p); }
            const double nzz = (double)norm_f(z, p) * sK;
            const double eps_primal = (sqrt(xn) > nzz ? sqrt(xn) : nzz) * eps_rel + spK * eps_abs;
            const double eps_dual = sqrt(yn) * eps_rel + spK * eps_abs;
#pragma omp parallel for schedule(static) num_threads(outer) if (outer > 1)
            for (int k = 0; k < K; ++k) {                                   /* worker next_x: PADMMLasso.h:17-31 */
                const float* Ak = Ause[k];
                const int m = rows[k];
                for (int j = 0; j < p; ++j) {
                    const float r0 = Ab[k][j] - y[k][j];
                    rhs[k][j] = z[j] != 0.f ? (float)((double)r0 + rho * (double)z[j]) : r0;
                }
                if (m >= p) {
                    memcpy(x[k], rhs[k], (size_t)p * sizeof(float));
                    if (Ld) llt_solve_fd(Ld[k], p, x[k], dwork[k]); else llt_solve_f(L[k], p, x[k]);
                } else {
                    axpy_cols_f(Ak, m, m, NULL, rhs[k], p, tv[k], inner);               /* t = A rhs */
                    if (Ld) llt_solve_fd(Ld[k], m, tv[k], dwork[k]); else llt_solve_f(L[k], m, tv[k]);   /* s = (AA' + rho I)^-1 t */
#pragma omp parallel for schedule(static) num_threads(inner) if (inner > 1)
                    for (int j = 0; j < p; ++j) g[k][j] = dot_f(Ak + (size_t)j * m, tv[k], m);   /* A's */
                    for (int j = 0; j < p; ++j) x[k][j] = (rhs[k][j] - g[k][j]) / rho_f;
                }
            }
            /* master next_z: PADMMLasso.h:99-108 */
            const double pen = lambda / (rho * (double)K);
            float dz2 = 0.f;
            for (int j = 0; j < p; ++j) {
                float v = 0.f;
                for (int k = 0; k < K; ++k) v = v + (x[k][j] + y[k][j] / rho_f);
                v = v / (float)K;
                const double vd = (double)v;
                nz[j] = vd > pen ? (float)(vd - pen) : (vd < -pen ? (float)(vd + pen) : 0.f);
                const float d = nz[j] - z[j];
                dz2 += d * d;
            }
            const double resid_dual = rho * sqrt((double)K * (double)dz2);                  /* :149-152 */
            memcpy(z, nz, (size_t)p * sizeof(float));
#pragma omp parallel for schedule(static) num_threads(outer) if (outer > 1)
            for (int k = 0; k < K; ++k) {                                   /* PADMMBase.h:70-78,200-214 */
                float s = 0.f;
                for (int j = 0; j < p; ++j) {
                    const float r = x[k][j] - z[j];
                    s += r * r;
                    y[k][j] = y[k][j] + rho_f * r;
                }
                sqr[k] = (double)s;
            }
            double coll = 0.0;
            for (int k = 0; k < K; ++k) coll += sqr[k];
            const double resid_primal = sqrt(coll);
            const int conv = resid_primal < eps_primal && resid_dual < eps_dual;
            if (trace && ntr < trace_cap) {
                double* tr = trace + 12 * (size_t)ntr++;
                tr[0] = l; tr[1] = i; tr[2] = eps_primal; tr[3] = eps_dual; tr[4] = resid_primal; tr[5] = resid_dual; tr[6] = rho; tr[7] = 0;
                tr[8] = conv ? 0 : 1; tr[9] = rho; tr[10] = rho; tr[11] = 0.0;
            }
            if (conv) { it = i + 1; break; }
            if (budget_s > 0 && now_s() - t0 > budget_s) { it = i + 1; cut = 1; break; }
        }
        niter_out[l] = it;
        if (beta_out) memcpy(beta_out + (size_t)l * p, z, (size_t)p * sizeof(float));       /* get_z(): ParLasso.cpp:98 */
        if (cut) { for (int q = l + 1; q < nlam; ++q) niter_out[q] = 0; break; }
    }
    *loop_seconds = now_s() - t0;
    if (ntrace) *ntrace = ntr;
    for (int k = 0; k < K; ++k) { free(x[k]); free(y[k]); free(rhs[k]); free(g[k]); free(tv[k]); free(Aown[k]); free(dwork[k]); }
    free(Ause); free(Aown); free(dwork);
    free(x); free(y); free(rhs); free(g); free(tv); free(sqr); free(z); free(nz);
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------------- LAD / BP
 * FADMMBase::solve in double for the two dense problems.
 *   prob 0 (LAD, dim = n):  M = X (n x p column-major, ld n, standardised), L = Cholesky factor of X'X (p x p lower), dvec = y;
 *                           x = X (X'X)^-1 X' (y - adj_y/rho + adj_z)   (ADMMLAD.h:62-78), extra_norm = ||y|| (:152-157)
 *   prob 1 (BP, dim = p):   M = B = L^-1 A (n x p column-major, ld n), dvec = A'(AA')^-1 b;
 *                           x = vec + dvec - B'(B vec)                  (ADMMBP.h:48-67)
 * out: the final adj_z, adj_y, z and rho (LAD's get_x() needs adj and rho, ADMMLAD.h:220-225; BP returns z). */
static void llt_solve_d(const double* L, int m, double* x) {
    for (int j = 0; j < m; ++j) {
        const double* c = L + (size_t)j * m;
        const double xj = x[j] / c[j];
        x[j] = xj;
        for (int i = j + 1; i < m; ++i) x[i] -= xj * c[i];
    }
    for (int j = m - 1; j >= 0; --j) {
        const double* c = L + (size_t)j * m;
        double s = 0.0;
        for (int i = j + 1; i < m; ++i) s += c[i] * x[i];
        x[j] = (x[j] - s) / c[j];
    }
}

int oracle_dense_loop(int prob, const double* M, int n, int p, const double* L, const double* dvec, double rho, double eps_abs, double eps_rel,
                      int maxit, int nthreads, double* z_out, double* adjz_out, double* adjy_out, double* rho_out, int* niter_out,
                      double* loop_seconds, double* trace, int trace_cap, int* ntrace, double budget_s) {
    if (nthreads < 1) nthreads = 1;
    const int dim = prob == 0 ? n : p;
    double* Mown = own_copy(M, (size_t)n * sizeof(double), p, nthreads);
    if (Mown) M = Mown;
    double* x = calloc((size_t)dim, sizeof(double));
    double* z = calloc((size_t)dim, sizeof(double));
    double* y = calloc((size_t)dim, sizeof(double));
    double* az = calloc((size_t)dim, sizeof(double));
    double* ay = calloc((size_t)dim, sizeof(double));
    double* oz = calloc((size_t)dim, sizeof(double));
    double* oy = calloc((size_t)dim, sizeof(double));
    double* vec = calloc((size_t)dim, sizeof(double));
    double* small = calloc((size_t)(n > p ? n : p), sizeof(double));
    double* g = calloc((size_t)dim, sizeof(double));
    if (!x || !z || !y || !az || !ay || !oz || !oy || !vec || !small || !g) return 1;
    const double extra = prob == 0 ? norm_d(dvec, n) : 0.0;
    const double sqd = sqrt((double)dim);
    double a = 1.0, c = 9999.0;
    int it = maxit + 1, ntr = 0;
    const double t0 = now_s();
    for (int i = 0; i < maxit; ++i) {
        memcpy(oz, z, (size_t)dim * sizeof(double));
        memcpy(oy, y, (size_t)dim * sizeof(double));
        const double nx = norm_d(x, dim), nzz = norm_d(z, dim);
        double mx = nx > nzz ? nx : nzz;
        if (extra > mx) mx = extra;
        const double eps_primal = mx * eps_rel + sqd * eps_abs;
        const double eps_dual = norm_d(y, dim) * eps_rel + sqd * eps_abs;
        /* next_x */
        if (prob == 0) {
            for (int k = 0; k < n; ++k) vec[k] = dvec[k] - ay[k] / rho + az[k];
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
            for (int j = 0; j < p; ++j) small[j] = dot_d(M + (size_t)j * n, vec, n);          /* X' vec */
            llt_solve_d(L, p, small);                                                        /* (X'X)^-1 */
            axpy_cols_d(M, n, n, small, p, x, nthreads);                                     /* X s */
        } else {
            for (int k = 0; k < p; ++k) vec[k] = -ay[k] / rho + az[k];
            axpy_cols_d(M, n, n, vec, p, small, nthreads);                                   /* w = B vec */
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
            for (int j = 0; j < p; ++j) g[j] = dot_d(M + (size_t)j * n, small, n);            /* B' w */
            for (int k = 0; k < p; ++k) x[k] = (vec[k] + dvec[k]) - g[k];
        }
        /* next_z, resid_dual, residual, y */
        const double pen = 1.0 / rho;
        double dz2 = 0.0, r2 = 0.0;
        for (int k = 0; k < dim; ++k) {
            const double v = prob == 0 ? x[k] - dvec[k] + ay[k] / rho : x[k] + ay[k] / rho;
            const double zn = v > pen ? v - pen : (v < -pen ? v + pen : 0.0);
            const double d = zn - oz[k];
            dz2 += d * d;
            z[k] = zn;
            const double r = prob == 0 ? x[k] - dvec[k] - zn : x[k] - zn;
            r2 += r * r;
            y[k] = ay[k] + rho * r;
        }
        const double resid_dual = rho * sqrt(dz2), resid_primal = sqrt(r2);
        const int conv = resid_primal < eps_primal && resid_dual < eps_dual;
        double* tr = (trace && ntr < trace_cap) ? trace + 12 * (size_t)ntr++ : NULL;
        if (tr) { tr[0] = 0; tr[1] = i; tr[2] = eps_primal; tr[3] = eps_dual; tr[4] = resid_primal; tr[5] = resid_dual; tr[6] = 0; tr[7] = c; tr[8] = 0; tr[9] = rho; tr[10] = rho; tr[11] = 0; }
        if (conv) { it = i + 1; break; }
        const double old_c = c;
        double daz2 = 0.0;
        for (int k = 0; k < dim; ++k) { const double d = z[k] - az[k]; daz2 += d * d; }
        c = rho * resid_primal * resid_primal + rho * daz2;
        if (tr) { tr[6] = c; tr[8] = c < 0.999 * old_c ? 1 : 2; }
        if (c < 0.999 * old_c) {
            const double old_a = a;
            a = 0.5 + 0.5 * sqrt(1.0 + 4.0 * old_a * old_a);
            const double ratio = (old_a - 1.0) / a, t1 = 1.0 + ratio;
            for (int k = 0; k < dim; ++k) { az[k] = t1 * z[k] - ratio * oz[k]; ay[k] = t1 * y[k] - ratio * oy[k]; }
        } else {
            a = 1.0;
            memcpy(az, oz, (size_t)dim * sizeof(double));
            memcpy(ay, oy, (size_t)dim * sizeof(double));
            c = old_c / 0.999;
        }
        if (i > 5) rho = rho_rule(rho, resid_primal, eps_primal, resid_dual, eps_dual);
        if (tr) tr[10] = rho;
        if (budget_s > 0 && now_s() - t0 > budget_s) { it = i + 1; break; }
    }
    *loop_seconds = now_s() - t0;
    *niter_out = it;
    *rho_out = rho;
    if (z_out) memcpy(z_out, z, (size_t)dim * sizeof(double));
    if (adjz_out) memcpy(adjz_out, az, (size_t)dim * sizeof(double));
    if (adjy_out) memcpy(adjy_out, ay, (size_t)dim * sizeof(double));
    if (ntrace) *ntrace = ntr;
    free(x); free(z); free(y); free(az); free(ay); free(oz); free(oy); free(vec); free(small); free(g); free(Mown);
    return 0;
}

/* Column-block ("sharing") basis pursuit: the loop of oracle/solvers.py SharingBP.solve (that class states what is the unbuilt
 * reference source's, src/TODO/PADMMBP.h, and what had to be modelled on the current PADMMBase_Master).  A: n x p column-major,
 * N blocks of p div N columns (the last takes the remainder, PADMMBP.h:150-167); sprad[i] = lambda_max(A_i'A_i) and rho are inputs.
 * trace: 7 doubles per iteration (iteration, eps_primal, eps_dual, resid_primal, resid_dual, regular, converged) as the NumPy class
 * records them.  nthreads > 1: the columns' dot products of a block spread over the threads; the blocks' A_i x_i one thread each. */
int oracle_sharing_loop(const double* A, long lda, int n, int p, int N, const double* b, const double* sprad, double rho, double eps_abs,
                        double eps_rel, int maxit, int nthreads, double* x_out, int* niter_out, double* loop_seconds, double* trace,
                        int trace_cap, int* ntrace, double budget_s) {
    if (nthreads < 1) nthreads = 1;
    double* Aown = lda == n ? own_copy(A, (size_t)n * sizeof(double), p, nthreads) : NULL;
    if (Aown) A = Aown;
    const int chunk = p / N;
    double* x = calloc((size_t)p, sizeof(double));
    double* Ax = calloc((size_t)N * n, sizeof(double));
    double* nAx = calloc((size_t)N * n, sizeof(double));
    double* y = calloc((size_t)n, sizeof(double));
    double* r = calloc((size_t)n, sizeof(double));
    double* S = calloc((size_t)n, sizeof(double));
    double* v = calloc((size_t)n, sizeof(double));
    double* Sn = calloc((size_t)n, sizeof(double));
    double* zbar = calloc((size_t)n, sizeof(double));
    if (!x || !Ax || !nAx || !y || !r || !S || !v || !Sn || !zbar) return 1;
    for (int k = 0; k < n; ++k) zbar[k] = b[k] / (double)N;
    const double dN = (double)N, sq = sqrt((double)n * (double)N);
    int it = maxit + 1, ntr = 0, counter = 0;
    const double t0 = now_s();
    for (int i = 0; i < maxit; ++i) {
        /* thresholds (SharingBP._eps) */
        double sax = 0.0, abar_r = 0.0, r2 = 0.0;
        for (int q = 0; q < N; ++q) sax += dot_d(Ax + (size_t)q * n, Ax + (size_t)q * n, n);
        for (int k = 0; k < n; ++k) {
            double a = 0.0;
            for (int q = 0; q < N; ++q) a += Ax[(size_t)q * n + k];
            abar_r += (a / dN) * r[k];
            r2 += r[k] * r[k];
        }
        const double sz = sax - 2.0 * dN * abar_r + dN * r2;
        double m = sax > sz ? sax : sz;
        if (m < 0.0) m = 0.0;
        const double eps_primal = eps_rel * sqrt(m) + sq * eps_abs;
        const double eps_dual = eps_rel * sqrt(dN) * norm_d(y, n) + sq * eps_abs;
        for (int k = 0; k < n; ++k) v[k] = y[k] / rho + r[k];
        const int regular = counter % 10 == 0;
        /* workers: x-update (every column on regular iterations, the current non-zeros otherwise) */
        for (int q = 0; q < N; ++q) {
            const int c0 = q * chunk, c1 = q == N - 1 ? p : c0 + chunk;
            const double gamma = 2.0 * rho + sprad[q], pen = 1.0 / (rho * gamma);
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
            for (int j = c0; j < c1; ++j) {
                if (!regular && x[j] == 0.0) continue;
                const double val = x[j] - dot_d(A + (size_t)j * lda, v, n) / gamma;
                x[j] = val > pen ? val - pen : (val < -pen ? val + pen : 0.0);
            }
        }
#pragma omp parallel for schedule(static) num_threads(nthreads < N ? nthreads : N) if (nthreads > 1)
        for (int q = 0; q < N; ++q) {                                   /* A_i x_i over the non-zeros, column order */
            const int c0 = q * chunk, c1 = q == N - 1 ? p : c0 + chunk;
            double* o = nAx + (size_t)q * n;
            for (int k = 0; k < n; ++k) o[k] = 0.0;
            for (int j = c0; j < c1; ++j) {
                const double xj = x[j];
                if (xj == 0.0) continue;
                const double* c = A + (size_t)j * lda;
                for (int k = 0; k < n; ++k) o[k] += xj * c[k];
            }
        }
        ++counter;
        double qd = 0.0, drdS = 0.0, dr2 = 0.0, rn2 = 0.0;
        for (int k = 0; k < n; ++k) {
            double sn = 0.0;
            for (int q = 0; q < N; ++q) sn += nAx[(size_t)q * n + k];
            Sn[k] = sn;
        }
        for (int q = 0; q < N; ++q)
            for (int k = 0; k < n; ++k) { const double d = nAx[(size_t)q * n + k] - Ax[(size_t)q * n + k]; qd += d * d; }
        for (int k = 0; k < n; ++k) {
            const double rn = Sn[k] / dN - zbar[k];
            const double dr = rn - r[k], dS = Sn[k] - S[k];
            drdS += dr * dS; dr2 += dr * dr; rn2 += rn * rn;
            r[k] = rn; S[k] = Sn[k];
            y[k] = y[k] + rho * rn;
        }
        memcpy(Ax, nAx, (size_t)N * n * sizeof(double));
        const double sd = qd - 2.0 * drdS + dN * dr2;
        const double resid_dual = rho * sqrt(sd > 0.0 ? sd : 0.0), resid_primal = sqrt(dN * rn2);
        const int conv = resid_primal < eps_primal && resid_dual < eps_dual;
        if (trace && ntr < trace_cap) {
            double* tr = trace + 7 * (size_t)ntr++;
            tr[0] = i; tr[1] = eps_primal; tr[2] = eps_dual; tr[3] = resid_primal; tr[4] = resid_dual; tr[5] = regular; tr[6] = conv;
        }
        if (conv) { it = i + 1; break; }
        if (budget_s > 0 && now_s() - t0 > budget_s) { it = i + 1; break; }
    }
    *loop_seconds = now_s() - t0;
    *niter_out = it;
    if (x_out) memcpy(x_out, x, (size_t)p * sizeof(double));
    if (ntrace) *ntrace = ntr;
    free(x); free(Ax); free(nAx); free(y); free(r); free(S); free(v); free(Sn); free(zbar); free(Aown);
    return 0;
}

int oracle_loops_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
