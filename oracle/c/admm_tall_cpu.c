/*
 * admm_tall_cpu.c -- CPU restatement in C of the reference's tall Lasso / Elastic-net ADMM loop.
 * TEST INFRASTRUCTURE / CPU BASELINE, NOT PRODUCT: only tests/, __graft_entry__.smoke() and the cpu_baseline
 * leg of bench.py may load the library built from this file (oracle/c/Makefile -> oracle/c/liboracle_tall.so,
 * through oracle/ctall.py).  Nothing under admm_amd/ links or loads it.
 *
 * Follows (reference = /root/reference, yixuan/ADMM 1.0):
 *   FADMMBase::solve                 src/FADMMBase.h:185-265   (update_x/z/y, converged, acceleration / restart)
 *   ADMMLassoTall::next_x            src/ADMMLassoTall.h:70-80 (rhs = X'y - adj_y + rho adj_z; LLT solve)
 *   ADMMLassoTall::next_z            src/ADMMLassoTall.h:55-69,81-85   (soft threshold, double compare)
 *   ADMMEnetTall::enet               src/ADMMEnet.h:24-45
 *   eps / residual closed forms      src/ADMMLassoTall.h:141-161
 *   init / init_warm                 src/ADMMLassoTall.h:179-231 (a, c kept across lambdas)
 *   lambda loop                      src/Lasso.cpp:97-124
 * The Gram matrix, rho and the factorisation are inputs (the Python side builds them with LAPACK, as the NumPy
 * oracle does): this file is the per-iteration hot loop, in the two CPU configurations bench.py reports:
 *   mode 0  "faithful": x = L^-T (L^-1 rhs), two triangular solves on the float Cholesky factor, ONE thread --
 *           the reference's effective configuration (Eigen LLT::solve is serial and Lasso.cpp:1 defines
 *           EIGEN_DONT_PARALLELIZE);
 *   mode 1  "best effort": x = Minv rhs with the cached float inverse, the mat-vec spread over OpenMP threads
 *           (what a tuned CPU implementation of this build's own x-update would do; not the reference's arithmetic).
 * Vector arithmetic is float with double scalars where the reference uses double (thresholds, rho products).
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

/* x <- L^-1 x then x <- L^-T x, L lower triangular column-major (ld = p): Eigen's LLT::solve on one right-hand side */
static void llt_solve(const float* L, int p, float* x) {
    for (int j = 0; j < p; ++j) {                       /* forward: column-oriented axpy, contiguous */
        const float* c = L + (size_t)j * p;
        const float xj = x[j] / c[j];
        x[j] = xj;
        for (int i = j + 1; i < p; ++i) x[i] -= xj * c[i];
    }
    for (int j = p - 1; j >= 0; --j) {                  /* backward with L': contiguous dot */
        const float* c = L + (size_t)j * p;
        float s = 0.f;
        for (int i = j + 1; i < p; ++i) s += c[i] * x[i];
        x[j] = (x[j] - s) / c[j];
    }
}

/* y = M x for a symmetric float matrix (both triangles stored, column-major): column j dotted with x gives y_j */
static void sym_matvec_omp(const float* M, int p, const float* x, float* y, int nthreads) {
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int j = 0; j < p; ++j) {
        const float* c = M + (size_t)j * p;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int i = 0;
        for (; i + 3 < p; i += 4) { s0 += c[i] * x[i]; s1 += c[i + 1] * x[i + 1]; s2 += c[i + 2] * x[i + 2]; s3 += c[i + 3] * x[i + 3]; }
        for (; i < p; ++i) s0 += c[i] * x[i];
        y[j] = (s0 + s1) + (s2 + s3);
    }
}

static float norm_f(const float* v, int p) {           /* VectorXf::norm(): float accumulation */
    float s = 0.f;
    for (int i = 0; i < p; ++i) s += v[i] * v[i];
    return sqrtf(s);
}

/*
 * One warm-started lambda path (Lasso.cpp:97-124) from a cold start.
 *   F        mode 0: Cholesky factor L of X'X + rho I (lower, column-major, ld p);  mode 1: (X'X + rho I)^-1 (full)
 *   XY       X'y (float, length p)
 *   lam      internal lambdas (lambda_user * n / scaleY), nlam of them, already rounded to float precision
 *   alpha    < 0: Lasso prox; in [0, 1]: elastic net
 *   beta_out nlam x p (row-major per lambda): get_z() after each solve (standardised scale)
 *   niter_out[nlam]; *loop_seconds = wall time of the iteration loops only
 *   trace    NULL, or room for trace_cap records of 8 doubles -- one per decision, what the reference's commented-out
 *            print_row would show: lambda index, iteration, eps_primal, eps_dual, resid_primal, resid_dual, c (0 on the exit
 *            iteration), outcome (0 converged, 1 accelerate, 2 restart); *ntrace = records written
 * Returns 0.
 */
int oracle_tall_path_traced(const float* F, const float* XY, int p, const double* lam, int nlam, double rho,
                            double eps_abs, double eps_rel, int maxit, double alpha, int mode, int nthreads,
                            float* beta_out, int* niter_out, double* loop_seconds, double* trace, int trace_cap, int* ntrace) {
    int ntr = 0;
    float* x = calloc((size_t)p, sizeof(float));
    float* z = calloc((size_t)p, sizeof(float));
    float* y = calloc((size_t)p, sizeof(float));
    float* adj_z = calloc((size_t)p, sizeof(float));
    float* adj_y = calloc((size_t)p, sizeof(float));
    float* old_z = calloc((size_t)p, sizeof(float));
    float* old_y = calloc((size_t)p, sizeof(float));
    float* rhs = calloc((size_t)p, sizeof(float));
    float* r = calloc((size_t)p, sizeof(float));
    if (!x || !z || !y || !adj_z || !adj_y || !old_z || !old_y || !rhs || !r) return 1;
    if (nthreads < 1) nthreads = 1;
    /* mode 1 on several threads: the mat-vec is memory-bound, so WHERE the matrix lives decides its rate.  The caller's array was
     * written by one thread (one NUMA node); the threads stream a copy whose pages each of them touched first, with the static
     * schedule of sym_matvec_omp (column j belongs to the same thread in both loops).  Together with OMP_PROC_BIND=spread /
     * OMP_PLACES=cores (set by bench.py before this library is loaded) that pins every column block to the memory of the core
     * that reads it.  Not timed (setup). */
    float* Fown = NULL;
    if (mode == 1 && nthreads > 1) {
        Fown = malloc((size_t)p * (size_t)p * sizeof(float));
        if (Fown) {
#pragma omp parallel for schedule(static) num_threads(nthreads)
            for (int j = 0; j < p; ++j) memcpy(Fown + (size_t)j * p, F + (size_t)j * p, (size_t)p * sizeof(float));
            F = Fown;
        }
    }
    const int enet = alpha >= 0.0;
    const float alpha_f = (float)alpha;
    const float rho_f = (float)rho;
    const double sqrt_p = sqrt((double)p);
    double adj_a = 1.0, adj_c = 9999.0;                                     /* init(): ADMMLassoTall.h:207-213 */
    const double t0 = now_s();
    for (int l = 0; l < nlam; ++l) {
        const double lambda = (double)(float)lam[l];                        /* Scalar lambda (float) */
        int it = maxit + 1;                                                 /* `return i + 1` after the loop ran out */
        for (int i = 0; i < maxit; ++i) {
            memcpy(old_z, z, (size_t)p * sizeof(float));                    /* FADMMBase.h:227-229 */
            memcpy(old_y, y, (size_t)p * sizeof(float));
            /* update_x: eps from the current iterate (FADMMBase.h:187-188, ADMMLassoTall.h:141-149) */
            const double nx = norm_f(x, p), nz = norm_f(z, p), ny = norm_f(y, p);
            const double eps_primal = (nx > nz ? nx : nz) * eps_rel + sqrt_p * eps_abs;
            const double eps_dual = ny * eps_rel + sqrt_p * eps_abs;
            for (int k = 0; k < p; ++k) {                                   /* next_x: ADMMLassoTall.h:70-80 */
                const float t = XY[k] - adj_y[k];
                rhs[k] = adj_z[k] != 0.f ? (float)((double)t + rho * (double)adj_z[k]) : t;
            }
            if (mode == 0) { memcpy(x, rhs, (size_t)p * sizeof(float)); llt_solve(F, p, x); }
            else sym_matvec_omp(F, p, rhs, x, nthreads);
            /* update_z: next_z + resid_dual (FADMMBase.h:195-202, ADMMLassoTall.h:81-85,150-153) */
            const double pen = lambda / rho;
            float dz2 = 0.f;
            for (int k = 0; k < p; ++k) {
                const float vec = x[k] + adj_y[k] / rho_f;
                float zn;
                if (!enet) {
                    const double v = (double)vec;
                    zn = v > pen ? (float)(v - pen) : (v < -pen ? (float)(v + pen) : 0.f);
                } else {                                                    /* ADMMEnet.h:24-40 */
                    const float thresh = (float)((double)alpha_f * pen);
                    const float denom = (float)(1.0 + pen * (1.0 - (double)alpha_f));
                    zn = vec > thresh ? (vec - thresh) / denom : (vec < -thresh ? (vec + thresh) / denom : 0.f);
                }
                const float d = zn - old_z[k];
                dz2 += d * d;
                z[k] = zn;
            }
            const double resid_dual = rho * sqrt((double)dz2);
            /* update_y (FADMMBase.h:203-211) */
            for (int k = 0; k < p; ++k) { r[k] = x[k] - z[k]; y[k] = adj_y[k] + rho_f * r[k]; }
            const double resid_primal = (double)norm_f(r, p);
            double* tr = (trace && ntr < trace_cap) ? trace + 8 * (size_t)ntr++ : NULL;
            if (tr) { tr[0] = l; tr[1] = i; tr[2] = eps_primal; tr[3] = eps_dual; tr[4] = resid_primal; tr[5] = resid_dual; tr[6] = 0.0; tr[7] = 0.0; }
            if (resid_primal < eps_primal && resid_dual < eps_dual) { it = i + 1; break; }      /* FADMMBase.h:237-238 */
            /* acceleration / restart (FADMMBase.h:240-256, ADMMLassoTall.h:154-161) */
            const double old_c = adj_c;
            float daz2 = 0.f;
            for (int k = 0; k < p; ++k) { const float d = z[k] - adj_z[k]; daz2 += d * d; }
            adj_c = rho * resid_primal * resid_primal + rho * (double)daz2;
            if (tr) { tr[6] = adj_c; tr[7] = adj_c < 0.999 * old_c ? 1.0 : 2.0; }
            if (adj_c < 0.999 * old_c) {
                const double old_a = adj_a;
                adj_a = 0.5 + 0.5 * sqrt(1.0 + 4.0 * old_a * old_a);
                const double ratio = (old_a - 1.0) / adj_a;
                const float t1 = (float)(1.0 + ratio), t = (float)ratio;
                for (int k = 0; k < p; ++k) {
                    adj_z[k] = t1 * z[k] - t * old_z[k];
                    adj_y[k] = t1 * y[k] - t * old_y[k];
                }
            } else {
                adj_a = 1.0;
                memcpy(adj_z, old_z, (size_t)p * sizeof(float));
                memcpy(adj_y, old_y, (size_t)p * sizeof(float));
                adj_c = old_c / 0.999;
            }
        }
        niter_out[l] = it;
        memcpy(beta_out + (size_t)l * p, z, (size_t)p * sizeof(float));     /* get_z(): Lasso.cpp:108 */
    }
    *loop_seconds = now_s() - t0;
    if (ntrace) *ntrace = ntr;
    free(x); free(z); free(y); free(adj_z); free(adj_y); free(old_z); free(old_y); free(rhs); free(r); free(Fown);
    return 0;
}

int oracle_tall_path(const float* F, const float* XY, int p, const double* lam, int nlam, double rho,
                     double eps_abs, double eps_rel, int maxit, double alpha, int mode, int nthreads,
                     float* beta_out, int* niter_out, double* loop_seconds) {
    return oracle_tall_path_traced(F, XY, p, lam, nlam, rho, eps_abs, eps_rel, maxit, alpha, mode, nthreads, beta_out, niter_out,
                                   loop_seconds, NULL, 0, NULL);
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
