// Host-side logic of the reference's single Spectra call (nev = 1, ncv = 3, maxit = 10, tol = 0.1)
// with the symmetric mat-vec running on the device.
//
//   Spectra::SymEigsSolver<float, LARGEST_ALGE, DenseSymMatProd<float>> eigs(&op, 1, 3);
//   eigs.init(); eigs.compute(10, 0.1); evals[0]
//     /root/reference/src/ADMMLassoTall.h:196-201, ADMMLassoWide.h:202-207
//
// Restated from the algorithm in src/Spectra/SymEigsSolver.h (factorize_from :201-280, restart
// :283-323, num_converged :326-335, nev_adjusted :338-353, retrieve_ritzpair :356-397, init
// :494-544, compute :564-587), SimpleRandom.h:38-76 (deterministic LCG start vector),
// LinAlg/TridiagEigen.h (QR iteration on the 3x3 tridiagonal) and TridiagQR in
// LinAlg/UpperHessenbergQR.h:415-600 (shifted QR step of the implicit restart).  The value is a
// loose UNDER-estimate of lambda_max by design; it sets rho (tall) and the step size (wide), so
// it is reproduced, not improved.  Vectors of length n stay on the host (3-5 mat-vecs in total).
#include "prep.h"
#include <algorithm>
#include <functional>
#include <limits>

namespace admm {
namespace {

constexpr int NCV = 3;

template <typename T>
std::vector<T> simple_random_vec(int n, unsigned long seed) {
    const long a = 16807, mx = 2147483647L;
    long r = seed ? (long)(seed & (unsigned long)mx) : 1;
    std::vector<T> out(n);
    for (int i = 0; i < n; ++i) {
        unsigned long lo = a * (long)(r & 0xFFFF);
        unsigned long hi = a * (long)((unsigned long)r >> 16);
        lo += (hi & 0x7FFF) << 16;
        if ((long)lo > mx) { lo &= mx; ++lo; }
        lo += hi >> 15;
        if ((long)lo > mx) { lo &= mx; ++lo; }
        r = (long)lo;
        out[i] = T(r) / T(mx) - T(0.5);
    }
    return out;
}

template <typename T>
T dotf(const T* a, const T* b, int n) {
    double s = 0;
    for (int i = 0; i < n; ++i) s += (double)a[i] * (double)b[i];
    return (T)s;
}
template <typename T>
T normf(const T* a, int n) { return (T)std::sqrt((double)dotf(a, a, n)); }

template <typename T>
void make_givens(T p, T q, T& c, T& s) {      // Eigen JacobiRotation::makeGivens (real)
    if (q == T(0.)) { c = p < T(0.) ? -T(1.) : T(1.); s = T(0.); }
    else if (p == T(0.)) { c = T(0.); s = q < T(0.) ? T(1.) : -T(1.); }
    else if (std::fabs(p) > std::fabs(q)) {
        T t = q / p, u = std::sqrt(T(1.) + t * t);
        if (p < T(0.)) u = -u;
        c = T(1.) / u; s = -t * c;
    } else {
        T t = p / q, u = std::sqrt(T(1.) + t * t);
        if (q < T(0.)) u = -u;
        s = -T(1.) / u; c = -t * s;
    }
}

// Eigen-decomposition of the symmetric tridiagonal part of H (3x3, row-major here).
template <typename T>
void tridiag_eigen(const T H[NCV][NCV], T evals[NCV], T Q[NCV][NCV]) {
    const int n = NCV;
    T d[NCV], e[NCV - 1];
    for (int i = 0; i < n; ++i) { d[i] = H[i][i]; for (int j = 0; j < n; ++j) Q[i][j] = (i == j) ? T(1.) : T(0.); }
    for (int i = 0; i < n - 1; ++i) e[i] = H[i + 1][i];
    const T prec = sizeof(T) == 4 ? T(1e-5) : T(1e-12);     // NumTraits<T>::dummy_precision()
    int end = n - 1, start = 0, iter = 0;
    while (end > 0) {
        for (int i = start; i < end; ++i) {
            T x = std::fabs(e[i]), y = std::fabs(d[i]) + std::fabs(d[i + 1]);
            if (x * x <= y * y * prec * prec) e[i] = T(0.);
        }
        while (end > 0 && e[end - 1] == T(0.)) end--;
        if (end <= 0) break;
        if (++iter > 30 * n) throw Error(ADMM_ERR_EIGS, "tridiagonal QR iteration did not converge");
        start = end - 1;
        while (start > 0 && e[start - 1] != T(0.)) start--;
        T td = (d[end - 1] - d[end]) * T(0.5), ee = e[end - 1], mu = d[end];
        if (td == T(0.)) mu -= std::fabs(ee);
        else {
            T e2 = ee * ee, h = std::hypot(td, ee);
            if (e2 == T(0.)) mu -= (ee / (td + (td > T(0.) ? T(1.) : -T(1.)))) * (ee / h);
            else mu -= e2 / (td + (td > T(0.) ? h : -h));
        }
        T x = d[start] - mu, z = e[start];
        for (int k = start; k < end; ++k) {
            T c, s;
            make_givens(x, z, c, s);
            T sdk = s * d[k] + c * e[k];
            T dkp1 = s * e[k] + c * d[k + 1];
            d[k] = c * (c * d[k] - s * e[k]) - s * (c * e[k] - s * d[k + 1]);
            d[k + 1] = s * sdk + c * dkp1;
            e[k] = c * sdk - s * dkp1;
            if (k > start) e[k - 1] = c * e[k - 1] - s * z;
            x = e[k];
            if (k < end - 1) { z = -s * e[k + 1]; e[k + 1] = c * e[k + 1]; }
            for (int i = 0; i < n; ++i) {           // Q = Q * G
                T qk = Q[i][k], qk1 = Q[i][k + 1];
                Q[i][k] = c * qk - s * qk1;
                Q[i][k + 1] = s * qk + c * qk1;
            }
        }
    }
    for (int i = 0; i < n; ++i) evals[i] = d[i];
}

// One shifted QR step on the tridiagonal H: H <- RQ (+shift handled by caller), Qacc <- Qacc * Q.
template <typename T>
void tridiag_qr_step(T H[NCV][NCV], T Qacc[NCV][NCV]) {
    const int n = NCV;
    T Tm[NCV][NCV] = {};
    for (int i = 0; i < n; ++i) Tm[i][i] = H[i][i];
    for (int i = 0; i < n - 1; ++i) { Tm[i][i + 1] = H[i + 1][i]; Tm[i + 1][i] = H[i + 1][i]; }
    T cs[NCV - 1], sn[NCV - 1];
    const T eps = std::numeric_limits<T>::epsilon();
    for (int i = 0; i < n - 1; ++i) {
        T a = Tm[i][i], b = Tm[i + 1][i];
        T r = std::sqrt(a * a + b * b), c, s;
        if (r <= eps) { r = T(0.); c = T(1.); s = T(0.); }
        else { c = a / r; s = -b / r; }
        cs[i] = c; sn[i] = s;
        Tm[i][i] = r; Tm[i + 1][i] = T(0.);
        T tmp = Tm[i][i + 1];
        Tm[i][i + 1] = c * tmp - s * Tm[i + 1][i + 1];
        Tm[i + 1][i + 1] = s * tmp + c * Tm[i + 1][i + 1];
        if (i < n - 2) {
            Tm[i][i + 2] = -s * Tm[i + 1][i + 2];
            Tm[i + 1][i + 2] = c * Tm[i + 1][i + 2];
        }
    }
    for (int i = 0; i < n - 1; ++i) {               // apply_YQ
        T c = cs[i], s = sn[i];
        for (int r = 0; r < n; ++r) {
            T yi = Qacc[r][i], yi1 = Qacc[r][i + 1];
            Qacc[r][i] = c * yi - s * yi1;
            Qacc[r][i + 1] = s * yi + c * yi1;
        }
    }
    T RQ[NCV][NCV] = {};                       // matrix_RQ: only diag and first super-diagonal of R are used
    for (int i = 0; i < n; ++i) RQ[i][i] = Tm[i][i];
    for (int i = 0; i < n - 1; ++i) RQ[i][i + 1] = Tm[i][i + 1];
    for (int i = 0; i < n - 1; ++i) {
        T c = cs[i], s = sn[i];
        T m11 = RQ[i][i], m12 = RQ[i][i + 1], m21 = RQ[i + 1][i], m22 = RQ[i + 1][i + 1];
        RQ[i][i] = c * m11 - s * m12;
        RQ[i + 1][i] = c * m21 - s * m22;
        RQ[i + 1][i + 1] = s * m21 + c * m22;
    }
    for (int i = 0; i < n - 1; ++i) RQ[i][i + 1] = RQ[i + 1][i];
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) H[i][j] = RQ[i][j];
}

}  // namespace

template <typename T>
static T lanczos_largest_impl(const std::function<void(const T*, T*)>& op, int n, int* nmatop_out) {
    const int nev = 1;
    const int ncv = NCV;
    if (n < ncv) throw Error(ADMM_ERR_EIGS, "matrix too small for the ncv=3 Lanczos estimate");
    const T prec = (T)std::pow((double)std::numeric_limits<T>::epsilon(), 2.0 / 3.0);
    const int maxit = 10;
    const T tol = T(0.1);
    std::vector<T> V((size_t)n * ncv, T(0.)), f(n), w(n), tmp(n);
    T H[NCV][NCV] = {};
    int nmatop = 0;
    auto Vc = [&](int c) { return V.data() + (size_t)c * n; };

    {   // init()
        std::vector<T> v = simple_random_vec<T>(n, 0);
        T vn = normf(v.data(), n);
        for (int i = 0; i < n; ++i) v[i] /= vn;
        op(v.data(), w.data()); nmatop++;
        H[0][0] = dotf(v.data(), w.data(), n);
        for (int i = 0; i < n; ++i) { f[i] = w[i] - v[i] * H[0][0]; Vc(0)[i] = v[i]; }
    }

    auto factorize_from = [&](int from_k, int to_m, const std::vector<T>& fk) {
        if (to_m <= from_k) return;
        f = fk;
        T beta = normf(f.data(), n);
        for (int i = 0; i < ncv; ++i) for (int j = from_k; j < ncv; ++j) H[i][j] = T(0.);
        for (int i = from_k; i < ncv; ++i) for (int j = 0; j < from_k; ++j) H[i][j] = T(0.);
        for (int i = from_k; i < to_m; ++i) {
            bool restart = false;
            if (beta < prec) {
                f = simple_random_vec<T>(n, 2 * i);
                for (int c = 0; c < i; ++c) tmp[c] = dotf(Vc(c), f.data(), n);
                for (int c = 0; c < i; ++c) { T t = tmp[c]; for (int r = 0; r < n; ++r) f[r] -= Vc(c)[r] * t; }
                beta = normf(f.data(), n);
                restart = true;
            }
            T* vi = Vc(i);
            for (int r = 0; r < n; ++r) vi[r] = f[r] / beta;
            H[i][i - 1] = restart ? T(0.) : beta;
            op(vi, w.data()); nmatop++;
            T Hii = dotf(vi, w.data(), n);
            H[i - 1][i] = H[i][i - 1];
            H[i][i] = Hii;
            if (restart) { for (int r = 0; r < n; ++r) f[r] = w[r] - Hii * vi[r]; }
            else { const T h = H[i][i - 1]; const T* vp = Vc(i - 1); for (int r = 0; r < n; ++r) f[r] = w[r] - h * vp[r] - Hii * vi[r]; }
            beta = normf(f.data(), n);
            T Vf[NCV];
            T vmax = T(0.);
            for (int c = 0; c <= i; ++c) { Vf[c] = dotf(Vc(c), f.data(), n); vmax = std::max(vmax, std::fabs(Vf[c])); }
            int count = 0;
            while (count < 5 && vmax > prec * beta) {
                for (int c = 0; c <= i; ++c) { T t = Vf[c]; const T* vc = Vc(c); for (int r = 0; r < n; ++r) f[r] -= vc[r] * t; }
                H[i - 1][i] += Vf[i - 1];
                H[i][i - 1] = H[i - 1][i];
                H[i][i] += Vf[i];
                beta = normf(f.data(), n);
                vmax = T(0.);
                for (int c = 0; c <= i; ++c) { Vf[c] = dotf(Vc(c), f.data(), n); vmax = std::max(vmax, std::fabs(Vf[c])); }
                count++;
            }
        }
    };

    T ritz_val[NCV], ritz_est[NCV];
    auto retrieve_ritzpair = [&]() {
        T ev[NCV], Q[NCV][NCV];
        tridiag_eigen(H, ev, Q);
        int ind[NCV] = {0, 1, 2};
        std::stable_sort(ind, ind + ncv, [&](int a, int b) { return -ev[a] < -ev[b]; });
        for (int i = 0; i < ncv; ++i) { ritz_val[i] = ev[ind[i]]; ritz_est[i] = Q[ncv - 1][ind[i]]; }
    };

    factorize_from(1, ncv, f);
    retrieve_ritzpair();
    int nconv = 0;
    for (int it = 0; it < maxit; ++it) {
        const T thresh = tol * std::max(std::fabs(ritz_val[0]), prec);
        const T resid = std::fabs(ritz_est[0]) * normf(f.data(), n);
        nconv = resid < thresh ? 1 : 0;
        if (nconv >= nev) break;
        int nev_new = nev;
        for (int i = nev; i < ncv; ++i) if (std::fabs(ritz_est[i]) < prec) nev_new++;
        nev_new += std::min(nconv, (ncv - nev_new) / 2);
        if (nev_new == 1 && ncv >= 6) nev_new = ncv / 2;
        else if (nev_new == 1 && ncv > 2) nev_new = 2;
        const int k = nev_new;
        if (k >= ncv) continue;
        T Q[NCV][NCV] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
        for (int i = k; i < ncv; ++i) {
            for (int d = 0; d < ncv; ++d) H[d][d] -= ritz_val[i];
            tridiag_qr_step(H, Q);
            for (int d = 0; d < ncv; ++d) H[d][d] += ritz_val[i];
        }
        std::vector<T> Vs((size_t)n * (k + 1));
        for (int i = 0; i < k; ++i) {
            const int nnz = ncv - k + i + 1;
            for (int r = 0; r < n; ++r) {
                T s = T(0.);
                for (int c = 0; c < nnz; ++c) s += Vc(c)[r] * Q[c][i];
                Vs[(size_t)i * n + r] = s;
            }
        }
        for (int r = 0; r < n; ++r) {
            T s = T(0.);
            for (int c = 0; c < ncv; ++c) s += Vc(c)[r] * Q[c][k];
            Vs[(size_t)k * n + r] = s;
        }
        std::memcpy(V.data(), Vs.data(), sizeof(T) * (size_t)n * (k + 1));
        std::vector<T> fk(n);
        const T q = Q[ncv - 1][k - 1], hk = H[k][k - 1];
        for (int r = 0; r < n; ++r) fk[r] = f[r] * q + Vc(k)[r] * hk;
        factorize_from(k, ncv, fk);
        retrieve_ritzpair();
    }
    if (nmatop_out) *nmatop_out = nmatop;
    if (nconv < nev)
        throw Error(ADMM_ERR_EIGS, "Lanczos (nev=1, ncv=3, 10 restarts, tol 0.1) did not converge; the reference reads an empty eigenvalue vector here");
    return ritz_val[0];
}


float lanczos_largest_f32(const std::function<void(const float*, float*)>& op, int n, int* nmatop_out) {
    const TraceRange trace_range("admm:lanczos");
    return lanczos_largest_impl<float>(op, n, nmatop_out);
}
// the same call with Scalar = double (src/TODO/ADMMDantzig.h:226-233)
double lanczos_largest_f64(const std::function<void(const double*, double*)>& op, int n, int* nmatop_out) {
    return lanczos_largest_impl<double>(op, n, nmatop_out);
}

}  // namespace admm
