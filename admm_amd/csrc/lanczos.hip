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

namespace admm {
namespace {

constexpr int NCV = 3;

std::vector<float> simple_random_vec(int n, unsigned long seed) {
    const long a = 16807, mx = 2147483647L;
    long r = seed ? (long)(seed & (unsigned long)mx) : 1;
    std::vector<float> out(n);
    for (int i = 0; i < n; ++i) {
        unsigned long lo = a * (long)(r & 0xFFFF);
        unsigned long hi = a * (long)((unsigned long)r >> 16);
        lo += (hi & 0x7FFF) << 16;
        if ((long)lo > mx) { lo &= mx; ++lo; }
        lo += hi >> 15;
        if ((long)lo > mx) { lo &= mx; ++lo; }
        r = (long)lo;
        out[i] = float(r) / float(mx) - 0.5f;
    }
    return out;
}

float dotf(const float* a, const float* b, int n) {
    double s = 0;
    for (int i = 0; i < n; ++i) s += (double)a[i] * (double)b[i];
    return (float)s;
}
float normf(const float* a, int n) { return (float)std::sqrt((double)dotf(a, a, n)); }

void make_givens(float p, float q, float& c, float& s) {      // Eigen JacobiRotation::makeGivens (real)
    if (q == 0.f) { c = p < 0.f ? -1.f : 1.f; s = 0.f; }
    else if (p == 0.f) { c = 0.f; s = q < 0.f ? 1.f : -1.f; }
    else if (std::fabs(p) > std::fabs(q)) {
        float t = q / p, u = std::sqrt(1.f + t * t);
        if (p < 0.f) u = -u;
        c = 1.f / u; s = -t * c;
    } else {
        float t = p / q, u = std::sqrt(1.f + t * t);
        if (q < 0.f) u = -u;
        s = -1.f / u; c = -t * s;
    }
}

// Eigen-decomposition of the symmetric tridiagonal part of H (3x3, row-major here).
void tridiag_eigen(const float H[NCV][NCV], float evals[NCV], float Q[NCV][NCV]) {
    const int n = NCV;
    float d[NCV], e[NCV - 1];
    for (int i = 0; i < n; ++i) { d[i] = H[i][i]; for (int j = 0; j < n; ++j) Q[i][j] = (i == j) ? 1.f : 0.f; }
    for (int i = 0; i < n - 1; ++i) e[i] = H[i + 1][i];
    const float prec = 1e-5f;     // NumTraits<float>::dummy_precision()
    int end = n - 1, start = 0, iter = 0;
    while (end > 0) {
        for (int i = start; i < end; ++i) {
            float x = std::fabs(e[i]), y = std::fabs(d[i]) + std::fabs(d[i + 1]);
            if (x * x <= y * y * prec * prec) e[i] = 0.f;
        }
        while (end > 0 && e[end - 1] == 0.f) end--;
        if (end <= 0) break;
        if (++iter > 30 * n) throw Error(ADMM_ERR_EIGS, "tridiagonal QR iteration did not converge");
        start = end - 1;
        while (start > 0 && e[start - 1] != 0.f) start--;
        float td = (d[end - 1] - d[end]) * 0.5f, ee = e[end - 1], mu = d[end];
        if (td == 0.f) mu -= std::fabs(ee);
        else {
            float e2 = ee * ee, h = std::hypot(td, ee);
            if (e2 == 0.f) mu -= (ee / (td + (td > 0.f ? 1.f : -1.f))) * (ee / h);
            else mu -= e2 / (td + (td > 0.f ? h : -h));
        }
        float x = d[start] - mu, z = e[start];
        for (int k = start; k < end; ++k) {
            float c, s;
            make_givens(x, z, c, s);
            float sdk = s * d[k] + c * e[k];
            float dkp1 = s * e[k] + c * d[k + 1];
            d[k] = c * (c * d[k] - s * e[k]) - s * (c * e[k] - s * d[k + 1]);
            d[k + 1] = s * sdk + c * dkp1;
            e[k] = c * sdk - s * dkp1;
            if (k > start) e[k - 1] = c * e[k - 1] - s * z;
            x = e[k];
            if (k < end - 1) { z = -s * e[k + 1]; e[k + 1] = c * e[k + 1]; }
            for (int i = 0; i < n; ++i) {           // Q = Q * G
                float qk = Q[i][k], qk1 = Q[i][k + 1];
                Q[i][k] = c * qk - s * qk1;
                Q[i][k + 1] = s * qk + c * qk1;
            }
        }
    }
    for (int i = 0; i < n; ++i) evals[i] = d[i];
}

// One shifted QR step on the tridiagonal H: H <- RQ (+shift handled by caller), Qacc <- Qacc * Q.
void tridiag_qr_step(float H[NCV][NCV], float Qacc[NCV][NCV]) {
    const int n = NCV;
    float T[NCV][NCV] = {};
    for (int i = 0; i < n; ++i) T[i][i] = H[i][i];
    for (int i = 0; i < n - 1; ++i) { T[i][i + 1] = H[i + 1][i]; T[i + 1][i] = H[i + 1][i]; }
    float cs[NCV - 1], sn[NCV - 1];
    const float eps = 1.1920929e-07f;
    for (int i = 0; i < n - 1; ++i) {
        float a = T[i][i], b = T[i + 1][i];
        float r = std::sqrt(a * a + b * b), c, s;
        if (r <= eps) { r = 0.f; c = 1.f; s = 0.f; }
        else { c = a / r; s = -b / r; }
        cs[i] = c; sn[i] = s;
        T[i][i] = r; T[i + 1][i] = 0.f;
        float tmp = T[i][i + 1];
        T[i][i + 1] = c * tmp - s * T[i + 1][i + 1];
        T[i + 1][i + 1] = s * tmp + c * T[i + 1][i + 1];
        if (i < n - 2) {
            T[i][i + 2] = -s * T[i + 1][i + 2];
            T[i + 1][i + 2] = c * T[i + 1][i + 2];
        }
    }
    for (int i = 0; i < n - 1; ++i) {               // apply_YQ
        float c = cs[i], s = sn[i];
        for (int r = 0; r < n; ++r) {
            float yi = Qacc[r][i], yi1 = Qacc[r][i + 1];
            Qacc[r][i] = c * yi - s * yi1;
            Qacc[r][i + 1] = s * yi + c * yi1;
        }
    }
    float RQ[NCV][NCV] = {};                       // matrix_RQ: only diag and first super-diagonal of R are used
    for (int i = 0; i < n; ++i) RQ[i][i] = T[i][i];
    for (int i = 0; i < n - 1; ++i) RQ[i][i + 1] = T[i][i + 1];
    for (int i = 0; i < n - 1; ++i) {
        float c = cs[i], s = sn[i];
        float m11 = RQ[i][i], m12 = RQ[i][i + 1], m21 = RQ[i + 1][i], m22 = RQ[i + 1][i + 1];
        RQ[i][i] = c * m11 - s * m12;
        RQ[i + 1][i] = c * m21 - s * m22;
        RQ[i + 1][i + 1] = s * m21 + c * m22;
    }
    for (int i = 0; i < n - 1; ++i) RQ[i][i + 1] = RQ[i + 1][i];
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) H[i][j] = RQ[i][j];
}

}  // namespace

float lanczos_largest_f32(const std::function<void(const float*, float*)>& op, int n, int* nmatop_out) {
    const int nev = 1;
    const int ncv = NCV;
    if (n < ncv) throw Error(ADMM_ERR_EIGS, "matrix too small for the ncv=3 Lanczos estimate");
    const float prec = std::pow(1.1920929e-07f, 2.0f / 3.0f);
    const int maxit = 10;
    const float tol = 0.1f;
    std::vector<float> V((size_t)n * ncv, 0.f), f(n), w(n), tmp(n);
    float H[NCV][NCV] = {};
    int nmatop = 0;
    auto Vc = [&](int c) { return V.data() + (size_t)c * n; };

    {   // init()
        std::vector<float> v = simple_random_vec(n, 0);
        float vn = normf(v.data(), n);
        for (int i = 0; i < n; ++i) v[i] /= vn;
        op(v.data(), w.data()); nmatop++;
        H[0][0] = dotf(v.data(), w.data(), n);
        for (int i = 0; i < n; ++i) { f[i] = w[i] - v[i] * H[0][0]; Vc(0)[i] = v[i]; }
    }

    auto factorize_from = [&](int from_k, int to_m, const std::vector<float>& fk) {
        if (to_m <= from_k) return;
        f = fk;
        float beta = normf(f.data(), n);
        for (int i = 0; i < ncv; ++i) for (int j = from_k; j < ncv; ++j) H[i][j] = 0.f;
        for (int i = from_k; i < ncv; ++i) for (int j = 0; j < from_k; ++j) H[i][j] = 0.f;
        for (int i = from_k; i < to_m; ++i) {
            bool restart = false;
            if (beta < prec) {
                f = simple_random_vec(n, 2 * i);
                for (int c = 0; c < i; ++c) tmp[c] = dotf(Vc(c), f.data(), n);
                for (int c = 0; c < i; ++c) { float t = tmp[c]; for (int r = 0; r < n; ++r) f[r] -= Vc(c)[r] * t; }
                beta = normf(f.data(), n);
                restart = true;
            }
            float* vi = Vc(i);
            for (int r = 0; r < n; ++r) vi[r] = f[r] / beta;
            H[i][i - 1] = restart ? 0.f : beta;
            op(vi, w.data()); nmatop++;
            float Hii = dotf(vi, w.data(), n);
            H[i - 1][i] = H[i][i - 1];
            H[i][i] = Hii;
            if (restart) { for (int r = 0; r < n; ++r) f[r] = w[r] - Hii * vi[r]; }
            else { const float h = H[i][i - 1]; const float* vp = Vc(i - 1); for (int r = 0; r < n; ++r) f[r] = w[r] - h * vp[r] - Hii * vi[r]; }
            beta = normf(f.data(), n);
            float Vf[NCV];
            float vmax = 0.f;
            for (int c = 0; c <= i; ++c) { Vf[c] = dotf(Vc(c), f.data(), n); vmax = std::max(vmax, std::fabs(Vf[c])); }
            int count = 0;
            while (count < 5 && vmax > prec * beta) {
                for (int c = 0; c <= i; ++c) { float t = Vf[c]; const float* vc = Vc(c); for (int r = 0; r < n; ++r) f[r] -= vc[r] * t; }
                H[i - 1][i] += Vf[i - 1];
                H[i][i - 1] = H[i - 1][i];
                H[i][i] += Vf[i];
                beta = normf(f.data(), n);
                vmax = 0.f;
                for (int c = 0; c <= i; ++c) { Vf[c] = dotf(Vc(c), f.data(), n); vmax = std::max(vmax, std::fabs(Vf[c])); }
                count++;
            }
        }
    };

    float ritz_val[NCV], ritz_est[NCV];
    auto retrieve_ritzpair = [&]() {
        float ev[NCV], Q[NCV][NCV];
        tridiag_eigen(H, ev, Q);
        int ind[NCV] = {0, 1, 2};
        std::stable_sort(ind, ind + ncv, [&](int a, int b) { return -ev[a] < -ev[b]; });
        for (int i = 0; i < ncv; ++i) { ritz_val[i] = ev[ind[i]]; ritz_est[i] = Q[ncv - 1][ind[i]]; }
    };

    factorize_from(1, ncv, f);
    retrieve_ritzpair();
    int nconv = 0;
    for (int it = 0; it < maxit; ++it) {
        const float thresh = tol * std::max(std::fabs(ritz_val[0]), prec);
        const float resid = std::fabs(ritz_est[0]) * normf(f.data(), n);
        nconv = resid < thresh ? 1 : 0;
        if (nconv >= nev) break;
        int nev_new = nev;
        for (int i = nev; i < ncv; ++i) if (std::fabs(ritz_est[i]) < prec) nev_new++;
        nev_new += std::min(nconv, (ncv - nev_new) / 2);
        if (nev_new == 1 && ncv >= 6) nev_new = ncv / 2;
        else if (nev_new == 1 && ncv > 2) nev_new = 2;
        const int k = nev_new;
        if (k >= ncv) continue;
        float Q[NCV][NCV] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
        for (int i = k; i < ncv; ++i) {
            for (int d = 0; d < ncv; ++d) H[d][d] -= ritz_val[i];
            tridiag_qr_step(H, Q);
            for (int d = 0; d < ncv; ++d) H[d][d] += ritz_val[i];
        }
        std::vector<float> Vs((size_t)n * (k + 1));
        for (int i = 0; i < k; ++i) {
            const int nnz = ncv - k + i + 1;
            for (int r = 0; r < n; ++r) {
                float s = 0.f;
                for (int c = 0; c < nnz; ++c) s += Vc(c)[r] * Q[c][i];
                Vs[(size_t)i * n + r] = s;
            }
        }
        for (int r = 0; r < n; ++r) {
            float s = 0.f;
            for (int c = 0; c < ncv; ++c) s += Vc(c)[r] * Q[c][k];
            Vs[(size_t)k * n + r] = s;
        }
        std::memcpy(V.data(), Vs.data(), sizeof(float) * (size_t)n * (k + 1));
        std::vector<float> fk(n);
        const float q = Q[ncv - 1][k - 1], hk = H[k][k - 1];
        for (int r = 0; r < n; ++r) fk[r] = f[r] * q + Vc(k)[r] * hk;
        factorize_from(k, ncv, fk);
        retrieve_ritzpair();
    }
    if (nmatop_out) *nmatop_out = nmatop;
    if (nconv < nev)
        throw Error(ADMM_ERR_EIGS, "Lanczos (nev=1, ncv=3, 10 restarts, tol 0.1) did not converge; the reference reads an empty eigenvalue vector here");
    return ritz_val[0];
}

}  // namespace admm
