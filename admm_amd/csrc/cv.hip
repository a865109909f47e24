// K-fold cross-validation of the Lasso / elastic-net lambda path ("cross-validation folds as
// independent replicas across GPUs").  Not in the reference (R package ADMM has no CV driver): the folds are ordinary
// fits -- the same plan as admm_hip_lasso / admm_hip_enet on the training rows, so each fold's coefficients are
// bit-identical to a user-side call on that row subset (unless the folds are formed as down-dates of the full-data Gram,
// second half of this file) -- plus two small kernels: a row gather that builds the training /
// held-out matrices on the device from one resident copy of X, and the held-out squared prediction error for every lambda.
#include "solvers.h"
#include "comm.h"
#include "device_utils.h"
#include "prep.h"

namespace admm {

// out[r, j] = x[idx[r], j]   (column-major, doubles: the input type of the C ABI)
__global__ void __launch_bounds__(256)
cv_gather_rows_kernel(const double* __restrict__ x, long long ldx, const int* __restrict__ idx, int m, int p,
                      double* __restrict__ out, long long ldo) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= m) return;
    const int src = idx[r];
    for (int j = blockIdx.y; j < p; j += gridDim.y) out[(size_t)j * ldo + r] = x[(size_t)j * ldx + src];
}

__global__ void __launch_bounds__(256)
cv_gather_vec_kernel(const double* __restrict__ y, const int* __restrict__ idx, int m, double* __restrict__ out) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < m) out[r] = y[idx[r]];
}

// Held-out squared error of every lambda: part[b][l] = sum over the rows of workgroup b of (y_i - beta0_l - x_i' beta_l)^2.
// One thread per row, kCvChunk lambdas at a time in registers; X is read with unit stride along the rows, the
// coefficients are wave-uniform (scalar loads).  Partials per workgroup, summed on the host in a fixed order.
constexpr int kCvChunk = 8;
__global__ void __launch_bounds__(256)
cv_score_kernel(const double* __restrict__ xt, long long ld, const double* __restrict__ yt, int m, int p,
                const float* __restrict__ beta /* (p + 1) x nlam, intercept first */, int nlam, double* __restrict__ part) {
    __shared__ double red[256 / 64];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < m;
    const int ic = valid ? i : m - 1;
    const double yi = yt[ic];
    for (int l0 = 0; l0 < nlam; l0 += kCvChunk) {
        double acc[kCvChunk];
#pragma unroll
        for (int c = 0; c < kCvChunk; ++c) acc[c] = l0 + c < nlam ? (double)beta[(size_t)(l0 + c) * (p + 1)] : 0.0;
        for (int j = 0; j < p; ++j) {
            const double xv = xt[(size_t)j * ld + ic];
#pragma unroll
            for (int c = 0; c < kCvChunk; ++c) {
                const int l = min(l0 + c, nlam - 1);
                acc[c] = fma(xv, (double)beta[(size_t)l * (p + 1) + 1 + j], acc[c]);
            }
        }
#pragma unroll
        for (int c = 0; c < kCvChunk; ++c) {
            if (l0 + c >= nlam) break;                                   // uniform
            const double e = valid ? yi - acc[c] : 0.0;
            double s = wave_sum(e * e);
            __syncthreads();
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
            __syncthreads();
            if (threadIdx.x == 0) {
                double t = 0;
                for (int w = 0; w < 256 / 64; ++w) t += red[w];
                part[(size_t)blockIdx.x * nlam + l0 + c] = t;
            }
        }
    }
}

void cv_gather(const double* x, long long ldx, const double* y, const int* d_idx, int m, int p, double* xo, double* yo, hipStream_t st) {
    if (m <= 0) return;
    const dim3 grid((m + 255) / 256, (unsigned)std::min(p, 4096));
    hipLaunchKernelGGL(cv_gather_rows_kernel, grid, dim3(256), 0, st, x, ldx, d_idx, m, p, xo, (long long)m);
    hipLaunchKernelGGL(cv_gather_vec_kernel, dim3((m + 255) / 256), dim3(256), 0, st, y, d_idx, m, yo);
    ADMM_HIP_CHECK(hipGetLastError());
}

// sse[l] = sum_i (y_i - beta0_l - x_i' beta_l)^2 over the m held-out rows
std::vector<double> cv_score(const double* xt, const double* yt, int m, int p, const float* beta_host, int nlam, hipStream_t st) {
    std::vector<double> sse(nlam, 0.0);
    if (m <= 0) return sse;
    const int nb = (m + 255) / 256;
    DevBuf<float> db((size_t)(p + 1) * nlam);
    DevBuf<double> part((size_t)nb * nlam);
    ADMM_HIP_CHECK(hipMemcpyAsync(db.get(), beta_host, (size_t)(p + 1) * nlam * sizeof(float), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(cv_score_kernel, dim3(nb), dim3(256), 0, st, xt, (long long)m, yt, m, p, db.get(), nlam, part.get());
    ADMM_HIP_CHECK(hipGetLastError());
    std::vector<double> hp((size_t)nb * nlam);
    ADMM_HIP_CHECK(hipMemcpyAsync(hp.data(), part.get(), hp.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    comm_stream_sync(st);
    for (int l = 0; l < nlam; ++l) {
        double s = 0;
        for (int b = 0; b < nb; ++b) s += hp[(size_t)b * nlam + l];
        sse[l] = s;
    }
    return sse;
}

// ================================================================================================ folds as down-dates
// Tall problems (every training set has more rows than columns): the setup cost of a fold is X_T'X_T, 2 n_T p^2 flop -- K - 1
// passes over the whole matrix for K folds.  Instead: standardise the FULL data once (Z, n x p float; statistics m, s), form
// G_all = Z'Z once, and for fold f form only the held-out block's Gram G_f = Z_f'Z_f (n/K rows): all folds together cost one
// more pass.  The training rows' own standardisation (DataStd on the training rows: mean m_T, scale s_T) is applied to the
// Gram algebraically.  With delta = mean_T(z) = (m_T - m) / s and sigma = s_T / s (per column, from the column sums of Z and
// Z.^2 over the fold, in double):
//      x_std,T = (z - c delta) / sigma          c = 1 if the flag centres (intercept), 0 if not;  sigma = 1 if it does not scale
//      X_T'X_T  = D^-1 (G_all - G_f - c n_T delta delta') D^-1,        D = diag(sigma)
//      X_T'y_T  = D^-1 (Z'w - c delta sum(w)),   w = the training rows' standardised response (the same kernels as a direct
//                 fit: bit-identical), scattered into a length-n vector that is zero on the held-out rows
// What differs from a direct fit of the training rows is rounding only: there the rows are re-standardised in float and the
// Gram is one matrix-core accumulation over n_T rows; here it is the difference of two such accumulations, corrected in double
// and rounded once -- entries agree to a few 1e-7 of the diagonal (tests/test_gpu_cv.py holds the system to 1e-6), which is
// the level at which two float Gram kernels with different summation orders differ anyway.
template <typename T>
__global__ void __launch_bounds__(256)
cv_gather_rows_t_kernel(const T* __restrict__ x, long long ldx, const int* __restrict__ idx, int m, int p, T* __restrict__ out, long long ldo) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= m) return;
    const int src = idx[r];
    for (int j = blockIdx.y; j < p; j += gridDim.y) out[(size_t)j * ldo + r] = x[(size_t)j * ldx + src];
}

// s1[j] = sum_i z_ij, s2[j] = sum_i z_ij^2 in double; one workgroup per column
__global__ void __launch_bounds__(256)
cv_colsum2_kernel(const float* __restrict__ Z, long long ldz, int n, double* __restrict__ s1, double* __restrict__ s2) {
    __shared__ double scratch[8];
    const float* col = Z + (size_t)blockIdx.x * ldz;
    double s[2] = {0.0, 0.0};
    for (int i = threadIdx.x; i < n; i += 256) { const double v = (double)col[i]; s[0] += v; s[1] += v * v; }
    block_sum<double, 2>(s, scratch);
    if (threadIdx.x == 0) { s1[blockIdx.x] = s[0]; s2[blockIdx.x] = s[1]; }
}

__global__ void __launch_bounds__(256)
cv_scatter_vec_kernel(const float* __restrict__ yt, const int* __restrict__ idx, int m, float* __restrict__ w) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < m) w[idx[r]] = yt[r];
}

// M = D^-1 (G_all - G_f - cn delta delta') D^-1 on the p x p block, zero in the padding (ld = round_up(p, 128))
__global__ void __launch_bounds__(256)
cv_downdate_kernel(const float* __restrict__ Gall, const float* __restrict__ Gf, long long ld, int p, double cn,
                   const double* __restrict__ delta, const double* __restrict__ isig, float* __restrict__ M) {
    const int j = blockIdx.y;                                    // column
    const double dj = delta[j < p ? j : 0], sj = isig[j < p ? j : 0];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < ld; i += gridDim.x * 256) {
        float v = 0.0f;
        if (i < p && j < p) {
            const size_t k = (size_t)j * ld + i;
            v = (float)((((double)Gall[k] - (double)Gf[k]) - cn * delta[i] * dj) * isig[i] * sj);
        }
        M[(size_t)j * ld + i] = v;
    }
}

__global__ void cv_xy_fix_kernel(const float* __restrict__ raw, int p, double csw, const double* __restrict__ delta,
                                 const double* __restrict__ isig, float* __restrict__ xy) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < p) xy[j] = (float)(((double)raw[j] - csw * delta[j]) * isig[j]);
}

void cv_downdate_prepare(CvBase& b, const double* xd, const double* yd, int n, int p, bool standardize, bool intercept, hipStream_t st) {
    const double t0 = now_s();
    upload_standardize<float>(b.full, xd, yd, n, p, ADMM_MEM_DEVICE, standardize, intercept, st);
    b.ldp = round_up(p, 128);
    b.Gall.alloc((size_t)b.ldp * b.ldp); b.Gall.zero(st);
    gram_full<float>(b.full.X.get(), b.full.ldx, n, p, true, b.Gall.get(), b.ldp, st);
    DevBuf<double> s((size_t)2 * p);
    hipLaunchKernelGGL(cv_colsum2_kernel, dim3(p), dim3(256), 0, st, b.full.X.get(), b.full.ldx, n, s.get(), s.get() + p);
    ADMM_HIP_CHECK(hipGetLastError());
    b.s1.resize(p); b.s2.resize(p);
    ADMM_HIP_CHECK(hipMemcpyAsync(b.s1.data(), s.get(), (size_t)p * sizeof(double), hipMemcpyDeviceToHost, st));
    ADMM_HIP_CHECK(hipMemcpyAsync(b.s2.data(), s.get() + p, (size_t)p * sizeof(double), hipMemcpyDeviceToHost, st));
    comm_stream_sync(st);
    b.t_prepare = now_s() - t0;
}

// The full-data fit in Gram form: the same Gram kernel and the same X'y product a direct call runs -> bit-identical to it.
void cv_downdate_full(DeviceData<float>& d, const CvBase& b, hipStream_t st) {
    const DeviceData<float>& f = b.full;
    d.n = f.n; d.p = f.p; d.n_total = f.n; d.ldx = 0; d.flag = f.flag;
    d.meanX = f.meanX; d.scaleX = f.scaleX; d.meanY = f.meanY; d.scaleY = f.scaleY;
    d.t_h2d = f.t_h2d; d.t_std = f.t_std;
    d.gram.alloc((size_t)b.ldp * b.ldp);
    ADMM_HIP_CHECK(hipMemcpyAsync(d.gram.get(), b.Gall.get(), (size_t)b.ldp * b.ldp * sizeof(float), hipMemcpyDeviceToDevice, st));
    d.ldgram = b.ldp;
    d.t_gram_tail = b.t_prepare;
    d.xy.alloc(b.ldp); d.xy.zero(st);
    gemv_t_simple<float>(f.X.get(), f.ldx, f.n, f.p, f.Y.get(), d.xy.get(), st);
    comm_stream_sync(st);
}

// Fold data in Gram form.  d_train / d_test: device row indices (ntr / nte of them); yd: the response as handed over.
void cv_downdate_fold(DeviceData<float>& d, const CvBase& b, const double* yd, const int* d_train, int ntr, const int* d_test, int nte, hipStream_t st) {
    const double t0 = now_s();
    const DeviceData<float>& f = b.full;
    const int n = f.n, p = f.p, flag = f.flag;
    const long long ldp = b.ldp;
    const bool centre = (flag & 2) != 0, scale = (flag & 1) != 0;
    // ---- held-out block: rows gathered from Z, its Gram and its column sums
    const long long ldf = round_up(nte, 32);
    DevBuf<float> Zf((size_t)ldf * p);
    Zf.zero(st);
    hipLaunchKernelGGL((cv_gather_rows_t_kernel<float>), dim3((nte + 255) / 256, (unsigned)std::min(p, 4096)), dim3(256), 0, st,
                       f.X.get(), f.ldx, d_test, nte, p, Zf.get(), ldf);
    DevBuf<float> Gf((size_t)ldp * ldp);
    Gf.zero(st);
    gram_full<float>(Zf.get(), ldf, nte, p, true, Gf.get(), ldp, st);
    DevBuf<double> s((size_t)2 * p);
    hipLaunchKernelGGL(cv_colsum2_kernel, dim3(p), dim3(256), 0, st, Zf.get(), ldf, nte, s.get(), s.get() + p);
    std::vector<double> f1(p), f2(p);
    ADMM_HIP_CHECK(hipMemcpyAsync(f1.data(), s.get(), (size_t)p * sizeof(double), hipMemcpyDeviceToHost, st));
    ADMM_HIP_CHECK(hipMemcpyAsync(f2.data(), s.get() + p, (size_t)p * sizeof(double), hipMemcpyDeviceToHost, st));
    // ---- the training rows' response: gathered and standardised as a direct fit does, then scattered to its rows
    DevBuf<double> yt(ntr);
    hipLaunchKernelGGL(cv_gather_vec_kernel, dim3((ntr + 255) / 256), dim3(256), 0, st, yd, d_train, ntr, yt.get());
    const long long ldt = round_up(ntr, 32);
    DevBuf<float> Yt((size_t)ldt);
    float meanY = 0, scaleY = 1;
    standardize_response_f32(yt.get(), ntr, flag, ntr, Yt.get(), ldt, &meanY, &scaleY, st);     // synchronises when flag != 0
    DevBuf<float> w((size_t)f.ldx);
    w.zero(st);
    hipLaunchKernelGGL(cv_scatter_vec_kernel, dim3((ntr + 255) / 256), dim3(256), 0, st, Yt.get(), d_train, ntr, w.get());
    std::vector<float> hY(ntr);
    read_back(hY.data(), Yt.get(), (size_t)ntr * sizeof(float), st);
    DevBuf<float> raw(ldp);
    raw.zero(st);
    gemv_t_simple<float>(f.X.get(), f.ldx, n, p, w.get(), raw.get(), st);
    comm_stream_sync(st);
    double sw = 0;
    for (int i = 0; i < ntr; ++i) sw += (double)hY[i];
    // ---- the training rows' column statistics in the coordinates of Z
    std::vector<double> hd((size_t)2 * p);                        // delta | 1 / sigma
    d.meanX.assign(p, 0.0f); d.scaleX.assign(p, 1.0f);
    for (int j = 0; j < p; ++j) {
        const double t1 = b.s1[j] - f1[j], t2 = b.s2[j] - f2[j];
        const double delta = t1 / ntr;
        const double var = (t2 - (double)ntr * delta * delta) / ntr;           // population variance of z over the training rows
        const double sigma = scale ? std::sqrt(var > 0 ? var : 0.0) : 1.0;
        hd[j] = delta;
        hd[p + j] = 1.0 / sigma;
        if (centre) d.meanX[j] = (float)((double)f.meanX[j] + (double)f.scaleX[j] * delta);
        if (scale) d.scaleX[j] = (float)((double)f.scaleX[j] * sigma);
    }
    DevBuf<double> dd((size_t)2 * p);
    ADMM_HIP_CHECK(hipMemcpyAsync(dd.get(), hd.data(), hd.size() * sizeof(double), hipMemcpyHostToDevice, st));
    d.n = ntr; d.p = p; d.n_total = ntr; d.ldx = 0; d.flag = flag;
    d.meanY = meanY; d.scaleY = scaleY;
    d.gram.alloc((size_t)ldp * ldp);
    d.ldgram = ldp;
    hipLaunchKernelGGL(cv_downdate_kernel, dim3((unsigned)((ldp + 255) / 256), (unsigned)ldp), dim3(256), 0, st, b.Gall.get(), Gf.get(), ldp, p,
                       centre ? (double)ntr : 0.0, dd.get(), dd.get() + p, d.gram.get());
    d.xy.alloc(ldp); d.xy.zero(st);
    hipLaunchKernelGGL(cv_xy_fix_kernel, dim3((p + 255) / 256), dim3(256), 0, st, raw.get(), p, centre ? sw : 0.0, dd.get(), dd.get() + p, d.xy.get());
    ADMM_HIP_CHECK(hipGetLastError());
    comm_stream_sync(st);
    d.t_gram_tail = now_s() - t0;
}

}  // namespace admm
