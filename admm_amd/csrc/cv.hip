// K-fold cross-validation of the Lasso / elastic-net lambda path ("cross-validation folds as
// independent replicas across GPUs").  Not in the reference (R package ADMM has no CV driver): the folds are ordinary
// fits -- the same plan as admm_hip_lasso / admm_hip_enet on the training rows, so each fold's coefficients are
// bit-identical to a user-side call on that row subset -- plus two small kernels: a row gather that builds the training /
// held-out matrices on the device from one resident copy of X, and the held-out squared prediction error for every lambda.
#include "solvers.h"
#include "comm.h"
#include "device_utils.h"

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
    ADMM_HIP_CHECK(hipStreamSynchronize(st));
    for (int l = 0; l < nlam; ++l) {
        double s = 0;
        for (int b = 0; b < nb; ++b) s += hp[(size_t)b * nlam + l];
        sse[l] = s;
    }
    return sse;
}

}  // namespace admm
