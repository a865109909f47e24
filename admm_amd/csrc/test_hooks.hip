// Test hooks behind the C ABI (include/admm_hip.h, "test hooks"): single kernels of the solvers run on caller data so
// that the test-suite can compare them with the oracle / NumPy.  No solver entry point calls anything in this file.
#include "symv_kernels.h"
#include "prep.h"
#include "gather_kernels.h"
#include <vector>

namespace admm {
void require_device();

// The consumer side of the symmetric mat-vec, with the tall tail kernel's geometry and summation order
// (kSySumLanes lanes per element, 256 threads per workgroup).
__global__ void __launch_bounds__(256)
test_symv_finish_kernel(const float* dot0, const float* dot1, const float* axp0, const float* axp1, long long ldo, int nrb, SymvSched sched, int p32,
                        int p, float* y0, float* y1) {
    const int sub = threadIdx.x & (kSySumLanes - 1);
    const int i = blockIdx.x * (256 / kSySumLanes) + threadIdx.x / kSySumLanes;
    float a, b;
    symv_sum_partials<kSySumLanes>(dot0, dot1, axp0, axp1, ldo, nrb, sched, p32, i, sub, i < p, a, b);
    if (i < p && sub == 0) { y0[i] = a; y1[i] = b; }
}

void test_symv(const float* A, int p, const float* v0, const float* v1, float* y0, float* y1) {
    require_device();
    Stream st;
    const long long lda = round_up(p, 128), ldv = round_up(p, 256);      // the tall plan's storage (lasso_tall.hip)
    DevBuf<float> dA((size_t)lda * lda), d0(ldv), d1(ldv), o0(ldv), o1(ldv);
    dA.zero(st.s); d0.zero(st.s); d1.zero(st.s);
    ADMM_HIP_CHECK(hipMemcpy2DAsync(dA.get(), lda * sizeof(float), A, (size_t)p * sizeof(float), (size_t)p * sizeof(float), p,
                                    hipMemcpyHostToDevice, st.s));
    ADMM_HIP_CHECK(hipMemcpyAsync(d0.get(), v0, (size_t)p * sizeof(float), hipMemcpyHostToDevice, st.s));
    ADMM_HIP_CHECK(hipMemcpyAsync(d1.get(), v1, (size_t)p * sizeof(float), hipMemcpyHostToDevice, st.s));
    SymvPlan sy;
    sy.init(p, st.s);
    sy.launch(dA.get(), lda, d0.get(), d1.get(), nullptr, st.s);
    const int per = 256 / kSySumLanes;
    hipLaunchKernelGGL(test_symv_finish_kernel, dim3((p + per - 1) / per), dim3(256), 0, st.s, sy.dot0.get(), sy.dot1.get(),
                       sy.axp0.get(), sy.axp1.get(), sy.ldo, sy.nrb, sy.sched, sy.p32, p, o0.get(), o1.get());
    ADMM_HIP_CHECK(hipGetLastError());
    ADMM_HIP_CHECK(hipMemcpyAsync(y0, o0.get(), (size_t)p * sizeof(float), hipMemcpyDeviceToHost, st.s));
    ADMM_HIP_CHECK(hipMemcpyAsync(y1, o1.get(), (size_t)p * sizeof(float), hipMemcpyDeviceToHost, st.s));
    st.sync();
}

// Gram matrix through the solvers' own path (gram_full: matrix-core SYRK kernels, split-K for small orders):
// G = A'A (atA) or AA' for a host matrix A (rows x cols, column-major, leading dimension rows); G host, order k, ld k.
template <typename T>
void test_gram(const T* A, int rows, int cols, bool atA, T* G) {
    require_device();
    Stream st;
    const long long lda = round_up(rows, 32);
    const int k = atA ? cols : rows;
    const long long ldc = round_up(k, 128);
    DevBuf<T> dA((size_t)lda * cols), dG((size_t)ldc * ldc);
    dA.zero(st.s); dG.zero(st.s);
    ADMM_HIP_CHECK(hipMemcpy2DAsync(dA.get(), lda * sizeof(T), A, (size_t)rows * sizeof(T), (size_t)rows * sizeof(T), cols, hipMemcpyHostToDevice, st.s));
    gram_full<T>(dA.get(), lda, rows, cols, atA, dG.get(), ldc, st.s);
    ADMM_HIP_CHECK(hipMemcpy2DAsync(G, (size_t)k * sizeof(T), dG.get(), ldc * sizeof(T), (size_t)k * sizeof(T), k, hipMemcpyDeviceToHost, st.s));
    st.sync();
}
template void test_gram<float>(const float*, int, int, bool, float*);
template void test_gram<double>(const double*, int, int, bool, double*);

// y = A' v through the solvers' streaming mat-vec (gemv_t_kernel: contiguous columns, 16 bytes per lane, K-split into row
// segments + ordered partial sums): A host, rows x cols column-major (ld rows), v length rows, y length cols.
template <typename T>
void test_gemv_t(const T* A, int rows, int cols, const T* v, T* y) {
    require_device();
    Stream st;
    const long long lda = round_up(rows, 32), ldy = round_up(cols, 32);
    DevBuf<T> dA((size_t)lda * cols), dv(lda), dy(ldy);
    dA.zero(st.s); dv.zero(st.s); dy.zero(st.s);
    ADMM_HIP_CHECK(hipMemcpy2DAsync(dA.get(), lda * sizeof(T), A, (size_t)rows * sizeof(T), (size_t)rows * sizeof(T), cols, hipMemcpyHostToDevice, st.s));
    ADMM_HIP_CHECK(hipMemcpyAsync(dv.get(), v, (size_t)rows * sizeof(T), hipMemcpyHostToDevice, st.s));
    gemv_t_simple<T>(dA.get(), lda, rows, cols, dv.get(), dy.get(), st.s);
    ADMM_HIP_CHECK(hipMemcpyAsync(y, dy.get(), (size_t)cols * sizeof(T), hipMemcpyDeviceToHost, st.s));
    st.sync();
}
template void test_gemv_t<float>(const float*, int, int, const float*, float*);
template void test_gemv_t<double>(const double*, int, int, const double*, double*);

// y = A v over the non-zeros of v through the gather mat-vec of the one-pass forms (gather_kernels.h): A host, rows x cols
// column-major (ld rows), v length cols, y length rows in DOUBLE (the kernel accumulates in double whatever T is); the column
// groups' partial rows are summed in group order, as the consumers do.
template <typename T>
void test_gather(const T* A, int rows, int cols, const T* v, double* y) {
    require_device();
    Stream st;
    const long long lda = round_up(rows, 32);
    DevBuf<T> dA((size_t)lda * cols), dv(round_up(cols, 32));
    dA.zero(st.s); dv.zero(st.s);
    ADMM_HIP_CHECK(hipMemcpy2DAsync(dA.get(), lda * sizeof(T), A, (size_t)rows * sizeof(T), (size_t)rows * sizeof(T), cols, hipMemcpyHostToDevice, st.s));
    ADMM_HIP_CHECK(hipMemcpyAsync(dv.get(), v, (size_t)cols * sizeof(T), hipMemcpyHostToDevice, st.s));
    const GatherPlan gp = plan_gather<T>(rows, cols);
    DevBuf<double> part((size_t)gp.ngroups * gp.pstride);
    part.zero(st.s);
    const GatherArgs<T> a = gather_args<T>(gp, dA.get(), lda, rows, cols, dv.get(), part.get(), nullptr);
    hipLaunchKernelGGL((gather_kernel<T>), dim3(gp.tiles, gp.ngroups), dim3(kGatherThreads), 0, st.s, a);
    std::vector<double> hp((size_t)gp.ngroups * gp.pstride);
    ADMM_HIP_CHECK(hipMemcpyAsync(hp.data(), part.get(), hp.size() * sizeof(double), hipMemcpyDeviceToHost, st.s));
    st.sync();
    ADMM_HIP_CHECK(hipGetLastError());
    for (int i = 0; i < rows; ++i) {
        double s = 0.0;
        for (int g = 0; g < gp.ngroups; ++g) s += hp[(size_t)g * gp.pstride + i];
        y[i] = s;
    }
}
template void test_gather<float>(const float*, int, int, const float*, double*);
template void test_gather<double>(const double*, int, int, const double*, double*);

// Symmetric inverse of an SPD host matrix (order n, ld n) through the solvers' own path: blocked Cholesky + inverse on
// the matrix cores (chol_inverse.h) for n >= 256.  via64: the float matrix factorised / inverted in double and rounded once.
template <typename T>
void test_spd_inverse(const T* A, int n, T* Ainv, bool via64) {
    require_device();
    Stream st;
    const long long lda = round_up(n, 128);
    DevBuf<T> dA((size_t)lda * lda);
    dA.zero(st.s);
    ADMM_HIP_CHECK(hipMemcpy2DAsync(dA.get(), lda * sizeof(T), A, (size_t)n * sizeof(T), (size_t)n * sizeof(T), n, hipMemcpyHostToDevice, st.s));
    if constexpr (std::is_same<T, float>::value) {
        if (via64) spd_inverse_f32_via_f64(dA.get(), lda, n, 0.0, st.s);
        else spd_inverse_f32(dA.get(), lda, n, st.s);
    } else {
        spd_inverse_f64(dA.get(), lda, n, st.s);
    }
    ADMM_HIP_CHECK(hipMemcpy2DAsync(Ainv, (size_t)n * sizeof(T), dA.get(), lda * sizeof(T), (size_t)n * sizeof(T), n, hipMemcpyDeviceToHost, st.s));
    st.sync();
}
template void test_spd_inverse<float>(const float*, int, float*, bool);
template void test_spd_inverse<double>(const double*, int, double*, bool);

}  // namespace admm
