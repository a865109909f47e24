// Dev tool: time gemv_t (fp64, one right-hand side) on the two stored layouts of the C5 LAD / BP matrices for the plan's knobs
// (columns per wave C, workgroups per CU, segment length, non-temporal loads) against a plain streaming read of the same bytes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I admm_amd/csrc scripts/gemv_sweep64.hip admm_amd/csrc/_obj/*.o -L/opt/rocm/lib -lrccl -o scripts/_bin/gemv_sweep64
#include "gemv_kernels.h"
#include <cstdio>
using namespace admm;

__global__ void __launch_bounds__(256) read_bw_kernel(const double2* __restrict__ a, size_t n2, double* out) {
    double s = 0.0;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    for (; i + 3 * stride < n2; i += 4 * stride) {
        const double2 v0 = load16_nt<double2>(a + i), v1 = load16_nt<double2>(a + i + stride), v2 = load16_nt<double2>(a + i + 2 * stride), v3 = load16_nt<double2>(a + i + 3 * stride);
        s += v0.x + v0.y + v1.x + v1.y + v2.x + v2.y + v3.x + v3.y;
    }
    for (; i < n2; i += stride) { const double2 v = a[i]; s += v.x + v.y; }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0 && s == 123.456) out[0] = s;
}

template <typename F>
double time_ms(F&& f, hipStream_t st, int reps) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    (void)hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i) f();
    (void)hipEventRecord(e1, st);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

template <int C, typename T = double>
void run_variant(int m, int k, long long lda, const T* A, const T* v, T* out, int wg_per_cu, int max_seg, int nt, hipStream_t st) {
    GemvTPlan pl = plan_gemv_t<T>(m, k, 1, C, max_seg, wg_per_cu);
    if (nt >= 0) pl.nt = nt != 0;
    const long long stride = round_up(k, 32);
    double ms = time_ms([&] { launch_gemv_t<T, 1, C>(pl, A, lda, m, k, v, nullptr, out, nullptr, stride, nullptr, st); }, st, 20);
    printf(sizeof(T) == 4 ? "f32 " : "f64 ");
    printf("m=%d k=%d C=%d wgpc=%d maxseg=%d nt=%d | nseg=%d seg=%d gpw=%d grid=%d lds=%zu : %.1f us  %.0f GB/s\n", m, k, C, wg_per_cu, max_seg, (int)pl.nt,
           pl.nseg, pl.seg_len, pl.groups_per_wg, pl.grid, pl.lds_bytes, ms * 1e3, (double)sizeof(T) * m * k / (ms * 1e-3) / 1e9);
}

int main() {
    hipStream_t st; (void)hipStreamCreate(&st);
    const size_t elems = (size_t)5024 * 50000 + 4096;
    DevBuf<double> A(elems), v(65536), out((size_t)64 * 50048), o2(16);
    (void)hipMemset(A.get(), 0, elems * 8); (void)hipMemset(v.get(), 0, 65536 * 8);
    for (int g : {1024, 2048, 4096, 8192}) {
        double ms = time_ms([&] { hipLaunchKernelGGL(read_bw_kernel, dim3(g), dim3(256), 0, st, (const double2*)A.get(), elems / 2, o2.get()); }, st, 20);
        printf("read_bw (nt) grid=%d : %.1f us %.0f GB/s\n", g, ms * 1e3, 8.0 * elems / (ms * 1e-3) / 1e9);
    }
    for (int shape = 0; shape < 2; ++shape) {
        const int m = shape == 0 ? 50000 : 5000, k = shape == 0 ? 5000 : 50000;
        const long long lda = round_up(m, 32);
        for (int wg : {2, 4, 8}) {
            run_variant<4>(m, k, lda, A.get(), v.get(), out.get(), wg, 0, -1, st);
            run_variant<2>(m, k, lda, A.get(), v.get(), out.get(), wg, 0, -1, st);
            run_variant<8>(m, k, lda, A.get(), v.get(), out.get(), wg, 0, -1, st);
        }
        run_variant<4>(m, k, lda, A.get(), v.get(), out.get(), 4, 0, 0, st);
        for (int seg : {1024, 2048, 8192, 16384}) run_variant<4>(m, k, lda, A.get(), v.get(), out.get(), 4, seg, -1, st);
    }
    // the p x p inverse of LAD at C5 (200 MB) -- evicted between two uses by the 4 GB of the two big products: flush with a read first
    for (int rep = 0; rep < 1; ++rep) {
        const int m = 5000, k = 5000; const long long lda = 5024;
        auto flush = [&] { hipLaunchKernelGGL(read_bw_kernel, dim3(4096), dim3(256), 0, st, (const double2*)(A.get() + (size_t)5024 * 5000), (size_t)100000000, o2.get()); };
        const double fl = time_ms(flush, st, 10);
        for (int nt : {0, 1}) for (int seg : {0, 1024, 512}) for (int wg : {4, 8}) {
            GemvTPlan pl = plan_gemv_t<double>(m, k, 1, 4, seg, wg); pl.nt = nt != 0;
            const double ms = time_ms([&] { flush(); launch_gemv_t<double, 1, 4>(pl, A.get(), lda, m, k, v.get(), nullptr, out.get(), nullptr, 5024, nullptr, st); }, st, 10) - fl;
            printf("f64 inverse 5000x5000 after a 1.6 GB flush: nt=%d maxseg=%d wgpc=%d nseg=%d grid=%d : %.1f us %.0f GB/s\n", nt, seg, wg, pl.nseg, pl.grid, ms * 1e3, 8.0 * m * k / (ms * 1e-3) / 1e9);
        }
    }
    // fp32, the consensus solver's blocks at BASELINE configs[3] (8 x 1250 rows of a 10^4 x 10^5 matrix): one 500 MB block, both layouts
    const float* Af = reinterpret_cast<const float*>(A.get());
    const float* vf = reinterpret_cast<const float*>(v.get());
    float* of = reinterpret_cast<float*>(out.get());
    for (int shape = 0; shape < 2; ++shape) {
        const int m = shape == 0 ? 100000 : 1250, k = shape == 0 ? 1250 : 100000;
        const long long lda = round_up(m, 32);
        for (int seg : {0, 2048, 4096, 8192}) run_variant<4, float>(m, k, lda, Af, vf, of, 4, seg, 1, st);
        run_variant<2, float>(m, k, lda, Af, vf, of, 4, 0, 1, st);
        run_variant<8, float>(m, k, lda, Af, vf, of, 4, 0, 1, st);
        run_variant<4, float>(m, k, lda, Af, vf, of, 8, 0, 1, st);
    }
    return 0;
}
