// Dev tool: time variants of the x-update mat-vec (gemv_t, 2 right-hand sides) on a p x p fp32 matrix.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I admm_amd/csrc scripts/gemv_sweep.hip admm_amd/csrc/prep.hip admm_amd/csrc/lanczos.hip -lrocblas -lrocsolver -o scripts/_bin/gemv_sweep
#include "gemv_kernels.h"
#include <cstdio>
using namespace admm;

// plain streaming read: upper bound for a read-only kernel
__global__ void __launch_bounds__(256) read_bw_kernel(const float4* __restrict__ a, size_t n4, float* out) {
    float s = 0.f;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        float4 v0 = a[i], v1 = a[i + stride], v2 = a[i + 2 * stride], v3 = a[i + 3 * stride];
        s += v0.x + v0.y + v0.z + v0.w + v1.x + v1.y + v1.z + v1.w + v2.x + v2.y + v2.z + v2.w + v3.x + v3.y + v3.z + v3.w;
    }
    for (; i < n4; i += stride) { float4 v = a[i]; s += v.x + v.y + v.z + v.w; }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0 && s == 123.456f) out[0] = s;
}

template <typename F>
double time_ms(F&& f, hipStream_t st, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) f();
    hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

template <int NRHS, int C>
void run_variant(const char* name, int p, long long ldp, const float* M, const float* u, const float* w, float* a, float* b, int wg_per_cu, int max_seg, hipStream_t st) {
    GemvTPlan pl = plan_gemv_t<float>(p, p, NRHS, C, max_seg, wg_per_cu);
    double ms = time_ms([&] { launch_gemv_t<float, NRHS, C>(pl, M, ldp, p, p, u, w, a, b, ldp, nullptr, st); }, st, 50);
    printf("%-28s NRHS=%d C=%d wgpc=%d nseg=%d seg=%d gpw=%d grid=%d lds=%zu : %.2f us  %.0f GB/s (4p^2)\n", name, NRHS, C, wg_per_cu,
           pl.nseg, pl.seg_len, pl.groups_per_wg, pl.grid, pl.lds_bytes, ms * 1e3, 4.0 * p * p / (ms * 1e-3) / 1e9);
}

int main(int argc, char** argv) {
    int p = argc > 1 ? atoi(argv[1]) : 10000;
    long long ldp = round_up(p, 32);
    hipStream_t st; hipStreamCreate(&st);
    DevBuf<float> M((size_t)ldp * p), u(ldp), w(ldp), a(8 * ldp), b(8 * ldp), out(16);
    std::vector<float> h((size_t)ldp * p);
    unsigned s = 1;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 65536.f - 0.5f; }
    hipMemcpy(M.get(), h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(u.get(), h.data(), ldp * 4, hipMemcpyHostToDevice);
    hipMemcpy(w.get(), h.data() + ldp, ldp * 4, hipMemcpyHostToDevice);
    size_t n4 = (size_t)ldp * p / 4;
    for (int g : {1024, 2048, 4096, 8192}) {
        double ms = time_ms([&] { hipLaunchKernelGGL(read_bw_kernel, dim3(g), dim3(256), 0, st, (const float4*)M.get(), n4, out.get()); }, st, 50);
        printf("read_bw grid=%d : %.2f us %.0f GB/s\n", g, ms * 1e3, 4.0 * ldp * p / (ms * 1e-3) / 1e9);
    }
    for (int wg : {2, 4, 8}) {
        run_variant<2, 4>("gemv_t", p, ldp, M.get(), u.get(), w.get(), a.get(), b.get(), wg, 0, st);
        run_variant<2, 8>("gemv_t", p, ldp, M.get(), u.get(), w.get(), a.get(), b.get(), wg, 0, st);
        run_variant<2, 2>("gemv_t", p, ldp, M.get(), u.get(), w.get(), a.get(), b.get(), wg, 0, st);
        run_variant<1, 4>("gemv_t", p, ldp, M.get(), u.get(), w.get(), a.get(), b.get(), wg, 0, st);
    }
    for (int seg : {1024, 2048, 2560, 5120}) run_variant<2, 4>("gemv_t seg", p, ldp, M.get(), u.get(), w.get(), a.get(), b.get(), 4, seg, st);
    return 0;
}
