// Dev tool: check symv2_lower_kernel against gemv_t on a random symmetric matrix and time both.
#include "gemv_kernels.h"
#include "symv_kernels.h"
#include <cstdio>
using namespace admm;

__global__ void symv_finish(const float* d0, const float* x0, const float* d1, const float* x1, long long ldo, int nrb, int ncb, int p, float* y0, float* y1) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    float a, b;
    symv_sum_partials<1>(d0, d1, x0, x1, ldo, nrb, ncb, i, 0, i < p, a, b);
    if (i < p) { y0[i] = a; y1[i] = b; }
}

template <typename F>
double time_ms(F&& f, hipStream_t st, int reps) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) f();
    (void)hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i) f();
    (void)hipEventRecord(e1, st);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main(int argc, char** argv) {
    int p = argc > 1 ? atoi(argv[1]) : 10000;
    long long ldp = round_up(p, 32);
    hipStream_t st; (void)hipStreamCreate(&st);
    std::vector<float> h((size_t)ldp * p, 0.f);
    unsigned s = 1;
    for (int j = 0; j < p; ++j)
        for (int i = j; i < p; ++i) {
            s = s * 1664525u + 1013904223u;
            float v = ((s >> 8) & 0xFFFF) / 65536.f - 0.5f;
            h[(size_t)j * ldp + i] = v; h[(size_t)i * ldp + j] = v;
        }
    DevBuf<float> M((size_t)ldp * p);
    (void)hipMemcpy(M.get(), h.data(), h.size() * 4, hipMemcpyHostToDevice);
    SymvPlan sp; sp.init(p, st);
    DevBuf<float> u(sp.ldo), w(sp.ldo), a(8 * ldp), b(8 * ldp), y0(sp.ldo), y1(sp.ldo);
    u.zero(st); w.zero(st);
    std::vector<float> hu(p), hw(p);
    for (int i = 0; i < p; ++i) { s = s * 1664525u + 1013904223u; hu[i] = ((s >> 8) & 0xFFFF) / 65536.f - 0.5f; s = s * 1664525u + 1013904223u; hw[i] = ((s >> 8) & 0xFFFF) / 65536.f; }
    (void)hipMemcpy(u.get(), hu.data(), p * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(w.get(), hw.data(), p * 4, hipMemcpyHostToDevice);
    GemvTPlan pl = plan_gemv_t<float>(p, p, 2, 4);
    launch_gemv_t<float, 2, 4>(pl, M.get(), ldp, p, p, u.get(), w.get(), a.get(), b.get(), ldp, nullptr, st);
    sp.launch(M.get(), ldp, u.get(), w.get(), nullptr, st, SymvNoExtra());
    hipLaunchKernelGGL(symv_finish, dim3((p + 255) / 256), dim3(256), 0, st, sp.dot0.get(), sp.axp0.get(), sp.dot1.get(), sp.axp1.get(), sp.ldo, sp.nrb, sp.ncb, p, y0.get(), y1.get());
    (void)hipStreamSynchronize(st);
    std::vector<float> ha((size_t)pl.nseg * ldp), hb((size_t)pl.nseg * ldp), hy0(p), hy1(p);
    (void)hipMemcpy(ha.data(), a.get(), ha.size() * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hb.data(), b.get(), hb.size() * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hy0.data(), y0.get(), p * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hy1.data(), y1.get(), p * 4, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, n0 = 0, n1 = 0;
    for (int i = 0; i < p; ++i) {
        float ra = 0, rb = 0;
        for (int sg = 0; sg < pl.nseg; ++sg) { ra += ha[(size_t)sg * ldp + i]; rb += hb[(size_t)sg * ldp + i]; }
        e0 = std::max(e0, (double)std::fabs(ra - hy0[i])); e1 = std::max(e1, (double)std::fabs(rb - hy1[i]));
        n0 = std::max(n0, (double)std::fabs(ra)); n1 = std::max(n1, (double)std::fabs(rb));
    }
    printf("p=%d tiles=%d  max|diff| u: %.3e (max %.3e)  w: %.3e (max %.3e)\n", p, sp.ntiles, e0, n0, e1, n1);
    double t_sym = time_ms([&] { sp.launch(M.get(), ldp, u.get(), w.get(), nullptr, st, SymvNoExtra()); }, st, 50);
    double t_fin = time_ms([&] { hipLaunchKernelGGL(symv_finish, dim3((p + 255) / 256), dim3(256), 0, st, sp.dot0.get(), sp.axp0.get(), sp.dot1.get(), sp.axp1.get(), sp.ldo, sp.nrb, sp.ncb, p, y0.get(), y1.get()); }, st, 50);
    double t_gemv = time_ms([&] { launch_gemv_t<float, 2, 4>(pl, M.get(), ldp, p, p, u.get(), w.get(), a.get(), b.get(), ldp, nullptr, st); }, st, 50);
    printf("symv %.2f us (%.0f GB/s of 4p^2, %.0f GB/s of 2p^2)   finish %.2f us   gemv_t %.2f us\n", t_sym * 1e3, 4.0 * p * p / (t_sym * 1e-3) / 1e9,
           2.0 * p * p / (t_sym * 1e-3) / 1e9, t_fin * 1e3, t_gemv * 1e3);
    return 0;
}
