// Dev tool: what a rank-K update C -= A B' of the blocked factorisation costs as a function of K (128-row tiles, lower triangle or
// full grid), to separate the per-launch fixed part (prologue, C read-modify-write) from the matrix-core part.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I admm_amd/csrc scripts/gemm_shortk.hip <objects except syrk_mfma.o> -L/opt/rocm/lib -lrccl -o scripts/_bin/gemm_shortk
#include "../admm_amd/csrc/syrk_mfma.hip"
#include <cstdio>
using namespace admm;

template <typename F>
static double time_us(F&& f, hipStream_t st, int reps) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    (void)hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i) f();
    (void)hipEventRecord(e1, st);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3 / reps;
}

int main() {
    hipStream_t st; (void)hipStreamCreate(&st);
    const int P = 10112;
    const long long ld = P;
    DevBuf<float> A((size_t)ld * 1024), C((size_t)ld * P);
    A.zero(st); C.zero(st);
    const int epi = 1;      // (round 6: the direct-store epilogue's A/B knob is gone; profiles/r04_gemm_shortk.md holds that comparison)
    for (int M : {1280, 2560, 5120, 7680, 9984}) {
        for (int lower = 1; lower >= 0; --lower) {
            const long long tiles = lower ? (long long)(M / 128) * (M / 128 + 1) / 2 : (long long)(M / 128) * (M / 128);
            for (int K : {16, 128, 256, 1024}) {
                for (float beta : {1.f, 0.f}) {
                    const double us = time_us([&] { launch_gemm_nt(lower != 0, A.get(), ld, A.get(), ld, C.get(), ld, M, M, K, -1.f, beta, false, false, st); }, st, 10);
                    printf("epi=%d M=%5d %s tiles=%5lld K=%4d beta=%g : %8.1f us  %6.1f TF/s  %6.3f us per tile-slot (tiles/256 CUs)\n", epi, M, lower ? "lower" : "full ", tiles, K, beta,
                           us, 2.0 * tiles * 128 * 128 * K / us * 1e-6, us / ((tiles + 255) / 256));
                }
            }
        }
    }
    return 0;
}
