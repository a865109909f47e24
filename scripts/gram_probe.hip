// Dev tool (GPU box): WHERE the fp16 x 2 Gram kernel's time goes.  The tile loop of gram_split_kernel<2> (admm_amd/csrc/gram_bf16x3.hip) with
// phases taken out one at a time, on the C2 launch shape (3160 lower tiles of 128 x 128, 516 K tiles of 16 per launch):
//   MODE 0  the kernel as it is (global loads three tiles ahead, LDS stores, barrier, fragment reads, 12 matrix instructions per wave and step)
//   MODE 1  no global loads (the staged registers keep their first values): LDS stores + barrier + fragment reads + matrix instructions
//   MODE 2  no global loads, no LDS stores, no barrier: fragment reads + matrix instructions
//   MODE 3  matrix instructions only (fragments stay in registers)
//   MODE 4  as MODE 0 with every tile reading the SAME two operand panels (everything hits the L2): the cost of the operand traffic beyond the L2
// Build: hipcc -O3 --offload-arch=gfx950 scripts/gram_probe.hip -o scripts/_bin/gram_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
constexpr int BM = 128, THREADS = 256;
struct G { const uint4* Z[2]; long long ldz; float* C; long long ldc; int M, ntiles, nkt; };
__device__ __forceinline__ void tri(int t, int& bi, int& bj) {
    int b = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while ((long long)(b + 1) * (b + 2) / 2 <= t) ++b;
    while ((long long)b * (b + 1) / 2 > t) --b;
    bi = b; bj = t - b * (b + 1) / 2;
}
__device__ __forceinline__ halfx8 hf(const uint4& u) { halfx8 f; __builtin_memcpy(&f, &u, 16); return f; }
template <int MODE>
__global__ void __launch_bounds__(THREADS, 2) probe(G g) {
    __shared__ uint4 lds[2][2][2][2][BM];
    const int per = (g.ntiles + 7) / 8;
    const int w_idx = (blockIdx.x % 8) * per + blockIdx.x / 8;
    if (w_idx >= g.ntiles) return;
    int bi, bj;
    tri(w_idx, bi, bj);
    if (MODE == 4) { bi = 1; bj = 0; }
    const int I0 = bi * BM, J0 = bj * BM;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wi = (wid >> 1) * 64, wj = (wid & 1) * 64;
    const int s_kg = tid >> 7, s_i = tid & 127;
    const size_t kstep = (size_t)2 * g.ldz;
    const size_t offA = (size_t)s_kg * g.ldz + I0 + s_i, offB = (size_t)s_kg * g.ldz + J0 + s_i;
    floatx16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    uint4 s0a0, s0a1, s0b0, s0b1, s1a0, s1a1, s1b0, s1b1, s2a0, s2a1, s2b0, s2b1;
#define GLOAD(S, t) { const size_t k_ = (size_t)(t) * kstep; S##a0 = g.Z[0][offA + k_]; S##a1 = g.Z[1][offA + k_]; S##b0 = g.Z[0][offB + k_]; S##b1 = g.Z[1][offB + k_]; }
#define LSTORE(S, buf) { lds[buf][0][0][s_kg][s_i] = S##a0; lds[buf][0][1][s_kg][s_i] = S##a1; lds[buf][1][0][s_kg][s_i] = S##b0; lds[buf][1][1][s_kg][s_i] = S##b1; }
    const int fk = lane >> 5, fi = lane & 31;
    uint4 ua[2][2], ub[2][2];
    auto frags = [&](int buf) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) { ua[a][pl] = lds[buf][0][pl][fk][wi + a * 32 + fi]; ub[a][pl] = lds[buf][1][pl][fk][wj + a * 32 + fi]; }
    };
    auto mm = [&]() {
#define TERM(PA, PB) \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hf(ua[0][PA]), hf(ub[0][PB]), acc[0][0], 0, 0, 0); \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hf(ua[0][PA]), hf(ub[1][PB]), acc[0][1], 0, 0, 0); \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hf(ua[1][PA]), hf(ub[0][PB]), acc[1][0], 0, 0, 0); \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hf(ua[1][PA]), hf(ub[1][PB]), acc[1][1], 0, 0, 0);
        TERM(0, 1) TERM(1, 0) TERM(0, 0)
#undef TERM
    };
    const int nk = g.nkt;
    GLOAD(s0, 0) GLOAD(s1, 1) GLOAD(s2, 2)
    LSTORE(s0, 0)
    __syncthreads();
    if (MODE == 3) frags(0);
#define STEP(T, SL, SS, BC, BS) { \
        if (MODE == 0 || MODE == 4) GLOAD(SL, min((T) + 3, nk - 1)) \
        if (MODE != 3) frags(BC); \
        mm(); \
        if (MODE == 3) { ua[0][0].x ^= (unsigned)(T) & 0u; } \
        if (MODE == 0 || MODE == 1 || MODE == 4) { LSTORE(SS, BS) __syncthreads(); } \
    }
    // MODE 5: the LDS stores of tile T + 1 FIRST (they overlap the matrix instructions that follow), then the requests, fragments, products, barrier
#define STEP5(T, SL, SS, BC, BS) { LSTORE(SS, BS) GLOAD(SL, min((T) + 3, nk - 1)) frags(BC); mm(); __syncthreads(); }
    // MODE 6: requests first, fragments, then the stores between the products' first and second term
#define STEP6(T, SL, SS, BC, BS) { GLOAD(SL, min((T) + 3, nk - 1)) frags(BC); LSTORE(SS, BS) mm(); __syncthreads(); }
    // MODE 7: the requests pinned at the start of the step (a scheduling barrier behind them: the compiler sinks them below the products otherwise)
#define STEP7(T, SL, SS, BC, BS) { GLOAD(SL, min((T) + 3, nk - 1)) __builtin_amdgcn_sched_barrier(0); frags(BC); mm(); LSTORE(SS, BS) __syncthreads(); }
    if (MODE == 7) {
        for (int kt = 0; kt < nk; kt += 6) {
            STEP7(kt, s0, s1, 0, 1) STEP7(kt + 1, s1, s2, 1, 0) STEP7(kt + 2, s2, s0, 0, 1)
            STEP7(kt + 3, s0, s1, 1, 0) STEP7(kt + 4, s1, s2, 0, 1) STEP7(kt + 5, s2, s0, 1, 0)
        }
    } else
    if (MODE == 5) {
        for (int kt = 0; kt < nk; kt += 6) {
            STEP5(kt, s0, s1, 0, 1) STEP5(kt + 1, s1, s2, 1, 0) STEP5(kt + 2, s2, s0, 0, 1)
            STEP5(kt + 3, s0, s1, 1, 0) STEP5(kt + 4, s1, s2, 0, 1) STEP5(kt + 5, s2, s0, 1, 0)
        }
    } else if (MODE == 6) {
        for (int kt = 0; kt < nk; kt += 6) {
            STEP6(kt, s0, s1, 0, 1) STEP6(kt + 1, s1, s2, 1, 0) STEP6(kt + 2, s2, s0, 0, 1)
            STEP6(kt + 3, s0, s1, 1, 0) STEP6(kt + 4, s1, s2, 0, 1) STEP6(kt + 5, s2, s0, 1, 0)
        }
    } else
    for (int kt = 0; kt < nk; kt += 6) {
        STEP(kt, s0, s1, 0, 1) STEP(kt + 1, s1, s2, 1, 0) STEP(kt + 2, s2, s0, 0, 1)
        STEP(kt + 3, s0, s1, 1, 0) STEP(kt + 4, s1, s2, 0, 1) STEP(kt + 5, s2, s0, 1, 0)
    }
    float s = 0.f;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    if (s == 12345.678f) g.C[(size_t)blockIdx.x * THREADS + tid] = s;      // keep the work alive
}
// 256 x 256 macro-tile: 8 waves as 4 x 2, a wave owns 64 rows x 128 columns (2 x 4 matrix-core tiles, 24 instructions per step), one workgroup per CU.
// Per K step the workgroup stages 512 operand rows for 4 x the products of a 128 x 128 tile: twice the flops per byte pulled from the L2.
__global__ void __launch_bounds__(512, 1) probe256(G g) {
    __shared__ uint4 lds[2][2][2][2][256];
    const int per = (g.ntiles + 7) / 8;
    const int w_idx = (blockIdx.x % 8) * per + blockIdx.x / 8;
    if (w_idx >= g.ntiles) return;
    int bi, bj;
    tri(w_idx, bi, bj);
    const int I0 = bi * 256, J0 = bj * 256;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wi = (wid >> 1) * 64, wj = (wid & 1) * 128;
    const int s_kg = tid >> 8, s_i = tid & 255;
    const size_t kstep = (size_t)2 * g.ldz;
    const size_t offA = (size_t)s_kg * g.ldz + I0 + s_i, offB = (size_t)s_kg * g.ldz + J0 + s_i;
    floatx16 acc[2][4];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    uint4 s0a0, s0a1, s0b0, s0b1, s1a0, s1a1, s1b0, s1b1, s2a0, s2a1, s2b0, s2b1;
    const int fk = lane >> 5, fi = lane & 31;
    const int nk = g.nkt;
    auto compute = [&](int buf) {
        uint4 ua[2][2], ub[4][2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
            for (int a = 0; a < 2; ++a) ua[a][pl] = lds[buf][0][pl][fk][wi + a * 32 + fi];
#pragma unroll
            for (int b = 0; b < 4; ++b) ub[b][pl] = lds[buf][1][pl][fk][wj + b * 32 + fi];
        }
#define TERM2(PA, PB) \
        _Pragma("unroll") for (int a = 0; a < 2; ++a) _Pragma("unroll") for (int b = 0; b < 4; ++b) \
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hf(ua[a][PA]), hf(ub[b][PB]), acc[a][b], 0, 0, 0);
        TERM2(0, 1) TERM2(1, 0) TERM2(0, 0)
#undef TERM2
    };
    GLOAD(s0, 0) GLOAD(s1, 1) GLOAD(s2, 2)
    LSTORE(s0, 0)
    __syncthreads();
#define STEPB(T, SL, SS, BC, BS) { GLOAD(SL, min((T) + 3, nk - 1)) compute(BC); LSTORE(SS, BS) __syncthreads(); }
    for (int kt = 0; kt < nk; kt += 6) {
        STEPB(kt, s0, s1, 0, 1) STEPB(kt + 1, s1, s2, 1, 0) STEPB(kt + 2, s2, s0, 0, 1)
        STEPB(kt + 3, s0, s1, 1, 0) STEPB(kt + 4, s1, s2, 0, 1) STEPB(kt + 5, s2, s0, 1, 0)
    }
    float s = 0.f;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    if (s == 12345.678f) g.C[(size_t)blockIdx.x * 512 + tid] = s;
}
static void run256(G g, int ntiles, const char* what) {
    g.ntiles = ntiles;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = (g.ntiles + 7) / 8 * 8;
    hipLaunchKernelGGL(probe256, dim3(grid), dim3(512), 0, 0, g);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(probe256, dim3(grid), dim3(512), 0, 0, g);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    const double flop = 3.0 * 2.0 * 256 * 256 * 16.0 * (double)g.nkt * g.ntiles;
    printf("256 x 256, %4d tiles  %-62s %8.3f ms  %6.3f PF/s  (%.2f of 2.5)\n", ntiles, what, ms, flop / ms / 1e12, flop / ms / 1e12 / 2.5);
}
template <int MODE> static void run(const G& g, const char* what) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = (g.ntiles + 7) / 8 * 8;
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(THREADS), 0, 0, g);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(THREADS), 0, 0, g);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    const double flop = 3.0 * 2.0 * 128 * 128 * 16.0 * (double)g.nkt * g.ntiles;
    printf("MODE %d  %-78s %8.3f ms  %6.3f PF/s  (%.2f of 2.5)\n", MODE, what, ms, flop / ms / 1e12, flop / ms / 1e12 / 2.5);
}
int main() {
    const int M = 10000, nkt = 516, nkg = nkt * 2;
    const long long ldz = 10112;
    G g; g.ldz = ldz; g.M = M; g.nkt = nkt; g.ldc = ldz;
    const int nb = (M + BM - 1) / BM; g.ntiles = nb * (nb + 1) / 2;
    uint4* z[2];
    for (int p = 0; p < 2; ++p) {
        hipMalloc(&z[p], (size_t)nkg * ldz * 16);
        hipMemset(z[p], p ? 0x1c : 0x3c, (size_t)nkg * ldz * 16);       // finite fp16 patterns
        if (getenv("PROBE_RANDOM")) {                                    // realistic operand bits (the matrix pipe's power, hence the clock, depends on them)
            const size_t nh = (size_t)nkg * ldz * 8;
            unsigned short* h = (unsigned short*)malloc(nh * 2);
            unsigned x = 12345u + 77u * p;
            for (size_t i = 0; i < nh; ++i) { x = x * 1664525u + 1013904223u; const unsigned m = (x >> 9) & 0x3ff, e = p ? 2 + ((x >> 20) & 3) : 12 + ((x >> 20) & 3); h[i] = (unsigned short)(((x >> 31) << 15) | (e << 10) | m); }
            hipMemcpy(z[p], h, nh * 2, hipMemcpyHostToDevice);
            free(h);
        }
        g.Z[p] = z[p];
    }
    hipMalloc(&g.C, (size_t)4096 * THREADS * 4);
    printf("tiles %d, K tiles %d per launch\n", g.ntiles, nkt);
    run<0>(g, "as the library runs it");
    run<4>(g, "the same, every tile on the same two panels (operands from the L2)");
    run<1>(g, "no global loads: LDS stores + barrier + fragment reads + matrix instructions");
    run<2>(g, "no global loads, LDS stores, barrier: fragment reads + matrix instructions");
    run<3>(g, "matrix instructions only");
    run<5>(g, "full, LDS stores of the next tile at the START of the step");
    run<6>(g, "full, LDS stores after the fragment reads, before the products");
    run<7>(g, "full, requests pinned at the start of the step");
    run256(g, 768, "three full rounds of 256 workgroups");
    run256(g, 820, "the lower triangle of order 10 000 (40 block rows)");
    run256(g, 256, "one round");
    run<0>(g, "as the library runs it (again)");
    return 0;
}
