// NOTE (round 2): written against the round-1 tiling (256 x 128 tiles, 32 columns per wave); the production kernel now uses
// 256 x 256 tiles (symv_kernels.h), so the tile list this tool gets from SymvPlan no longer matches its kernels.  Kept for the record of
// the round-1 measurements quoted in symv_kernels.h.
// Dev tool: timing-only variants of the lower-triangle symv (layout, chunk width, occupancy) over a sweep of p.
// hipcc -O3 --offload-arch=gfx950 -I admm_amd/csrc -I include scripts/symv_tune.hip -o /tmp/symv_tune
#include "symv_kernels.h"
#include <cstdio>
using namespace admm;

struct TArgs { const float* A; long long lda; int p; const float* v0; const float* v1; float* dot0; float* dot1; float* axp0; float* axp1; long long ldo; const int2* tiles; };

// PACKED: tile t occupies 256*128 contiguous floats (column-major inside the tile). CH: columns per chunk (loads in flight).
template <bool PACKED, int CH, int OCC, bool PIPE>
__global__ void __launch_bounds__(256, OCC) symv_var(TArgs a) {
    __shared__ float4 red[2][256];
    const int2 t = a.tiles[blockIdx.x];
    const int rb = t.x, cb = t.y;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int row = rb * 256 + lane * 4;
    const int col0 = cb * 128 + wid * 32;
    const int p4 = (a.p + 3) & ~3;
    const bool active = row < p4;
    float4 aU = make_float4(0.f, 0.f, 0.f, 0.f), aW = aU;
    if (col0 < a.p) {
        const float4 uI = active ? *reinterpret_cast<const float4*>(a.v0 + row) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 wI = active ? *reinterpret_cast<const float4*>(a.v1 + row) : make_float4(0.f, 0.f, 0.f, 0.f);
        const int cj = col0 + (lane & 31);
        const float uj = cj < a.p ? a.v0[cj] : 0.f;
        const float wj = cj < a.p ? a.v1[cj] : 0.f;
        const bool diag = col0 + 31 >= rb * 256;
        const float* base; size_t cs;
        if (PACKED) { base = a.A + (size_t)blockIdx.x * (256 * 128) + (size_t)(wid * 32) * 256 + lane * 4; cs = 256; }
        else { base = a.A + (size_t)col0 * a.lda + row; cs = a.lda; }
        float4 nx[CH];
        if (PIPE) {
#pragma unroll
            for (int k = 0; k < CH; ++k) { nx[k] = make_float4(0.f, 0.f, 0.f, 0.f); if (active && col0 + k < a.p) nx[k] = *reinterpret_cast<const float4*>(base + (size_t)k * cs); }
        }
#pragma unroll 1
        for (int q = 0; q < 32 / CH; ++q) {
            float4 av[CH];
            if (PIPE) {
#pragma unroll
                for (int k = 0; k < CH; ++k) av[k] = nx[k];
                if (q + 1 < 32 / CH) {
#pragma unroll
                    for (int k = 0; k < CH; ++k) { nx[k] = make_float4(0.f, 0.f, 0.f, 0.f); if (active && col0 + (q + 1) * CH + k < a.p) nx[k] = *reinterpret_cast<const float4*>(base + (size_t)((q + 1) * CH + k) * cs); }
                }
            } else {
#pragma unroll
                for (int k = 0; k < CH; ++k) { av[k] = make_float4(0.f, 0.f, 0.f, 0.f); if (active && col0 + q * CH + k < a.p) av[k] = *reinterpret_cast<const float4*>(base + (size_t)(q * CH + k) * cs); }
            }
#pragma unroll
            for (int h = 0; h < CH / 8; ++h) {
                float dU[8], dW[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int kk = h * 8 + k;
                    const int col = col0 + q * CH + kk;
                    float4 v = av[kk]; float4 ax = v;
                    if (diag) {
                        if (row + 0 < col) v.x = 0.f; if (row + 1 < col) v.y = 0.f; if (row + 2 < col) v.z = 0.f; if (row + 3 < col) v.w = 0.f;
                        ax = v;
                        if (row + 0 == col) ax.x = 0.f; if (row + 1 == col) ax.y = 0.f; if (row + 2 == col) ax.z = 0.f; if (row + 3 == col) ax.w = 0.f;
                    }
                    dU[k] = fmaf(v.x, uI.x, fmaf(v.y, uI.y, fmaf(v.z, uI.z, v.w * uI.w)));
                    dW[k] = fmaf(v.x, wI.x, fmaf(v.y, wI.y, fmaf(v.z, wI.z, v.w * wI.w)));
                    const float ujc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(uj), (q * CH + kk) & 31));
                    const float wjc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wj), (q * CH + kk) & 31));
                    aU.x = fmaf(ax.x, ujc, aU.x); aU.y = fmaf(ax.y, ujc, aU.y); aU.z = fmaf(ax.z, ujc, aU.z); aU.w = fmaf(ax.w, ujc, aU.w);
                    aW.x = fmaf(ax.x, wjc, aW.x); aW.y = fmaf(ax.y, wjc, aW.y); aW.z = fmaf(ax.z, wjc, aW.z); aW.w = fmaf(ax.w, wjc, aW.w);
                }
                const float du = butterfly8(dU, lane), dw = butterfly8(dW, lane);
                if ((lane & 7) == 0) {
                    const int col = col0 + q * CH + h * 8 + (lane >> 3);
                    if (col < a.p) { a.dot0[(size_t)rb * a.ldo + col] = du; a.dot1[(size_t)rb * a.ldo + col] = dw; }
                }
            }
        }
    }
    red[0][threadIdx.x] = aU; red[1][threadIdx.x] = aW;
    __syncthreads();
    if (wid < 2) {
        float4 s = red[wid][lane];
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) { const float4 o = red[wid][ww * 64 + lane]; s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
        float* dst = (wid == 0 ? a.axp0 : a.axp1) + (size_t)cb * a.ldo + row;
        *reinterpret_cast<float4*>(dst) = s;
    }
}

template <typename F>
double time_us(F&& f, hipStream_t st, int reps) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) f();
    (void)hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i) f();
    (void)hipEventRecord(e1, st);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1e3;
}

int main(int argc, char** argv) {
    hipStream_t st; (void)hipStreamCreate(&st);
    int ps[] = {6000, 8000, 10000, 12000, 16000};
    for (int p : ps) {
        long long ldp = round_up(p, 128);
        SymvPlan sp; sp.init(p, st);
        size_t nA = (size_t)ldp * ldp; if ((size_t)sp.ntiles * 256 * 128 > nA) nA = (size_t)sp.ntiles * 256 * 128;
        DevBuf<float> M(nA); (void)hipMemsetAsync(M.get(), 0, nA * 4, st);
        DevBuf<float> u(sp.ldo), w(sp.ldo); u.zero(st); w.zero(st);
        TArgs a{M.get(), ldp, p, u.get(), w.get(), sp.dot0.get(), sp.dot1.get(), sp.axp0.get(), sp.axp1.get(), sp.ldo, sp.tiles.get()};
        const double gb = 2.0 * p * p / 1e3;   // bytes / us -> GB/s when divided by us
        printf("p=%d tiles=%d (%.0f MB triangle)\n", p, sp.ntiles, 2.0 * p * p / 1e6);
#define RUN(PK, CH, OCC, PIPE) { double t = time_us([&] { hipLaunchKernelGGL((symv_var<PK, CH, OCC, PIPE>), dim3(sp.ntiles), dim3(256), 0, st, a); }, st, 40); \
        printf("  packed=%d ch=%2d occ=%d pipe=%d : %7.2f us  %6.0f GB/s\n", PK, CH, OCC, PIPE, t, gb / t); }
        RUN(false, 8, 4, false) RUN(true, 8, 4, false)
        RUN(false, 8, 4, true) RUN(true, 8, 4, true)
        RUN(false, 16, 2, false) RUN(true, 16, 2, false)
        RUN(false, 8, 2, false) RUN(true, 8, 2, true)
        (void)hipStreamSynchronize(st);
    }
    return 0;
}
