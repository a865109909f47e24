// Symmetric mat-vec that reads only the lower triangle, for the tall x-update.
//
//   y_r = A v_r  (r = 0, 1)  for a symmetric p x p fp32 matrix stored column-major (both triangles
//   are in memory, only tiles on or below the diagonal are read): 2 p^2 bytes instead of 4 p^2.
//
// Tiling: a workgroup owns 256 rows x a column segment of its row strip (round 3: 32 .. 256 columns, SymvSched below;
// round 2: always 128); each of its 4 waves owns the same 256 rows (one float4 per lane) and a quarter of the columns.
// Measured on C2, same box (round 2, fixed widths): 128-column tiles 35.1 us per launch, 23.2 k ADMM iterations/s;
// 256-column tiles with 64 columns per wave halve the axpy partial rows (tail 6.6 instead of 7.1 us) but the longer
// per-wave column loop streams worse: 37.7 us, 22.0 k it/s; 512-column tiles leave too few workgroups: 51 us.
// Also measured and rejected (round 2): both right-hand sides as pairs through v_pk_fma_f32 plus separate loop copies for
// diagonal / interior / ragged tiles (half the FMA instructions, 32 instead of 74 selects per chunk in the hot copy):
// 36.7 us -- the loop is not VALU-bound, and the scheduling of its 8 loads in flight is what the time depends on.  For every element a_ij (i > j) it loads, a wave
// does both halves of the symmetric product:
//     dot  part:  y_j += a_ij v_i   -> per-lane partials of 8 columns at a time, combined across
//                                      the 64 lanes with a halving butterfly (10 shuffles per 8 columns)
//     axpy part:  y_i += a_ij v_j   -> 4 per-lane accumulators (v_j is wave-uniform, v_readlane)
// The diagonal element contributes once (dot part).  Results are written as partials:
//     dot[rb][j]  (rb = row block of 256)      axp[cb][i]  (cb = column block of 128)
// and summed by the consumer (`symv_sum_partials` below, fused into the tall tail kernel):
//     y_i = sum_{rb >= cb(i)/2} dot[rb][i] + sum_{cb <= 2 rb(i) + 1} axp[cb][i].
// Deterministic (no atomics).  Partial traffic: (p/256 + p/128) * p * 8 bytes per launch, written
// once and read once (about 9 % of the triangle at p = 10^4).
//
// Measured and rejected (scripts/symv_tune.hip, scripts/symv_check.hip, in-situ A/B of the tall loop):
//  * packing the triangle tile by tile (each 128 KB tile contiguous): +-2..6 % depending on p, -1 % in the
//    loop at p = 10^4;
//  * alternating the tile order between launches so that an XCD reads first what it read last (order of the groups of 8
//    tiles reversed on odd launches; a tile keeps its XCD): no change at all (24.10 vs 24.10 k it/s) -- nothing of the
//    matrix survives a kernel boundary in the XCD L2s.  In-kernel timestamps (scripts/tall_probe.py) put the streaming
//    phase itself at 200 MB / 31 us = 6.45 TB/s, the Infinity Cache ceiling; the rest of the 34.5 us is ramp and drain;
//  * 5 or 6 workgroups per CU (launch bounds) so that all 1600 tiles of p = 10^4 are resident at once and end together (the
//    probe shows ~2.6 us between the end of the last tile in the list and the true last end): 96 / 80 registers spill the
//    column chunk -- 47 / 113 us per launch;
//  * two or three tiles per workgroup (half / a third of the workgroups, same tile code): 38.1 / 38.3 instead of 35.6 us
//    per launch -- the second round of small work units is what balances the end of the launch;
//  * fusing the consumer into this launch ("last tile of a block finalises it", arrival counters): correct,
//    but cross-XCD visibility needs either agent-scope fences (whole-L2 write-back per wave: 5x slower) or
//    uncached partial arrays plus an acknowledged-store wait and an atomic round trip per tile (1.65x slower
//    than two launches).
#pragma once
#include "admm_internal.h"
#include "device_utils.h"
#include <hip/hip_ext.h>

namespace admm {

constexpr int kSyRB = 256;     // rows per tile
constexpr int kSyCB = 128;     // default columns per workgroup tile (round 2's fixed tile)
constexpr int kSyCBMax = 256;  // widest column segment a workgroup may own (64 columns per wave: one register of right-hand entries)
constexpr int kSyThreads = 256;

// Round 3: a tile is 256 rows x a column SEGMENT [c0, c0 + width) of its row strip, width a multiple of 32 up to kSyCBMax,
// each of the 4 waves owning width / 4 consecutive columns.  The strips are cut with one of two widths (SymvSched): the
// long strips, dispatched first, in wide segments; the short strips, dispatched last, in narrow ones -- the end of the
// launch is then made of small work units (the integer number of tiles a CU gets no longer leaves some CUs a whole
// 256 x 128 tile short of the others), and the wide segments leave fewer axpy partial rows for the consumer to sum.
// width_big = width_small = 128 reproduces round 2's tiling and partial layout exactly.
struct SymvSched {
    int width_big = kSyCB, width_small = kSyCB;
    int rb_split = 0;              // strips rb >= rb_split are cut in width_big, the others in width_small
    __host__ __device__ int width(int rb) const { return rb >= rb_split ? width_big : width_small; }
    // segments of strip rb: its columns 0 .. min((rb + 1) * 256, p32) - 1 in pieces of width(rb)   (p32 = p rounded up to 32)
    __host__ __device__ int nseg(int rb, int p32) const {
        const int cols = min((rb + 1) * kSyRB, p32), w = width(rb);
        return (cols + w - 1) / w;
    }
};
constexpr int kSySumLanes = 8;  // lanes that share one element when the consumer sums the partials (symv_sum_partials)

struct SymvArgs {
    const float* A; long long lda; int p;
    const float* v0; const float* v1;      // right-hand vectors, allocated (and zero padded) to a multiple of 256
    float* dot0; float* dot1;              // [nrb][ldo]
    float* axp0; float* axp1;              // [ncb][ldo]
    long long ldo;
    const int4* tiles;                     // (row block, first column, width, segment index within the strip) of every tile
    const int* skip;
#ifdef ADMM_HIP_PROBE
    long long* probe; int probe_idx;       // dev build only (probe.h): entry / end stamps of the first, middle and last tile
#endif
};

// Sum 8 per-lane values over the 64 lanes: afterwards every lane l holds the total of value (l >> 3).
template <typename T>
__device__ __forceinline__ T butterfly8(const T (&v)[8], int lane) {
    T a4[4], a2[2];
    {
        const int b = (lane >> 5) & 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const T keep = b ? v[k + 4] : v[k], send = b ? v[k] : v[k + 4];
            a4[k] = keep + __shfl_xor(send, 32, 64);
        }
    }
    {
        const int b = (lane >> 4) & 1;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const T keep = b ? a4[k + 2] : a4[k], send = b ? a4[k] : a4[k + 2];
            a2[k] = keep + __shfl_xor(send, 16, 64);
        }
    }
    const int b = (lane >> 3) & 1;
    const T keep = b ? a2[1] : a2[0], send = b ? a2[0] : a2[1];
    T r = keep + __shfl_xor(send, 8, 64);
    r += __shfl_xor(r, 4, 64);
    r += __shfl_xor(r, 2, 64);
    r += __shfl_xor(r, 1, 64);
    return r;
}

// How a tile reads the right-hand vectors.  Plain: ordinary loads (the vectors were written by an earlier launch).
// Bypass: agent-scope relaxed atomic loads that skip this XCD's L2 -- for vectors written with write-through stores by
// OTHER workgroups of the SAME launch (the single-launch tall iteration, lasso_tall.hip).
struct SymvPlainVec {
    __device__ __forceinline__ float4 load4(const float* p) const { return *reinterpret_cast<const float4*>(p); }
    __device__ __forceinline__ float load1(const float* p) const { return *p; }
};
struct SymvBypassVec {
    __device__ __forceinline__ float4 load4(const float* p) const {
        const unsigned long long lo = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long hi = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return make_float4(__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)), __uint_as_float((unsigned)hi), __uint_as_float((unsigned)(hi >> 32)));
    }
    __device__ __forceinline__ float load1(const float* p) const {
        return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned int*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
};
struct SymvNoWait { __device__ __forceinline__ void operator()() const {} };

// One tile (kSyRB rows x kSyCB columns, 4 waves) of the symmetric product.  The first 8 matrix columns of every wave are
// requested BEFORE `wait()` -- they do not depend on the right-hand vectors -- so that a caller whose vectors are still
// being produced (wait() = poll a flag + barrier) already has its share of the stream in flight.  wait() is called by
// every thread of the workgroup exactly once.
// PRE = false (the two-launch path): no wait, right-hand entries first, every 8-column chunk loaded at the top of its loop
// iteration -- the shape that streams best (with the PRE shape the same 128-column kernel ran at 39.3 instead of 35.1 us on
// C2: a first chunk carried into the loop in registers and a conditional reload defeat the scheduling of the loads).
template <bool PRE, typename Wait, typename VecLoad, bool NT = false>      // NT: matrix read with non-temporal loads (triangle larger than the Infinity Cache)
__device__ __forceinline__ void symv2_tile(const SymvArgs& a, const int4 t, Wait wait, VecLoad vl,
                                           float4 (*red)[kSyThreads], float (*sdot)[kSyCBMax]) {
    const int rb = t.x, seg = t.w;
    const int cw = t.z >> 2;               // columns per wave: a multiple of 8, at most 64
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int row = rb * kSyRB + lane * 4;
    const int col0 = t.y + wid * cw;
    const int p4 = (a.p + 3) & ~3;
    const bool active = row < p4;
    const bool has = col0 < a.p;           // every wave of a listed tile has all its columns <= the block's last row
    const float* base = a.A + (size_t)col0 * a.lda + row;
    float4 av[8];
    if (PRE) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            av[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (has && active && col0 + k < a.p) av[k] = *reinterpret_cast<const float4*>(base + (size_t)k * a.lda);
        }
        wait();
    }
    float4 aU = make_float4(0.f, 0.f, 0.f, 0.f), aW = aU;

    if (has) {
        const float4 uI = active ? vl.load4(a.v0 + row) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 wI = active ? vl.load4(a.v1 + row) : make_float4(0.f, 0.f, 0.f, 0.f);
        const int cj = col0 + lane;                                          // the wave's (<= 64) right-hand entries, one per lane
        const float uj = (lane < cw && cj < a.p) ? vl.load1(a.v0 + cj) : 0.f;
        const float wj = (lane < cw && cj < a.p) ? vl.load1(a.v1 + cj) : 0.f;
        const bool diag = col0 + (cw - 1) >= rb * kSyRB;         // this wave's block meets the diagonal
        const int nq = cw >> 3;
#pragma unroll 1
        for (int q = 0; q < nq; ++q) {
            if (!PRE || q > 0) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int col = col0 + q * 8 + k;
                    av[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if constexpr (NT) { if (active && col < a.p) av[k] = load16_nt<float4>(base + (size_t)(q * 8 + k) * a.lda); }
                    else { if (active && col < a.p) av[k] = *reinterpret_cast<const float4*>(base + (size_t)(q * 8 + k) * a.lda); }
                }
            }
            float dU[8], dW[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int col = col0 + q * 8 + k;
                float4 v = av[k];
                float4 ax = v;                                    // axpy part excludes the diagonal
                if (diag) {
                    if (row + 0 < col) v.x = 0.f;
                    if (row + 1 < col) v.y = 0.f;
                    if (row + 2 < col) v.z = 0.f;
                    if (row + 3 < col) v.w = 0.f;
                    ax = v;
                    if (row + 0 == col) ax.x = 0.f;
                    if (row + 1 == col) ax.y = 0.f;
                    if (row + 2 == col) ax.z = 0.f;
                    if (row + 3 == col) ax.w = 0.f;
                }
                dU[k] = fmaf(v.x, uI.x, fmaf(v.y, uI.y, fmaf(v.z, uI.z, v.w * uI.w)));
                dW[k] = fmaf(v.x, wI.x, fmaf(v.y, wI.y, fmaf(v.z, wI.z, v.w * wI.w)));
                const float ujc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(uj), (q * 8 + k) & 63));
                const float wjc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wj), (q * 8 + k) & 63));
                aU.x = fmaf(ax.x, ujc, aU.x); aU.y = fmaf(ax.y, ujc, aU.y); aU.z = fmaf(ax.z, ujc, aU.z); aU.w = fmaf(ax.w, ujc, aU.w);
                aW.x = fmaf(ax.x, wjc, aW.x); aW.y = fmaf(ax.y, wjc, aW.y); aW.z = fmaf(ax.z, wjc, aW.z); aW.w = fmaf(ax.w, wjc, aW.w);
            }
            // dot part of these 8 columns: every lane ends with column (lane >> 3)
            const float du = butterfly8(dU, lane);
            const float dw = butterfly8(dW, lane);
            if ((lane & 7) == 0) {
                sdot[0][wid * cw + q * 8 + (lane >> 3)] = du;
                sdot[1][wid * cw + q * 8 + (lane >> 3)] = dw;
            }
        }
    } else {
        for (int c = lane; c < cw; c += 64) { sdot[0][wid * cw + c] = 0.f; sdot[1][wid * cw + c] = 0.f; }
    }
    // axpy part: add the 4 waves (same rows, different columns); dot part: one 512-byte row per array
    red[0][threadIdx.x] = aU;
    red[1][threadIdx.x] = aW;
    __syncthreads();
    if (wid < 2) {
        float4 s = red[wid][lane];
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) {
            const float4 o = red[wid][ww * 64 + lane];
            s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
        }
        float* dst = (wid == 0 ? a.axp0 : a.axp1) + (size_t)seg * a.ldo + row;
        *reinterpret_cast<float4*>(dst) = s;
    } else {
        float* dst = (wid == 2 ? a.dot0 : a.dot1) + (size_t)rb * a.ldo + t.y;                          // t.y + width <= (rb + 1) * 256 <= ldo
        for (int c = lane * 4; c < t.z; c += 256) *reinterpret_cast<float4*>(dst + c) = *reinterpret_cast<const float4*>(&sdot[wid - 2][c]);
    }
}

// `Extra`: a functor run by ONE additional workgroup (block 0) concurrently with the tiles; the tall
// solver uses it for its scalar iteration control, which then costs no launch and no latency.
struct SymvNoExtra { __device__ void operator()() const {} };

template <typename Extra, bool NT = false>
__global__ void __launch_bounds__(kSyThreads, 4)      // 4 waves/SIMD: 2 or 4 measure the same, 8 spills; non-temporal loads are 8 % slower (the 2p^2 bytes stay in the Infinity Cache)
symv2_lower_kernel(SymvArgs a, Extra extra) {
    if (blockIdx.x == 0) { extra(); return; }
    if (a.skip != nullptr && *a.skip != 0) return;
    __shared__ float4 red[2][kSyThreads];
    __shared__ __attribute__((aligned(16))) float sdot[2][kSyCBMax];
#ifdef ADMM_HIP_PROBE
    const long long pt0 = wall_clock64();
#endif
    symv2_tile<false, SymvNoWait, SymvPlainVec, NT>(a, a.tiles[blockIdx.x - 1], SymvNoWait(), SymvPlainVec(), red, sdot);
#ifdef ADMM_HIP_PROBE
    if (a.probe != nullptr && threadIdx.x == 0) {
        const int nt = (int)gridDim.x - 1, t = (int)blockIdx.x - 1;
        const int which = t == 0 ? 0 : (t == nt / 2 ? 1 : (t == nt - 1 ? 2 : (t == (3 * nt) / 4 ? 3 : -1)));
        if (which >= 0) {
            long long* d = a.probe + ((size_t)(a.probe_idx & 4095) * 4 + 3) * 8 + which * 2;
            d[0] = pt0; d[1] = wall_clock64();
        }
    }
#endif
}

// Mixed-precision refinement of the tall x-update (opt-in, ADMM_HIP_REFINE=1; lasso_tall.hip): the products M x0, M x1 of the
// float system matrix M = X'X + rho I (lower triangle read, both halves of the symmetric product per loaded element, as
// above) with the two float vectors x0, x1, every product and every sum in DOUBLE -- the residual of a refinement step has
// to be formed more accurately than the solve it corrects.  Same tiles, same partial layout, double partial arrays.  Not
// tuned like the float kernel (a refined iteration streams the triangle three times anyway).
struct SymvArgsD {
    const float* A; long long lda; int p;
    const float* v0; const float* v1;
    double* dot0; double* dot1; double* axp0; double* axp1;
    long long ldo;
    const int4* tiles;
    const int* skip;
};
static __global__ void __launch_bounds__(kSyThreads, 2)
symv2_lower_f64acc_kernel(SymvArgsD a) {
    if (a.skip != nullptr && *a.skip != 0) return;
    __shared__ double4 red[2][kSyThreads];
    __shared__ __attribute__((aligned(16))) double sdot[2][kSyCBMax];
    const int4 t = a.tiles[blockIdx.x];
    const int rb = t.x, seg = t.w, cw = t.z >> 2;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int row = rb * kSyRB + lane * 4;
    const int col0 = t.y + wid * cw;
    const int p4 = (a.p + 3) & ~3;
    const bool active = row < p4, has = col0 < a.p;
    const float* base = a.A + (size_t)col0 * a.lda + row;
    double4 aU = make_double4(0, 0, 0, 0), aW = aU;
    if (has) {
        const float4 uI = active ? *reinterpret_cast<const float4*>(a.v0 + row) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 wI = active ? *reinterpret_cast<const float4*>(a.v1 + row) : make_float4(0.f, 0.f, 0.f, 0.f);
        const int cj = col0 + lane;
        const float uj = (lane < cw && cj < a.p) ? a.v0[cj] : 0.f;
        const float wj = (lane < cw && cj < a.p) ? a.v1[cj] : 0.f;
        const bool diag = col0 + (cw - 1) >= rb * kSyRB;
        for (int q = 0; q < (cw >> 3); ++q) {
            double dU[8], dW[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int col = col0 + q * 8 + k;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (active && col < a.p) v = *reinterpret_cast<const float4*>(base + (size_t)(q * 8 + k) * a.lda);
                float4 ax = v;
                if (diag) {
                    if (row + 0 < col) v.x = 0.f;
                    if (row + 1 < col) v.y = 0.f;
                    if (row + 2 < col) v.z = 0.f;
                    if (row + 3 < col) v.w = 0.f;
                    ax = v;
                    if (row + 0 == col) ax.x = 0.f;
                    if (row + 1 == col) ax.y = 0.f;
                    if (row + 2 == col) ax.z = 0.f;
                    if (row + 3 == col) ax.w = 0.f;
                }
                dU[k] = (double)v.x * uI.x + (double)v.y * uI.y + (double)v.z * uI.z + (double)v.w * uI.w;
                dW[k] = (double)v.x * wI.x + (double)v.y * wI.y + (double)v.z * wI.z + (double)v.w * wI.w;
                const double ujc = (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(uj), (q * 8 + k) & 63));
                const double wjc = (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(wj), (q * 8 + k) & 63));
                aU.x += ax.x * ujc; aU.y += ax.y * ujc; aU.z += ax.z * ujc; aU.w += ax.w * ujc;
                aW.x += ax.x * wjc; aW.y += ax.y * wjc; aW.z += ax.z * wjc; aW.w += ax.w * wjc;
            }
            const double du = butterfly8(dU, lane), dw = butterfly8(dW, lane);
            if ((lane & 7) == 0) { sdot[0][wid * cw + q * 8 + (lane >> 3)] = du; sdot[1][wid * cw + q * 8 + (lane >> 3)] = dw; }
        }
    } else {
        for (int c = lane; c < cw; c += 64) { sdot[0][wid * cw + c] = 0.0; sdot[1][wid * cw + c] = 0.0; }
    }
    red[0][threadIdx.x] = aU;
    red[1][threadIdx.x] = aW;
    __syncthreads();
    if (wid < 2) {
        double4 s = red[wid][lane];
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) { const double4 o = red[wid][ww * 64 + lane]; s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
        double* dst = (wid == 0 ? a.axp0 : a.axp1) + (size_t)seg * a.ldo + row;
        dst[0] = s.x; dst[1] = s.y; dst[2] = s.z; dst[3] = s.w;
    } else {
        double* dst = (wid == 2 ? a.dot0 : a.dot1) + (size_t)rb * a.ldo + t.y;
        for (int c = lane; c < t.z; c += 64) dst[c] = sdot[wid - 2][c];
    }
}

// Number of row blocks, the segment schedule and the tile list (host).
struct SymvPlan {
    int p = 0, nrb = 0, ncb = 0, ntiles = 0, p32 = 0;
    int nax_rows = 0;              // rows of the axpy partial arrays = the largest number of segments of one strip
    SymvSched sched;
    long long ldo = 0;
    DevBuf<int4> tiles;
    std::vector<int4> htiles;      // host copy of this plan's tile list (the distributed setup derives from it which tiles of the inverse a rank reads)
    bool nt = false;
    DevBuf<float> dot0, dot1, axp0, axp1;
#ifdef ADMM_HIP_PROBE
    long long* probe = nullptr; mutable int probe_idx = 0;
#endif
    // part / nparts: this plan launches only the tiles whose 128-column group (index in the full list) % nparts == part -- the row-sharded
    // tall x-update gives every rank an equal share of the triangle; the partial arrays keep the full shape (slots of
    // tiles owned by other ranks stay zero) so that the same consumer code sums them.
    void init(int p_, hipStream_t st, int part = 0, int nparts = 1) {
        p = p_;
        p32 = (p + 31) / 32 * 32;
        nrb = (p + kSyRB - 1) / kSyRB;
        ncb = (p + kSyCB - 1) / kSyCB;
        ldo = (long long)nrb * kSyRB;
        // Segment schedule (measured on one MI355X with scripts/symv_sched_sweep.py, profiles/r03_symv_sched.md; it/s of the
        // whole 100-lambda loop against round 2's fixed 256 x 128 tiles):
        //   * while the triangle is small enough, the narrowest segments that still fit ONE resident round of workgroups
        //     (4 per CU): more workgroups in flight and no second round -- p = 2048 / 3000: 32 columns, +41 % / +32 %;
        //     p = 4096: 64 columns, +14 % (32 columns would need a second round there: -17 %);
        //   * beyond that two classes: wide segments for the long strips, dispatched first, narrow ones (64) for the short
        //     strips at the end of the launch, so that its end is made of small work units -- p = 6000 / 8000: 128 | 64,
        //     +8 % / +7 %; p = 10^4: 192 | 64, +5..7 % (38.8-39.6 instead of 41.7 us per iteration); p = 16000: +8 %.
        // ADMM_HIP_SYMV_SCHED=big,small,split_permille overrides it ("128,128,0" = round 2's tiling).
        auto count = [&](const SymvSched& sc) { long long t = 0; for (int rb = 0; rb < nrb; ++rb) t += sc.nseg(rb, p32); return t; };
        int cus = 256;
        { int dev = 0; hipDeviceProp_t prop; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount; }
        const long long slots = 4ll * cus * std::max(1, nparts);      // a rank of the row-sharded solver launches 1 / nparts of the tiles
        auto uniform = [](int w) { SymvSched sc; sc.width_big = sc.width_small = w; sc.rb_split = 0; return sc; };
        auto two = [&](int wb, int ws) { SymvSched sc; sc.width_big = wb; sc.width_small = ws; sc.rb_split = nrb / 2; return sc; };
        if (4 * count(uniform(32)) <= 3 * slots) sched = uniform(32);
        else if (count(uniform(64)) <= slots) sched = uniform(64);
        else if (10 * count(two(128, 64)) <= 14 * slots) sched = two(128, 64);
        else sched = two(192, 64);
        if (const char* e = option("SYMV_SCHED")) {
            int wb = 0, ws = 0, pm = 0;
            if (std::sscanf(e, "%d,%d,%d", &wb, &ws, &pm) == 3 && wb >= 32 && wb <= kSyCBMax && wb % 32 == 0 && ws >= 32 && ws <= kSyCBMax && ws % 32 == 0 &&
                pm >= 0 && pm <= 1000) {
                sched.width_big = wb; sched.width_small = ws; sched.rb_split = (int)((long long)pm * nrb / 1000);
            }
        }
        std::vector<int4> h;
        nax_rows = 1;
        // long row strips first so that the end of the launch is made of the short strips' (narrow) segments
        for (int rb = nrb - 1; rb >= 0; --rb) {
            const int w = sched.width(rb), ns = sched.nseg(rb, p32);
            nax_rows = std::max(nax_rows, ns);
            for (int sg = 0; sg < ns; ++sg) {
                const int c0 = sg * w;
                const int cols = std::min((rb + 1) * kSyRB, p32);
                h.push_back(make_int4(rb, c0, std::min(w, cols - c0), sg));
            }
        }
        if (nparts > 1) {
            // Dealt out in groups of segments that cover whole 128-column blocks of one strip (segment widths of 32 / 64 / 128: 128
            // columns; 192: 384), not segment by segment: the distributed factorisation (chol_inverse.h) forms, of the inverse,
            // only the 128 x 128 tiles a rank reads here -- interleaved 32-column segments made every rank read every tile
            // (measured: 0.78 p^3 flops per rank of 2 where 0.59 p^3 is the share).
            auto gcd = [](int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; };
            std::vector<int4> mine;
            int group = -1, last_rb = -1, last_cg = -1;
            for (const int4& t : h) {
                const int w = sched.width(t.x);
                const int gw = w / gcd(w, 128) * 128;
                const int cg = t.y / gw;
                if (t.x != last_rb || cg != last_cg) { ++group; last_rb = t.x; last_cg = cg; }
                if (group % nparts == part) mine.push_back(t);
            }
            h.swap(mine);
        }
        ntiles = (int)h.size();
        htiles = h;
        // The whole triangle is re-read every iteration.  Plain loads while most of it stays in the 256 MB Infinity Cache
        // between two passes, non-temporal beyond.  Measured crossover (one MI355X, plain vs nt, TB/s on 2p^2 bytes):
        // p = 10000 5.7 vs 5.3 | 11000 5.90 vs 5.24 | 12000 6.08 vs 5.41 | 13000 4.31 vs 5.50 | 16000 4.1-4.4 vs 5.3-5.4.
        // (a rank of the row-sharded solver re-reads only its 1 / nparts share)
        nt = (size_t)2 * (size_t)p * (size_t)p / (size_t)std::max(1, nparts) > (size_t)310000000;
        if (const char* e = option("SYMV_NT")) nt = std::string(e) == "1";
        tiles.alloc(std::max<size_t>(h.size(), 1));
        if (!h.empty()) ADMM_HIP_CHECK(hipMemcpyAsync(tiles.get(), h.data(), h.size() * sizeof(int4), hipMemcpyHostToDevice, st));
        dot0.alloc((size_t)nrb * ldo); dot1.alloc((size_t)nrb * ldo);
        axp0.alloc((size_t)nax_rows * ldo); axp1.alloc((size_t)nax_rows * ldo);
        dot0.zero(st); dot1.zero(st); axp0.zero(st); axp1.zero(st);
        ADMM_HIP_CHECK(hipStreamSynchronize(st));
    }
    SymvArgs args(const float* A, long long lda, const float* v0, const float* v1, const int* skip) const {
        SymvArgs a;
        a.A = A; a.lda = lda; a.p = p; a.v0 = v0; a.v1 = v1;
        a.dot0 = dot0.get(); a.dot1 = dot1.get(); a.axp0 = axp0.get(); a.axp1 = axp1.get();
        a.ldo = ldo; a.tiles = tiles.get(); a.skip = skip;
#ifdef ADMM_HIP_PROBE
        a.probe = probe; a.probe_idx = probe_idx++;
#endif
        return a;
    }
    template <typename Extra = SymvNoExtra>
    void launch(const float* A, long long lda, const float* v0, const float* v1, const int* skip, hipStream_t st, Extra extra = Extra(),
                hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr) {
        const SymvArgs a = args(A, lda, v0, v1, skip);
        // start/stop events (when given) time exactly this kernel on its stream (hipExtLaunchKernel)
        if (nt) {
            if (ev_start == nullptr && ev_stop == nullptr) hipLaunchKernelGGL((symv2_lower_kernel<Extra, true>), dim3(ntiles + 1), dim3(kSyThreads), 0, st, a, extra);
            else hipExtLaunchKernelGGL((symv2_lower_kernel<Extra, true>), dim3(ntiles + 1), dim3(kSyThreads), 0, st, ev_start, ev_stop, 0, a, extra);
            return;
        }
        if (ev_start == nullptr && ev_stop == nullptr) hipLaunchKernelGGL((symv2_lower_kernel<Extra>), dim3(ntiles + 1), dim3(kSyThreads), 0, st, a, extra);
        else hipExtLaunchKernelGGL((symv2_lower_kernel<Extra>), dim3(ntiles + 1), dim3(kSyThreads), 0, st, ev_start, ev_stop, 0, a, extra);
    }
};

// y_i of both right-hand sides from the partial arrays.  NL lanes (a power of two <= 64, consecutive lanes of one
// wave, `sub` = this lane's index among them) share element i: each issues up to 16 partial loads per array at once
// (one memory round trip up to 16 * NL partials), then the lanes combine with shuffles.  The summation order is
// fixed, so the result is bit-reproducible; every lane of the group returns the total.  Used by the tall tail
// kernel (lasso_tall.hip), by the row-sharded x-update (tall_shard.hip) and by the test hook admm_hip_test_symv.
template <int NL, typename T = float>
__device__ __forceinline__ void symv_sum_partials(const T* __restrict__ dot0, const T* __restrict__ dot1,
                                                  const T* __restrict__ axp0, const T* __restrict__ axp1,
                                                  long long ldo, int nrb, const SymvSched sched, int p32, int i, int sub, bool valid, T& a, T& b) {
    // Branch-free requests: every slot loads from a clamped (always valid) address and out-of-range slots are replaced by
    // zero afterwards, so the 32 loads of a pass are issued back to back.  (The round-2 form guarded each load by two range
    // tests: ~25 instructions of exec-mask bookkeeping per load, ~2 us of issue time in the tall tail kernel before the
    // last request left.)  Offsets fit 32 bits: (nrb + ncb) * ldo < 2^31 up to p ~ 5e5, far beyond what a p x p matrix allows.
    const int ic = valid ? i : 0;
    const int rbi = ic / kSyRB;
    const int rb0 = rbi;                                                // first row strip whose segments reach column ic
    const int ndot = nrb - rb0;
    const int nax = sched.nseg(rbi, p32);                               // segments of row strip rbi (up to its diagonal block)
    const int ntot = valid ? ndot + nax : 0;
    const unsigned ld = (unsigned)ldo;
    a = T(0); b = T(0);
    for (int k0 = 0; k0 < ntot; k0 += 16 * NL) {
        T va[16], vb[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int k = k0 + j * NL + sub;
            const bool isdot = k < ndot;
            const int row = isdot ? rb0 + k : min(k, ntot - 1) - ndot;      // clamped into the axpy rows when k >= ntot
            const unsigned o = (unsigned)row * ld + (unsigned)ic;
            va[j] = (isdot ? dot0 : axp0)[o];
            vb[j] = (isdot ? dot1 : axp1)[o];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const bool in = k0 + j * NL + sub < ntot;
            a += in ? va[j] : T(0); b += in ? vb[j] : T(0);
        }
    }
#pragma unroll
    for (int m = 1; m < NL; m <<= 1) { a += __shfl_xor(a, m, 64); b += __shfl_xor(b, m, 64); }
}

}  // namespace admm
