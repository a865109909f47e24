// Bandwidth-bound dense matrix-vector kernels for gfx950 (CDNA4), shared by every solver.
//
// gemv_t: out_r[s][j] = sum_{i in row segment s} A[i + j*lda] * v_r[i]   (r < NRHS)
//   A is column-major; the summation index runs along the contiguous dimension, so one wave
//   streams a column with 16-byte-per-lane loads (1 KiB per wave instruction) and the only
//   partial results are nseg values per output.  The right-hand vectors' segment lives in LDS
//   and is shared by the 4 waves of the workgroup; a wave owns C columns at a time, so each
//   LDS read feeds C*NRHS FMAs and C independent 16-B global loads are in flight per lane per
//   pass.  HBM roofline: lda*k*sizeof(T) bytes read once.
//
// Used for: the tall x-update (A = cached (X'X+rho I)^-1, NRHS = 2: the accelerate and
// restart candidates of the right-hand side, see lasso_tall.hip), X'y, Lanczos SYMVs, the wide solver's X't, PADMM / LAD / BP
// products (on the stored transpose where the reference multiplies by the matrix itself).
#pragma once
#include <cstdlib>
#include "admm_internal.h"
#include "device_utils.h"
#include "gemv_plan.h"
#include <hip/hip_ext.h>

namespace admm {

constexpr int kGemvThreads = 256;          // 4 waves per workgroup
constexpr int kGemvWaves = kGemvThreads / kWave;

template <typename T>
struct GemvTArgs {
    const T* A;
    long long lda;
    int m;              // rows (length of the dot products)
    int k;              // columns (number of outputs)
    const T* v[2];
    int vparts;         // > 1: v[0] is given as `vparts` partial rows (the un-reduced output of another gemv_t),
    long long vstride;  //      summed in order while the segment is staged -- no separate reduction launch
    T* out[2];          // out[r][s * out_stride + j]
    long long out_stride;
    int seg_len;        // rows per segment (multiple of 32 elements: 128-B aligned starts)
    int seg_alloc;      // LDS stride per right-hand segment: seg_len rounded up to a wave pass
    int nseg;
    int groups_per_wg;  // column groups (of C columns) per workgroup
    const int* skip;    // optional device flag: non-zero -> kernel is a no-op
};

// `Extra`: a functor run by ONE additional workgroup (the last block) concurrently with the streaming
// workgroups; the tall solver uses it for its scalar iteration control (no launch, no latency).
struct GemvNoExtra { static constexpr bool kHas = false; __device__ void operator()() const {} };

// NT: the matrix is read with non-temporal loads (GemvTPlan::nt: operands larger than kGemvNtBytes, device_utils.h).
// The work of one workgroup (index bx of the product's own grid): shared by the one-product launch and the batched launch.
template <typename T, int NRHS, int C, bool NT>
__device__ __forceinline__ void gemv_t_body(const GemvTArgs<T>& a, int bx) {
    using VT = Vec16<T>;
    using V = typename VT::type;
    constexpr int VN = VT::N;
    constexpr int PASS = kWave * VN;        // rows covered by one wave pass

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* rhs = reinterpret_cast<T*>(smem_raw);     // [NRHS][seg_alloc]

    const int s = bx % a.nseg;
    const int cb = bx / a.nseg;
    const int r0 = s * a.seg_len;
    const int len = min(a.seg_len, a.m - r0);
    const int lpad = (len + PASS - 1) / PASS * PASS;

    // Stage the right-hand segment(s), zero padded so that padded rows contribute nothing.
#pragma unroll
    for (int r = 0; r < NRHS; ++r) {
        const T* src = a.v[r] + r0;
        T* dst = rhs + (size_t)r * a.seg_alloc;
        if (r == 0 && a.vparts > 1) {
            // The right-hand vector arrives as `vparts` partial rows (the un-reduced output of the previous product).  Four
            // elements per thread at a time, 8 partial rows each: 32 independent loads in flight, every element summed in
            // row order.  (One element at a time was a chain of two dependent round trips per element -- ~7 of the 9.5 us
            // of the consensus solver's 1250 x 1250 product.)
            for (int i0 = threadIdx.x; i0 < lpad; i0 += 4 * kGemvThreads) {
                T acc[4] = {T(0), T(0), T(0), T(0)};
                for (int q0 = 0; q0 < a.vparts; q0 += 8) {
                    T t[4][8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int i = min(i0 + e * kGemvThreads, len - 1);         // clamped: never stored beyond lpad, zeroed beyond len
#pragma unroll
                        for (int u = 0; u < 8; ++u) t[e][u] = src[(size_t)min(q0 + u, a.vparts - 1) * a.vstride + i];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int u = 0; u < 8; ++u) acc[e] += (q0 + u < a.vparts) ? t[e][u] : T(0);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = i0 + e * kGemvThreads;
                    if (i < lpad) dst[i] = (i < len) ? acc[e] : T(0);
                }
            }
        } else {
            for (int i = threadIdx.x; i < lpad; i += kGemvThreads) dst[i] = (i < len) ? src[i] : T(0);
        }
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int ngroups = (a.k + C - 1) / C;
    const int g_end = min((cb + 1) * a.groups_per_wg, ngroups);
    const int len_v = (len + VN - 1) / VN * VN;          // rows touched by vector loads
    const int nfull = len_v / PASS;                      // unmasked passes

    for (int g = cb * a.groups_per_wg + wid; g < g_end; g += kGemvWaves) {
        const int col0 = g * C;
        const T* colp[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int col = min(col0 + c, a.k - 1);      // clamp: the duplicate is never written
            colp[c] = a.A + (size_t)col * a.lda + r0;
        }
        T acc[C][NRHS];
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int r = 0; r < NRHS; ++r) acc[c][r] = T(0);

        int row = lane * VN;
#pragma unroll 2
        for (int it = 0; it < nfull; ++it, row += PASS) {
            V av[C];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if constexpr (NT) av[c] = load16_nt<V>(colp[c] + row);       // `if constexpr`: the plain variant's code is exactly what it was
                else av[c] = *reinterpret_cast<const V*>(colp[c] + row);
            }
            V rv[NRHS];
#pragma unroll
            for (int r = 0; r < NRHS; ++r) rv[r] = *reinterpret_cast<const V*>(rhs + (size_t)r * a.seg_alloc + row);
#pragma unroll
            for (int c = 0; c < C; ++c)
#pragma unroll
                for (int r = 0; r < NRHS; ++r) acc[c][r] = VT::dot(av[c], rv[r], acc[c][r]);
        }
        if (row < len_v) {   // masked tail pass
            V rv[NRHS];
#pragma unroll
            for (int r = 0; r < NRHS; ++r) rv[r] = *reinterpret_cast<const V*>(rhs + (size_t)r * a.seg_alloc + row);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                V av = *reinterpret_cast<const V*>(colp[c] + row);
#pragma unroll
                for (int r = 0; r < NRHS; ++r) acc[c][r] = VT::dot(av, rv[r], acc[c][r]);
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int r = 0; r < NRHS; ++r) acc[c][r] = wave_sum(acc[c][r]);
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if (col0 + c < a.k) {
#pragma unroll
                    for (int r = 0; r < NRHS; ++r) a.out[r][(size_t)s * a.out_stride + col0 + c] = acc[c][r];
                }
            }
        }
    }
}

template <typename T, int NRHS, int C, typename Extra = GemvNoExtra, bool NT = false>
__global__ void __launch_bounds__(kGemvThreads)
gemv_t_kernel(GemvTArgs<T> a, Extra extra) {
    if (Extra::kHas && blockIdx.x == gridDim.x - 1) { extra(); return; }
    if (a.skip != nullptr && *a.skip != 0) return;
    gemv_t_body<T, NRHS, C, NT>(a, (int)blockIdx.x);
}

// Several independent products in ONE launch: blockIdx.y picks the product (its argument block lives in device memory,
// fixed for the life of the solver), blockIdx.x is the workgroup of that product's own grid (the launch is as wide as the
// widest).  Same body, same partial order: bit-identical to separate launches -- what goes away is a fill / drain and a
// kernel boundary per product (the consensus solver with several row blocks on one GPU: 24 launches per iteration -> 3).
template <typename T, int NRHS, int C, bool NT>
__global__ void __launch_bounds__(kGemvThreads)
gemv_t_batch_kernel(const GemvTArgs<T>* __restrict__ batch) {
    const GemvTArgs<T> a = batch[blockIdx.y];
    if (a.skip != nullptr && *a.skip != 0) return;
    const int ngroups = (a.k + C - 1) / C;
    const int grid = a.nseg * ((ngroups + a.groups_per_wg - 1) / a.groups_per_wg);
    if ((int)blockIdx.x >= grid) return;
    gemv_t_body<T, NRHS, C, NT>(a, (int)blockIdx.x);
}

template <typename T>
inline GemvTPlan plan_gemv_t(int m, int k, int nrhs, int C, int max_seg_rows = 0, int wg_per_cu = 4) {
    constexpr int VN = 16 / (int)sizeof(T);
    const int PASS = kWave * VN;
    GemvTPlan pl;
    // LDS budget: at most 160 KiB / wg_per_cu per workgroup for the staged right-hand segments.
    size_t budget = (size_t)(160 * 1024) / (size_t)wg_per_cu - 512;
    int max_rows = (int)(budget / ((size_t)nrhs * sizeof(T)));
    max_rows = max_rows / PASS * PASS;
    if (max_seg_rows > 0) max_rows = std::min(max_rows, std::max(PASS, max_seg_rows / PASS * PASS));
    pl.nseg = (m + max_rows - 1) / max_rows;
    pl.seg_len = round_up((m + pl.nseg - 1) / pl.nseg, 32);      // balanced, 128-B aligned starts
    pl.nseg = (m + pl.seg_len - 1) / pl.seg_len;
    pl.seg_alloc = round_up(pl.seg_len, PASS);
    const int ngroups = (k + C - 1) / C;
    const int target_wg = device_info().num_cu * wg_per_cu;
    int cb = std::max(1, target_wg / pl.nseg);
    int gpw = (ngroups + cb - 1) / cb;
    gpw = std::max(kGemvWaves, round_up(gpw, kGemvWaves));
    pl.groups_per_wg = gpw;
    pl.num_cb = (ngroups + gpw - 1) / gpw;
    pl.grid = pl.num_cb * pl.nseg;
    pl.lds_bytes = (size_t)nrhs * pl.seg_alloc * sizeof(T);
    pl.nt = (size_t)m * (size_t)k * sizeof(T) > kGemvNtBytes;
    return pl;
}

template <typename T, int NRHS, int C, typename Extra = GemvNoExtra>
inline void launch_gemv_t(const GemvTPlan& pl, const T* A, long long lda, int m, int k,
                          const T* v0, const T* v1, T* out0, T* out1, long long out_stride,
                          const int* skip, hipStream_t st, Extra extra = Extra(),
                          hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr, int vparts = 1, long long vstride = 0) {
    GemvTArgs<T> a;
    a.A = A; a.lda = lda; a.m = m; a.k = k;
    a.v[0] = v0; a.v[1] = v1; a.vparts = vparts; a.vstride = vstride; a.out[0] = out0; a.out[1] = out1;
    a.out_stride = out_stride; a.seg_len = pl.seg_len; a.seg_alloc = pl.seg_alloc; a.nseg = pl.nseg;
    a.groups_per_wg = pl.groups_per_wg; a.skip = skip;
    const dim3 grid(pl.grid + (Extra::kHas ? 1 : 0)), block(kGemvThreads);
    const std::uint32_t lds = (std::uint32_t)pl.lds_bytes;
    if (ev_start == nullptr && ev_stop == nullptr) {
        if (pl.nt) hipLaunchKernelGGL((gemv_t_kernel<T, NRHS, C, Extra, true>), grid, block, lds, st, a, extra);
        else hipLaunchKernelGGL((gemv_t_kernel<T, NRHS, C, Extra, false>), grid, block, lds, st, a, extra);
    } else {
        if (pl.nt) hipExtLaunchKernelGGL((gemv_t_kernel<T, NRHS, C, Extra, true>), grid, block, lds, st, ev_start, ev_stop, 0, a, extra);
        else hipExtLaunchKernelGGL((gemv_t_kernel<T, NRHS, C, Extra, false>), grid, block, lds, st, ev_start, ev_stop, 0, a, extra);
    }
}

// Sum the nseg partial rows of a gemv_t result: y[j] = sum_s part[s*stride + j].
template <typename T>
__global__ void reduce_partials_kernel(const T* part, long long stride, int nseg, int k, T* y, const int* skip = nullptr) {
    if (skip != nullptr && *skip != 0) return;
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= k) return;
    T s = 0;
    for (int i = 0; i < nseg; ++i) s += part[(size_t)i * stride + j];
    y[j] = s;
}

// y = A' v for a fixed column-major matrix with preallocated partial buffers (single right-hand side).
template <typename T>
struct GemvT {
    const T* A = nullptr;
    long long lda = 0, stride = 0;
    int m = 0, k = 0;
    GemvTPlan pl;
    DevBuf<T> part, red;                                          // red: the previous product's partial rows summed (run_partials_from, long chains only)
    static constexpr int kChainRows = 12;
    size_t bytes() const { return (size_t)m * (size_t)k * sizeof(T); }
    void set_nt(bool nt) { pl.nt = nt; }                           // streaming policy from the solver's working set (gemv_plan.h)
    void init(const T* A_, long long lda_, int m_, int k_, int wg_per_cu = 4) {
        A = A_; lda = lda_; m = m_; k = k_;
        // fp64 matrices beyond the Infinity Cache (LAD / BP / Dantzig at the C5 shapes): row segments of ~2048 -- 16 KB of LDS per
        // workgroup instead of 36 KB -- measured with scripts/gemv_sweep64.hip on one MI355X (2 GB, non-temporal loads):
        // 50000 x 5000: 312 -> 291 us (6.41 -> 6.88 TB/s); 5000 x 50000: 301 -> 297 us; a plain streaming read of the same bytes: 322 us
        // (with few columns -- Dantzig's 50000 x 2000 -- the extra partial rows cost more than the shorter segments gain: 1770 -> 1700 it/s)
        const bool big64 = sizeof(T) == 8 && (size_t)m * (size_t)k * sizeof(T) > kGemvNtBytes && m > 4096 && k > 4096;
        // (fp32, the consensus solver's 500 MB blocks: segments of 4096 rows take a single 100000 x 1250 product from 83.3 to 77.6 us, but the
        // batched launch of the eight workers' products does not gain: C4 782-790 against 801 it/s -- not applied)
        int seg = big64 ? 2048 : 0;
        if (const char* e = option("GEMV_SEG64")) { if (sizeof(T) == 8) seg = std::atoi(e); }      // A/B knob: 0 = the LDS-budget segments
        pl = plan_gemv_t<T>(m, k, 1, 4, seg, wg_per_cu);
        stride = round_up(k, 32);
        part.alloc((size_t)pl.nseg * stride);
    }
    // partials only: part[s * stride + j]
    void run_partials(const T* v, const int* skip, hipStream_t st) {
        launch_gemv_t<T, 1, 4>(pl, A, lda, m, k, v, nullptr, part.get(), nullptr, stride, skip, st);
    }
    // the same with the right-hand vector taken from the un-reduced partials of `prev` (prev.k == m): a chain of
    // products needs no reduction launches in between
    void run_partials_from(const GemvT<T>& prev, const int* skip, hipStream_t st) {
        const int chain_rows = option_int("GEMV_CHAIN", kChainRows);      // A/B knob
        if (prev.pl.nseg > chain_rows) {
            // many partial rows (the 2048-row segments of a tall fp64 operand leave 25): EVERY workgroup of this product would sum
            // them all for its right-hand segment -- more L2 traffic than the matrix is HBM traffic.  One small launch sums them once,
            // in the same row order (bit-identical), and this product stages a single row.
            if (!red.get()) red.alloc((size_t)round_up(m, 32));
            hipLaunchKernelGGL((reduce_partials_kernel<T>), dim3((m + 255) / 256), dim3(256), 0, st, prev.part.get(), prev.stride, prev.pl.nseg, m, red.get(), skip);
            run_partials(red.get(), skip, st);
            return;
        }
        launch_gemv_t<T, 1, 4>(pl, A, lda, m, k, prev.part.get(), nullptr, part.get(), nullptr, stride, skip, st, GemvNoExtra(),
                               nullptr, nullptr, prev.pl.nseg, prev.stride);
    }
    void run(const T* v, T* y, const int* skip, hipStream_t st) {
        if (pl.nseg == 1) {                                      // one segment: the partial row is the result
            launch_gemv_t<T, 1, 4>(pl, A, lda, m, k, v, nullptr, y, nullptr, stride, skip, st);
            return;
        }
        run_partials(v, skip, st);
        hipLaunchKernelGGL((reduce_partials_kernel<T>), dim3((k + 255) / 256), dim3(256), 0, st, part.get(), stride, pl.nseg, k, y, skip);
    }
    // the argument block of run_partials / run_partials_from, for a batched launch (gemv_t_batch_kernel)
    GemvTArgs<T> args_partials(const T* v, const int* skip, int vparts = 1, long long vstride = 0) {
        GemvTArgs<T> a;
        a.A = A; a.lda = lda; a.m = m; a.k = k;
        a.v[0] = v; a.v[1] = nullptr; a.vparts = vparts; a.vstride = vstride; a.out[0] = part.get(); a.out[1] = nullptr;
        a.out_stride = stride; a.seg_len = pl.seg_len; a.seg_alloc = pl.seg_alloc; a.nseg = pl.nseg;
        a.groups_per_wg = pl.groups_per_wg; a.skip = skip;
        return a;
    }
    GemvTArgs<T> args_partials_from(const GemvT<T>& prev, const int* skip) { return args_partials(prev.part.get(), skip, prev.pl.nseg, prev.stride); }
    // y from the partials of `prev` as right-hand vector
    void run_from(const GemvT<T>& prev, T* y, const int* skip, hipStream_t st) {
        if (pl.nseg == 1) {
            launch_gemv_t<T, 1, 4>(pl, A, lda, m, k, prev.part.get(), nullptr, y, nullptr, stride, skip, st, GemvNoExtra(),
                                   nullptr, nullptr, prev.pl.nseg, prev.stride);
            return;
        }
        run_partials_from(prev, skip, st);
        hipLaunchKernelGGL((reduce_partials_kernel<T>), dim3((k + 255) / 256), dim3(256), 0, st, part.get(), stride, pl.nseg, k, y, skip);
    }
};

// One launch for `count` products whose argument blocks sit in device memory (single right-hand side, C = 4).
template <typename T>
inline void launch_gemv_t_batch(const GemvTArgs<T>* d_batch, int count, int grid_x, size_t lds_bytes, bool nt, hipStream_t st) {
    const dim3 grid(grid_x, count), block(kGemvThreads);
    if (nt) hipLaunchKernelGGL((gemv_t_batch_kernel<T, 1, 4, true>), grid, block, (std::uint32_t)lds_bytes, st, d_batch);
    else hipLaunchKernelGGL((gemv_t_batch_kernel<T, 1, 4, false>), grid, block, (std::uint32_t)lds_bytes, st, d_batch);
}

}  // namespace admm
