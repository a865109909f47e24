// Gather mat-vec for gfx950: out = A v over the NON-ZERO entries of v only (A column-major, columns contiguous).
//
//   part[c][i] = sum_{j in column group c, v_j != 0} v_j * A[i + j*lda]          (accumulated in double)
//
// Why it exists (round 5): the consensus solver's Woodbury workers and basis pursuit both multiply by their matrix AND by its
// transpose every iteration (PADMMLasso.h:23-29, ADMMBP.h:65-66) -- two streaming passes over the largest object in HBM.  In
// both, the first product is algebraically available from n-sized state plus a product with the SPARSE prox output z:
//   consensus:  A_k rhs_k = A_k A_k'b_k - A_k y_k + rho A_k z,   A_k y_k by the recurrence q_k += rho (s_k - A_k z)   (A_k x_k = s_k)
//   BP:         B v = B adj_z - B adj_y / rho,   B y = B adj_y + rho (L^-1 b - B z)                                 (B x = L^-1 b: B B' = I)
// so that only ONE product per iteration streams the matrix; A z costs 4|8 * rows * nnz(z) bytes.  This kernel is that product
// (padmm_lasso.hip / fadmm_dense.hip hold the recurrences and the argument why their errors do not accumulate).
//
// Work decomposition: a workgroup = one row tile (4 waves x 64 lanes x 16 bytes of rows) x one column group.  It scans its
// group's entries of v 256 at a time, compacts the non-zeros by ballot into LDS in column order and adds the listed columns
// for its rows, 8 column requests in flight per lane.  Every sum has a fixed order (columns ascending inside a group, groups
// summed in order by the consumer): bit-reproducible.  A dense v makes it a plain streaming mat-vec (used once at setup for
// A_k (A_k'b_k)); plain loads, so that the few MB of support columns an iteration re-reads can stay in the Infinity Cache.
#pragma once
#include "admm_internal.h"
#include "device_utils.h"

namespace admm {

constexpr int kGatherThreads = 256;
constexpr int kGatherUnroll = 8;

template <typename T>
struct GatherArgs {
    const T* A;
    long long lda;          // multiple of 32 elements; rows [rows, lda) may hold anything (never stored)
    int rows, cols;
    const T* v;             // cols entries
    double* part;           // part[c * pstride + i], c < ngroups, i < rows
    long long pstride;
    int ngroups;
    int cols_per_group;     // multiple of kGatherThreads
    const int* skip;        // optional device flag: non-zero -> no-op
    const int* only_if;     // optional device flag: zero -> no-op (the consensus workers' per-iteration fall-back pass)
};

template <typename T>
__device__ __forceinline__ void gather_body(const GatherArgs<T>& a, int tile, int group) {
    using VT = Vec16<T>;
    using V = typename VT::type;
    constexpr int VN = VT::N;
    constexpr int TILE = kGatherThreads * VN;
    __shared__ int lj[2][kGatherThreads];
    __shared__ T lv[2][kGatherThreads];
    __shared__ int wcnt[2][kGatherThreads / kWave];

    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const long long row = (long long)tile * TILE + (long long)threadIdx.x * VN;
    const bool live = row < a.lda;                       // whole 16-byte pieces: lda is a multiple of 32 elements
    const T* base = a.A + (live ? row : 0);
    const int j0 = group * a.cols_per_group;
    const int j1 = min(j0 + a.cols_per_group, a.cols);
    double acc[VN];
#pragma unroll
    for (int e = 0; e < VN; ++e) acc[e] = 0.0;

    int buf = 0;
    for (int jc = j0; jc < j1; jc += kGatherThreads, buf ^= 1) {
        const int j = jc + (int)threadIdx.x;
        const T vj = j < j1 ? a.v[j] : T(0);
        const bool nz = vj != T(0);
        const unsigned long long bal = __ballot(nz);
        if (lane == 0) wcnt[buf][wid] = __popcll(bal);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kGatherThreads / kWave; ++w) {
            const int c = wcnt[buf][w];
            before += w < wid ? c : 0;
            total += c;
        }
        if (nz) {
            const int slot = before + __popcll(bal & ((1ull << lane) - 1ull));
            lj[buf][slot] = j; lv[buf][slot] = vj;
        }
        __syncthreads();          // (the next chunk writes the OTHER buffer: two barriers per chunk suffice)
        if (!live) continue;
        for (int e0 = 0; e0 < total; e0 += kGatherUnroll) {
            V av[kGatherUnroll];
            T xv[kGatherUnroll];
#pragma unroll
            for (int u = 0; u < kGatherUnroll; ++u) {
                const int e = min(e0 + u, total - 1);                    // clamped duplicate: weighted with zero below
                av[u] = *reinterpret_cast<const V*>(base + (size_t)lj[buf][e] * a.lda);
                xv[u] = e0 + u < total ? lv[buf][e] : T(0);
            }
#pragma unroll
            for (int u = 0; u < kGatherUnroll; ++u) {
                const T* ae = reinterpret_cast<const T*>(&av[u]);
#pragma unroll
                for (int e = 0; e < VN; ++e) acc[e] = fma((double)xv[u], (double)ae[e], acc[e]);
            }
        }
    }
    if (live) {
        double* out = a.part + (size_t)group * a.pstride + row;
#pragma unroll
        for (int e = 0; e < VN; ++e)
            if (row + e < a.rows) out[e] = acc[e];
    }
}

// grid: (row tiles, column groups, products); the argument blocks live in device memory (fixed for the life of the solver)
template <typename T>
__global__ void __launch_bounds__(kGatherThreads)
gather_batch_kernel(const GatherArgs<T>* __restrict__ batch) {
    const GatherArgs<T> a = batch[blockIdx.z];
    if (a.skip != nullptr && load_flag_vector(a.skip) != 0) return;
    if (a.only_if != nullptr && load_flag_vector(a.only_if) == 0) return;
    if ((int)blockIdx.y >= a.ngroups) return;
    constexpr int TILE = kGatherThreads * Vec16<T>::N;
    if ((long long)blockIdx.x * TILE >= a.rows) return;
    gather_body<T>(a, (int)blockIdx.x, (int)blockIdx.y);
}

template <typename T>
__global__ void __launch_bounds__(kGatherThreads)
gather_kernel(GatherArgs<T> a) {
    if (a.skip != nullptr && load_flag_vector(a.skip) != 0) return;
    gather_body<T>(a, (int)blockIdx.x, (int)blockIdx.y);
}

// Geometry: row tiles x column groups.  The number of groups fixes the summation order, so it is a function of the shape only.
struct GatherPlan {
    int tiles = 0, ngroups = 0, cols_per_group = 0;
    long long pstride = 0;
};
template <typename T>
inline GatherPlan plan_gather(int rows, int cols, int products = 1) {
    constexpr int TILE = kGatherThreads * (16 / (int)sizeof(T));
    GatherPlan g;
    g.tiles = (rows + TILE - 1) / TILE;
    // about four workgroups per CU over all products of the launch, 8 ... 64 groups (the consumer sums `ngroups` partial rows)
    int want = (4 * device_info().num_cu + g.tiles * products - 1) / (g.tiles * products);
    if (const char* e = option("GATHER_GROUPS")) { if (std::atoi(e) > 0) want = std::atoi(e); }      // A/B knob
    want = std::max(8, std::min(64, want));
    g.cols_per_group = round_up((cols + want - 1) / want, kGatherThreads);
    g.ngroups = (cols + g.cols_per_group - 1) / g.cols_per_group;
    g.pstride = round_up(rows, 32);
    return g;
}

template <typename T>
inline GatherArgs<T> gather_args(const GatherPlan& g, const T* A, long long lda, int rows, int cols, const T* v, double* part, const int* skip) {
    GatherArgs<T> a;
    a.A = A; a.lda = lda; a.rows = rows; a.cols = cols; a.v = v; a.part = part; a.pstride = g.pstride;
    a.ngroups = g.ngroups; a.cols_per_group = g.cols_per_group; a.skip = skip; a.only_if = nullptr;
    return a;
}

}  // namespace admm
