// fp32 matrix-core kernels for the one-time setup of the tall path: the Gram matrix X'X and the
// cached inverse (X'X + rho I)^-1.  These are the only GEMM-shaped pieces of the whole path.
//
// Replaces (reference, all single-threaded Eigen under the default NO_FLOAT_BLAS):
//   Linalg::cross_prod_lower        BlasWrapper.h:73-112   (called from ADMMLassoTall.h:191-192)
//   Eigen::LLT::compute / solve     ADMMLassoTall.h:204-205, :79  (here: factor once, cache the inverse)
//
// One kernel does all the flops: gemm_nt_mfma_kernel computes C = alpha * A B' + beta * C on
// 128 x 128 tiles where BOTH operands are stored with the output index contiguous (A[i, k] at
// i + k lda, B[j, k] at j + k ldb), so for a fixed summation index k the operands of both factors
// are contiguous 512-byte rows.  A workgroup = 4 waves (2 x 2), each wave 64 x 64 = 2 x 2 MFMA
// tiles of v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD = the 157 TF/s fp32 peak; there
// is no xf32/TF32 on gfx950).  K tiles of 16 go global -> registers -> LDS ([k][i] rows of 128
// floats, so the fragment reads lds[(kk + lane/32) * 128 + i0 + lane%32] are bank-conflict free),
// double buffered (a K tile of 32 -- half the barriers per flop, 64 KB of LDS per workgroup -- was measured in round 3:
// Gram setup 127 ms against 117 ms, factorisation 45.7 against 44.2 ms: slower).  Tile ids are remapped so that each XCD (block id % 8) walks a contiguous range
// of tiles and re-uses its operand panels from its own L2.
//
//   Gram:      Z = X' (one transpose pass), C = Z Z', lower tiles + mirrored store (both triangles).
//   Cholesky:  right-looking on 128-blocks: potf2_inv (one workgroup, in LDS: L_kk and L_kk^-1),
//              panel L_ik = A_ik L_kk^-T (in place), trailing A_ij -= L_ik L_jk' (lower tiles).
//   Inverse:   the same block loop eliminates [L | I] right-looking (W_k <- L_kk^-1 W_k, W_i -= L_ik W_k),
//              stored transposed (U = L^-T) so that both updates are NT products over many tiles; then
//              (X'X + rho I)^-1 = U U' (lower tiles + mirror; products start at k = tile row since
//              U is upper triangular).
// No rocBLAS / rocSOLVER on this path (their handle creation alone costs 0.1-0.2 s per process).
#include "prep.h"
#include "chol_inverse.h"
#include "comm.h"

namespace admm {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int SK_BM = 128;      // tile rows / cols
constexpr int SK_BK = 16;       // K tile
constexpr int SK_THREADS = 256;

struct GemmNT {
    const float* A; long long lda;
    const float* B; long long ldb;
    float* C; long long ldc;
    int M, N, K;                 // K multiple of SK_BK; operand rows readable up to the next multiple of 128
    float alpha, beta;
    int nbi, nbj, ntiles;
    int mirror;                  // LOWER mode: also store the transposed tile (both triangles)
    int kstart_row;              // start the K loop at the tile's first row (A, B upper triangular)
    int ksplit;                  // > 1: split s of the K range writes its own partial C + s * cstride (alpha = 1, beta = 0)
    long long cstride;
    const int* tilemap;          // optional [ntiles]: work item -> (bi << 16 | bj); NULL = row-major triangle / column-major grid
    const int* qmap;             // optional [nq]: QUARTER items (qi << 16 | qj, in units of 64 rows / columns), run by the blocks past the
    int nq, grid_full;           //   first grid_full ones (the tail of a deep-K launch, see gram_mfma_f32); LOWER + mirror semantics
};

__device__ __forceinline__ void tri_decode(int t, int& bi, int& bj) {   // t -> (bi, bj), bi >= bj, row-major triangle
    int b = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while ((long long)(b + 1) * (b + 2) / 2 <= t) ++b;
    while ((long long)b * (b + 1) / 2 > t) --b;
    bi = b; bj = t - b * (b + 1) / 2;
}

// A 64 x 64 piece of C (a QUARTER of a tile) by one workgroup: wave w owns the 32 x 32 block (w >> 1, w & 1).  Same K order, same
// instruction, same accumulation per element as the full tile: the values are bit-identical, only the shape of the work item
// differs.  Used for the last, partly filled round of a deep-K launch (gram_mfma_f32): a quarter item is a quarter of the flops
// and the round it runs in is latency-bound (at most one workgroup per CU), so global loads run two K tiles ahead.
__device__ __forceinline__ void gemm_nt_quarter(const GemmNT& g, float (*lds)[2][SK_BK][SK_BM], int qi, int qj) {
    const int I0 = qi * 64, J0 = qj * 64;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wi = (wid >> 1) * 32, wj = (wid & 1) * 32;
    const int s_row = tid >> 4;               // 0..15 (k row)
    const int s_col = (tid & 15) * 4;         // 0..60 (i)
    const float* gA = g.A + (size_t)s_row * g.lda + I0 + s_col;
    const float* gB = g.B + (size_t)s_row * g.ldb + J0 + s_col;
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int kbeg = g.kstart_row ? (max(I0, J0) / SK_BK) * SK_BK : 0;
    const int ntile_k = (g.K - kbeg) / SK_BK;
    const int fk = lane >> 5, fi = lane & 31;
    // tile t travels through register set t & 1 (named variables: arrays indexed through lambdas end up in scratch memory)
    float4 qa0, qb0, qa1, qb1;
#define SK_QLOAD(S, t)                                                                                       \
    {                                                                                                        \
        qa##S = *reinterpret_cast<const float4*>(gA + (size_t)(kbeg + (t) * SK_BK) * g.lda);                 \
        qb##S = *reinterpret_cast<const float4*>(gB + (size_t)(kbeg + (t) * SK_BK) * g.ldb);                 \
    }
#define SK_QSTORE(S, buf)                                                                                    \
    {                                                                                                        \
        *reinterpret_cast<float4*>(&lds[buf][0][s_row][s_col]) = qa##S;                                      \
        *reinterpret_cast<float4*>(&lds[buf][1][s_row][s_col]) = qb##S;                                      \
    }
    auto qcompute = [&](int buf) {                                       // all fragments of the K tile first, then the eight dependent instructions
        float fa[SK_BK / 2], fb[SK_BK / 2];
#pragma unroll
        for (int u = 0; u < SK_BK / 2; ++u) { fa[u] = lds[buf][0][2 * u + fk][wi + fi]; fb[u] = lds[buf][1][2 * u + fk][wj + fi]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < SK_BK / 2; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[u], fb[u], acc, 0, 0, 0);
    };
    if (ntile_k > 0) {
        SK_QLOAD(0, 0)
        if (ntile_k > 1) SK_QLOAD(1, 1)
        SK_QSTORE(0, 0)
        __syncthreads();
        for (int kt = 0; kt < ntile_k; kt += 2) {
            if (kt + 2 < ntile_k) SK_QLOAD(0, kt + 2)
            qcompute(0);
            if (kt + 1 < ntile_k) {
                SK_QSTORE(1, 1)
                __syncthreads();
                if (kt + 3 < ntile_k) SK_QLOAD(1, kt + 3)
                qcompute(1);
                if (kt + 2 < ntile_k) { SK_QSTORE(0, 0) __syncthreads(); }
            }
        }
    }
#undef SK_QLOAD
#undef SK_QSTORE
    const bool offdiag = I0 != J0;
    const int col = J0 + wj + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = I0 + wi + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < g.M && col < g.N) {
            float v = g.alpha * acc[r];
            float* dst = g.C + (size_t)col * g.ldc + row;
            if (g.beta != 0.f) v += g.beta * *dst;
            *dst = v;
            if (g.mirror && offdiag) g.C[(size_t)row * g.ldc + col] = v;
        }
    }
}

// EPI = 1: the output tile leaves through LDS so that C is read and written in whole 512-byte column pieces (launcher: every
// call without the mirrored store).  The matrix cores hand a lane ONE column and rows four at a time, eight rows apart: written
// from the registers, a wave's load or store touches 32 columns with 16 bytes each -- measured (scripts/gemm_shortk.hip) 10 us
// per 128 x 128 tile for the stores and 10 more for the loads of beta C, against 4 us for the K = 128 product itself: the
// rank-128 updates of the blocked factorisation were 5/6 epilogue.  Same arithmetic per element, bit-identical
// (held bit-identical to the direct stores in round 5).
template <int LOWER, int EPI = 0>
__global__ void __launch_bounds__(SK_THREADS, 2)
gemm_nt_mfma_kernel(GemmNT g) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][SK_BK][SK_BM];      // [buffer][A/B][k][i]
    if (LOWER && g.nq > 0 && (int)blockIdx.x >= g.grid_full) {                   // the quarter items of the tail, one XCD range each as below
        const int b = (int)blockIdx.x - g.grid_full;
        const int perq = (g.nq + 7) / 8;
        const int q = (b % 8) * perq + b / 8;
        if (q >= g.nq) return;
        const int m = g.qmap[q];
        gemm_nt_quarter(g, lds, m >> 16, m & 0xffff);
        return;
    }
    // XCD-aware remap: block b runs on XCD b % 8; give every XCD a contiguous range of tiles
    const int nwork = g.ntiles * g.ksplit;
    const int per = (nwork + 7) / 8;
    // (kstart_row -- the inverse U U': a tile's K range shrinks with its row, from pp down to 128, and row-major ranges gave the
    // first XCD ten times the work of the last: 6.5 ms at p = 10^4 where the flops need 2.5.  Tiles in launch order instead:
    // longest first, consecutive tiles on different XCDs.)
    const int w_idx = g.kstart_row ? (int)blockIdx.x : (blockIdx.x % 8) * per + blockIdx.x / 8;
    if (w_idx >= nwork) return;
    const int t_idx = w_idx % g.ntiles, split = w_idx / g.ntiles;
    int bi, bj;
    if (g.tilemap != nullptr) { const int m = g.tilemap[t_idx]; bi = m >> 16; bj = m & 0xffff; }
    else if (LOWER) tri_decode(t_idx, bi, bj);
    else { bi = t_idx % g.nbi; bj = t_idx / g.nbi; }
    const int I0 = bi * SK_BM, J0 = bj * SK_BM;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wi = (wid >> 1) * 64, wj = (wid & 1) * 64;      // wave's 64 x 64 sub-tile

    // staging map: a K tile of one operand is 16 rows x 128 floats = 512 float4; 2 per thread
    const int s_row0 = tid >> 5;              // 0..7   (k row), second load: +8
    const int s_col = (tid & 31) * 4;         // 0..124 (i)
    const float* gA = g.A + (size_t)s_row0 * g.lda + I0 + s_col;
    const float* gB = g.B + (size_t)s_row0 * g.ldb + J0 + s_col;
    const size_t a8 = (size_t)8 * g.lda, b8 = (size_t)8 * g.ldb;

    floatx16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // Global loads run one K tile ahead of the matrix cores: tile t travels through register set t & 1 and is written to the
    // other LDS buffer before the barrier.  Round 4 measured the alternatives on the C2 Gram (one round of 1024 tiles, K = 10^5,
    // 25.1-25.8 ms as is; the matrix pipe alone needs 21.7 ms at the 154.9 TF/s scripts/mfma_peak.hip sustains on the box):
    // loads two tiles ahead 25.8; global -> LDS directly (global_load_lds_dwordx4, no staging registers, no ds_write pass)
    // 27.2; the fragment reads of k step kk + 2 ahead of the instructions of step kk (sched_barrier) 25.5-25.9; LDS reads and
    // matrix instructions alone, no loads / writes / barriers, 23.0; K = 12 500 costs the same per K tile as K = 10^5 (so the
    // 51-fold over-fetch -- tiles of a square drifting apart in K -- is not what the time goes to).  None of them pays: kept as is.
    // (named registers and macros: register arrays or structs handed to lambdas by reference end up in scratch memory)
    float4 s0a0, s0a1, s0b0, s0b1, s1a0, s1a1, s1b0, s1b1;
    const int kchunk = g.K / g.ksplit;                                  // multiple of SK_BK (launcher)
    const int kbeg = g.kstart_row ? (max(I0, J0) / SK_BK) * SK_BK : split * kchunk;
    const int ntile_k = ((g.ksplit > 1 ? kbeg + kchunk : g.K) - kbeg) / SK_BK;
    const int fk = lane >> 5, fi = lane & 31;
#define SK_GLOAD(S, t)                                                                   \
    {                                                                                    \
        const size_t k0_ = (size_t)(kbeg + (t) * SK_BK);                                 \
        S##a0 = *reinterpret_cast<const float4*>(gA + k0_ * g.lda);                      \
        S##a1 = *reinterpret_cast<const float4*>(gA + k0_ * g.lda + a8);                 \
        S##b0 = *reinterpret_cast<const float4*>(gB + k0_ * g.ldb);                      \
        S##b1 = *reinterpret_cast<const float4*>(gB + k0_ * g.ldb + b8);                 \
    }
#define SK_LSTORE(S, buf)                                                                \
    {                                                                                    \
        *reinterpret_cast<float4*>(&lds[buf][0][s_row0][s_col]) = S##a0;                 \
        *reinterpret_cast<float4*>(&lds[buf][0][s_row0 + 8][s_col]) = S##a1;            \
        *reinterpret_cast<float4*>(&lds[buf][1][s_row0][s_col]) = S##b0;                 \
        *reinterpret_cast<float4*>(&lds[buf][1][s_row0 + 8][s_col]) = S##b1;            \
    }
    auto compute = [&](int buf) {
#pragma unroll
        for (int kk = 0; kk < SK_BK; kk += 2) {
            const float a0 = lds[buf][0][kk + fk][wi + fi];
            const float a1 = lds[buf][0][kk + fk][wi + 32 + fi];
            const float b0 = lds[buf][1][kk + fk][wj + fi];
            const float b1 = lds[buf][1][kk + fk][wj + 32 + fi];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    };
    if (ntile_k > 0) {
        SK_GLOAD(s0, 0)
        SK_LSTORE(s0, 0)
        __syncthreads();
        for (int kt = 0; kt < ntile_k; kt += 2) {                       // two K tiles per trip: even tiles through s0 / buffer 0, odd ones through s1 / buffer 1
            if (kt + 1 < ntile_k) SK_GLOAD(s1, kt + 1)                  // next tile in flight while this one computes
            compute(0);
            if (kt + 1 < ntile_k) {
                SK_LSTORE(s1, 1)
                __syncthreads();
                if (kt + 2 < ntile_k) SK_GLOAD(s0, kt + 2)
                compute(1);
                if (kt + 2 < ntile_k) { SK_LSTORE(s0, 0) __syncthreads(); }
            }
        }
    }
#undef SK_GLOAD
#undef SK_LSTORE

    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    if (EPI) {
        // Two passes of 64 columns x 128 rows through the 32 KB of the staging buffers: pass b takes the b-th 32-column block of
        // every wave.  T[c][row], rows contiguous, float4 slot q = row / 4 stored at q ^ (c & 31): the writers (8 lanes of
        // different columns per LDS cycle) and the readers (32 lanes walking down one column) are both conflict free.
        float* T = &lds[0][0][0][0];
        float* Cs = g.C + (size_t)split * g.cstride;
        const bool vec = (g.ldc & 3) == 0 && (reinterpret_cast<size_t>(Cs) & 15) == 0;
        const int cw = (wid & 1) * 32 + (lane & 31);
        __syncthreads();                                                // the last K tile has been read by every wave
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if (b) __syncthreads();                                     // pass 0 has been read
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int q = (wi + a * 32 + 8 * rq + 4 * (lane >> 5)) >> 2;
                    *reinterpret_cast<float4*>(&T[cw * SK_BM + ((q ^ (cw & 31)) << 2)]) =
                        make_float4(acc[a][b][4 * rq], acc[a][b][4 * rq + 1], acc[a][b][4 * rq + 2], acc[a][b][4 * rq + 3]);
                }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int idx = i * SK_THREADS + tid;
                const int c = idx >> 5, q = idx & 31;
                const float4 t = *reinterpret_cast<const float4*>(&T[c * SK_BM + ((q ^ (c & 31)) << 2)]);
                const int col = J0 + (c >> 5) * 64 + b * 32 + (c & 31), row = I0 + q * 4;
                if (col >= g.N || row >= g.M) continue;
                float* dst = Cs + (size_t)col * g.ldc + row;
                const float tv[4] = {t.x, t.y, t.z, t.w};
                if (vec && row + 3 < g.M) {
                    float o[4];
                    float4 c4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (g.beta != 0.f) c4 = *reinterpret_cast<const float4*>(dst);
                    const float cv[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = g.alpha * tv[e];
                        if (g.beta != 0.f) v += g.beta * cv[e];
                        o[e] = v;
                    }
                    *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (row + e < g.M) {
                            float v = g.alpha * tv[e];
                            if (g.beta != 0.f) v += g.beta * dst[e];
                            dst[e] = v;
                        }
                }
            }
        }
        return;
    }
    const bool offdiag = I0 != J0;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = J0 + wj + b * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = I0 + wi + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < g.M && col < g.N) {
                    float v = g.alpha * acc[a][b][r];
                    float* Cs = g.C + (size_t)split * g.cstride;
                    float* dst = Cs + (size_t)col * g.ldc + row;
                    if (g.beta != 0.f) v += g.beta * *dst;
                    *dst = v;
                    if (LOWER && g.mirror && offdiag) Cs[(size_t)row * g.ldc + col] = v;     // mirrored tile: both triangles
                }
            }
        }
}

// Work order of a lower-triangle launch with a deep K (the Gram): the 64 tiles that are resident on one XCD at a time
// (32 CUs x 2 workgroups) should form a SQUARE of the tile grid, not a row -- they walk K roughly in step, so an 8 x 8
// square streams 8 + 8 operand panels through that XCD's L2 for 64 tiles (each loaded line feeds 8 tiles), while 64
// consecutive tiles of one row stream 1 + 64 panels (the B panels feed one tile each).  Block b runs on XCD b % 8 and is
// the (b / 8)-th work item of that XCD; super-blocks of 8 x 8 tiles (triangular on the diagonal) are dealt out to the
// XCDs in turn.  Returns a device array of ntiles entries (bi << 16 | bj) indexed like w_idx in the kernel.
static std::vector<std::vector<int>> square_tile_lists(int nb) {
    const int ntiles = nb * (nb + 1) / 2;
    const int per = (ntiles + 7) / 8;
    std::vector<std::vector<int>> q(8);
    const int S = 8;
    int turn = 0;
    for (int sbi = 0; sbi * S < nb; ++sbi)
        for (int sbj = 0; sbj <= sbi; ++sbj) {
            // give the super-block to the XCD with the shortest list so far (keeps the 8 lists within one super-block of each other)
            int x = turn;
            for (int k = 0; k < 8; ++k) if (q[k].size() < q[x].size()) x = k;
            turn = (turn + 1) & 7;
            for (int bi = sbi * S; bi < std::min(nb, (sbi + 1) * S); ++bi)
                for (int bj = sbj * S; bj < std::min(nb, (sbj + 1) * S); ++bj)
                    if (bj <= bi) q[x].push_back(bi << 16 | bj);
        }
    // the kernel gives XCD x the index range [x * per, (x + 1) * per): rebalance so that no list exceeds `per`
    std::vector<int> flat;
    for (int x = 0; x < 8; ++x) while ((int)q[x].size() > per) { flat.push_back(q[x].back()); q[x].pop_back(); }
    for (int x = 0; x < 8; ++x) while ((int)q[x].size() < per && !flat.empty()) { q[x].push_back(flat.back()); flat.pop_back(); }
    return q;
}

static DevBuf<int> upload_ints(const std::vector<int>& v, hipStream_t st) {
    DevBuf<int> d(std::max<size_t>(v.size(), 1));
    if (!v.empty()) ADMM_HIP_CHECK(hipMemcpyAsync(d.get(), v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice, st));
    comm_stream_sync(st);
    return d;
}

static DevBuf<int> make_square_tilemap(int nb, hipStream_t st) {
    // The kernel gives XCD x the work items [x * per, (x + 1) * per) of tilemap: lay the 8 lists out back to back (a
    // permutation of all tiles; where a list is a little shorter than `per` a few tiles slide to the neighbouring XCD).
    std::vector<int> order;
    for (const auto& l : square_tile_lists(nb)) for (int v : l) order.push_back(v);
    return upload_ints(order, st);
}

// Work lists of a deep-K lower-triangle launch (the Gram) that ends without a straggling round.  All tiles cost the same and
// `slots` workgroups are resident at a time (4 per CU: 128 VGPRs, 32 KB of LDS), so T tiles take ceil(T / slots) rounds: at C2 (p = 10^4:
// 3160 tiles, 1024 slots) the fourth round runs 88 tiles on 1024 slots -- 3.09 rounds of work in the time of 4 (84 ms where 3.09
// rounds of 21.3 ms are 66), and 79 of those tiles are the ragged
// last block row (16 of their 128 rows are real).  Here: whole rounds of full tiles, and the remainder -- the ragged block row when
// at most 64 of its rows are real, plus the tiles of a last round that would be less than `tail_max` full -- cut into QUARTER items
// (64 x 64, one workgroup each: same values bit for bit, gemm_nt_quarter), dispatched after the full tiles: one short round in
// which every CU holds at most a workgroup or two.  `full`: the tile lists of the 8 XCDs back to back; `quarters`: qi << 16 | qj.
static void gram_work_lists(int M, int slots, double tail_max, std::vector<int>& full, std::vector<int>& quarters) {
    const int nb = (M + SK_BM - 1) / SK_BM;
    const int ragged = M - (nb - 1) * SK_BM;                           // real rows of the last block row (1 .. 128)
    const int nbf = (ragged <= 64 && nb > 1) ? nb - 1 : nb;            // block rows done as full tiles
    full.clear(); quarters.clear();
    std::vector<std::vector<int>> q = square_tile_lists(nbf);
    size_t total = 0;
    for (const auto& l : q) total += l.size();
    const size_t rem = total % (size_t)slots;
    size_t cut = (total > (size_t)slots && (double)rem < tail_max * slots) ? rem : 0;
    // even the lists out first (lengths differ by at most one afterwards), then take the cut from the back of the longest lists
    auto longest = [&]() { int x = 0; for (int k = 1; k < 8; ++k) if (q[k].size() > q[x].size()) x = k; return x; };
    auto shortest = [&]() { int x = 0; for (int k = 1; k < 8; ++k) if (q[k].size() < q[x].size()) x = k; return x; };
    while (q[longest()].size() > q[shortest()].size() + 1) { const int a = longest(), b = shortest(); q[b].push_back(q[a].back()); q[a].pop_back(); }
    for (; cut > 0; --cut) {
        const int x = longest();
        const int m = q[x].back(); q[x].pop_back();
        const int bi = m >> 16, bj = m & 0xffff;
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) if (2 * bi + a >= 2 * bj + b) quarters.push_back((2 * bi + a) << 16 | (2 * bj + b));
    }
    if (nbf < nb) for (int qj = 0; qj <= 2 * nbf; ++qj) quarters.push_back((2 * nbf) << 16 | qj);
    // the kernel's XCD ranges are [x * per, (x + 1) * per) with per = ceil(count / 8): lists of equal length map onto them exactly;
    // where they differ by one, a tile slides to the neighbouring XCD (harmless)
    for (int x = 0; x < 8; ++x) for (int v : q[x]) full.push_back(v);
}

static void launch_gemm_nt(bool lower, const float* A, long long lda, const float* B, long long ldb, float* C, long long ldc,
                           int M, int N, int K, float alpha, float beta, bool mirror, bool kstart_row, hipStream_t st,
                           int ksplit = 1, long long cstride = 0, const int* tilemap = nullptr, int ntiles_listed = -1,
                           const int* qmap = nullptr, int nq = 0) {
    if (M <= 0 || N <= 0) return;
    GemmNT g;
    g.ksplit = ksplit; g.cstride = cstride; g.tilemap = tilemap; g.qmap = qmap; g.nq = lower ? nq : 0; g.grid_full = 0;
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.alpha = alpha; g.beta = beta; g.mirror = mirror ? 1 : 0; g.kstart_row = kstart_row ? 1 : 0;
    g.nbi = (M + SK_BM - 1) / SK_BM; g.nbj = (N + SK_BM - 1) / SK_BM;
    g.ntiles = lower ? g.nbi * (g.nbi + 1) / 2 : g.nbi * g.nbj;
    if (tilemap != nullptr && ntiles_listed >= 0) g.ntiles = ntiles_listed;     // only the listed tiles (the distributed inverse: this rank's share)
    if (g.ntiles <= 0 && g.nq <= 0) return;
    g.grid_full = (g.ntiles * ksplit + 7) / 8 * 8;
    const int grid = g.grid_full + (g.nq + 7) / 8 * 8;
    if (!mirror && g.nq == 0) {                              // (the quarter items of the Gram's tail come with the mirrored store)
        if (lower) hipLaunchKernelGGL((gemm_nt_mfma_kernel<1, 1>), dim3(grid), dim3(SK_THREADS), 0, st, g);
        else hipLaunchKernelGGL((gemm_nt_mfma_kernel<0, 1>), dim3(grid), dim3(SK_THREADS), 0, st, g);
    } else if (lower) hipLaunchKernelGGL((gemm_nt_mfma_kernel<1, 0>), dim3(grid), dim3(SK_THREADS), 0, st, g);
    else hipLaunchKernelGGL((gemm_nt_mfma_kernel<0, 0>), dim3(grid), dim3(SK_THREADS), 0, st, g);
}

static void launch_gemm_nt_f32(bool lower, const float* A, long long lda, const float* B, long long ldb, float* C, long long ldc,
                               int M, int N, int K, float alpha, float beta, bool mirror, bool kstart_row, hipStream_t st) {
    launch_gemm_nt(lower, A, lda, B, ldb, C, ldc, M, N, K, alpha, beta, mirror, kstart_row, st);
}

// ---------------------------------------------------------------------------------------------- Gram
// out(i, j) = sum_s part[s](i, j) for i >= j, mirrored (both triangles)
__global__ void __launch_bounds__(256) sum_splits_mirror_kernel(const float* __restrict__ part, long long ldp, long long stride, int nsplit,
                                                                float* __restrict__ C, long long ldc, int m) {
    const int j = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < m && i >= j) {
        float v = 0.f;
        for (int s = 0; s < nsplit; ++s) v += part[(size_t)s * stride + (size_t)j * ldp + i];
        C[(size_t)j * ldc + i] = v;
        C[(size_t)i * ldc + j] = v;
    }
}

__global__ void __launch_bounds__(256) pad_copy_f32_kernel(const float* __restrict__ in, long long ldi, int rows, int cols,
                                                           float* __restrict__ out, long long ldo) {
    const int c = blockIdx.y;
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < rows) out[(size_t)c * ldo + r] = in[(size_t)c * ldi + r];
}

// C (both triangles) = A'A (atA; C is cols x cols) or A A' (C is rows x rows) for A rows x cols column-major.
// The operand is copied once into a zero-padded buffer with the OUTPUT index contiguous (A' for atA, A itself
// otherwise).  When the lower triangle has too few 128 x 128 tiles to fill 256 CUs (wide solver: order n = 2000,
// K = p = 2 * 10^5; consensus blocks; small tall problems) the summation range is split over several workgroups
// per tile and the partial tiles are summed in a fixed order (deterministic, no atomics).
void gram_mfma_f32(const float* A, long long lda, int rows, int cols, bool atA, float* C, long long ldc, hipStream_t st) {
    const int M = atA ? cols : rows;
    const int Kd = atA ? rows : cols;
    const long long ldz = round_up(M, SK_BM);
    const int nb = (int)(ldz / SK_BM);
    const int ntiles = nb * (nb + 1) / 2;
    int ksplit = 1;
    if (ntiles < 512) ksplit = std::min(std::min(16, (1024 + ntiles - 1) / ntiles), std::max(1, Kd / 2048));
    if (atA && ksplit == 1 && ntiles >= 512 && gram_split_mode() != 0) {
        // deep-K lower-triangle Gram of a tall matrix (>= 512 tiles: order >= ~4000): bf16 matrix cores, three-way split
        // (gram_bf16x3.hip); tiles in the same XCD-local square order
        GramSplit3 z3;
        z3.alloc(M, Kd, st);
        z3.split_cols(A, lda, rows, 0, cols, st);
        // fp16 form: 256 x 256 macro-tiles (round 6; option GRAM_B3_TILE=128 keeps the 128 x 128 tiles: the A/B, bit-identical)
        bool big = z3.npl == 2;
        if (const char* e = option("GRAM_B3_TILE")) big = big && std::atoi(e) != 128;
        if (big) {
            // One macro-tile workgroup is resident per CU: T macro-tiles take ceil(T / CUs) rounds (p = 10^4: 820 on 256 CUs, the fourth round
            // a fifth full).  Whole rounds of macro-tiles; the rest -- taken evenly from the back of the 8 XCD lists -- as 128 x 128 tiles after
            // them (four per CU are resident: one short round).  GRAM_B3_TAIL=0: every tile a macro-tile.
            const int nb2 = (M + 255) / 256;
            std::vector<std::vector<int>> q = square_tile_lists(nb2);
            size_t total = 0;
            for (const auto& l : q) total += l.size();
            const size_t slots = (size_t)device_info().num_cu;
            size_t cut = total > slots ? total % slots : 0;
            if (const char* e = option("GRAM_B3_TAIL")) { if (std::atoi(e) == 0) cut = 0; }
            if (cut * 4 > slots * 3) cut = 0;                     // a last round three quarters full is left alone
            std::vector<int> tail;
            auto longest = [&]() { int x = 0; for (int k = 1; k < 8; ++k) if (q[k].size() > q[x].size()) x = k; return x; };
            for (; cut > 0; --cut) {
                const int x = longest();
                const int m = q[x].back(); q[x].pop_back();
                const int bi = m >> 16, bj = m & 0xffff;
                for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b)
                    if (2 * bi + a >= 2 * bj + b && 2 * bi + a < nb) tail.push_back((2 * bi + a) << 16 | (2 * bj + b));
            }
            std::vector<int> order;
            for (int x = 0; x < 8; ++x) for (int v : q[x]) order.push_back(v);
            DevBuf<int> tmap = upload_ints(order, st), ttail = upload_ints(tail, st);
            z3.gram_lower256(C, ldc, tmap.get(), (int)order.size(), st, ttail.get(), (int)tail.size());
        } else {
            DevBuf<int> tmap = make_square_tilemap(nb, st);
            z3.gram_lower(C, ldc, tmap.get(), ntiles, st);      // (round 6 A/B: the plain row-major triangle order 35.8 against 34.7 ms)
        }
        ADMM_HIP_CHECK(hipGetLastError());
        comm_stream_sync(st);
        return;
    }
    const int nk = (int)round_up(Kd, (long long)SK_BK * ksplit);
    DevBuf<float> Z((size_t)ldz * nk);
    Z.zero(st);
    if (atA) transpose<float>(A, lda, rows, cols, Z.get(), ldz, st);
    else hipLaunchKernelGGL(pad_copy_f32_kernel, dim3((rows + 255) / 256, cols), dim3(256), 0, st, A, lda, rows, cols, Z.get(), ldz);
    if (ksplit == 1) {
        // square-ish tile order with quarter items for the tail (the row-major order and the no-quarter form lost their A/B in round 4)
        {
            std::vector<int> full, quarters;
            int wg_per_cu = 2;                               // resident workgroups per CU (registers / LDS: 4 as compiled today)
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&wg_per_cu, gemm_nt_mfma_kernel<1, 0>, SK_THREADS, 0) != hipSuccess || wg_per_cu < 1) wg_per_cu = 2;
            gram_work_lists(M, wg_per_cu * device_info().num_cu, 0.55, full, quarters);
            DevBuf<int> tmap = upload_ints(full, st), qmap = upload_ints(quarters, st);
            launch_gemm_nt(true, Z.get(), ldz, Z.get(), ldz, C, ldc, M, M, nk, 1.f, 0.f, true, false, st, 1, 0, tmap.get(), (int)full.size(),
                           qmap.get(), (int)quarters.size());
            comm_stream_sync(st);
        }
        ADMM_HIP_CHECK(hipGetLastError());
        comm_stream_sync(st);      // Z and the tile map are freed on return
        return;
    }
    const long long stride = ldz * ldz;
    DevBuf<float> part((size_t)stride * ksplit);
    launch_gemm_nt(true, Z.get(), ldz, Z.get(), ldz, part.get(), ldz, M, M, nk, 1.f, 0.f, false, false, st, ksplit, stride);
    hipLaunchKernelGGL(sum_splits_mirror_kernel, dim3((M + 255) / 256, M), dim3(256), 0, st, part.get(), ldz, stride, ksplit, C, ldc, M);
    ADMM_HIP_CHECK(hipGetLastError());
    comm_stream_sync(st);
}

// Block row of a Gram matrix, for the pipelined host-input setup: C[r0 : r0 + nr, 0 : r0 + nr] = Z[r0 : r0 + nr, :] Z[0 : r0 + nr, :]'
// (Z with the output index contiguous, K a multiple of 16, r0 a multiple of 4, rows readable up to the next multiple of
// 128 past r0 + nr).  Same K order per element as the one-shot Gram: bit-identical values.
void gram_rows_mfma_f32(const float* Z, long long ldz, int r0, int nr, int K, float* C, long long ldc, hipStream_t st) {
    launch_gemm_nt(false, Z + r0, ldz, Z, ldz, C + r0, ldc, nr, r0 + nr, K, 1.f, 0.f, false, false, st);
}

// ---------------------------------------------------------------------------------------------- Cholesky + inverse
// (diagonal-block kernel and block driver: chol_inverse.h)
void spd_inverse_mfma_f32(float* A, long long lda, int p, hipStream_t st) {
    spd_inverse_blocked<float>(A, lda, p, st, launch_gemm_nt_f32);
}

// The same with the block columns of the factorisation dealt out to the ranks of the attached communicator (chol_inverse.h,
// cholesky_linvt_blocked_dist) and, of the inverse U U', only the lower 128 x 128 tiles listed in `need` (bi << 16 | bj): the tiles
// this rank's share of the sharded x-update reads.  Everything a rank computes is bit-identical to the single-process result.
void spd_inverse_mfma_f32_dist(float* A, long long lda, int p, const std::vector<int>& need, double* flops, hipStream_t st) {
    const TraceRange trace_range("admm:factor+inverse (distributed)");
    const CommInfo ci = comm_info();
    double fl = 0;
    DevBuf<float> U = cholesky_linvt_blocked_dist<float>(A, lda, p, st, launch_gemm_nt_f32, ci.nranks, ci.rank,
                                                        [](float* b, size_t n, int root, hipStream_t s) { broadcast_f32(b, n, root, s); }, &fl);
    comm_check();
    const int pp = (p + 127) / 128 * 128;
    DevBuf<int> tmap(std::max<size_t>(need.size(), 1));
    if (!need.empty()) ADMM_HIP_CHECK(hipMemcpyAsync(tmap.get(), need.data(), need.size() * sizeof(int), hipMemcpyHostToDevice, st));
    launch_gemm_nt(true, U.get(), lda, U.get(), lda, A, lda, p, p, pp, 1.f, 0.f, false, true, st, 1, 0, tmap.get(), (int)need.size());
    for (int m : need) { const int bi = m >> 16; fl += 2.0 * 128 * 128 * (double)(pp - bi * 128); }
    ADMM_HIP_CHECK(hipGetLastError());
    comm_stream_sync(st);
    if (flops) *flops = fl;
}

}  // namespace admm
