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
// double buffered.  Tile ids are remapped so that each XCD (block id % 8) walks a contiguous range
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
};

__device__ __forceinline__ void tri_decode(int t, int& bi, int& bj) {   // t -> (bi, bj), bi >= bj, row-major triangle
    int b = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while ((long long)(b + 1) * (b + 2) / 2 <= t) ++b;
    while ((long long)b * (b + 1) / 2 > t) --b;
    bi = b; bj = t - b * (b + 1) / 2;
}

template <int LOWER>
__global__ void __launch_bounds__(SK_THREADS, 2)
gemm_nt_mfma_kernel(GemmNT g) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][SK_BK][SK_BM];      // [buffer][A/B][k][i]
    // XCD-aware remap: block b runs on XCD b % 8; give every XCD a contiguous range of tiles
    const int nwork = g.ntiles * g.ksplit;
    const int per = (nwork + 7) / 8;
    const int w_idx = (blockIdx.x % 8) * per + blockIdx.x / 8;
    if (w_idx >= nwork) return;
    const int t_idx = w_idx % g.ntiles, split = w_idx / g.ntiles;
    int bi, bj;
    if (LOWER) tri_decode(t_idx, bi, bj);
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

    float4 ra0, ra1, rb0, rb1;
    auto gload = [&](int k0) {
        ra0 = *reinterpret_cast<const float4*>(gA + (size_t)k0 * g.lda);
        ra1 = *reinterpret_cast<const float4*>(gA + (size_t)k0 * g.lda + a8);
        rb0 = *reinterpret_cast<const float4*>(gB + (size_t)k0 * g.ldb);
        rb1 = *reinterpret_cast<const float4*>(gB + (size_t)k0 * g.ldb + b8);
    };
    auto lstore = [&](int buf) {
        *reinterpret_cast<float4*>(&lds[buf][0][s_row0][s_col]) = ra0;
        *reinterpret_cast<float4*>(&lds[buf][0][s_row0 + 8][s_col]) = ra1;
        *reinterpret_cast<float4*>(&lds[buf][1][s_row0][s_col]) = rb0;
        *reinterpret_cast<float4*>(&lds[buf][1][s_row0 + 8][s_col]) = rb1;
    };

    const int kchunk = g.K / g.ksplit;                                  // multiple of SK_BK (launcher)
    const int kbeg = g.kstart_row ? (max(I0, J0) / SK_BK) * SK_BK : split * kchunk;
    const int ntile_k = ((g.ksplit > 1 ? kbeg + kchunk : g.K) - kbeg) / SK_BK;
    const int fk = lane >> 5, fi = lane & 31;
    if (ntile_k > 0) {
        gload(kbeg);
        lstore(0);
        __syncthreads();
        for (int kt = 0; kt < ntile_k; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < ntile_k) gload(kbeg + (kt + 1) * SK_BK);      // next tile in flight while this one computes
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
            if (kt + 1 < ntile_k) {
                lstore(buf ^ 1);
                __syncthreads();
            }
        }
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
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

static void launch_gemm_nt(bool lower, const float* A, long long lda, const float* B, long long ldb, float* C, long long ldc,
                           int M, int N, int K, float alpha, float beta, bool mirror, bool kstart_row, hipStream_t st,
                           int ksplit = 1, long long cstride = 0) {
    if (M <= 0 || N <= 0) return;
    GemmNT g;
    g.ksplit = ksplit; g.cstride = cstride;
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.alpha = alpha; g.beta = beta; g.mirror = mirror ? 1 : 0; g.kstart_row = kstart_row ? 1 : 0;
    g.nbi = (M + SK_BM - 1) / SK_BM; g.nbj = (N + SK_BM - 1) / SK_BM;
    g.ntiles = lower ? g.nbi * (g.nbi + 1) / 2 : g.nbi * g.nbj;
    const int grid = (g.ntiles * ksplit + 7) / 8 * 8;
    if (lower) hipLaunchKernelGGL(gemm_nt_mfma_kernel<1>, dim3(grid), dim3(SK_THREADS), 0, st, g);
    else hipLaunchKernelGGL(gemm_nt_mfma_kernel<0>, dim3(grid), dim3(SK_THREADS), 0, st, g);
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
    const int nk = (int)round_up(Kd, (long long)SK_BK * ksplit);
    DevBuf<float> Z((size_t)ldz * nk);
    Z.zero(st);
    if (atA) transpose<float>(A, lda, rows, cols, Z.get(), ldz, st);
    else hipLaunchKernelGGL(pad_copy_f32_kernel, dim3((rows + 255) / 256, cols), dim3(256), 0, st, A, lda, rows, cols, Z.get(), ldz);
    if (ksplit == 1) {
        launch_gemm_nt(true, Z.get(), ldz, Z.get(), ldz, C, ldc, M, M, nk, 1.f, 0.f, true, false, st);
        ADMM_HIP_CHECK(hipStreamSynchronize(st));      // Z is freed on return
        return;
    }
    const long long stride = ldz * ldz;
    DevBuf<float> part((size_t)stride * ksplit);
    launch_gemm_nt(true, Z.get(), ldz, Z.get(), ldz, part.get(), ldz, M, M, nk, 1.f, 0.f, false, false, st, ksplit, stride);
    hipLaunchKernelGGL(sum_splits_mirror_kernel, dim3((M + 255) / 256, M), dim3(256), 0, st, part.get(), ldz, stride, ksplit, C, ldc, M);
    ADMM_HIP_CHECK(hipGetLastError());
    ADMM_HIP_CHECK(hipStreamSynchronize(st));
}

// Block row of a Gram matrix, for the pipelined host-input setup: C[r0 : r0 + nr, 0 : r0 + nr] = Z[r0 : r0 + nr, :] Z[0 : r0 + nr, :]'
// (Z with the output index contiguous, K a multiple of 16, r0 a multiple of 4, rows readable up to the next multiple of
// 128 past r0 + nr).  Same K order per element as the one-shot Gram: bit-identical values.
void gram_rows_mfma_f32(const float* Z, long long ldz, int r0, int nr, int K, float* C, long long ldc, hipStream_t st) {
    launch_gemm_nt(false, Z + r0, ldz, Z, ldz, C + r0, ldc, nr, r0 + nr, K, 1.f, 0.f, false, false, st);
}

// ---------------------------------------------------------------------------------------------- Cholesky of a diagonal block
// One workgroup: Cholesky of the nbk x nbk diagonal block (nbk <= 128), written back in place (lower), plus
// the inverse of the factor into Dinv (128 x 128, zeros above the diagonal; rows/cols beyond nbk form an
// identity so that products with padded panels stay exact).
//
// Register tiled: thread (bi, bj) keeps the 4 x 4 sub-blocks L[4bi.., 4bj..] and W[4bi.., 4bj..] (W becomes
// L^-1 by the forward elimination of [L | I]) in registers for the whole factorisation.  A step j only moves
// the pivot column of L and the pivot row of W through LDS (double buffered: ONE barrier per step, three
// 16-byte LDS reads per thread instead of one read-modify-write per matrix element).  The scalings by
// 1 / l_jj are deferred: pivot column and pivot row stay unscaled, the update factors carry 1 / l_jj^2, and
// the outputs are scaled once at the end.  (First version: both matrices resident in LDS, 258 us per block,
// LDS-bandwidth bound; 20 ms of the 50 ms factorisation at p = 10^4.)
constexpr int PF_BLOCKS = 32 * 33 / 2;         // 4 x 4 sub-blocks on or below the diagonal
constexpr int PF_THREADS = 576;                // 9 waves >= 528 sub-blocks

__global__ void __launch_bounds__(PF_THREADS)
potf2_inv_kernel(float* __restrict__ A, long long lda, int nbk, float* __restrict__ Dinv, int* __restrict__ info, int base) {
    __shared__ __attribute__((aligned(16))) float colbuf[2][128];     // unscaled pivot column of L
    __shared__ __attribute__((aligned(16))) float rowbuf[2][128];     // unscaled pivot row of W
    __shared__ float invs[128];                                       // 1 / l_jj
    const int tid = threadIdx.x;
    // sub-blocks enumerated column by column (bj = 0: bi = 0..31, bj = 1: bi = 1..31, ...): the lanes of a wave
    // share bj and own consecutive row blocks
    int bj = 0, first = 0;
    while (bj < 31 && tid >= first + (32 - bj)) { first += 32 - bj; ++bj; }
    const int bi = bj + (tid - first);
    const bool act = tid < PF_BLOCKS;
    const int r0 = 4 * bi, c0 = 4 * bj;
    float l[4][4], w[4][4];            // [column][row]; entries above the diagonal of a diagonal sub-block are don't-cares
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = r0 + r, cc = c0 + c;
            float v = (rr == cc) ? 1.f : 0.f;
            if (act && rr < nbk && cc < nbk && rr >= cc) v = A[(size_t)cc * lda + rr];
            l[c][r] = v;
            w[c][r] = (rr == cc) ? 1.f : 0.f;
        }
    for (int k = tid; k < 128; k += PF_THREADS) { colbuf[1][k] = 0.f; rowbuf[1][k] = 0.f; rowbuf[0][k] = 0.f; }
    __syncthreads();
    if (act && bj == 0) {              // publish pivot column 0 / pivot row 0
#pragma unroll
        for (int r = 0; r < 4; ++r) colbuf[0][r0 + r] = l[0][r];
        if (bi == 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c) rowbuf[0][c] = w[c][0];
        }
    }
    for (int j = 0; j < 128; ++j) {
        __syncthreads();
        const int cur = j & 1, nxt = cur ^ 1;
        float d = colbuf[cur][j];
        if (!(d > 0.f) || !isfinite(d)) {
            if (tid == 0) atomicCAS(info, 0, base + j + 1);
            d = 1.f;
        }
        if (tid == 0) invs[j] = 1.f / sqrtf(d);
        if (act && r0 + 3 > j) {
            const float inv2 = 1.f / d;
            const float4 lr4 = *reinterpret_cast<const float4*>(&colbuf[cur][r0]);
            const float4 lc4 = *reinterpret_cast<const float4*>(&colbuf[cur][c0]);
            const float4 wj4 = *reinterpret_cast<const float4*>(&rowbuf[cur][c0]);
            // masks folded into the factors: rows <= j get f = 0; columns <= j take the W update, columns > j the L update
            float f[4], lc[4], wj[4];
            f[0] = r0 + 0 > j ? lr4.x * inv2 : 0.f; f[1] = r0 + 1 > j ? lr4.y * inv2 : 0.f;
            f[2] = r0 + 2 > j ? lr4.z * inv2 : 0.f; f[3] = r0 + 3 > j ? lr4.w * inv2 : 0.f;
            lc[0] = c0 + 0 > j ? lc4.x : 0.f; lc[1] = c0 + 1 > j ? lc4.y : 0.f;
            lc[2] = c0 + 2 > j ? lc4.z : 0.f; lc[3] = c0 + 3 > j ? lc4.w : 0.f;
            wj[0] = c0 + 0 > j ? 0.f : wj4.x; wj[1] = c0 + 1 > j ? 0.f : wj4.y;
            wj[2] = c0 + 2 > j ? 0.f : wj4.z; wj[3] = c0 + 3 > j ? 0.f : wj4.w;
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    l[c][r] = fmaf(-f[r], lc[c], l[c][r]);      // trailing update of L   (rows, columns > j)
                    w[c][r] = fmaf(-f[r], wj[c], w[c][r]);      // W_r -= l_rj W_j        (rows > j, columns <= j)
                }
        }
        // publish the next pivot column / row (final as of this step) into the other buffer
        const int jn = j + 1;
        if (act && jn < 128) {
            if (bj == (jn >> 2)) {
                const int cs = jn & 3;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    colbuf[nxt][r0 + r] = cs == 0 ? l[0][r] : (cs == 1 ? l[1][r] : (cs == 2 ? l[2][r] : l[3][r]));
            }
            if (bi == (jn >> 2)) {
                const int rs = jn & 3;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    rowbuf[nxt][c0 + c] = rs == 0 ? w[c][0] : (rs == 1 ? w[c][1] : (rs == 2 ? w[c][2] : w[c][3]));
            }
        }
    }
    __syncthreads();
    // outputs: L(r, c) = l(r, c) / l_cc (diagonal: d / sqrt(d)), Linv(r, c) = w(r, c) / l_rr; zeros above the diagonal of Dinv
    if (act) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int cc = c0 + c;
            const float ic = invs[cc];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = r0 + r;
                if (rr >= cc && rr < nbk && cc < nbk) A[(size_t)cc * lda + rr] = l[c][r] * ic;
                Dinv[(size_t)cc * 128 + rr] = (rr >= cc) ? w[c][r] * invs[rr] : 0.f;
                if (bi != bj) Dinv[(size_t)rr * 128 + cc] = 0.f;           // the mirrored sub-block above the diagonal
            }
        }
    }
}

__global__ void set_identity_kernel(float* U, long long ldu, int p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < p) U[(size_t)i * ldu + i] = 1.f;
}

// In place: A (p x p, SPD, lower triangle valid, leading dimension lda >= round_up(p, 128), allocation of
// round_up(p, 128) columns with zero padding) -> full symmetric inverse.  Throws ADMM_ERR_NOT_SPD.
void spd_inverse_mfma_f32(float* A, long long lda, int p, hipStream_t st) {
    const int nb = (p + SK_BM - 1) / SK_BM;
    const int pp = nb * SK_BM;
    ADMM_REQUIRE(lda >= pp, "spd_inverse_mfma_f32: leading dimension must cover whole 128-row blocks");
    DevBuf<float> Dinv((size_t)nb * 128 * 128), U((size_t)lda * pp);
    DevBuf<int> info(1);
    info.zero(st); U.zero(st);
    // ---- right-looking blocked Cholesky, fused with the right-looking block elimination of [L | I]:
    // at block step k   W_k <- L_kk^-1 W_k ;  W_i -= L_ik W_k (i > k),   stored transposed (U = W' = L^-T) so that
    // every update is an NT product with K = 128 over many tiles (no serial triangular-inverse sweep).
    hipLaunchKernelGGL(set_identity_kernel, dim3((p + 255) / 256), dim3(256), 0, st, U.get(), lda, p);
    for (int k = 0; k < nb; ++k) {
        const int r0 = k * SK_BM;
        const int nbk = std::min(SK_BM, p - r0);
        float* Akk = A + (size_t)r0 * lda + r0;
        float* Dk = Dinv.get() + (size_t)k * 128 * 128;
        hipLaunchKernelGGL(potf2_inv_kernel, dim3(1), dim3(PF_THREADS), 0, st, Akk, lda, nbk, Dk, info.get(), r0);
        float* Ukb = U.get() + (size_t)r0 * lda;                               // column block k of U, rows 0 .. r0 + nbk
        // U[:, k] <- U[:, k] L_kk^-T   (in place: a tile only reads its own rows)
        launch_gemm_nt(false, Ukb, lda, Dk, 128, Ukb, lda, r0 + nbk, nbk, 128, 1.f, 0.f, false, false, st);
        const int M = p - (r0 + SK_BM);
        if (M > 0) {
            float* Apan = A + (size_t)r0 * lda + r0 + SK_BM;                   // rows below the diagonal block, its 128 columns
            // L_ik = A_ik L_kk^-T : C[i, j] = sum_t A_ik[i, t] Linv[j, t]; in place
            launch_gemm_nt(false, Apan, lda, Dk, 128, Apan, lda, M, nbk, 128, 1.f, 0.f, false, false, st);
            // A_ij -= L_ik L_jk' on the lower tiles of the trailing matrix
            float* Atr = A + (size_t)(r0 + SK_BM) * lda + r0 + SK_BM;
            launch_gemm_nt(true, Apan, lda, Apan, lda, Atr, lda, M, M, 128, -1.f, 1.f, false, false, st);
            // U[:, i] -= U[:, k] L_ik'  for all row blocks i > k at once
            launch_gemm_nt(false, Ukb, lda, Apan, lda, U.get() + (size_t)(r0 + SK_BM) * lda, lda, r0 + SK_BM, M, 128, -1.f, 1.f, false, false, st);
        }
    }
    {
        int h = 0;
        ADMM_HIP_CHECK(hipMemcpyAsync(&h, info.get(), sizeof(int), hipMemcpyDeviceToHost, st));
        ADMM_HIP_CHECK(hipStreamSynchronize(st));
        if (h != 0) throw Error(ADMM_ERR_NOT_SPD, "Cholesky: matrix is not positive definite (pivot " + std::to_string(h) + ")");
    }
    // ---- A^-1 = L^-T L^-1 = U U'  (both triangles)
    launch_gemm_nt(true, U.get(), lda, U.get(), lda, A, lda, p, p, pp, 1.f, 0.f, true, true, st);
    ADMM_HIP_CHECK(hipStreamSynchronize(st));
}

}  // namespace admm
