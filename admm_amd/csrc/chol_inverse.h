// Blocked Cholesky + inverse on the matrix cores, generic in the element type (float: syrk_mfma.hip, double:
// gemm_f64_mfma.hip).  The caller supplies its NT-GEMM launcher
//     gemm(lower, A, lda, B, ldb, C, ldc, M, N, K, alpha, beta, mirror, kstart_row, stream)
// (C = alpha A B' + beta C on 128 x 128 tiles, operands with the output index contiguous).
#pragma once
#include "admm_internal.h"
#include "device_utils.h"
#include "comm.h"

namespace admm {

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
__device__ __forceinline__ void lds_load4(const float* p, float (&v)[4]) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void lds_load4(const double* p, double (&v)[4]) {
    const double2 a = *reinterpret_cast<const double2*>(p), b = *reinterpret_cast<const double2*>(p + 2);
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
}

constexpr int PF_BLOCKS = 32 * 33 / 2;         // 4 x 4 sub-blocks on or below the diagonal
constexpr int PF_THREADS = 576;                // 9 waves >= 528 sub-blocks

template <typename T>
__global__ void __launch_bounds__(PF_THREADS)
potf2_inv_kernel(T* __restrict__ A, long long lda, int nbk, T* __restrict__ Dinv, int* __restrict__ info, int base) {
    __shared__ __attribute__((aligned(16))) T colbuf[2][128];     // unscaled pivot column of L
    __shared__ __attribute__((aligned(16))) T rowbuf[2][128];     // unscaled pivot row of W
    __shared__ T invs[128];                                       // 1 / l_jj
    const int tid = threadIdx.x;
    // sub-blocks enumerated column by column (bj = 0: bi = 0..31, bj = 1: bi = 1..31, ...): the lanes of a wave
    // share bj and own consecutive row blocks
    int bj = 0, first = 0;
    while (bj < 31 && tid >= first + (32 - bj)) { first += 32 - bj; ++bj; }
    const int bi = bj + (tid - first);
    const bool act = tid < PF_BLOCKS;
    const int r0 = 4 * bi, c0 = 4 * bj;
    T l[4][4], w[4][4];            // [column][row]; entries above the diagonal of a diagonal sub-block are don't-cares
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = r0 + r, cc = c0 + c;
            T v = (rr == cc) ? T(1) : T(0);
            if (act && rr < nbk && cc < nbk && rr >= cc) v = A[(size_t)cc * lda + rr];
            l[c][r] = v;
            w[c][r] = (rr == cc) ? T(1) : T(0);
        }
    for (int k = tid; k < 128; k += PF_THREADS) { colbuf[1][k] = T(0); rowbuf[1][k] = T(0); rowbuf[0][k] = T(0); }
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
        T d = colbuf[cur][j];
        if (!(d > T(0)) || !isfinite(d)) {
            if (tid == 0) atomicCAS(info, 0, base + j + 1);
            d = T(1);
        }
        if (tid == 0) invs[j] = T(1) / sqrt(d);
        if (act && r0 + 3 > j) {
            const T inv2 = T(1) / d;
            // masks folded into the factors: rows <= j get f = 0; columns <= j take the W update, columns > j the L update
            // unconditional vector reads first (one 16- / 32-byte LDS read per group), masks afterwards
            T lr[4], lcv[4], wjv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { lr[k] = T(0); lcv[k] = T(0); wjv[k] = T(0); }
            lds_load4(&colbuf[cur][r0], lr); lds_load4(&colbuf[cur][c0], lcv); lds_load4(&rowbuf[cur][c0], wjv);
            T f[4], lc[4], wj[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                f[k] = r0 + k > j ? lr[k] * inv2 : T(0);
                lc[k] = c0 + k > j ? lcv[k] : T(0);
                wj[k] = c0 + k > j ? T(0) : wjv[k];
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    l[c][r] = fma(-f[r], lc[c], l[c][r]);       // trailing update of L   (rows, columns > j)
                    w[c][r] = fma(-f[r], wj[c], w[c][r]);       // W_r -= l_rj W_j        (rows > j, columns <= j)
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
            const T ic = invs[cc];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = r0 + r;
                if (rr >= cc && rr < nbk && cc < nbk) A[(size_t)cc * lda + rr] = l[c][r] * ic;
                Dinv[(size_t)cc * 128 + rr] = (rr >= cc) ? w[c][r] * invs[rr] : T(0);
                if (bi != bj) Dinv[(size_t)rr * 128 + cc] = T(0);          // the mirrored sub-block above the diagonal
            }
        }
    }
}

template <typename T>
__global__ void set_identity_kernel(T* U, long long ldu, int p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < p) U[(size_t)i * ldu + i] = T(1);
}

// In place: A (p x p, SPD, lower triangle valid, leading dimension lda >= round_up(p, 128), that many zero-padded
// columns allocated) -> its Cholesky factor L in the lower triangle, and U = L^-T (upper triangular, lda x round_up(p, 128),
// returned).  Right-looking on 128-blocks, fused with the right-looking block elimination of [L | I]:
//     at block step k   W_k <- L_kk^-1 W_k ;  W_i -= L_ik W_k (i > k),   stored transposed (U = W' = L^-T)
// so that every update is an NT product with K = 128 over many tiles (no serial triangular-inverse sweep).
// Measured and rejected: a second stream for the bulk updates with a look-ahead of one block column (the serial
// diagonal-block kernel of step k + 1 under the bulk updates of step k) -- correct, but the cross-stream event waits
// cost more than the overlap gains (38 -> 53 ms at p = 10^4, and a second stream costs small calls 80 ms).
// Throws ADMM_ERR_NOT_SPD.
template <typename T, typename Gemm>
DevBuf<T> cholesky_linvt_blocked(T* A, long long lda, int p, hipStream_t st, Gemm gemm) {
    const int nb = (p + 127) / 128;
    const int pp = nb * 128;
    ADMM_REQUIRE(lda >= pp, "blocked Cholesky: leading dimension must cover whole 128-row blocks");
    DevBuf<T> Dinv((size_t)nb * 128 * 128), U((size_t)lda * pp);
    DevBuf<int> info(1);
    info.zero(st); U.zero(st);
    hipLaunchKernelGGL((set_identity_kernel<T>), dim3((p + 255) / 256), dim3(256), 0, st, U.get(), lda, p);
    for (int k = 0; k < nb; ++k) {
        const int r0 = k * 128;
        const int nbk = std::min(128, p - r0);
        T* Akk = A + (size_t)r0 * lda + r0;
        T* Dk = Dinv.get() + (size_t)k * 128 * 128;
        hipLaunchKernelGGL((potf2_inv_kernel<T>), dim3(1), dim3(PF_THREADS), 0, st, Akk, lda, nbk, Dk, info.get(), r0);
        T* Ukb = U.get() + (size_t)r0 * lda;                                   // column block k of U, rows 0 .. r0 + nbk
        // U[:, k] <- U[:, k] L_kk^-T   (in place: a tile only reads its own rows)
        gemm(false, Ukb, lda, Dk, 128, Ukb, lda, r0 + nbk, nbk, 128, T(1), T(0), false, false, st);
        const int M = p - (r0 + 128);
        if (M > 0) {
            T* Apan = A + (size_t)r0 * lda + r0 + 128;                         // rows below the diagonal block, its 128 columns
            // L_ik = A_ik L_kk^-T : C[i, j] = sum_t A_ik[i, t] Linv[j, t]; in place
            gemm(false, Apan, lda, Dk, 128, Apan, lda, M, nbk, 128, T(1), T(0), false, false, st);
            // A_ij -= L_ik L_jk' on the lower tiles of the trailing matrix
            T* Atr = A + (size_t)(r0 + 128) * lda + r0 + 128;
            gemm(true, Apan, lda, Apan, lda, Atr, lda, M, M, 128, T(-1), T(1), false, false, st);
            // U[:, i] -= U[:, k] L_ik'  for all row blocks i > k at once
            gemm(false, Ukb, lda, Apan, lda, U.get() + (size_t)(r0 + 128) * lda, lda, r0 + 128, M, 128, T(-1), T(1), false, false, st);
        }
    }
    int h = 0;
    ADMM_HIP_CHECK(hipMemcpyAsync(&h, info.get(), sizeof(int), hipMemcpyDeviceToHost, st));
    ADMM_HIP_CHECK(hipStreamSynchronize(st));
    if (h != 0) throw Error(ADMM_ERR_NOT_SPD, "Cholesky: matrix is not positive definite (pivot " + std::to_string(h) + ")");
    return U;
}

// ---- the same factorisation with its block columns dealt out to the ranks of a communicator
// Block column k (128 columns of A's lower triangle and of U) belongs to rank k mod N.  Step k: the owner factorises the diagonal
// block, finishes its column of U and the panel L_ik, and BROADCASTS both (one message: 128 x (pp + 128) floats); every rank then
// applies the two rank-128 updates to the block columns IT owns (A_ij -= L_ik L_jk', U[:, i] -= U[:, k] L_ik').  Every tile sees
// exactly the updates, in exactly the order (k ascending, K = 128 per update), of the single-process driver above, from operands
// that are bit-identical (the broadcast moves them unchanged): the factor, U and everything built from them are BIT-IDENTICAL to
// the single-process result.  Per rank: 1 / N of the update flops (2 x p^3 / 3 in all); the diagonal-block kernel and the panel
// stay serial at the owner (nb x ~0.1 ms), and every rank RECEIVES every panel (p^2 floats in all: what a 1-D column distribution
// costs; a 2-D block-cyclic one would cut it to p^2 / sqrt(N) at the price of row broadcasts).  After the loop every rank holds
// all of U (each column was broadcast when it became final).  `flops`: update flops this rank performed (for the scaling table).
template <typename T>
__global__ void __launch_bounds__(256) panel_pack_kernel(const T* __restrict__ A, const T* __restrict__ U, long long lda, int r0, int pp, T* __restrict__ stage, int unpack,
                                                         T* __restrict__ Aw, T* __restrict__ Uw) {
    const int c = blockIdx.y;                               // column of the block (0 .. 127)
    const int len = pp + 128;                               // (pp - r0) rows of L (diagonal block included) + (r0 + 128) rows of U
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= len) return;
    const size_t col = (size_t)(r0 + c) * lda;
    const int nl = pp - r0;
    if (!unpack) stage[(size_t)c * len + idx] = idx < nl ? A[col + r0 + idx] : U[col + (idx - nl)];
    else { const T v = stage[(size_t)c * len + idx]; if (idx < nl) Aw[col + r0 + idx] = v; else Uw[col + (idx - nl)] = v; }
}

// the not-SPD flag of the distributed factorisation as a double, so that the exchange layer's all-reduce can carry it
static __global__ void info_to_double_kernel(const int* info, double* out) { if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)*info; }

template <typename T, typename Gemm, typename Bcast>
DevBuf<T> cholesky_linvt_blocked_dist(T* A, long long lda, int p, hipStream_t st, Gemm gemm, int nranks, int rank, Bcast bcast, double* flops) {
    const int nb = (p + 127) / 128;
    const int pp = nb * 128;
    ADMM_REQUIRE(lda >= pp, "blocked Cholesky: leading dimension must cover whole 128-row blocks");
    DevBuf<T> Dinv((size_t)128 * 128), U((size_t)lda * pp), stage((size_t)128 * (pp + 128));
    DevBuf<int> info(1);
    info.zero(st); U.zero(st);
    hipLaunchKernelGGL((set_identity_kernel<T>), dim3((p + 255) / 256), dim3(256), 0, st, U.get(), lda, p);
    double fl = 0;
    for (int k = 0; k < nb; ++k) {
        const int r0 = k * 128;
        const int nbk = std::min(128, p - r0);
        const int owner = k % nranks;
        T* Ukb = U.get() + (size_t)r0 * lda;
        const int M = p - (r0 + 128);
        T* Apan = A + (size_t)r0 * lda + r0 + 128;
        if (rank == owner) {
            T* Akk = A + (size_t)r0 * lda + r0;
            hipLaunchKernelGGL((potf2_inv_kernel<T>), dim3(1), dim3(PF_THREADS), 0, st, Akk, lda, nbk, Dinv.get(), info.get(), r0);
            gemm(false, Ukb, lda, Dinv.get(), 128, Ukb, lda, r0 + nbk, nbk, 128, T(1), T(0), false, false, st);
            if (M > 0) gemm(false, Apan, lda, Dinv.get(), 128, Apan, lda, M, nbk, 128, T(1), T(0), false, false, st);
            fl += 2.0 * 128 * nbk * (double)(r0 + nbk + std::max(M, 0));
        }
        if (nranks > 1) {
            const dim3 grid((unsigned)((pp + 128 + 255) / 256), 128);
            if (rank == owner) hipLaunchKernelGGL((panel_pack_kernel<T>), grid, dim3(256), 0, st, A, U.get(), lda, r0, pp, stage.get(), 0, A, U.get());
            bcast(stage.get(), (size_t)128 * (pp + 128), owner, st);
            if (rank != owner) hipLaunchKernelGGL((panel_pack_kernel<T>), grid, dim3(256), 0, st, A, U.get(), lda, r0, pp, stage.get(), 1, A, U.get());
        }
        // the block columns j > k this rank owns: trailing update of A (rows >= j) and of U (rows < r0 + 128)
        for (int j = k + 1; j < nb; ++j) {
            if (j % nranks != rank) continue;
            const int c0 = j * 128;
            const int nj = std::min(128, p - c0);
            const T* Lj = Apan + (c0 - (r0 + 128));            // rows of block j (and below) of the panel
            gemm(false, Lj, lda, Lj, lda, A + (size_t)c0 * lda + c0, lda, p - c0, nj, 128, T(-1), T(1), false, false, st);
            gemm(false, Ukb, lda, Lj, lda, U.get() + (size_t)c0 * lda, lda, r0 + 128, nj, 128, T(-1), T(1), false, false, st);
            fl += 2.0 * 128 * nj * (double)((p - c0) + (r0 + 128));
        }
    }
    int h = 0;
    if (nranks > 1) {
        // only the owner of a failing diagonal block sees the flag: summed over the ranks, so that EVERY rank throws the same error
        // instead of N - 1 of them running on into the solver's exchanges and timing out there
        DevBuf<double> hd(1);
        hipLaunchKernelGGL(info_to_double_kernel, dim3(1), dim3(64), 0, st, info.get(), hd.get());
        allreduce_sum_f64(hd.get(), 1, st);
        double hs = 0;
        ADMM_HIP_CHECK(hipMemcpyAsync(&hs, hd.get(), sizeof(double), hipMemcpyDeviceToHost, st));
        ADMM_HIP_CHECK(hipStreamSynchronize(st));
        h = hs != 0.0 ? (int)hs : 0;
        if (hs != 0.0 && h == 0) h = -1;
    } else {
        ADMM_HIP_CHECK(hipMemcpyAsync(&h, info.get(), sizeof(int), hipMemcpyDeviceToHost, st));
        ADMM_HIP_CHECK(hipStreamSynchronize(st));
    }
    if (h != 0) throw Error(ADMM_ERR_NOT_SPD, "Cholesky: matrix is not positive definite (pivot " + std::to_string(h) + ")");
    if (flops) *flops = fl;
    return U;
}

// In place: A -> full symmetric inverse  A^-1 = L^-T L^-1 = U U'  (both triangles; products start at k = tile row
// since U is upper triangular).
template <typename T, typename Gemm>
void spd_inverse_blocked(T* A, long long lda, int p, hipStream_t st, Gemm gemm) {
    DevBuf<T> U = cholesky_linvt_blocked<T>(A, lda, p, st, gemm);
    const int pp = (p + 127) / 128 * 128;
    gemm(true, U.get(), lda, U.get(), lda, A, lda, p, p, pp, T(1), T(0), true, true, st);
    ADMM_HIP_CHECK(hipStreamSynchronize(st));
}

}  // namespace admm
