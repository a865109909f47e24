// fp64 matrix-core Gram for the one-time setup of the LAD / BP solvers (X'X, AA').
//
// Replaces (reference):  Linalg::cross_prod_lower / tcross_prod_lower  BlasWrapper.h:73-152
//   (called from ADMMLAD.h:186-190 and ADMMBP.h:167-170), single-threaded Eigen there.
//
// Gram: C = Z Z' on 128 x 128 lower tiles (+ mirrored store), Z stored with the OUTPUT index contiguous (Z[i, k] at
// i + k ldz), so for a fixed summation index k both factors are contiguous 1 KiB rows.  A workgroup = 4 waves
// (2 x 2), each wave 64 x 64 = 4 x 4 tiles of v_mfma_f64_16x16x4_f64 (32 FLOP/clk/SIMD: the 78.6 TF/s fp64
// matrix peak).  K tiles of 8 go global -> registers -> LDS, double buffered; rows are padded by 8 doubles so
// that the fragment reads (16 consecutive doubles from each of 4 k-rows) touch every bank exactly twice.
// fp64 matrix work is so compute dense (one 64-cycle MFMA per 16-byte LDS read) that nothing else matters.
#include "prep.h"
#include "chol_inverse.h"

namespace admm {

typedef double doublex4 __attribute__((ext_vector_type(4)));

constexpr int DK_BM = 128;
constexpr int DK_BK = 8;
constexpr int DK_LD = DK_BM + 8;
constexpr int DK_THREADS = 256;

struct GemmNTd {
    const double* A; long long lda;      // operands readable for rows < round_up(M / N, 128), columns < K (K multiple of 8)
    const double* B; long long ldb;
    double* C; long long ldc;
    int M, N, K;
    double alpha, beta;
    int nbi, nbj, ntiles;
    int mirror;                          // LOWER mode: also store the transposed tile (both triangles)
    int kstart_row;                      // start the K loop at the tile's first row (A, B upper triangular)
    int ksplit; long long cstride;       // > 1: blockIdx.y takes the y-th share of the K range and writes its own copy of C (C + y cstride): summed afterwards
    int kend_col;                        // B is LOWER triangular (B[j, k] = 0 for k > j): end the K loop after the tile's last column --
                                         //   the skipped products are exact zeros, the result is bit-identical, half the flops
};

__device__ __forceinline__ void tri_decode_d(int t, int& bi, int& bj) {
    int b = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while ((long long)(b + 1) * (b + 2) / 2 <= t) ++b;
    while ((long long)b * (b + 1) / 2 > t) --b;
    bi = b; bj = t - b * (b + 1) / 2;
}

// C = alpha A B' + beta C on 128 x 128 tiles (LOWER: only tiles on or below the diagonal of a square C)
// EPI = 1 (every launch without the mirrored store): the output tile leaves through LDS, as in the float kernel
// (syrk_mfma.hip: C read and written in whole column pieces instead of 32-byte pieces of 16 columns per instruction); same
// arithmetic per element, bit-identical (measured against the direct stores in round 5).
template <int LOWER, int EPI = 0>
__global__ void __launch_bounds__(DK_THREADS, 2)
gemm_nt_mfma_f64_kernel(GemmNTd g) {
    __shared__ __attribute__((aligned(16))) double lds[2][2][DK_BK][DK_LD];      // [buffer][A / B][k][i]
    const int per = (g.ntiles + 7) / 8;                                           // XCD b % 8 walks a contiguous range of tiles
    // (kend_col: the tiles' K ranges grow with the column block -- dealt out to the XCDs in turn, longest first, instead of in ranges)
    // (kstart_row: the inverse U U', K range shrinking with the tile's row -- the same remedy, see syrk_mfma.hip)
    const int t_idx = g.kend_col || g.kstart_row ? (int)blockIdx.x : (blockIdx.x % 8) * per + blockIdx.x / 8;
    if (t_idx >= g.ntiles) return;
    int bi, bj;
    if (LOWER) tri_decode_d(t_idx, bi, bj);
    else { bi = t_idx % g.nbi; bj = t_idx / g.nbi; if (g.kend_col) bj = g.nbj - 1 - bj; }
    const int I0 = bi * DK_BM, J0 = bj * DK_BM;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wi = (wid >> 1) * 64, wj = (wid & 1) * 64;

    // staging: a K tile of one panel is 8 rows x 128 doubles = 512 double2; 2 per thread
    const int s_row0 = tid >> 6;              // 0..3, second load +4
    const int s_col = (tid & 63) * 2;
    const double* gA = g.A + (size_t)s_row0 * g.lda + I0 + s_col;
    const double* gB = g.B + (size_t)s_row0 * g.ldb + J0 + s_col;
    const size_t a4 = (size_t)4 * g.lda, b4 = (size_t)4 * g.ldb;

    doublex4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.0;

    double2 ra0, ra1, rb0, rb1;
    auto gload = [&](int k0) {
        ra0 = *reinterpret_cast<const double2*>(gA + (size_t)k0 * g.lda);
        ra1 = *reinterpret_cast<const double2*>(gA + (size_t)k0 * g.lda + a4);
        rb0 = *reinterpret_cast<const double2*>(gB + (size_t)k0 * g.ldb);
        rb1 = *reinterpret_cast<const double2*>(gB + (size_t)k0 * g.ldb + b4);
    };
    auto lstore = [&](int buf) {
        *reinterpret_cast<double2*>(&lds[buf][0][s_row0][s_col]) = ra0;
        *reinterpret_cast<double2*>(&lds[buf][0][s_row0 + 4][s_col]) = ra1;
        *reinterpret_cast<double2*>(&lds[buf][1][s_row0][s_col]) = rb0;
        *reinterpret_cast<double2*>(&lds[buf][1][s_row0 + 4][s_col]) = rb1;
    };

    int k_lo = 0, k_hi = g.K;
    if (g.ksplit > 1) {                                     // (Gram only: neither triangular shortcut)
        const int per_k = (g.K / DK_BK + g.ksplit - 1) / g.ksplit * DK_BK;
        k_lo = min(g.K, (int)blockIdx.y * per_k); k_hi = min(g.K, k_lo + per_k);
        g.C += (size_t)blockIdx.y * g.cstride;
    }
    const int kbeg = g.kstart_row ? (max(I0, J0) / DK_BK) * DK_BK : k_lo;
    const int kend = g.kend_col ? min(g.K, (J0 + DK_BM + DK_BK - 1) / DK_BK * DK_BK) : k_hi;
    const int ntile_k = (kend - kbeg) / DK_BK;
    const int fk = lane >> 4, fi = lane & 15;            // A / B fragment: one f64 per lane, [i = lane & 15][k = lane >> 4]
    if (ntile_k > 0) {
        gload(kbeg);
        lstore(0);
        __syncthreads();
        for (int kt = 0; kt < ntile_k; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < ntile_k) gload(kbeg + (kt + 1) * DK_BK);
#pragma unroll
            for (int kk = 0; kk < DK_BK; kk += 4) {
                double af[4], bf[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    af[m] = lds[buf][0][kk + fk][wi + 16 * m + fi];
                    bf[m] = lds[buf][1][kk + fk][wj + 16 * m + fi];
                }
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0);
            }
            if (kt + 1 < ntile_k) {
                lstore(buf ^ 1);
                __syncthreads();
            }
        }
    }

    // epilogue: C/D layout of the f64 16x16 MFMA: col = lane & 15, row = (lane >> 4) + 4 * r
    if (EPI) {
        // Four passes of 32 columns x 128 rows through the staging buffers (4096 of their 4352 doubles): pass b takes the b-th
        // 16-column block of every wave.  T[c][row]; the 4-row group row / 4 is stored at group ^ c, so that the 16 columns a
        // write instruction touches spread over the banks and a reader walking down a column meets every group once.
        double* T = &lds[0][0][0][0];
        const bool vec = (g.ldc & 1) == 0 && (reinterpret_cast<size_t>(g.C) & 15) == 0;
        const int cw = (wid & 1) * 16 + (lane & 15);
        __syncthreads();                                                // the last K tile has been read by every wave
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (b) __syncthreads();                                     // the previous pass has been read
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wi + 16 * a + 4 * r;                // + (lane >> 4): stays inside the 4-row group
                    T[cw * DK_BM + ((((row >> 2) ^ cw) & 31) << 2) + (lane >> 4)] = acc[a][b][r];
                }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int idx = i * DK_THREADS + tid;
                const int c = idx >> 6, r2 = idx & 63;
                const int rowl = 2 * r2;
                const double2 t = *reinterpret_cast<const double2*>(&T[c * DK_BM + ((((rowl >> 2) ^ c) & 31) << 2) + (rowl & 3)]);
                const int col = J0 + (c >> 4) * 64 + b * 16 + (c & 15), row = I0 + rowl;
                if (col >= g.N || row >= g.M) continue;
                double* dst = g.C + (size_t)col * g.ldc + row;
                const double tv[2] = {t.x, t.y};
                if (vec && row + 1 < g.M) {
                    double2 c2 = make_double2(0.0, 0.0);
                    if (g.beta != 0.0) c2 = *reinterpret_cast<const double2*>(dst);
                    const double cv[2] = {c2.x, c2.y};
                    double o[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        double v = g.alpha * tv[e];
                        if (g.beta != 0.0) v += g.beta * cv[e];
                        o[e] = v;
                    }
                    *reinterpret_cast<double2*>(dst) = make_double2(o[0], o[1]);
                } else {
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        if (row + e < g.M) {
                            double v = g.alpha * tv[e];
                            if (g.beta != 0.0) v += g.beta * dst[e];
                            dst[e] = v;
                        }
                }
            }
        }
        return;
    }
    const bool offdiag = I0 != J0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int col = J0 + wj + 16 * b + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = I0 + wi + 16 * a + (lane >> 4) + 4 * r;
                if (row < g.M && col < g.N) {
                    double v = g.alpha * acc[a][b][r];
                    double* dst = g.C + (size_t)col * g.ldc + row;
                    if (g.beta != 0.0) v += g.beta * *dst;
                    *dst = v;
                    if (LOWER && g.mirror && offdiag) g.C[(size_t)row * g.ldc + col] = v;
                }
            }
        }
}

static void launch_gemm_nt_f64(bool lower, const double* A, long long lda, const double* B, long long ldb, double* C, long long ldc,
                               int M, int N, int K, double alpha, double beta, bool mirror, bool kstart_row, hipStream_t st, bool kend_col,
                               int ksplit = 1, long long cstride = 0) {
    if (M <= 0 || N <= 0) return;
    GemmNTd g;
    g.ksplit = ksplit; g.cstride = cstride;
    g.kend_col = kend_col && !lower ? 1 : 0;
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.alpha = alpha; g.beta = beta; g.mirror = mirror ? 1 : 0; g.kstart_row = kstart_row ? 1 : 0;
    g.nbi = (M + DK_BM - 1) / DK_BM; g.nbj = (N + DK_BM - 1) / DK_BM;
    g.ntiles = lower ? g.nbi * (g.nbi + 1) / 2 : g.nbi * g.nbj;
    const int grid = (g.ntiles + 7) / 8 * 8;
    if (!mirror) {                                          // output tile handed over through LDS (whole column pieces); the mirrored store keeps the direct stores
        if (lower) hipLaunchKernelGGL((gemm_nt_mfma_f64_kernel<1, 1>), dim3(grid, ksplit), dim3(DK_THREADS), 0, st, g);
        else hipLaunchKernelGGL((gemm_nt_mfma_f64_kernel<0, 1>), dim3(grid), dim3(DK_THREADS), 0, st, g);
    } else if (lower) hipLaunchKernelGGL((gemm_nt_mfma_f64_kernel<1, 0>), dim3(grid), dim3(DK_THREADS), 0, st, g);
    else hipLaunchKernelGGL((gemm_nt_mfma_f64_kernel<0, 0>), dim3(grid), dim3(DK_THREADS), 0, st, g);
}

static void launch_gemm_nt_f64_plain(bool lower, const double* A, long long lda, const double* B, long long ldb, double* C, long long ldc,
                                     int M, int N, int K, double alpha, double beta, bool mirror, bool kstart_row, hipStream_t st) {
    launch_gemm_nt_f64(lower, A, lda, B, ldb, C, ldc, M, N, K, alpha, beta, mirror, kstart_row, st, false);
}

template <bool TRANSPOSE>
__global__ void __launch_bounds__(256)
pad_copy_f64_kernel(const double* __restrict__ in, long long ldi, int rows, int cols, double* __restrict__ out, long long ldo) {
    // out (ldo x K, zero padded) = in (rows x cols) or its transpose; 32 x 32 tiles through LDS
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    if (TRANSPOSE) {
        for (int c = ty; c < 32; c += 8) {
            const int r = r0 + tx, cc = c0 + c;
            tile[c][tx] = (r < rows && cc < cols) ? in[(size_t)cc * ldi + r] : 0.0;
        }
        __syncthreads();
        for (int c = ty; c < 32; c += 8) {
            const int orow = c0 + tx, ocol = r0 + c;          // out(orow, ocol) = in(ocol, orow)
            if (orow < cols && ocol < rows) out[(size_t)ocol * ldo + orow] = tile[tx][c];
        }
    } else {
        for (int c = ty; c < 32; c += 8) {
            const int r = r0 + tx, cc = c0 + c;
            if (r < rows && cc < cols) out[(size_t)cc * ldo + r] = in[(size_t)cc * ldi + r];
        }
    }
}

// C (both triangles) = A'A (atA, C is cols x cols) or A A' (C is rows x rows), A rows x cols column-major.
// out(i, j) = sum_s part[s](i, j) for i >= j, mirrored (both triangles)
__global__ void __launch_bounds__(256) sum_splits_mirror_f64_kernel(const double* __restrict__ part, long long ldp, long long stride, int nsplit,
                                                                    double* __restrict__ C, long long ldc, int m) {
    const int j = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < m && i >= j) {
        double v = 0.0;
        for (int s = 0; s < nsplit; ++s) v += part[(size_t)s * stride + (size_t)j * ldp + i];
        C[(size_t)j * ldc + i] = v;
        C[(size_t)i * ldc + j] = v;
    }
}

void gram_mfma_f64(const double* A, long long lda, int rows, int cols, bool atA, double* C, long long ldc, hipStream_t st) {
    const int M = atA ? cols : rows;          // order of C
    const int Kd = atA ? rows : cols;         // summation length
    const long long ldz = round_up(M, DK_BM);
    const int K = (int)round_up(Kd, DK_BK);
    DevBuf<double> Z((size_t)ldz * K);        // padded operand with the output index contiguous
    Z.zero(st);
    if (atA) hipLaunchKernelGGL((pad_copy_f64_kernel<true>), dim3((rows + 31) / 32, (cols + 31) / 32), dim3(256), 0, st, A, lda, rows, cols, Z.get(), ldz);
    else hipLaunchKernelGGL((pad_copy_f64_kernel<false>), dim3((rows + 31) / 32, (cols + 31) / 32), dim3(256), 0, st, A, lda, rows, cols, Z.get(), ldz);
    // The lower triangle's 128 x 128 tiles rarely fill whole rounds of the chip's 2 x 256 resident workgroups (order 5000: 820 tiles = 1.6
    // rounds, the second 60 % idle): the K range is cut into S shares inside ONE launch (tiles x S work items: S = 3 -> 4.8 rounds of a
    // third each), every share into its own copy of C, summed in share order afterwards (deterministic).  S = 1 keeps the single sweep.
    const int nb = (M + DK_BM - 1) / DK_BM;
    const long long tiles = (long long)nb * (nb + 1) / 2, slots = 2LL * device_info().num_cu;
    int S = 1;
    double best = (double)((tiles + slots - 1) / slots);
    for (int c = 2; c <= 4; ++c) {
        if (K / c < 4096) break;
        const double cost = (double)((tiles * c + slots - 1) / slots) / c;
        if (cost < best * 0.93) { best = cost; S = c; }
    }
    if (const char* e = option("GRAM64_KSPLIT")) { const int v = std::atoi(e); if (v >= 1 && v <= 8) S = v; }
    if (S == 1) {
        launch_gemm_nt_f64(true, Z.get(), ldz, Z.get(), ldz, C, ldc, M, M, K, 1.0, 0.0, true, false, st, false);
    } else {
        const long long stride = ldz * ldz;
        DevBuf<double> part((size_t)stride * S);
        launch_gemm_nt_f64(true, Z.get(), ldz, Z.get(), ldz, part.get(), ldz, M, M, K, 1.0, 0.0, false, false, st, false, S, stride);
        hipLaunchKernelGGL(sum_splits_mirror_f64_kernel, dim3((M + 255) / 256, M), dim3(256), 0, st, part.get(), ldz, stride, S, C, ldc, M);
        ADMM_HIP_CHECK(hipGetLastError());
        ADMM_HIP_CHECK(hipStreamSynchronize(st));
    }
    ADMM_HIP_CHECK(hipGetLastError());
    ADMM_HIP_CHECK(hipStreamSynchronize(st));     // Z is freed on return
}

// ---------------------------------------------------------------------------------------------- Cholesky + inverse (chol_inverse.h)
// A (lda >= round_up(n, 128), that many zero-padded columns allocated) -> A^-1, both triangles.
void spd_inverse_mfma_f64(double* A, long long lda, int n, hipStream_t st) {
    spd_inverse_blocked<double>(A, lda, n, st, launch_gemm_nt_f64_plain);
}

// A -> Cholesky factor L (lower triangle) in place; returns U = L^-T (lda x round_up(n, 128), upper triangular).
DevBuf<double> cholesky_linvt_mfma_f64(double* A, long long lda, int n, hipStream_t st) {
    return cholesky_linvt_blocked<double>(A, lda, n, st, launch_gemm_nt_f64_plain);
}

// C (M x N, ldc) = A B' for operands with the output index contiguous (rows readable up to the next multiple of 128,
// K a multiple of 8).
void gemm_nt_f64(const double* A, long long lda, const double* B, long long ldb, double* C, long long ldc, int M, int N, int K, hipStream_t st, bool b_lower) {
    launch_gemm_nt_f64(false, A, lda, B, ldb, C, ldc, M, N, K, 1.0, 0.0, false, false, st, b_lower);
}

}  // namespace admm
