// One-time Gram matrix C = X'X (fp32) on the matrix cores: the only GEMM-shaped work of the path.
//
// Replaces Linalg::cross_prod_lower (BlasWrapper.h:73-112; under the default NO_FLOAT_BLAS a
// single-threaded Eigen `triangularView<Lower>() = X'X`), called from ADMMLassoTall.h:191-192.
//
// Structure (CDNA4): X (n x p column-major) is first transposed to Z = X' (p x n, leading dimension
// padded to 128) so that for a fixed summation index k the operands of both factors are contiguous:
// C[i, j] = sum_k Z[i, k] Z[j, k].  One workgroup (4 waves, 2 x 2) owns a 128 x 128 tile of the
// LOWER triangle; each wave accumulates 64 x 64 = 2 x 2 MFMA tiles with v_mfma_f32_32x32x2_f32
// (exact fp32, 64 FLOP/clk/SIMD = the 157 TF/s fp32 peak; there is no xf32/TF32 on gfx950).
// K tiles of 16 are staged global -> registers -> LDS ([k][i] rows of 128 floats, so the MFMA
// fragment reads `lds[(kk + lane/32) * 128 + i0 + lane%32]` are bank-conflict free) and double
// buffered.  Off-diagonal tiles are written twice (tile and transposed tile) so that the result
// has both triangles, which the symmetric mat-vec and Lanczos expect.  Tile order is remapped so
// that each XCD (block id % 8) walks a contiguous range of the row-major triangle and re-uses its
// row panel from its own L2.  n * p * (p + 128) flop, compute bound.
#include "prep.h"

namespace admm {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int SK_BM = 128;      // tile rows / cols
constexpr int SK_BK = 16;       // K tile
constexpr int SK_THREADS = 256;

__global__ void __launch_bounds__(SK_THREADS, 2)
syrk_lower_mfma_kernel(const float* __restrict__ Z, long long ldz, int p, int nk /* padded n, multiple of SK_BK */,
                       float* __restrict__ C, long long ldc, const int2* __restrict__ tiles, int ntiles) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][SK_BK][SK_BM];      // [buffer][A/B][k][i]
    // XCD-aware remap: block b runs on XCD b % 8; give every XCD a contiguous range of tiles
    const int per = (ntiles + 7) / 8;
    const int t_idx = (blockIdx.x % 8) * per + blockIdx.x / 8;
    if (t_idx >= ntiles) return;
    const int2 t = tiles[t_idx];
    const int I0 = t.x * SK_BM, J0 = t.y * SK_BM;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wi = (wid >> 1) * 64, wj = (wid & 1) * 64;      // wave's 64 x 64 sub-tile

    // staging map: a K tile of one operand is 16 rows x 128 floats = 512 float4; 2 per thread
    const int s_row0 = tid >> 5;              // 0..7   (k row), second load: +8
    const int s_col = (tid & 31) * 4;         // 0..124 (i)
    const float* gA = Z + (size_t)s_row0 * ldz + I0 + s_col;
    const float* gB = Z + (size_t)s_row0 * ldz + J0 + s_col;
    const size_t k8 = (size_t)8 * ldz;

    floatx16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float4 ra0, ra1, rb0, rb1;
    auto gload = [&](int k0) {
        const size_t off = (size_t)k0 * ldz;
        ra0 = *reinterpret_cast<const float4*>(gA + off);
        ra1 = *reinterpret_cast<const float4*>(gA + off + k8);
        rb0 = *reinterpret_cast<const float4*>(gB + off);
        rb1 = *reinterpret_cast<const float4*>(gB + off + k8);
    };
    auto lstore = [&](int buf) {
        *reinterpret_cast<float4*>(&lds[buf][0][s_row0][s_col]) = ra0;
        *reinterpret_cast<float4*>(&lds[buf][0][s_row0 + 8][s_col]) = ra1;
        *reinterpret_cast<float4*>(&lds[buf][1][s_row0][s_col]) = rb0;
        *reinterpret_cast<float4*>(&lds[buf][1][s_row0 + 8][s_col]) = rb1;
    };

    gload(0);
    lstore(0);
    __syncthreads();
    const int ntile_k = nk / SK_BK;
    const int fk = lane >> 5, fi = lane & 31;
    for (int kt = 0; kt < ntile_k; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < ntile_k) gload((kt + 1) * SK_BK);          // next tile in flight while this one computes
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
                if (row < p && col < p) {
                    const float v = acc[a][b][r];
                    C[(size_t)col * ldc + row] = v;
                    if (offdiag) C[(size_t)row * ldc + col] = v;     // mirrored tile: both triangles
                }
            }
        }
}

// out (cols x rows, ldo, zero padded) = in' for in (rows x cols, ldi); declared in prep.h
void gram_xtx_mfma_f32(const float* X, long long ldx, int n, int p, float* C, long long ldc, hipStream_t st) {
    const long long ldz = round_up(p, SK_BM);
    const int nk = round_up(n, SK_BK);
    DevBuf<float> Z((size_t)ldz * nk);
    Z.zero(st);
    transpose<float>(X, ldx, n, p, Z.get(), ldz, st);
    const int nb = (p + SK_BM - 1) / SK_BM;
    std::vector<int2> h;
    for (int bi = 0; bi < nb; ++bi)
        for (int bj = 0; bj <= bi; ++bj) h.push_back(make_int2(bi, bj));
    DevBuf<int2> tiles(h.size());
    ADMM_HIP_CHECK(hipMemcpyAsync(tiles.get(), h.data(), h.size() * sizeof(int2), hipMemcpyHostToDevice, st));
    const int ntiles = (int)h.size();
    const int grid = (ntiles + 7) / 8 * 8;
    hipLaunchKernelGGL(syrk_lower_mfma_kernel, dim3(grid), dim3(SK_THREADS), 0, st, Z.get(), ldz, p, nk, C, ldc, tiles.get(), ntiles);
    ADMM_HIP_CHECK(hipStreamSynchronize(st));      // Z and the tile list are freed on return
}

}  // namespace admm
