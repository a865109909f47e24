// The tall solver's Gram X'X on the bf16 matrix cores with a 3-way split of every fp32 entry (round 5).
//
// Replaces Linalg::cross_prod_lower (BlasWrapper.h:89-112, called from ADMMLassoTall.h:191-192) for deep-K lower-triangle Grams;
// syrk_mfma.hip (exact fp32 MFMA, 157 TF/s peak) stays for everything else.  Why: once the loop runs at the bandwidth roofline the
// one-time setup decides sec-to-eps, 75 of its 116 ms were this product at 0.84 of the fp32 matrix rate, and the bf16 matrix rate of
// the chip is 16 x higher.  An fp32 value is the exact sum of three bf16 terms up to 2^-27 of itself,
//     x = h + m + l,   h = bf16(x),  m = bf16(x - h),  l = bf16(x - h - m)        (the two subtractions are exact in fp32)
// and of the nine cross products of two such sums the six with weight >= 2^-18 are kept:
//     x y ~= h h' + h m' + m h' + m m' + h l' + l h'                              (dropped: m l' + l m' + l l' <= 2^-25 |x y|)
// -- each a bf16 x bf16 product, EXACT in the fp32 accumulator of v_mfma_f32_32x32x16_bf16.  Six bf16 instructions of K = 16 replace
// eight fp32 instructions of K = 2: 768 instead of 2048 matrix-pipe cycles per 64 x 64 x 16 block of a wave.  The result carries
// the rounding of an fp32 accumulation over K like the fp32 kernel's (tests/test_gpu_kernels.py holds both to the same bound
// against float64).
//
// Data layout: the operand Z = X' is stored as three bf16 planes [plane][k / 8][i][8]: for a fixed group of eight k the eight values
// of one output index i are 16 contiguous bytes = one lane's A / B fragment of the instruction (lane l: index l % 32, k group l / 32),
// and 128 consecutive i are 2 KB = one row of a tile's K slice -- global loads, LDS stores and fragment reads are all 16 bytes per
// lane, consecutive lanes consecutive addresses.  A workgroup = 4 waves (2 x 2), 128 x 128 outputs, K tiles of 16 double buffered
// (48 KB of LDS, 2 workgroups per CU), XCD-local square tile order as in syrk_mfma.hip; lower tiles + mirrored store.
// The pipelined host-input setup (prep.hip) uses the rectangular mode on block rows: same K order per element, bit-identical.
//
// Second variant, the DEFAULT (ADMM_HIP_GRAM_SPLIT=f16x2): two fp16 planes, x = 2^e_j (h + l) with h = fp16(x 2^-e_j), l = fp16(rest),
// 2^e_j <= max|x_j| < 2^(e_j+1) per column (exact scaling; fp16 has no range to spare, bf16 has fp32's), three products
// h h' + h l' + l h' (dropped l l' <= 2^-22 |x y|; the operand is represented to 2^-23 of itself, i.e. the Gram is that of X with
// every entry moved by less than one fp32 ulp -- random, averaging out over K -- while the fp32 accumulation over K, the error that
// dominates either kernel's result, is unchanged).  Half the matrix instructions of the bf16 form: measured on C2 in
// profiles/r05_gram_split.md.  The bf16 form needs no scaling and stays selectable (ADMM_HIP_GRAM_SPLIT=bf16x3).
#include "prep.h"
#include "device_utils.h"
#include <string>

namespace admm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int GB_BM = 128;
constexpr int GB_BK = 16;
constexpr int GB_KPAD = 6 * GB_BK;      // K is zero-padded to whole groups of six K tiles (the kernel's unrolled ring)
constexpr int GB_THREADS = 256;

struct GramB3 {
    const uint4* Z[3];           // planes h, m, l (bf16x3) or h, l (f16x2): [k / 8][ldz] entries of 8 values (16 bytes)
    long long ldz;               // entries per k group (multiple of 128)
    const uint4* Zb[3];          // second operand (the same planes for the Gram; row offsets differ in the block-row mode)
    const float* rs;             // f16x2: 2^e_i per output index (the planes hold x 2^-e_i), applied to the result; NULL: none
    int ioff, joff;              // first output row of operand A / B in units of entries
    float* C; long long ldc;
    int M, N, K;                 // K multiple of 16
    int kt0, kt1;                // K tiles [kt0, kt1) of this launch; kt0 > 0: accumulate into C (the launches of one product run back to back)
    int lower, mirror;
    int nbi, nbj, ntiles;
    const int* tilemap;          // optional [ntiles]: bi << 16 | bj
};

__device__ __forceinline__ void gb_tri_decode(int t, int& bi, int& bj) {
    int b = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while ((long long)(b + 1) * (b + 2) / 2 <= t) ++b;
    while ((long long)b * (b + 1) / 2 > t) --b;
    bi = b; bj = t - b * (b + 1) / 2;
}

__device__ __forceinline__ bf16x8 gb_frag(const uint4& u) {
    bf16x8 f;
    __builtin_memcpy(&f, &u, 16);
    return f;
}
__device__ __forceinline__ halfx8 gb_hfrag(const uint4& u) {
    halfx8 f;
    __builtin_memcpy(&f, &u, 16);
    return f;
}

// Epilogue (round 6): the output tile leaves through LDS in whole column pieces.  In the matrix-core register layout a lane holds 16
// values of ONE column at rows 8 apart and the 32 lanes of a half-wave hold 32 different columns: stored directly, every instruction
// touched 64 different cache lines with 4 bytes each, and the accumulation over the K ranges read them back the same way -- 0.38 of a
// launch's 2.56 ms at C2 (scripts/gram_probe.hip runs the same loop without an epilogue).  Now: two passes of 64 columns; the two waves
// that own them write their accumulators (scaled by the exact powers of two of the fp16 form) into a 64 x 128 float tile (float4
// index XOR column: the 32 columns of an instruction spread over 8 banks), all 256 threads then add the earlier K ranges and store
// float4 pieces down the columns (512 contiguous bytes per column), and -- last launch of a lower-triangle product only -- the final
// values go back into the tile and are read across for the mirrored store (contiguous along the other index).  Same value per
// element, bit for bit (scale, then add the earlier ranges, as before).
__device__ __forceinline__ void gb_store_tile(const GramB3& g, const floatx16 (&acc)[2][2], float* T, int I0, int J0, int wi, int wj, int tid) {
    const int lane = tid & 63;
    const bool diag = g.lower && I0 == J0;
    for (int h = 0; h < 2; ++h) {
        __syncthreads();                                     // the tile is free: K loop finished / previous half consumed
        if (wj == 64 * h) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int c = b * 32 + (lane & 31);
                    const int col = J0 + 64 * h + c;
                    const float cs = (g.rs != nullptr && col < g.N) ? g.rs[g.joff + col] : 1.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rl = wi + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        const int row = I0 + rl;
                        float v = acc[a][b][r];
                        if (g.rs != nullptr && row < g.M && col < g.N) v *= g.rs[g.ioff + row] * cs;          // powers of two: exact
                        T[c * 128 + ((((rl >> 2) ^ (c & 31)) << 2) | (rl & 3))] = v;
                    }
                }
        }
        __syncthreads();
        const bool fin = g.lower && g.mirror;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int piece = it * GB_THREADS + tid;
            const int c = piece >> 5, q = piece & 31;
            const int col = J0 + 64 * h + c, row0 = I0 + 4 * q;
            if (col >= g.N || row0 >= g.M) continue;
            float* tp = T + c * 128 + ((q ^ (c & 31)) << 2);
            float4 v = *reinterpret_cast<const float4*>(tp);
            float* cp = g.C + (size_t)col * g.ldc + row0;
            if (row0 + 3 < g.M && (!diag || row0 >= col)) {       // a whole piece on / below the diagonal (or no diagonal in this tile)
                if (g.kt0 > 0) { const float4 o = *reinterpret_cast<const float4*>(cp); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                *reinterpret_cast<float4*>(cp) = v;
            } else {
                float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int row = row0 + k;
                    if (row >= g.M || (diag && row < col)) continue;        // lower mode: the upper triangle is ALWAYS the mirror of the lower one
                    if (g.kt0 > 0) e[k] += cp[k];
                    cp[k] = e[k];
                }
                v = make_float4(e[0], e[1], e[2], e[3]);
            }
            if (fin) *reinterpret_cast<float4*>(tp) = v;          // the final values, for the mirrored store below
        }
        if (fin) {
            __syncthreads();
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int piece = it * GB_THREADS + tid;
                const int rl = piece >> 4, c4 = piece & 15;           // row of the tile, group of four columns of this half
                const int row = I0 + rl, col0 = J0 + 64 * h + 4 * c4;
                if (row >= g.M || col0 >= g.N) continue;
                float e[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { const int c = 4 * c4 + k; e[k] = T[c * 128 + ((((rl >> 2) ^ (c & 31)) << 2) | (rl & 3))]; }
                float* mp = g.C + (size_t)row * g.ldc + col0;
                if (col0 + 3 < g.N && (!diag || row > col0 + 3)) {
                    *reinterpret_cast<float4*>(mp) = make_float4(e[0], e[1], e[2], e[3]);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (col0 + k < g.N && (!diag || row > col0 + k)) mp[k] = e[k];
                }
            }
        }
    }
}

template <int NPL>      // 3: bf16 planes, six products; 2: fp16 planes, three products
__global__ void __launch_bounds__(GB_THREADS, 2)
gram_split_kernel(GramB3 g) {
    __shared__ uint4 lds[2][2][NPL][2][GB_BM];           // [buffer][A/B][plane][k group][i]: 48 KB (bf16x3) / 32 KB (f16x2)
    const int per = (g.ntiles + 7) / 8;
    const int w_idx = (blockIdx.x % 8) * per + blockIdx.x / 8;          // XCD b % 8 walks a contiguous range of the tile list
    if (w_idx >= g.ntiles) return;
    int bi, bj;
    if (g.tilemap != nullptr) { const int m = g.tilemap[w_idx]; bi = m >> 16; bj = m & 0xffff; }
    else if (g.lower) gb_tri_decode(w_idx, bi, bj);
    else { bi = w_idx % g.nbi; bj = w_idx / g.nbi; }
    const int I0 = bi * GB_BM, J0 = bj * GB_BM;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wi = (wid >> 1) * 64, wj = (wid & 1) * 64;
    const int s_kg = tid >> 7, s_i = tid & 127;          // staging: one 16-byte entry per thread, plane and operand
    const size_t kstep = (size_t)2 * g.ldz;              // entries per K tile (two k groups)
    const size_t offA = (size_t)s_kg * g.ldz + g.ioff + I0 + s_i + (size_t)g.kt0 * kstep;
    const size_t offB = (size_t)s_kg * g.ldz + g.joff + J0 + s_i + (size_t)g.kt0 * kstep;

    floatx16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // Global loads run THREE K tiles ahead of the matrix cores through a ring of three register sets (tile t travels through set
    // t % 3 and LDS buffer t % 2).  One tile ahead -- the fp32 kernel's distance -- left this kernel at half of the matrix rate: a
    // K tile takes 1.2 us here (2.7 x faster than there), 38 % of the operand requests miss the XCD's L2 (PMC: 183 GB fetched for
    // 487 GB requested) and come back from the Infinity Cache / HBM later than that.
    uint4 s0a0, s0a1, s0a2, s0b0, s0b1, s0b2, s1a0, s1a1, s1a2, s1b0, s1b1, s1b2, s2a0, s2a1, s2a2, s2b0, s2b1, s2b2;
#define GB_GLOAD(S, t)                                          \
    {                                                           \
        const size_t k_ = (size_t)(t) * kstep;                  \
        S##a0 = g.Z[0][offA + k_]; S##a1 = g.Z[1][offA + k_];                                   \
        S##b0 = g.Zb[0][offB + k_]; S##b1 = g.Zb[1][offB + k_];                                 \
        if (NPL == 3) { S##a2 = g.Z[2][offA + k_]; S##b2 = g.Zb[2][offB + k_]; }                \
    }
#define GB_LSTORE(S, buf)                                       \
    {                                                           \
        lds[buf][0][0][s_kg][s_i] = S##a0; lds[buf][0][1][s_kg][s_i] = S##a1;                   \
        lds[buf][1][0][s_kg][s_i] = S##b0; lds[buf][1][1][s_kg][s_i] = S##b1;                   \
        if (NPL == 3) { lds[buf][0][NPL - 1][s_kg][s_i] = S##a2; lds[buf][1][NPL - 1][s_kg][s_i] = S##b2; }         \
    }
    const int fk = lane >> 5, fi = lane & 31;
    auto compute = [&](int buf) {
        uint4 ua[2][NPL], ub[2][NPL];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
                ua[a][pl] = lds[buf][0][pl][fk][wi + a * 32 + fi];
                ub[a][pl] = lds[buf][1][pl][fk][wj + a * 32 + fi];
            }
        // the kept products, smallest weights first; the four accumulators of the wave between two dependent instructions
        if constexpr (NPL == 3) {
#define GB_TERM(PA, PB)                                                                                                               \
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gb_frag(ua[0][PA]), gb_frag(ub[0][PB]), acc[0][0], 0, 0, 0);           \
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gb_frag(ua[0][PA]), gb_frag(ub[1][PB]), acc[0][1], 0, 0, 0);           \
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gb_frag(ua[1][PA]), gb_frag(ub[0][PB]), acc[1][0], 0, 0, 0);           \
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gb_frag(ua[1][PA]), gb_frag(ub[1][PB]), acc[1][1], 0, 0, 0);
            GB_TERM(0, 2) GB_TERM(2, 0) GB_TERM(1, 1) GB_TERM(0, 1) GB_TERM(1, 0) GB_TERM(0, 0)
#undef GB_TERM
        } else {
#define GB_TERM(PA, PB)                                                                                                               \
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gb_hfrag(ua[0][PA]), gb_hfrag(ub[0][PB]), acc[0][0], 0, 0, 0);          \
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gb_hfrag(ua[0][PA]), gb_hfrag(ub[1][PB]), acc[0][1], 0, 0, 0);          \
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gb_hfrag(ua[1][PA]), gb_hfrag(ub[0][PB]), acc[1][0], 0, 0, 0);          \
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gb_hfrag(ua[1][PA]), gb_hfrag(ub[1][PB]), acc[1][1], 0, 0, 0);
            GB_TERM(0, 1) GB_TERM(1, 0) GB_TERM(0, 0)
#undef GB_TERM
        }
    };
    const int ntile_k = g.kt1 - g.kt0;
    // step T: request tile T + 3 (its register set held tile T, which is in LDS), run tile T from LDS[T % 2], then write tile T + 1 to the
    // other buffer -- every wave left that buffer before the barrier that ended step T - 1 -- and close the step with ONE barrier.
    // STRAIGHT-LINE: the launcher makes the K tiles of a launch a multiple of six (K is zero-padded to 96), requests past the end are
    // clamped to the last tile and the final store goes to a buffer nobody reads -- with the requests and stores under `if`s the
    // compiler's wait-counter pass gave up at the joins and put s_waitcnt vmcnt(0) in front of every request (no prefetch at all).
#define GB_STEP(T, SLOAD, SSTORE, BUFC, BUFS)                   \
    {                                                           \
        GB_GLOAD(SLOAD, min((T) + 3, ntile_k - 1))              \
        compute(BUFC);                                          \
        GB_LSTORE(SSTORE, BUFS)                                 \
        __syncthreads();                                        \
    }
    if (ntile_k > 0) {
        GB_GLOAD(s0, 0)
        GB_GLOAD(s1, min(1, ntile_k - 1))
        GB_GLOAD(s2, min(2, ntile_k - 1))
        GB_LSTORE(s0, 0)
        __syncthreads();
        for (int kt = 0; kt < ntile_k; kt += 6) {
            GB_STEP(kt, s0, s1, 0, 1)
            GB_STEP(kt + 1, s1, s2, 1, 0)
            GB_STEP(kt + 2, s2, s0, 0, 1)
            GB_STEP(kt + 3, s0, s1, 1, 0)
            GB_STEP(kt + 4, s1, s2, 0, 1)
            GB_STEP(kt + 5, s2, s0, 1, 0)
        }
    }
#undef GB_STEP
#undef GB_GLOAD
#undef GB_LSTORE
    // C/D layout of the 32x32 instruction: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    gb_store_tile(g, acc, reinterpret_cast<float*>(&lds[0][0][0][0][0]), I0, J0, wi, wj, tid);
}

// Round 6: 256 x 256 macro-tiles for the lower-triangle product of the fp16 form.  scripts/gram_probe.hip takes the 128 x 128 loop apart:
// matrix instructions alone 1.11 ms per C2 launch (0.92 of the fp16 peak), + fragment reads 1.14, + LDS stores and the barrier 1.37,
// + the operand requests 1.83 even when every one of them hits the L2, 2.19 with the real traffic -- a 128 x 128 tile pulls 16 KB
// through the L2 -> L1 path per 1.6 Mflop (96 flop per byte): at the matrix rate that is ~27 TB/s out of the L2s, which they do not
// deliver.  A 256 x 256 tile stages 512 operand rows for four times the products (192 flop per byte): the same loop runs at 0.71 of
// the peak in whole rounds.  8 waves as 4 x 2, a wave owns 64 rows x 128 columns (2 x 4 matrix-core tiles, 24 instructions per K
// step, 128 accumulator registers), one workgroup per CU, 64 KB of LDS; same ring of three register sets, same products in the same
// K order per output element as gram_split_kernel<2> -- bit-identical, so the block-row mode of the pipelined host-input setup keeps
// the 128 x 128 kernel (tests/test_gpu_gram.py holds the two setups bit for bit).  820 tiles on 256 workgroups are 3.2 rounds run
// as 4: 1.82 ms per launch against 2.19 (the tail is what is left on the table).
constexpr int GB2_BM = 256;
constexpr int GB2_THREADS = 512;

__device__ __forceinline__ void gb_store_tile256(const GramB3& g, const floatx16 (&acc)[2][4], float* T, int I0, int J0, int wi, int wj, int tid) {
    const int lane = tid & 63;
    const bool diag = g.lower && I0 == J0;
    const bool fin = g.lower && g.mirror;
    for (int h = 0; h < 4; ++h) {                               // four passes of 64 columns through a 64 x 256 float tile (64 KB)
        __syncthreads();
        if (wj == 128 * (h >> 1)) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    const int b = 2 * (h & 1) + bb;
                    const int c = bb * 32 + (lane & 31);
                    const int col = J0 + 64 * h + c;
                    const float cs = (g.rs != nullptr && col < g.N) ? g.rs[g.joff + col] : 1.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rl = wi + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        const int row = I0 + rl;
                        // (b is a compile-time index after unrolling: the accumulators stay in registers)
                        float v = (h & 1) ? acc[a][2 + bb][r] : acc[a][bb][r];
                        (void)b;
                        if (g.rs != nullptr && row < g.M && col < g.N) v *= g.rs[g.ioff + row] * cs;
                        T[c * 256 + ((((rl >> 2) ^ (c & 31)) << 2) | (rl & 3))] = v;
                    }
                }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int piece = it * GB2_THREADS + tid;
            const int c = piece >> 6, q = piece & 63;
            const int col = J0 + 64 * h + c, row0 = I0 + 4 * q;
            if (col >= g.N || row0 >= g.M) continue;
            float* tp = T + c * 256 + ((q ^ (c & 31)) << 2);
            float4 v = *reinterpret_cast<const float4*>(tp);
            float* cp = g.C + (size_t)col * g.ldc + row0;
            if (row0 + 3 < g.M && (!diag || row0 >= col)) {
                if (g.kt0 > 0) { const float4 o = *reinterpret_cast<const float4*>(cp); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                *reinterpret_cast<float4*>(cp) = v;
            } else {
                float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int row = row0 + k;
                    if (row >= g.M || (diag && row < col)) continue;
                    if (g.kt0 > 0) e[k] += cp[k];
                    cp[k] = e[k];
                }
                v = make_float4(e[0], e[1], e[2], e[3]);
            }
            if (fin) *reinterpret_cast<float4*>(tp) = v;
        }
        if (fin) {
            __syncthreads();
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int piece = it * GB2_THREADS + tid;
                const int rl = piece >> 4, c4 = piece & 15;
                const int row = I0 + rl, col0 = J0 + 64 * h + 4 * c4;
                if (row >= g.M || col0 >= g.N) continue;
                float e[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { const int c = 4 * c4 + k; e[k] = T[c * 256 + ((((rl >> 2) ^ (c & 31)) << 2) | (rl & 3))]; }
                float* mp = g.C + (size_t)row * g.ldc + col0;
                if (col0 + 3 < g.N && (!diag || row > col0 + 3)) {
                    *reinterpret_cast<float4*>(mp) = make_float4(e[0], e[1], e[2], e[3]);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (col0 + k < g.N && (!diag || row > col0 + k)) mp[k] = e[k];
                }
            }
        }
    }
}

__global__ void __launch_bounds__(GB2_THREADS, 1)
gram_split256_kernel(GramB3 g) {
    __shared__ uint4 lds[2][2][2][2][GB2_BM];             // [buffer][A/B][plane][k group][i]: 64 KB
    const int per = (g.ntiles + 7) / 8;
    const int w_idx = (blockIdx.x % 8) * per + blockIdx.x / 8;
    if (w_idx >= g.ntiles) return;
    int bi, bj;
    if (g.tilemap != nullptr) { const int m = g.tilemap[w_idx]; bi = m >> 16; bj = m & 0xffff; }
    else gb_tri_decode(w_idx, bi, bj);
    const int I0 = bi * GB2_BM, J0 = bj * GB2_BM;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wi = (wid >> 1) * 64, wj = (wid & 1) * 128;
    const int s_kg = tid >> 8, s_i = tid & 255;
    const size_t kstep = (size_t)2 * g.ldz;
    const size_t offA = (size_t)s_kg * g.ldz + g.ioff + I0 + s_i + (size_t)g.kt0 * kstep;
    const size_t offB = (size_t)s_kg * g.ldz + g.joff + J0 + s_i + (size_t)g.kt0 * kstep;

    floatx16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    uint4 s0a0, s0a1, s0b0, s0b1, s1a0, s1a1, s1b0, s1b1;      // a ring of TWO register sets (three spill: 128 accumulator registers per lane)
#define G2_GLOAD(S, t)                                          \
    {                                                           \
        const size_t k_ = (size_t)(t) * kstep;                  \
        S##a0 = g.Z[0][offA + k_]; S##a1 = g.Z[1][offA + k_];                                   \
        S##b0 = g.Zb[0][offB + k_]; S##b1 = g.Zb[1][offB + k_];                                 \
    }
#define G2_LSTORE(S, buf)                                       \
    {                                                           \
        lds[buf][0][0][s_kg][s_i] = S##a0; lds[buf][0][1][s_kg][s_i] = S##a1;                   \
        lds[buf][1][0][s_kg][s_i] = S##b0; lds[buf][1][1][s_kg][s_i] = S##b1;                   \
    }
    const int fk = lane >> 5, fi = lane & 31;
    auto compute = [&](int buf) {
        uint4 ua[2][2], ub[4][2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
            for (int a = 0; a < 2; ++a) ua[a][pl] = lds[buf][0][pl][fk][wi + a * 32 + fi];
#pragma unroll
            for (int b = 0; b < 4; ++b) ub[b][pl] = lds[buf][1][pl][fk][wj + b * 32 + fi];
        }
        // the kept products in the order of gram_split_kernel<2> (per output element: h l', l h', h h' of every K step)
#define G2_TERM(PA, PB)                                                                                                                  \
        _Pragma("unroll") for (int a = 0; a < 2; ++a) _Pragma("unroll") for (int b = 0; b < 4; ++b)                                        \
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gb_hfrag(ua[a][PA]), gb_hfrag(ub[b][PB]), acc[a][b], 0, 0, 0);
        G2_TERM(0, 1) G2_TERM(1, 0) G2_TERM(0, 0)
#undef G2_TERM
    };
    const int ntile_k = g.kt1 - g.kt0;
#define G2_STEP(T, SLOAD, SSTORE, BUFC, BUFS)                   \
    {                                                           \
        G2_GLOAD(SLOAD, min((T) + 2, ntile_k - 1))              \
        compute(BUFC);                                          \
        G2_LSTORE(SSTORE, BUFS)                                 \
        __syncthreads();                                        \
    }
    if (ntile_k > 0) {
        // step T: request tile T + 2 into the set that held tile T (in LDS since the end of step T - 1), run tile T from LDS[T % 2], write
        // tile T + 1 to the other buffer, one barrier (the launcher makes the K tiles of a launch a multiple of six, hence even)
        G2_GLOAD(s0, 0)
        G2_GLOAD(s1, min(1, ntile_k - 1))
        G2_LSTORE(s0, 0)
        __syncthreads();
        for (int kt = 0; kt < ntile_k; kt += 2) {
            G2_STEP(kt, s0, s1, 0, 1)
            G2_STEP(kt + 1, s1, s0, 1, 0)
        }
    }
#undef G2_STEP
#undef G2_GLOAD
#undef G2_LSTORE
    gb_store_tile256(g, acc, reinterpret_cast<float*>(&lds[0][0][0][0][0]), I0, J0, wi, wj, tid);
}

// Split columns [0, nc) of X (rows x nc, column-major, ld ldx, rows beyond `rows` zero up to a multiple of 8) into the planes of
// Z = X': entry (k group kg, index i0 + j) of plane pl holds rows 8 kg .. 8 kg + 7 of column j.  A workgroup transposes a
// 64-row x 64-column piece through LDS so that the reads run down the columns of X and the writes along i.
__device__ __forceinline__ unsigned short gb_bf16_rn(float x) {      // round to nearest even (finite inputs)
    const unsigned u = __float_as_uint(x);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ unsigned short gb_f16_bits(_Float16 h) {
    unsigned short u;
    __builtin_memcpy(&u, &h, 2);
    return u;
}
template <int NPL>
__global__ void __launch_bounds__(256)
split_kernel(const float* __restrict__ X, long long ldx, int rows, int nc, uint4* __restrict__ Z0, uint4* __restrict__ Z1, uint4* __restrict__ Z2,
             long long ldz, int i0, int nkg, const float* __restrict__ rinv) {
    __shared__ float tile[64][65];
    const int R0 = blockIdx.x * 64, J0 = blockIdx.y * 64;
    const int tid = threadIdx.x;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int q = tid + 256 * u;
        const int col = q >> 4, r4 = (q & 15) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (J0 + col < nc && R0 + r4 < rows) v = *reinterpret_cast<const float4*>(X + (size_t)(J0 + col) * ldx + R0 + r4);     // (ldx is a multiple of 32: whole float4 inside the column)
        tile[col][r4] = v.x; tile[col][r4 + 1] = v.y; tile[col][r4 + 2] = v.z; tile[col][r4 + 3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int col = tid & 63, kgl = (tid >> 6) + 4 * u;
        const int kg = R0 / 8 + kgl;
        if (J0 + col >= nc || kg >= nkg) continue;
        unsigned short h[8], m[8], l[8];
        const float sc = NPL == 2 ? rinv[i0 + J0 + col] : 1.f;           // 2^-e_j (exact)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int r = R0 + kgl * 8 + e;
            const float x = r < rows ? tile[col][kgl * 8 + e] * sc : 0.f;
            if constexpr (NPL == 3) {
                h[e] = gb_bf16_rn(x);
                const float r1 = x - __uint_as_float((unsigned)h[e] << 16);
                m[e] = gb_bf16_rn(r1);
                const float r2 = r1 - __uint_as_float((unsigned)m[e] << 16);
                l[e] = gb_bf16_rn(r2);
            } else {
                const _Float16 hh = (_Float16)x;                          // round to nearest even; |x| < 2: no overflow
                const _Float16 ll = (_Float16)(x - (float)hh);             // exact difference, then rounded (subnormals kept)
                h[e] = gb_f16_bits(hh); m[e] = gb_f16_bits(ll);
            }
        }
        auto pack = [](const unsigned short (&s)[8]) {
            return make_uint4((unsigned)s[0] | ((unsigned)s[1] << 16), (unsigned)s[2] | ((unsigned)s[3] << 16),
                              (unsigned)s[4] | ((unsigned)s[5] << 16), (unsigned)s[6] | ((unsigned)s[7] << 16));
        };
        const size_t o = (size_t)kg * ldz + i0 + J0 + col;
        Z0[o] = pack(h); Z1[o] = pack(m);
        if (NPL == 3) Z2[o] = pack(l);
    }
}

// f16x2: 2^e_j with 2^e_j <= max|x_j| < 2^(e_j + 1) (1 for a null column) and its reciprocal, per column -- exact scale factors
__global__ void __launch_bounds__(256)
colscale_kernel(const float* __restrict__ X, long long ldx, int rows, float* __restrict__ rs, float* __restrict__ rinv) {
    __shared__ float red[4];
    const float* c = X + (size_t)blockIdx.x * ldx;
    float m = 0.f;
    for (int i = threadIdx.x; i < rows; i += 256) m = fmaxf(m, fabsf(c[i]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        int e = 0;
        if (m > 0.f && m < 3.0e38f) (void)frexpf(m, &e), e -= 1;       // m = f 2^(e+1), f in [0.5, 1)  ->  2^e <= m < 2^(e+1)
        e = max(-100, min(100, e));
        rs[blockIdx.x] = ldexpf(1.f, e);
        rinv[blockIdx.x] = ldexpf(1.f, -e);
    }
}

int gram_split_mode() {
    const char* e = option("GRAM_SPLIT");       // read per call (the A/B tests flip it inside one process)
    if (e == nullptr) return 2;
    const std::string v(e);
    if (v == "0" || v == "fp32") return 0;
    if (v == "bf16x3") return 3;
    return 2;
}

void GramSplit3::alloc(int order, int kdepth, hipStream_t st) {
    M = order;
    npl = gram_split_mode() == 3 ? 3 : 2;
    ldz = round_up(order, GB2_BM);                        // whole 256-row macro-tiles (the 128 x 128 kernel needs whole 128s)
    nkg = (int)(round_up(kdepth, GB_KPAD) / 8);
    planes.alloc((size_t)npl * nkg * ldz * 8);
    planes.zero(st);
    if (npl == 2) { rs.alloc(2 * (size_t)ldz); rs.zero(st); }
}

void GramSplit3::split_cols(const float* X, long long ldx, int rows, int c0, int nc, hipStream_t st) {
    uint4* base = reinterpret_cast<uint4*>(planes.get());
    const size_t pl = (size_t)nkg * ldz;
    const dim3 grid((rows + 63) / 64, (nc + 63) / 64);
    if (npl == 3) {
        hipLaunchKernelGGL((split_kernel<3>), grid, dim3(256), 0, st, X, ldx, rows, nc, base, base + pl, base + 2 * pl, ldz, c0, nkg, (const float*)nullptr);
    } else {
        hipLaunchKernelGGL(colscale_kernel, dim3(nc), dim3(256), 0, st, X, ldx, rows, rs.get() + c0, rs.get() + ldz + c0);
        hipLaunchKernelGGL((split_kernel<2>), grid, dim3(256), 0, st, X, ldx, rows, nc, base, base + pl, (uint4*)nullptr, ldz, c0, nkg, (const float*)(rs.get() + ldz));
    }
}

static void launch_gram_b3(const GramSplit3& z, int ioff, int joff, float* C, long long ldc, int M, int N, bool lower, const int* tilemap, int ntiles_listed, hipStream_t st,
                           bool big = false, const int* tail = nullptr, int ntail = 0) {
    GramB3 g;
    const uint4* base = reinterpret_cast<const uint4*>(z.planes.get());
    const size_t pl = (size_t)z.nkg * z.ldz;
    for (int k = 0; k < 3; ++k) { g.Z[k] = base + std::min(k, z.npl - 1) * pl; g.Zb[k] = g.Z[k]; }
    g.rs = z.npl == 2 ? z.rs.get() : nullptr;
    g.ldz = z.ldz; g.ioff = ioff; g.joff = joff; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = z.nkg * 8;
    g.lower = lower ? 1 : 0; g.mirror = lower ? 1 : 0;
    const int bm = big ? GB2_BM : GB_BM;
    g.nbi = (M + bm - 1) / bm; g.nbj = (N + bm - 1) / bm;
    g.ntiles = lower ? g.nbi * (g.nbi + 1) / 2 : g.nbi * g.nbj;
    g.tilemap = tilemap;
    if (tilemap != nullptr && ntiles_listed >= 0) g.ntiles = ntiles_listed;
    if (g.ntiles <= 0) return;
    // The K range goes out in several launches (C accumulated between them): the tiles that share operand panels in an XCD's L2 drift
    // apart in K inside one long launch -- at 2.7 x the fp32 kernel's pace the operand over-fetch (57-fold there) is what the launch
    // waits for -- and start together again with every launch.  ADMM_HIP_GRAM_B3_KTILES=<K tiles per launch> (A/B; 0 = one launch).
    const int nkt = g.K / GB_BK;
    int per_launch = 516;
    if (const char* e = option("GRAM_B3_KTILES")) per_launch = std::atoi(e);
    if (per_launch <= 0) per_launch = nkt;
    per_launch = (per_launch + 5) / 6 * 6;
    for (int kt = 0; kt < nkt; kt += per_launch) {
        g.kt0 = kt; g.kt1 = std::min(nkt, kt + per_launch);
        g.mirror = (lower && g.kt1 == nkt) ? 1 : 0;           // the mirrored store once, with the final values (it was written by every launch: 12 x 400 MB at C2)
        if (big) {
            hipLaunchKernelGGL(gram_split256_kernel, dim3((g.ntiles + 7) / 8 * 8), dim3(GB2_THREADS), 0, st, g);
            if (ntail > 0) {                                   // the macro-tiles that would make a straggling round, as 128 x 128 tiles (same values)
                GramB3 t = g;
                t.nbi = (M + GB_BM - 1) / GB_BM; t.nbj = (N + GB_BM - 1) / GB_BM;
                t.tilemap = tail; t.ntiles = ntail;
                hipLaunchKernelGGL((gram_split_kernel<2>), dim3((ntail + 7) / 8 * 8), dim3(GB_THREADS), 0, st, t);
            }
        } else if (z.npl == 3) hipLaunchKernelGGL((gram_split_kernel<3>), dim3((g.ntiles + 7) / 8 * 8), dim3(GB_THREADS), 0, st, g);
        else hipLaunchKernelGGL((gram_split_kernel<2>), dim3((g.ntiles + 7) / 8 * 8), dim3(GB_THREADS), 0, st, g);
    }
}

// C (both triangles, order x order) = Z Z' from the planes; tiles in the XCD-local square order of syrk_mfma.hip
void GramSplit3::gram_lower(float* C, long long ldc, const int* tilemap, int ntiles, hipStream_t st) const {
    launch_gram_b3(*this, 0, 0, C, ldc, M, M, true, tilemap, ntiles, st);
}
// the same with 256 x 256 macro-tiles (fp16 form only; `tilemap` lists 256-blocks): bit-identical to gram_lower
void GramSplit3::gram_lower256(float* C, long long ldc, const int* tilemap, int ntiles, hipStream_t st, const int* tail, int ntail) const {
    launch_gram_b3(*this, 0, 0, C, ldc, M, M, true, tilemap, ntiles, st, true, tail, ntail);
}
// block row: C[r0 : r0 + nr, 0 : r0 + nr] = Z[r0 : r0 + nr, :] Z[0 : r0 + nr, :]'  (r0 a multiple of 128)
void GramSplit3::gram_rows(int r0, int nr, float* C, long long ldc, hipStream_t st) const {
    launch_gram_b3(*this, r0, 0, C + r0, ldc, nr, r0 + nr, false, nullptr, -1, st);
}

}  // namespace admm
