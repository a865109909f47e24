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
#include "prep.h"

namespace admm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int GB_BM = 128;
constexpr int GB_BK = 16;
constexpr int GB_THREADS = 256;

struct GramB3 {
    const uint4* Z[3];           // planes h, m, l: [k / 8][ldz] entries of 8 bf16 (16 bytes)
    long long ldz;               // entries per k group (multiple of 128)
    const uint4* Zb[3];          // second operand (the same planes for the Gram; row offsets differ in the block-row mode)
    int ioff, joff;              // first output row of operand A / B in units of entries
    float* C; long long ldc;
    int M, N, K;                 // K multiple of 16
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

__global__ void __launch_bounds__(GB_THREADS, 2)
gram_bf16x3_kernel(GramB3 g) {
    __shared__ uint4 lds[2][2][3][2][GB_BM];             // [buffer][A/B][plane][k group][i]: 48 KB
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
    const size_t offA = (size_t)s_kg * g.ldz + g.ioff + I0 + s_i;
    const size_t offB = (size_t)s_kg * g.ldz + g.joff + J0 + s_i;
    const size_t kstep = (size_t)2 * g.ldz;              // entries per K tile (two k groups)

    floatx16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    uint4 s0a0, s0a1, s0a2, s0b0, s0b1, s0b2, s1a0, s1a1, s1a2, s1b0, s1b1, s1b2;
#define GB_GLOAD(S, t)                                          \
    {                                                           \
        const size_t k_ = (size_t)(t) * kstep;                  \
        S##a0 = g.Z[0][offA + k_]; S##a1 = g.Z[1][offA + k_]; S##a2 = g.Z[2][offA + k_];        \
        S##b0 = g.Zb[0][offB + k_]; S##b1 = g.Zb[1][offB + k_]; S##b2 = g.Zb[2][offB + k_];     \
    }
#define GB_LSTORE(S, buf)                                       \
    {                                                           \
        lds[buf][0][0][s_kg][s_i] = S##a0; lds[buf][0][1][s_kg][s_i] = S##a1; lds[buf][0][2][s_kg][s_i] = S##a2;   \
        lds[buf][1][0][s_kg][s_i] = S##b0; lds[buf][1][1][s_kg][s_i] = S##b1; lds[buf][1][2][s_kg][s_i] = S##b2;   \
    }
    const int fk = lane >> 5, fi = lane & 31;
    auto compute = [&](int buf) {
        bf16x8 fa[2][3], fb[2][3];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                fa[a][pl] = gb_frag(lds[buf][0][pl][fk][wi + a * 32 + fi]);
                fb[a][pl] = gb_frag(lds[buf][1][pl][fk][wj + a * 32 + fi]);
            }
        // the six kept products, smallest weights first; the four accumulators of the wave between two dependent instructions
#define GB_TERM(PA, PB)                                                                                         \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][PA], fb[0][PB], acc[0][0], 0, 0, 0);           \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][PA], fb[1][PB], acc[0][1], 0, 0, 0);           \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][PA], fb[0][PB], acc[1][0], 0, 0, 0);           \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][PA], fb[1][PB], acc[1][1], 0, 0, 0);
        GB_TERM(0, 2) GB_TERM(2, 0) GB_TERM(1, 1) GB_TERM(0, 1) GB_TERM(1, 0) GB_TERM(0, 0)
#undef GB_TERM
    };
    const int ntile_k = g.K / GB_BK;
    if (ntile_k > 0) {
        GB_GLOAD(s0, 0)
        GB_LSTORE(s0, 0)
        __syncthreads();
        for (int kt = 0; kt < ntile_k; kt += 2) {
            if (kt + 1 < ntile_k) GB_GLOAD(s1, kt + 1)
            compute(0);
            if (kt + 1 < ntile_k) {
                GB_LSTORE(s1, 1)
                __syncthreads();
                if (kt + 2 < ntile_k) GB_GLOAD(s0, kt + 2)
                compute(1);
                if (kt + 2 < ntile_k) { GB_LSTORE(s0, 0) __syncthreads(); }
            }
        }
    }
#undef GB_GLOAD
#undef GB_LSTORE
    // C/D layout of the 32x32 instruction: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
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
                    // lower mode: the upper triangle is ALWAYS the mirror of the lower one, inside the diagonal tiles too -- the pairs
                    // (h l', l h') / (h m', m h') enter the accumulator in an order that is not symmetric in (i, j), so (i, j) and (j, i)
                    // computed directly would differ in the last bit; the block-row mode is followed by symmetrize_from_lower (prep.hip)
                    if (g.lower && !offdiag && row < col) continue;
                    const float v = acc[a][b][r];
                    g.C[(size_t)col * g.ldc + row] = v;
                    if (g.lower && g.mirror && row != col) g.C[(size_t)row * g.ldc + col] = v;
                }
            }
        }
}

// Split columns [0, nc) of X (rows x nc, column-major, ld ldx, rows beyond `rows` zero up to a multiple of 8) into the three bf16
// planes of Z = X': entry (k group kg, index i0 + j) of plane pl holds rows 8 kg .. 8 kg + 7 of column j.  A workgroup transposes a
// 64-row x 64-column piece through LDS so that the reads run down the columns of X and the writes along i.
__device__ __forceinline__ unsigned short gb_bf16_rn(float x) {      // round to nearest even (finite inputs)
    const unsigned u = __float_as_uint(x);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__global__ void __launch_bounds__(256)
split3_kernel(const float* __restrict__ X, long long ldx, int rows, int nc, uint4* __restrict__ Zh, uint4* __restrict__ Zm, uint4* __restrict__ Zl,
              long long ldz, int i0, int nkg) {
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
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int r = R0 + kgl * 8 + e;
            const float x = r < rows ? tile[col][kgl * 8 + e] : 0.f;
            h[e] = gb_bf16_rn(x);
            const float r1 = x - __uint_as_float((unsigned)h[e] << 16);
            m[e] = gb_bf16_rn(r1);
            const float r2 = r1 - __uint_as_float((unsigned)m[e] << 16);
            l[e] = gb_bf16_rn(r2);
        }
        auto pack = [](const unsigned short (&s)[8]) {
            return make_uint4((unsigned)s[0] | ((unsigned)s[1] << 16), (unsigned)s[2] | ((unsigned)s[3] << 16),
                              (unsigned)s[4] | ((unsigned)s[5] << 16), (unsigned)s[6] | ((unsigned)s[7] << 16));
        };
        const size_t o = (size_t)kg * ldz + i0 + J0 + col;
        Zh[o] = pack(h); Zm[o] = pack(m); Zl[o] = pack(l);
    }
}

bool gram_bf16x3_enabled() {
    const char* e = std::getenv("ADMM_HIP_GRAM_BF16");       // read per call (the A/B test flips it inside one process)
    return !(e && e[0] == '0');
}

void GramSplit3::alloc(int order, int kdepth, hipStream_t st) {
    M = order;
    ldz = round_up(order, GB_BM);
    nkg = (int)(round_up(kdepth, GB_BK) / 8);
    planes.alloc((size_t)3 * nkg * ldz * 8);
    planes.zero(st);
}

void GramSplit3::split_cols(const float* X, long long ldx, int rows, int c0, int nc, hipStream_t st) {
    uint4* base = reinterpret_cast<uint4*>(planes.get());
    const size_t pl = (size_t)nkg * ldz;
    hipLaunchKernelGGL(split3_kernel, dim3((rows + 63) / 64, (nc + 63) / 64), dim3(256), 0, st, X, ldx, rows, nc, base, base + pl, base + 2 * pl, ldz, c0, nkg);
}

static void launch_gram_b3(const GramSplit3& z, int ioff, int joff, float* C, long long ldc, int M, int N, bool lower, const int* tilemap, int ntiles_listed, hipStream_t st) {
    GramB3 g;
    const uint4* base = reinterpret_cast<const uint4*>(z.planes.get());
    const size_t pl = (size_t)z.nkg * z.ldz;
    for (int k = 0; k < 3; ++k) { g.Z[k] = base + k * pl; g.Zb[k] = base + k * pl; }
    g.ldz = z.ldz; g.ioff = ioff; g.joff = joff; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = z.nkg * 8;
    g.lower = lower ? 1 : 0; g.mirror = lower ? 1 : 0;
    g.nbi = (M + GB_BM - 1) / GB_BM; g.nbj = (N + GB_BM - 1) / GB_BM;
    g.ntiles = lower ? g.nbi * (g.nbi + 1) / 2 : g.nbi * g.nbj;
    g.tilemap = tilemap;
    if (tilemap != nullptr && ntiles_listed >= 0) g.ntiles = ntiles_listed;
    if (g.ntiles <= 0) return;
    hipLaunchKernelGGL(gram_bf16x3_kernel, dim3((g.ntiles + 7) / 8 * 8), dim3(GB_THREADS), 0, st, g);
}

// C (both triangles, order x order) = Z Z' from the planes; tiles in the XCD-local square order of syrk_mfma.hip
void GramSplit3::gram_lower(float* C, long long ldc, const int* tilemap, int ntiles, hipStream_t st) const {
    launch_gram_b3(*this, 0, 0, C, ldc, M, M, true, tilemap, ntiles, st);
}
// block row: C[r0 : r0 + nr, 0 : r0 + nr] = Z[r0 : r0 + nr, :] Z[0 : r0 + nr, :]'  (r0 a multiple of 128)
void GramSplit3::gram_rows(int r0, int nr, float* C, long long ldc, hipStream_t st) const {
    launch_gram_b3(*this, r0, 0, C + r0, ldc, nr, r0 + nr, false, nullptr, -1, st);
}

}  // namespace admm
