// Basis pursuit with the COLUMNS of A split into blocks -- "sharing" ADMM (fp64), device resident; admm_hip_parbp.
//
// Replaces what R's admm_bp()$parallel(nthread)$fit() asks for, .Call("admm_parbp", x, y, nthread, opts)
// (/root/reference/R/10_admm_bp.R:111-116): a symbol the reference never builds.  Its source is
//   /root/reference/src/TODO/PADMMBP.h:19-61 (worker: linearised x-update, active-set form on 9 of 10 iterations),
//   :137-140 (z-bar = b / N), :150-168 (partition), :181-186 (rho = 1 / (rho_ratio mean sprad)), TODO/ParBP.cppp:26-71 (entry)
// written against a master / worker base class that no longer exists; the loop, the residuals and the thresholds are
// restated on the shape of the CURRENT PADMMBase_Master (/root/reference/src/PADMMBase.h:118-142,174-237) -- the
// derivation and every choice that is ours is in oracle/solvers.py (class SharingBP), which this file follows line by line.
// SURVEY.md section 8f rows n2 ("feature-block sharing ADMM") and n3 (admm_parbp).
//
//   min ||x||_1  s.t.  sum_i A_i x_i = b          i = 1..N column blocks
//   v = y / rho + r                                r = mean_i(A_i x_i) - b / N   (the shared primal residual)
//   x_i <- soft(x_i - A_i'v / gamma_i, 1 / (rho gamma_i))    gamma_i = 2 rho + lambda_max(A_i'A_i)
//          every column on iterations 0, 10, 20, ...; only the current non-zeros otherwise
//   y <- y + rho r
//
// Launches.  Regular iteration (0, 10, 20, ...): `xreg` (the decision of the previous iteration, evaluated identically by every
// workgroup from the norm partials; then the x-update of EVERY column as a streaming transposed mat-vec, 6.5 TB/s at the C5
// shape), `list` (the blocks' non-zero lists, ascending, from per-workgroup counts), `xact<AXONLY>` (the workgroup partials of
// A_i x_i over the lists), `tail`.  Active-set iteration: `xact` (decision; x-update of the listed non-zeros and the partials
// of A_i x_i in the same launch), `tail` (block sums of the partials, S = sum_i A_i x_i, the two block norms; r, y, v and the
// norm partials).  With a communicator attached the column blocks are spread over the ranks and ONE sum all-reduce of
// n + 2 nT doubles sits between `tail` and `tail_b`: S and the two block sums -- the dual residual is evaluated as
// sum_i ||A_i dx_i - dr||^2 = sum_i ||A_i dx_i||^2 - 2 dr'dS + N ||dr||^2  so that it needs nothing else.  The host enqueues
// iterations in batches and polls a sticky flag.  Measured (C5 shape n = 5000, p = 50 000, 8 blocks on one MI355X, 545
// non-zeros at the end, 5754 iterations): regular iteration 308 + 5 + 11 + 11 us, active-set iteration 20 + 11 us, loop 0.36 s.
// One process (round 5): the nine active-set iterations between two regular ones run in GRAM SPACE instead -- one launch each
// that never touches A (section "Gram space" below: sbp_gs_*), and the regular iteration itself then needs only `xreg`, `list`
// and three Gram-space launches; the `xact` / `tail` launches above remain for the sharded solver, for supports too large for
// the Gram matrix and as the A/B (ADMM_HIP_SBP_GRAM=0).  Same shape: 9.3 us per active-set iteration, loop 0.257 s.
#include <system_error>
#include "prep.h"
#include "gemv_kernels.h"
#include "solvers.h"
#include "loop_driver.h"
#include "comm.h"
#include "device_utils.h"
#include "gather_kernels.h"

#include <cmath>
#include <exception>
#include <thread>

namespace admm {

struct SbpCtl {                                   // 64 bytes: whole 16-byte words (load_ctl_vector)
    double eps_primal, eps_dual, rp, rd;
    int iter, done, niter, conv, total, pad0, pad1, pad2;
};

struct SbpBlk {                                   // 32 bytes: one vector round trip (load_ctl_vector)
    int c0, pb;                                   // first local column, columns
    double gamma, pen;                            // 2 rho + sprad_i, 1 / (rho gamma_i)
    long long pad;
};

// Gram-space state of the active-set iterations (see "Gram space" below).  U = the columns that have been non-zero since the
// last reset, in the order they appeared; everything indexed by a < count is in that order.
struct SbpGram {
    int cap, ldg;                                 // most entries of U; leading dimension of G (= cap)
    int* umap;                                    // [pl] index into U or -1
    int* ucol; int* ubid;                         // [cap] local column, local block
    int* ust;                                     // [8]: 0 count, 1 count before this stretch's additions, 2 this stretch runs in Gram space, 3 abort, 4 iteration of the abort, 5 resets, 6 stretches
    double* G;                                    // [cap][ldg] A_U'A_U (symmetric, both halves stored)
    double* gz;                                   // [cap] A_U'zbar
    double* ugp;                                  // [cap][2] 1 / gamma and the threshold of the entry's block
    double* xs; double* hr; double* gy;           // [2][cap] x_U, A_U'r, A_U'y (double-buffered by the iteration's parity)
    double* sx;                                   // [cap] sum of the stretch's iterates
    double* sxd;                                  // [pl] dense: sum_t (x_t - x_last) over the stretch, zero outside U
    double* Ps;                                   // [2][cap / 8][8] workgroup partials of the seven sums a decision needs: a launch reads the
                                                  // buffer of its parity and writes the other (a workgroup that finishes early must not overwrite
                                                  // what a workgroup of the SAME launch that started late has yet to read)
    double* sc;                                   // [2][4] ||r||^2, ||y||^2, y'zbar carried from decision to decision; [8] y'zbar of the stretch's start
    double zz;                                    // zbar'zbar
    double* partP; double* partT;                 // [NL][ngroups][pstride] partials of A_i x_i and of A_i sxd_i (gather_kernels.h)
    int ngroups, pad; long long pstride;
    int test_delay, pad2;                         // test hook: ticks (10 ns) by which every even workgroup of a Gram-space launch starts late
};

struct SbpParams {
    int n, npad, N, NL, maxit, G, nT, min_share;
    double eps_abs, eps_rel, rho, sqrt_nN, sqrtN, dN;
    double invN, inv_rho;                        // Gram space only (the direct launches divide, as the reference does)
    const double* A; long long lda;              // this rank's columns, n x pl column-major, rows padded with zeros to npad
    const unsigned short* Ah; long long ldh;     // the regular iterations' screen: A rounded to fp16 ([pl][ldh], ldh a multiple of 8, rows [n, ldh) zero), or NULL
    const float* scr_s;                          // [pl] per-column bound (sbp_screen_prep_kernel)
    unsigned long long* scr_stat;                // optional [2]: columns screened, columns that took the exact path (SBP_SCREEN_STATS)
    int Gb, pad1;                                // workgroups per local block (the same for every block: block = g / Gb, no table)
    const SbpBlk* blk;                           // [NL]
    double* x;                                   // [pl]
    int* list; int* cnt;                         // [pl] block-relative indices of the non-zeros, ascending; [NL]
    int* wcount;                                 // [G]
    double* xl;                                  // [pl] the values of the listed entries, in list order
    double* P;                                   // [G][npad] workgroup partials of A_i x_i
    double* Axo;                                 // [NL][npad]
    double* S; double* Qa;                       // exchange buffer: [npad] | [nT][2]
    double* Sold; double* y; double* r; double* v; const double* zbar;   // [npad]
    double* Q;                                   // [nT][8]
    SbpCtl* ctl; int* done; int* hflag;
    double* trace; long long trace_cap;
    SbpGram gs;
};

constexpr int kSbpThreads = 256;

__device__ __forceinline__ double sbp_soft(double v, double pen) {
    return v > pen ? v - pen : (v < -pen ? v + pen : 0.0);
}

// Entries of a block's work list per workgroup.  Regular iterations: the columns in Gb equal shares.  Active-set iterations: at
// least `min_share` (4: one round of the four waves) list entries per workgroup, so that a few hundred non-zeros occupy a few
// dozen workgroups and the tail sums that many partial rows instead of Gb (measured: 1 -> 0.354 s, 4 -> 0.373, 8 -> 0.430, 16 -> 0.573).
__host__ __device__ __forceinline__ int sbp_share(int total, int Gb, int min_share) {
    const int per = (total + Gb - 1) / Gb;
    return per > min_share ? per : min_share;                       // min_share = 0: regular iterations, equal shares of the columns
}

// The decision every iteration starts with, evaluated identically by every workgroup of the iteration's first launch: the
// residuals of the iteration just finished against the thresholds it ran with (PADMMBase.h:223-231), then the thresholds of the
// coming one (:118-136).  Returns false when the loop is over (all threads agree).  red: 28 doubles of LDS.
// `core`: from the seven sums to the new control block (all threads agree); `sbp_decide`: the sums from the norm partials the
// tail left; `sbp_decide_gram` (below): the same seven from Gram-space quadratic forms.
__device__ __forceinline__ bool sbp_decide_core(const SbpParams& q, const SbpCtl& in, SbpCtl* outp, SbpCtl& out,
                                                double drdS, double dr2, double r2, double y2, double abar_r, double sax, double qq) {
    double code = ADMM_TRACE_COLD;
    if (in.iter > 0) {
        const double sd = qq - 2.0 * drdS + q.dN * dr2;
        out.rp = sqrt(q.dN * r2);
        out.rd = q.rho * sqrt(sd > 0.0 ? sd : 0.0);
        code = ADMM_TRACE_CONTINUE;
        if (out.rp < in.eps_primal && out.rd < in.eps_dual) { out.done = 1; out.conv = 1; out.niter = in.iter; code = ADMM_TRACE_CONVERGED; }
        else if (in.iter >= q.maxit) { out.done = 1; out.conv = 0; out.niter = q.maxit + 1; }      // `return i + 1` after the loop (PADMMBase.h:236)
    }
    {
        const double sz = sax - 2.0 * q.dN * abar_r + q.dN * r2;                       // sum_i ||z_i||^2, z_i = A_i x_i - r
        const double m = fmax(fmax(sax, sz), 0.0);
        out.eps_primal = q.eps_rel * sqrt(m) + q.sqrt_nN * q.eps_abs;
        out.eps_dual = q.eps_rel * q.sqrtN * sqrt(y2) + q.sqrt_nN * q.eps_abs;
    }
    if (!out.done) out.iter = in.iter + 1;
    out.total = in.total + 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *outp = out;
        if (out.done) { *q.done = 1; if (q.hflag) *q.hflag = 1; }
        if (q.trace != nullptr && in.total < q.trace_cap) {
            double* t = q.trace + (size_t)in.total * ADMM_TRACE_FIELDS;
            t[0] = 0.0; t[1] = in.iter - 1; t[2] = in.eps_primal; t[3] = in.eps_dual; t[4] = out.rp; t[5] = out.rd;
            t[6] = 0.0; t[7] = 0.0; t[8] = code; t[9] = q.rho; t[10] = q.rho; t[11] = in.iter > 0 && ((in.iter - 1) % 10) == 0 ? 1.0 : 0.0;      // the judged iteration was a regular one
        }
    }
    return !out.done;
}

__device__ __forceinline__ bool sbp_decide(const SbpParams& q, int par, SbpCtl& out, double* red) {
    const SbpCtl in = load_ctl_vector(q.ctl + par);
    SbpCtl* outp = &q.ctl[par ^ 1];
    out = in;
    if (in.done) {
        if (blockIdx.x == 0 && threadIdx.x == 0) *outp = in;
        return false;
    }
    double s[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int w = threadIdx.x; w < q.nT; w += kSbpThreads) {
#pragma unroll
        for (int k = 0; k < 5; ++k) s[k] += q.Q[w * 8 + k];
        s[5] += q.Qa[w * 2]; s[6] += q.Qa[w * 2 + 1];
    }
    block_sum<double, 7>(s, red);
    return sbp_decide_core(q, in, outp, out, s[0], s[1], s[2], s[3], s[4], s[5], s[6]);
}

// Regular iterations (0, 10, 20, ...), first launch: x_j <- soft(x_j - A_j'v / gamma, pen) for EVERY column -- a streaming
// transposed mat-vec: v staged in LDS, a wave owns two columns at a time and reads them with 16-byte loads, no barrier in the
// loop.  A_i x_i is NOT formed here (after the threshold nearly every column is zero): the list launch builds the non-zero
// lists from the per-workgroup counts and the step launch below adds x_j A_j over them (AXONLY).
template <bool NT>                                                  // NT: the matrix does not stay in the Infinity Cache between two regular iterations
__global__ void __launch_bounds__(kSbpThreads)
sbp_xreg_kernel(SbpParams q, int par) {
    extern __shared__ __attribute__((aligned(16))) double vsh[];    // npad doubles
    __shared__ double red[8 * 4];
    __shared__ int wnz[4];
    SbpCtl c;
    if (!sbp_decide(q, par, c, red)) return;
    for (int k = threadIdx.x; k < q.npad; k += kSbpThreads) vsh[k] = q.v[k];
    __syncthreads();
    const int g = blockIdx.x;
    const int Gb = q.Gb, b = g / Gb, sub = g - b * Gb;
    const SbpBlk bi = load_ctl_vector(q.blk + b);
    const int c0 = bi.c0, pb = bi.pb;
    const double gamma = bi.gamma, pen = bi.pen;
    const int per = sbp_share(pb, Gb, 0);
    const int lo = sub * per, hi = min(pb, lo + per);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nk = q.npad / 128;                                    // 16-byte loads: 128 rows per wave instruction
    int nzc = 0;
    for (int e = lo + 2 * wid; e < hi; e += 8) {
        const bool two = e + 1 < hi;
        const double2* a0 = reinterpret_cast<const double2*>(q.A + (size_t)(c0 + e) * q.lda) + lane;
        const double2* a1 = reinterpret_cast<const double2*>(q.A + (size_t)(c0 + e + (two ? 1 : 0)) * q.lda) + lane;
        const double2* vv = reinterpret_cast<const double2*>(vsh) + lane;
        double d0 = 0.0, d1 = 0.0;
        int k = 0;
        for (; k + 4 <= nk; k += 4) {
            double2 x0[4], x1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { x0[u] = NT ? load16_nt<double2>(a0 + (size_t)(k + u) * 64) : a0[(size_t)(k + u) * 64]; x1[u] = NT ? load16_nt<double2>(a1 + (size_t)(k + u) * 64) : a1[(size_t)(k + u) * 64]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double2 w = vv[(k + u) * 64];
                d0 = fma(x0[u].x, w.x, d0); d0 = fma(x0[u].y, w.y, d0);
                d1 = fma(x1[u].x, w.x, d1); d1 = fma(x1[u].y, w.y, d1);
            }
        }
        for (; k < nk; ++k) {
            const double2 x0 = NT ? load16_nt<double2>(a0 + (size_t)k * 64) : a0[(size_t)k * 64], x1 = NT ? load16_nt<double2>(a1 + (size_t)k * 64) : a1[(size_t)k * 64], w = vv[k * 64];
            d0 = fma(x0.x, w.x, d0); d0 = fma(x0.y, w.y, d0);
            d1 = fma(x1.x, w.x, d1); d1 = fma(x1.y, w.y, d1);
        }
        d0 = wave_sum(d0); d1 = wave_sum(d1);
        double xn0, xn1;
        {
#pragma clang fp contract(off)
            xn0 = sbp_soft(q.x[c0 + e] - d0 / gamma, pen);
            xn1 = two ? sbp_soft(q.x[c0 + e + 1] - d1 / gamma, pen) : 0.0;
        }
        if (lane == 0) { q.x[c0 + e] = xn0; if (two) q.x[c0 + e + 1] = xn1; }
        nzc += (xn0 != 0.0) + (xn1 != 0.0);
    }
    if (lane == 0) wnz[wid] = nzc;
    __syncthreads();
    if (threadIdx.x == 0) q.wcount[g] = wnz[0] + wnz[1] + wnz[2] + wnz[3];
}

// Regular iterations, first launch, SCREENED (round 6; the argument is wide_x_kernel's, lasso_wide.hip): the prox leaves all but a per
// cent of the columns at zero, and a product with the column ROUNDED to fp16 proves it for nearly all of them --
//     |fl64(A_j'v)| <= |fl32(Ah_j'vf)| + s_j ||v||_2,   s_j >= ||A_j - Ah_j||_2 + ((n + 2) 2^-24 + n 2^-53) max(||A_j||_2, ||Ah_j||_2)
// (rounding of the column by Cauchy-Schwarz; the float inner product's gamma_n; v rounded to float, 2^-24 per entry; the double inner
// product's own gamma_n).  A column with x_j = 0 whose bound stays below gamma * pen * (1 - 1e-9) keeps the zero it holds (the
// exact step divides by gamma -- one rounding of 2^-53 -- and compares); every other column takes the exact step on the double
// column, IN THE ARITHMETIC OF sbp_xreg_kernel (one accumulator per lane, rows ascending, then the wave sum): x and the
// per-workgroup counts are bit-identical to the unscreened launch's, and the launch streams 2 n p bytes instead of 8 n p.
__device__ __forceinline__ double sbp_col_dot_seq(const double* __restrict__ col, const double* vsh, int nk, int lane) {
    const double2* a = reinterpret_cast<const double2*>(col) + lane;
    const double2* vv = reinterpret_cast<const double2*>(vsh) + lane;
    double d = 0.0;
    for (int k0 = 0; k0 < nk; k0 += 16) {
        double2 x[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) x[u] = k0 + u < nk ? a[(size_t)(k0 + u) * 64] : make_double2(0.0, 0.0);
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (k0 + u < nk) { const double2 w = vv[(k0 + u) * 64]; d = fma(x[u].x, w.x, d); d = fma(x[u].y, w.y, d); }
    }
    return wave_sum(d);
}

__global__ void __launch_bounds__(kSbpThreads)
sbp_xreg_screen_kernel(SbpParams q, int par) {
    extern __shared__ __attribute__((aligned(16))) double vsh[];    // npad doubles
    __shared__ double red[8 * 4];
    __shared__ int wnz[4];
    SbpCtl c;
    if (!sbp_decide(q, par, c, red)) return;
    for (int k = threadIdx.x; k < q.npad; k += kSbpThreads) vsh[k] = q.v[k];
    __syncthreads();
    const int g = blockIdx.x;
    const int Gb = q.Gb, b = g / Gb, sub = g - b * Gb;
    const SbpBlk bi = load_ctl_vector(q.blk + b);
    const int c0 = bi.c0, pb = bi.pb;
    const double gamma = bi.gamma, pen = bi.pen;
    const int per = sbp_share(pb, Gb, 0);
    const int lo = sub * per, hi = min(pb, lo + per);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nk = q.npad / 128;
    double vs = 0.0;
    for (int i = lane; i < q.npad; i += 64) vs = fma(vsh[i], vsh[i], vs);
    const double V = sqrt(wave_sum(vs)) * (1.0 + 1e-12);             // >= ||v||_2
    const double Gthr = gamma * pen * (1.0 - 1e-9);
    const int nkh = (int)((q.ldh + 511) / 512);                      // 16-byte pieces per lane of a rounded column
    typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
    constexpr int kScrUn = 8;                                        // pieces of each of the two columns requested together (16 KB per wave in flight)
    int nzc = 0;
    unsigned long long seen = 0, exact = 0;
    for (int e = lo + 2 * wid; e < hi; e += 8) {
        const bool two = e + 1 < hi;
        const int j0 = c0 + e, j1 = c0 + e + (two ? 1 : 0);
        const double xo0 = q.x[j0], xo1 = q.x[j1];
        const float s0 = q.scr_s[j0], s1 = q.scr_s[j1];
        const unsigned short* h0 = q.Ah + (size_t)j0 * q.ldh + lane * 8;
        const unsigned short* h1 = q.Ah + (size_t)j1 * q.ldh + lane * 8;
        float d0 = 0.f, d1 = 0.f;
        for (int k0 = 0; k0 < nkh; k0 += kScrUn) {
            uint4 a0[kScrUn], a1[kScrUn];
#pragma unroll
            for (int u = 0; u < kScrUn; ++u) {
                const long long r = (long long)(k0 + u) * 512 + lane * 8;
                a0[u] = make_uint4(0u, 0u, 0u, 0u); a1[u] = make_uint4(0u, 0u, 0u, 0u);
                if (k0 + u < nkh && r < q.ldh) { a0[u] = load16_nt<uint4>(h0 + (size_t)(k0 + u) * 512); a1[u] = load16_nt<uint4>(h1 + (size_t)(k0 + u) * 512); }
            }
#pragma unroll
            for (int u = 0; u < kScrUn; ++u) {
                const long long r = (long long)(k0 + u) * 512 + lane * 8;
                if (k0 + u < nkh && r < q.ldh) {                     // ldh <= npad: the eight rows are inside vsh
                    const double2* vp = reinterpret_cast<const double2*>(vsh + r);
                    const double2 w0 = vp[0], w1 = vp[1], w2 = vp[2], w3 = vp[3];
                    const float vf[8] = {(float)w0.x, (float)w0.y, (float)w1.x, (float)w1.y, (float)w2.x, (float)w2.y, (float)w3.x, (float)w3.y};
                    const half8_t x0 = __builtin_bit_cast(half8_t, a0[u]), x1 = __builtin_bit_cast(half8_t, a1[u]);
#pragma unroll
                    for (int t = 0; t < 8; ++t) { d0 = fmaf((float)x0[t], vf[t], d0); d1 = fmaf((float)x1[t], vf[t], d1); }
                }
            }
        }
        d0 = wave_sum(d0); d1 = wave_sum(d1);
        seen += two ? 2 : 1;
        double xn0 = 0.0, xn1 = 0.0;
        if (!(xo0 == 0.0 && (double)fabsf(d0) + (double)s0 * V <= Gthr)) {              // (a NaN anywhere: the exact step)
            const double d = sbp_col_dot_seq(q.A + (size_t)j0 * q.lda, vsh, nk, lane);
            {
#pragma clang fp contract(off)
                xn0 = sbp_soft(xo0 - d / gamma, pen);
            }
            if (lane == 0) q.x[j0] = xn0;
            ++exact;
        }
        if (two && !(xo1 == 0.0 && (double)fabsf(d1) + (double)s1 * V <= Gthr)) {
            const double d = sbp_col_dot_seq(q.A + (size_t)j1 * q.lda, vsh, nk, lane);
            {
#pragma clang fp contract(off)
                xn1 = sbp_soft(xo1 - d / gamma, pen);
            }
            if (lane == 0) q.x[j1] = xn1;
            ++exact;
        }
        nzc += (xn0 != 0.0) + (xn1 != 0.0);
    }
    if (lane == 0) wnz[wid] = nzc;
    __syncthreads();
    if (threadIdx.x == 0) q.wcount[g] = wnz[0] + wnz[1] + wnz[2] + wnz[3];
    if (q.scr_stat != nullptr && lane == 0) { atomicAdd(q.scr_stat, seen); atomicAdd(q.scr_stat + 1, exact); }
}

// Setup of the screen: column j of A rounded to fp16 (a wave per column; a rounding that is not finite is stored as zero and counts
// as rounding error in full) and its bound s_j, sums in double, rounded up to float; a column holding a NaN / Inf gets +Inf.
__global__ void __launch_bounds__(256)
sbp_screen_prep_kernel(const double* __restrict__ A, long long lda, int n, int pl, unsigned short* __restrict__ Ah, long long ldh, float* __restrict__ s) {
    const int lane = threadIdx.x & 63;
    const long long j = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= pl) return;
    const double* col = A + (size_t)j * lda;
    unsigned short* hc = Ah + (size_t)j * ldh;
    typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
    double e2 = 0.0, x2 = 0.0, h2 = 0.0;
    for (long long r = (long long)lane * 8; r < ldh; r += 512) {
        half8_t hv;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const double x = r + e < n ? col[r + e] : 0.0;
            _Float16 hh = (_Float16)(float)x;
            double back = (double)(float)hh;
            if (!(fabs(back) <= 65504.0)) { hh = (_Float16)0.f; back = 0.0; }
            hv[e] = hh;
            e2 = fma(x - back, x - back, e2); x2 = fma(x, x, x2); h2 = fma(back, back, h2);
        }
        *reinterpret_cast<uint4*>(hc + r) = __builtin_bit_cast(uint4, hv);
    }
    e2 = wave_sum(e2); x2 = wave_sum(x2); h2 = wave_sum(h2);
    if (lane == 0) {
        const double sd = (sqrt(e2) + 1.01 * ((double)n + 4.0) * 5.9604644775390625e-8 * sqrt(fmax(x2, h2))) * (1.0 + 1e-6);
        float sv = __double2float_ru(sd);
        if (!(sv < __builtin_huge_valf())) sv = __builtin_huge_valf();
        s[j] = sv;
    }
}

// Active-set iterations, first launch (AXONLY = false): the decision, then x_j <- soft(x_j - A_j'v / gamma, pen) for the blocks'
// current non-zeros only (PADMMBP.h:19-44) and their share of A_i x_i.  The block's list is cut into shares of at least
// `min_share` entries per workgroup; in a round of four entries a wave streams one column (16-byte loads, v
// staged in LDS, the column requested in chunks of 24 x 16 bytes per lane), the four new values are exchanged through LDS, and then every thread adds  sum_c x_c A_c  for ITS rows (the
// columns are still in L2; entry order, so the additions have a fixed order) into registers -- the workgroup's partial of
// A_i x_i, stored to P[g] at the end.  Latency is what this launch costs (a dependent round trip to memory is ~1.5 us):
// the block record, the list length, the list entries and the first column are requested BEFORE the decision, whose own two
// round trips (control block, norm partials) they overlap.  Measured on the way here (C5 shape, 545 non-zeros, one MI355X):
// block-per-column dots with the partial in registers 45 us; wave-per-column + a row-gather launch 13 + 22 us (every 512-byte
// piece of the gather on another page of the 2 GB matrix); wave-per-column with the waves taking turns on an LDS partial 23 us.
// Regular iterations, third launch (AXONLY = true): the same over the lists just built, x untouched.
constexpr int kSbpChunk = 24;                                       // double2 per lane requested together when a wave streams a column
// RP: double2 of the partial per thread -- 16 (up to 8192 rows) or 32 (up to 16384 rows: the column pieces of the axpy half
// are then requested eight at a time instead of all at once)
template <bool AXONLY, int RP>
__global__ void __launch_bounds__(kSbpThreads)
sbp_xact_kernel(SbpParams q, int par) {
    constexpr int kSbpRowPairs = RP;
    constexpr int kSub = RP == 16 ? 16 : 8;
    extern __shared__ __attribute__((aligned(16))) double vsh[];    // npad doubles: v (not used by AXONLY)
    __shared__ double red[8 * 4];
    __shared__ double sx[4];
    __shared__ int sj[4];
    const int g = blockIdx.x;
    const int Gb = q.Gb, b = g / Gb, sub = g - b * Gb;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    // ---- requests that do not depend on the decision
    const SbpBlk bi = load_ctl_vector(q.blk + b);
    const int total = load_flag_vector(q.cnt + b);
    const int c0 = bi.c0;
    const int per = sbp_share(total, Gb, q.min_share);
    const int lo = sub * per, hi = min(total, lo + per);
    int j0 = 0; double x0 = 0.0;
    if (lo + wid < hi) { j0 = q.list[c0 + lo + wid]; x0 = q.xl[c0 + lo + wid]; }
    if (!AXONLY && lo < total)
        for (int k = threadIdx.x; k < q.npad; k += kSbpThreads) vsh[k] = q.v[k];      // made visible by the barriers of the decision
    __builtin_amdgcn_sched_barrier(0);
    SbpCtl out;
    if (AXONLY) {
        out = load_ctl_vector(q.ctl + par);                         // written by this iteration's xreg launch
        if (out.done) return;
    } else {
        if (!sbp_decide(q, par, out, red)) return;
    }
    if (lo >= total) return;                                        // only the first ceil(total / per) workgroups of the block work (the tail knows)
    const double gamma = bi.gamma, pen = bi.pen;
    const int nk = q.npad / 128;                                    // double2 per lane and column
    const int np2 = q.npad / 2;                                     // double2 per column
    const double2* vv = reinterpret_cast<const double2*>(vsh) + lane;
    double2 acc[kSbpRowPairs];
#pragma unroll
    for (int k = 0; k < kSbpRowPairs; ++k) acc[k] = make_double2(0.0, 0.0);
    for (int e4 = lo; e4 < hi; e4 += 4) {                           // uniform trip count: the barriers below are the workgroup's
        const int e = e4 + wid;
        double xn = 0.0;
        int j = 0;
        if (e < hi) {
            j = __builtin_amdgcn_readfirstlane(e4 == lo ? j0 : q.list[c0 + e]);
            const double xj = readlane_f64(e4 == lo ? x0 : q.xl[c0 + e], 0);
            if (AXONLY) {
                xn = xj;
            } else if (xj != 0.0) {                                 // an entry the active set has already pruned stays zero (PADMMBP.h:43)
                const double2* a = reinterpret_cast<const double2*>(q.A + (size_t)(c0 + j) * q.lda) + lane;
                double d0 = 0.0, d1 = 0.0;
                for (int k0 = 0; k0 < nk; k0 += kSbpChunk) {       // a chunk of the column entirely in flight: one round trip per 3072 rows
                    double2 x[kSbpChunk];
#pragma unroll
                    for (int u = 0; u < kSbpChunk; ++u) x[u] = k0 + u < nk ? a[(size_t)(k0 + u) * 64] : make_double2(0.0, 0.0);
#pragma unroll
                    for (int u = 0; u < kSbpChunk; u += 2) {
                        if (k0 + u < nk) { const double2 w = vv[(k0 + u) * 64]; d0 = fma(x[u].x, w.x, d0); d0 = fma(x[u].y, w.y, d0); }
                        if (k0 + u + 1 < nk) { const double2 w = vv[(k0 + u + 1) * 64]; d1 = fma(x[u + 1].x, w.x, d1); d1 = fma(x[u + 1].y, w.y, d1); }
                    }
                }
                const double d = wave_sum(d0 + d1);
                {
#pragma clang fp contract(off)
                    xn = sbp_soft(xj - d / gamma, pen);
                }
                if (lane == 0) { q.x[c0 + j] = xn; q.xl[c0 + e] = xn; }
            }
        }
        __syncthreads();                                            // (the previous round's readers of sx / sj are done)
        if (lane == 0) { sx[wid] = xn; sj[wid] = j; }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 4; ++c) {                               // entry order: every thread adds x_c A_c for its rows
            const double xc = sx[c];
            if (xc == 0.0) continue;                                // uniform
            const double2* ac = reinterpret_cast<const double2*>(q.A + (size_t)(c0 + sj[c]) * q.lda) + threadIdx.x;
#pragma unroll
            for (int kb = 0; kb < kSbpRowPairs; kb += kSub) {
                double2 t[kSub];
#pragma unroll
                for (int k = 0; k < kSub; ++k) t[k] = (kb + k) * kSbpThreads + (int)threadIdx.x < np2 ? ac[(size_t)(kb + k) * kSbpThreads] : make_double2(0.0, 0.0);
#pragma unroll
                for (int k = 0; k < kSub; ++k) { acc[kb + k].x = fma(xc, t[k].x, acc[kb + k].x); acc[kb + k].y = fma(xc, t[k].y, acc[kb + k].y); }
            }
        }
    }
    double2* P = reinterpret_cast<double2*>(q.P + (size_t)g * q.npad) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < kSbpRowPairs; ++k)
        if (k * kSbpThreads + (int)threadIdx.x < np2) P[(size_t)k * kSbpThreads] = acc[k];
    (void)gamma; (void)pen; (void)vv; (void)nk;
}

// Regular iterations only: the non-zero lists of the blocks, ascending, from the counts the step left per workgroup.
__global__ void __launch_bounds__(kSbpThreads)
sbp_list_kernel(SbpParams q, int par) {
    __shared__ int sh[4];
    __shared__ int sbase;
    const SbpCtl c = load_ctl_vector(q.ctl + par);                  // written by this iteration's step
    if (c.done) return;
    const int g = blockIdx.x;
    const int Gb = q.Gb, b = g / Gb, sub = g - b * Gb, g0 = b * Gb;
    const SbpBlk bi = load_ctl_vector(q.blk + b);
    const int c0 = bi.c0, pb = bi.pb;
    const int per = (pb + Gb - 1) / Gb;
    const int lo = sub * per, hi = min(pb, lo + per);
    int off = 0;
    for (int k = threadIdx.x; k < sub; k += kSbpThreads) off += q.wcount[g0 + k];
    off = wave_sum(off);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) sh[wid] = off;
    __syncthreads();
    if (threadIdx.x == 0) sbase = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    int base = sbase;
    for (int e0 = lo; e0 < hi; e0 += kSbpThreads) {
        const int e = e0 + threadIdx.x;
        const bool nz = e < hi && q.x[c0 + e] != 0.0;
        const unsigned long long m = __ballot(nz);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        __syncthreads();
        if (lane == 0) sh[wid] = __popcll(m);
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wid; ++w) woff += sh[w];
        if (nz) { q.list[c0 + base + woff + before] = e; q.xl[c0 + base + woff + before] = q.x[c0 + e]; }
        base += sh[0] + sh[1] + sh[2] + sh[3];
    }
    if (sub == Gb - 1 && threadIdx.x == 0) q.cnt[b] = base;
}

// Every iteration, last launch(es): A_i x_i of every local block = the sum of the workgroup partials the x-update launch left
// (the first ceil(list length / share) workgroups of the block wrote one) -- 64 rows per workgroup, eight waves; blocks are
// taken eight at a time, their list lengths requested together and then wave w's partials w, w + 8, ... of ALL eight blocks
// (ascending within a block): two dependent round trips per eight blocks.  Wave t then adds the eight waves' sums of block t in
// wave order and books what changed; wave 0 adds the blocks in order: S = sum_i A_i x_i of the local blocks and the two sums.
// FUSE_B (one process): the r / y / v update and the norm partials of tail_b in the same launch.
constexpr int kSbpTailWaves = 8;
constexpr int kSbpTailBlocks = 8;
template <bool FUSE_B>
__global__ void __launch_bounds__(64 * kSbpTailWaves)
sbp_tail_kernel(SbpParams q, int par) {
    __shared__ double sh[kSbpTailBlocks][kSbpTailWaves][64];        // 32 KB
    __shared__ double shS[kSbpTailBlocks][3][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int row = blockIdx.x * 64 + lane;                         // < npad (npad is a multiple of 128)
    const int Gb = q.Gb;
    int nact0[kSbpTailBlocks];
#pragma unroll
    for (int t = 0; t < kSbpTailBlocks; ++t) {                      // requested before the control block arrives
        const int total = t < q.NL ? load_flag_vector(q.cnt + t) : 0;
        const int per = sbp_share(total, Gb, q.min_share);
        nact0[t] = min(Gb, (total + per - 1) / per);
    }
    const SbpCtl c = load_ctl_vector(q.ctl + par);                  // written by this iteration's first launch
    if (c.done) return;
    double S = 0.0, sax = 0.0, qq = 0.0;                            // wave 0 only
    for (int b0 = 0; b0 < q.NL; b0 += kSbpTailBlocks) {
        const int nb = min(kSbpTailBlocks, q.NL - b0);
        double a[kSbpTailBlocks];
#pragma unroll
        for (int t = 0; t < kSbpTailBlocks; ++t) {
            a[t] = 0.0;
            if (t >= nb) continue;
            int nact = nact0[t];
            if (b0 > 0) {
                const int total = load_flag_vector(q.cnt + b0 + t);
                const int per = sbp_share(total, Gb, q.min_share);
                nact = min(Gb, (total + per - 1) / per);
            }
            const double* P = q.P + (size_t)(b0 + t) * Gb * q.npad + row;
            for (int g0 = grp; g0 < nact; g0 += 8 * kSbpTailWaves) {       // predicated, not counted: all eight requests leave together
                double pv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int g = g0 + u * kSbpTailWaves; pv[u] = g < nact ? P[(size_t)g * q.npad] : 0.0; }
#pragma unroll
                for (int u = 0; u < 8; ++u) a[t] += pv[u];
            }
        }
        __syncthreads();                                            // (the previous chunk's readers are done)
#pragma unroll
        for (int t = 0; t < kSbpTailBlocks; ++t) sh[t][grp][lane] = a[t];
        __syncthreads();
        if (grp < nb) {                                             // wave t finishes block b0 + t
            double v = sh[grp][0][lane];
#pragma unroll
            for (int w = 1; w < kSbpTailWaves; ++w) v += sh[grp][w][lane];
            double* ao = q.Axo + (size_t)(b0 + grp) * q.npad + row;
            const double d = v - *ao;
            *ao = v;
            shS[grp][0][lane] = v; shS[grp][1][lane] = v * v; shS[grp][2][lane] = d * d;
        }
        __syncthreads();
        if (grp == 0)
            for (int t = 0; t < nb; ++t) { S += shS[t][0][lane]; sax += shS[t][1][lane]; qq += shS[t][2][lane]; }
    }
    if (grp != 0) return;
    sax = wave_sum(sax); qq = wave_sum(qq);
    if (lane == 0) { q.Qa[blockIdx.x * 2] = sax; q.Qa[blockIdx.x * 2 + 1] = qq; }
    if (!FUSE_B) { q.S[row] = S; return; }
    double drdS, dr2, r2, y2, abar_r;
    {
#pragma clang fp contract(off)
        const double dS = S - q.Sold[row];
        q.Sold[row] = S;
        const double abar = S / q.dN;
        const double rn = abar - q.zbar[row];
        const double dr = rn - q.r[row];
        q.r[row] = rn;
        const double yn = q.y[row] + q.rho * rn;
        q.y[row] = yn;
        q.v[row] = yn / q.rho + rn;
        drdS = dr * dS; dr2 = dr * dr; r2 = rn * rn; y2 = yn * yn; abar_r = abar * rn;
    }
    drdS = wave_sum(drdS); dr2 = wave_sum(dr2); r2 = wave_sum(r2); y2 = wave_sum(y2); abar_r = wave_sum(abar_r);
    if (lane == 0) {
        double* Q = q.Q + (size_t)blockIdx.x * 8;
        Q[0] = drdS; Q[1] = dr2; Q[2] = r2; Q[3] = y2; Q[4] = abar_r;
    }
}

// tail_b: r, y, v and the norm partials, from S summed over ALL blocks (all-reduced between the two tails when the blocks
// are spread over ranks: every rank then computes the same rows from the same numbers).
__global__ void __launch_bounds__(64)
sbp_tail_b_kernel(SbpParams q, int par) {
    const SbpCtl c = load_ctl_vector(q.ctl + par);
    if (c.done) return;
    const int row = blockIdx.x * 64 + threadIdx.x;
    double drdS, dr2, r2, y2, abar_r;
    {
#pragma clang fp contract(off)
        const double S = q.S[row];
        const double dS = S - q.Sold[row];
        q.Sold[row] = S;
        const double abar = S / q.dN;
        const double rn = abar - q.zbar[row];
        const double dr = rn - q.r[row];
        q.r[row] = rn;
        const double yn = q.y[row] + q.rho * rn;
        q.y[row] = yn;
        q.v[row] = yn / q.rho + rn;
        drdS = dr * dS; dr2 = dr * dr; r2 = rn * rn; y2 = yn * yn; abar_r = abar * rn;
    }
    drdS = wave_sum(drdS); dr2 = wave_sum(dr2); r2 = wave_sum(r2); y2 = wave_sum(y2); abar_r = wave_sum(abar_r);
    if (threadIdx.x == 0) {
        double* Q = q.Q + (size_t)blockIdx.x * 8;
        Q[0] = drdS; Q[1] = dr2; Q[2] = r2; Q[3] = y2; Q[4] = abar_r;
    }
}

// ------------------------------------------------------------------------------------------------ Gram space
// Active-set iterations without the matrix (round 5).  Between two regular iterations only the columns U that have been
// non-zero take part, a few hundred at the C5 shape, and an iteration needs of A only inner products of those columns:
//   A_j'v = A_j'y / rho + A_j'r,        A_U'r = G x_U / N - gz,       G = A_U'A_U,  gz = A_U'zbar,
//   A_U'y  by the recurrence  gy <- gy + rho A_U'r   from the exact product at the start of the stretch,
// and every sum the decision needs is a quadratic form (dx = the iteration's change of x_U, block = same column block):
//   ||A dx||^2 = dx'G dx = D          dr'dS = D / N      ||dr||^2 = D / N^2      sum_i ||A_i dx_i||^2 = dx'G_block dx
//   ||r+||^2 = ||r||^2 + 2 (A_U'r)'dx / N + D / N^2                               sum_i ||A_i x_i||^2 = x'G_block x
//   abar'r = x'(A_U'r) / N       y'r+ = gy'x+ / N - y'zbar       ||y+||^2 = ||y||^2 + 2 rho y'r+ + rho^2 ||r+||^2
// (||r||^2 and ||y||^2 are carried from their exact values at the stretch's start by these recurrences -- never formed as the
// difference of O(||b||^2) terms, which would cancel as r -> 0).  One launch per iteration: every workgroup repeats the decision
// and the |U| soft-thresholds, then owns 8 rows of G for the two mat-vecs G x+ and G dx (the same loads) and writes 7 partial
// sums.  After the ninth iteration the n-vectors are materialised for the regular iteration that follows:  A_i x_i  and
// A sum_t (x_t - x_last)  by the gather mat-vec (gather_kernels.h) over the dense x, then
//   y <- y + rho (m r_last + A sum_t (x_t - x_last) / N)       m = 9 iterates, 10 in a carried stretch    (r_t = r_last + A (x_t - x_last) / N)
// and r, v, the exact norms.  G grows by the columns that enter U (a transposed mat-vec of U's columns against each new
// one); entries whose x returned to zero stay in U and cost nothing (the lists the mat-vecs run over hold non-zeros only).
// Every sum has a fixed order: U is appended to in (block, column) order by one workgroup, the lists are compacted in U order.
// Differences to the direct launches are rounding only (measured: trace scalars agree with the oracle's to 2e-15 on the test
// problems, to 6e-13 over a soak of 1000 random ones).
// A stretch that follows a Gram-space stretch does not go back to n-space at its regular iteration (MODE 2 of sbp_gs_kernel):
// x is gathered from the dense vector `xreg` wrote, A_U'y continues its recurrence (exact dots only for the columns that just
// entered U, against the y and r the previous stretch's tail materialised), the regular iteration's decision is made from the
// same quadratic forms, and the carried ||r||^2, ||y||^2, y'zbar are re-anchored on the exact values that tail computed.  The
// first stretch, and one that follows direct launches, starts from the direct tail's n-vectors (dots<1>, MODE 1).
// Ways out.  More non-zeros than G holds (or, in a carried stretch, more new columns than fit): the merge launch marks both
// control blocks done -- every enqueued launch becomes a no-op -- and the host, seeing ust[3], lifts the halt, makes the regular
// iteration's n-vectors if the stretch was a carried one, and goes on with the direct launches; Gram space is tried again 100,
// 200, 400, ... iterations later.  U full of columns that came and went (not in a carried stretch): rebuilt from the lists.
constexpr int kGsCapMax = 1024;
constexpr int kGsRows = 8;

// After the regular iteration's list launch: append the listed columns that are not in U yet, in (block, column) order.  One
// workgroup, the lists of all blocks flattened over its threads (at most kGsCapMax entries: four per thread): three dependent
// round trips (list lengths, list entries, their slots in U).
constexpr int kGsMaxBlocks = 256;
static_assert(kGsCapMax == 4 * kSbpThreads, "the merge launch flattens the blocks' lists over its threads, four entries per thread (ADVICE r5)");
static_assert(kGsMaxBlocks <= kSbpThreads, "one thread per block in the merge launch's prefix scan");
__global__ void __launch_bounds__(kSbpThreads)
sbp_gs_merge_kernel(SbpParams q, int par, int g, int may_reset) {
    __shared__ int pre[kGsMaxBlocks + 1], c0s[kGsMaxBlocks];
    __shared__ int wcnt[4][4];
    const SbpGram& s = q.gs;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    int cb = 0;
    if (tid < q.NL) { cb = q.cnt[tid]; c0s[tid] = q.blk[tid].c0; }
    int uc = s.ust[0];
    const SbpCtl c = load_ctl_vector(q.ctl + par);                  // written by this iteration's xreg launch
    if (c.done) return;
    if (tid < q.NL) pre[tid + 1] = cb;
    __syncthreads();
    if (tid == 0) { pre[0] = 0; for (int b = 0; b < q.NL; ++b) pre[b + 1] += pre[b]; }
    __syncthreads();
    const int T = pre[q.NL];
    bool halt = T > s.cap;
    if (!halt && !may_reset) {
        // a stretch that carries its state over from the previous one cannot have U re-numbered under it: count first
        int newc = 0;
        for (int i = tid; i < T; i += kSbpThreads) {
            int lo = 0, hi = q.NL - 1;
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (pre[mid] <= i) lo = mid; else hi = mid - 1; }
            newc += s.umap[c0s[lo] + q.list[c0s[lo] + i - pre[lo]]] < 0 ? 1 : 0;
        }
        newc = wave_sum(newc);
        if (lane == 0) wcnt[0][wid] = newc;
        __syncthreads();
        halt = uc + wcnt[0][0] + wcnt[0][1] + wcnt[0][2] + wcnt[0][3] > s.cap;
        __syncthreads();
    }
    if (halt) {
        // more non-zeros than G has room for: halt the enqueued launches (they all honour `done`); the host resumes from
        // iteration g + 1 with the direct launches
        if (tid == 0) {
            s.ust[2] = 0; s.ust[3] = 1; s.ust[4] = g;
            SbpCtl h = c;                                           // BOTH copies: the Gram-space launches that follow return on
            h.done = 1;                                             // ust[2] == 0 without forwarding the control block
            q.ctl[par] = h; q.ctl[par ^ 1] = h;
            *q.done = 1; if (q.hflag) *q.hflag = 1;
        }
        return;
    }
    int col[4], blkv[4];
    bool nw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = tid + k * kSbpThreads;
        col[k] = -1; blkv[k] = 0;
        if (i < T) {
            int lo = 0, hi = q.NL - 1;                              // the block of flattened entry i: pre[b] <= i < pre[b + 1]
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (pre[mid] <= i) lo = mid; else hi = mid - 1; }
            blkv[k] = lo;
            col[k] = c0s[lo] + q.list[c0s[lo] + i - pre[lo]];
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) nw[k] = col[k] >= 0 && s.umap[col[k]] < 0;
    for (int pass = 0; pass < 2; ++pass) {
        unsigned long long bal[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { bal[k] = __ballot(nw[k]); if (lane == 0) wcnt[k][wid] = __popcll(bal[k]); }
        __syncthreads();
        int newc = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            for (int w = 0; w < 4; ++w) newc += wcnt[k][w];
        if (pass == 0 && uc + newc > s.cap) {                       // forget U and start from the current lists (T <= cap)
            for (int a = tid; a < uc; a += kSbpThreads) { const int cc = s.ucol[a]; s.umap[cc] = -1; s.sxd[cc] = 0.0; }
            if (tid == 0) s.ust[5] += 1;
            uc = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) nw[k] = col[k] >= 0;
            __syncthreads();
            continue;
        }
        int base = uc;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int before = 0, tot = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { const int cw = wcnt[k][w]; before += w < wid ? cw : 0; tot += cw; }
            if (nw[k]) {
                const int a = base + before + __popcll(bal[k] & ((1ull << lane) - 1ull));
                const SbpBlk bi = q.blk[blkv[k]];
                s.umap[col[k]] = a; s.ucol[a] = col[k]; s.ubid[a] = blkv[k];
                s.ugp[2 * a] = 1.0 / bi.gamma; s.ugp[2 * a + 1] = bi.pen;
            }
            base += tot;
        }
        if (tid == 0) { s.ust[1] = uc; s.ust[0] = base; s.ust[2] = 1; s.ust[6] += 1; }
        break;
    }
}

// One column against a vector in LDS (a wave; 16-byte loads, a chunk of the column in flight).
__device__ __forceinline__ double sbp_wave_dot(const double* __restrict__ col, const double* vsh, int nk, int lane) {
    const double2* a = reinterpret_cast<const double2*>(col) + lane;
    const double2* vv = reinterpret_cast<const double2*>(vsh) + lane;
    double d0 = 0.0, d1 = 0.0;
    for (int k0 = 0; k0 < nk; k0 += kSbpChunk) {
        double2 x[kSbpChunk];
#pragma unroll
        for (int u = 0; u < kSbpChunk; ++u) x[u] = k0 + u < nk ? a[(size_t)(k0 + u) * 64] : make_double2(0.0, 0.0);
#pragma unroll
        for (int u = 0; u < kSbpChunk; u += 2) {
            if (k0 + u < nk) { const double2 w = vv[(k0 + u) * 64]; d0 = fma(x[u].x, w.x, d0); d0 = fma(x[u].y, w.y, d0); }
            if (k0 + u + 1 < nk) { const double2 w = vv[(k0 + u + 1) * 64]; d1 = fma(x[u + 1].x, w.x, d1); d1 = fma(x[u + 1].y, w.y, d1); }
        }
    }
    return wave_sum(d0 + d1);
}

// MODE 0 (grid x: shares of U, y: shares of the new entries): column c of G and gz for every entry c that joined U at this
// regular iteration -- the new column staged in LDS, a wave per column of U.  Pairs of two new entries are computed from both
// sides with the same products in the same order: the two stores carry the same bits.
// MODE 1 (grid x: U four entries at a time): the start of a stretch -- gy = A_U'y, x_U gathered from x, the sum of iterates cleared,
// y'zbar.
template <int MODE>
__global__ void __launch_bounds__(kSbpThreads)
sbp_gs_dots_kernel(SbpParams q, int idx) {
    extern __shared__ __attribute__((aligned(16))) double vsh[];    // npad doubles
    const SbpGram& s = q.gs;
    if (load_flag_vector(q.done) != 0) return;
    if (load_flag_vector(s.ust + 2) == 0) return;
    const int uc = load_flag_vector(s.ust), uold = load_flag_vector(s.ust + 1);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nk = q.npad / 128;
    if (MODE == 1) {
        const bool last = blockIdx.x == gridDim.x - 1;              // (never one of U's: the grid has one workgroup more)
        if ((int)blockIdx.x * 4 >= uc && !last) return;
        for (int k = threadIdx.x; k < q.npad; k += kSbpThreads) vsh[k] = q.y[k];
        __syncthreads();
        const int a = blockIdx.x * 4 + wid;
        if (a < uc) {
            const int col = s.ucol[a];
            const double xa = q.x[col];
            const double d = sbp_wave_dot(q.A + (size_t)col * q.lda, vsh, nk, lane);
            if (lane == 0) { s.gy[idx * s.cap + a] = d; s.xs[idx * s.cap + a] = xa; s.sx[a] = 0.0; }
        }
        if (last && wid == 0) {
            const double d = sbp_wave_dot(q.zbar, vsh, nk, lane);
            if (lane == 0) s.sc[8] = d;
        }
    } else {
        if (uold >= uc) return;
        for (int c = uold + blockIdx.y; c < uc; c += gridDim.y) {
            __syncthreads();                                        // (the previous column's readers are done)
            const double* ac = q.A + (size_t)s.ucol[c] * q.lda;
            for (int k = threadIdx.x; k < q.npad; k += kSbpThreads) vsh[k] = ac[k];
            __syncthreads();
            for (int a = blockIdx.x * 4 + wid; a < uc; a += gridDim.x * 4) {
                const double d = sbp_wave_dot(q.A + (size_t)s.ucol[a] * q.lda, vsh, nk, lane);
                if (lane == 0) { s.G[(size_t)c * s.ldg + a] = d; s.G[(size_t)a * s.ldg + c] = d; }
            }
            if (blockIdx.x == 0) {
                // the entry's place in the vectors a stretch that follows a Gram-space stretch starts from (buffer idx): it was
                // zero there, A_c'y and A_c'r of the n-vectors the previous stretch's tail left are exact products
                const double* vec = wid == 0 ? q.zbar : (wid == 1 ? q.y : q.r);
                if (wid < 3) {
                    const double d = sbp_wave_dot(vec, vsh, nk, lane);
                    if (lane == 0) { if (wid == 0) s.gz[c] = d; else if (wid == 1) s.gy[idx * s.cap + c] = d; else s.hr[idx * s.cap + c] = d; }
                } else if (lane == 0) {
                    s.xs[idx * s.cap + c] = 0.0;
                }
            }
        }
    }
}

// A workgroup barrier that orders LDS traffic only.  __syncthreads() is a fence: it also waits for every outstanding global
// load (s_waitcnt vmcnt(0)) -- here that would be the 32 requests for G per thread that are in flight across the decision.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int NV>
__device__ __forceinline__ void block_sum_lds(double (&v)[NV], double* scratch) {      // block_sum (device_utils.h), 4 waves, LDS-only barriers
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
    lds_barrier();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) scratch[i * 4 + wid] = v[i];
    }
    lds_barrier();
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = ((scratch[i * 4] + scratch[i * 4 + 1]) + scratch[i * 4 + 2]) + scratch[i * 4 + 3];
}

// The decision from the Gram-space partial sums of the previous launch (pp: this thread's row of them, already requested)
// and the carried norms.
__device__ __forceinline__ bool sbp_decide_gram(const SbpParams& q, const SbpCtl& in, int par, int idx, double (&p)[7],
                                                double R, double Y, double yz, SbpCtl& out, double* red) {
    const SbpGram& s = q.gs;
    SbpCtl* outp = &q.ctl[par ^ 1];
    out = in;
    if (in.done) {
        if (blockIdx.x == 0 && threadIdx.x == 0) *outp = in;
        return false;
    }
    block_sum_lds<7>(p, red);
    const double D = p[0], hdx = p[1], xh = p[2], sax = p[3], qq = p[4], gyx = p[5], gzx = p[6];
    const double iN = q.invN;                                       // (a double division is ~0.2 us of a wave's time: none in this launch)
    const double Rn = fmax(R + 2.0 * hdx * iN + D * (iN * iN), 0.0);
    const double yr = gyx * iN - yz, rz = gzx * iN - s.zz;
    const double Yn = fmax(Y + 2.0 * q.rho * yr + q.rho * q.rho * Rn, 0.0);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double* o = s.sc + (idx ^ 1) * 4;
        o[0] = Rn; o[1] = Yn; o[2] = yz + q.rho * rz;
    }
    return sbp_decide_core(q, in, outp, out, D * iN, D * (iN * iN), Rn, Yn, xh * iN, sax, qq);
}

// One active-set iteration in Gram space (INIT: the start of a stretch -- A_U'r for the x the regular iteration left, the
// carried norms from the tail's exact partials).  grid: cap / 8 workgroups; workgroup g owns rows [8 g, 8 g + 8) of G.
// par: the control block's parity, idx: the buffer this launch reads (it writes idx ^ 1), first: the decision reads the
// regular iteration's norm partials, nth: the iterates summed after this launch.
// Latency is what the launch costs: everything that does not depend on the decision -- U's state, the previous launch's
// partials, the rows' own entries -- is requested before the first wait (the buffers are `cap` long, so the requests need not
// know U's length), and the G requests of the mat-vecs are all in flight together.
constexpr int kGsSlices = 64;                                       // slices of the list: a thread owns one slice and two of the workgroup's eight rows
constexpr int kGsFlight = kGsCapMax / kGsSlices;                    // 16-byte G requests per thread: the whole list in one round trip
// "This value is needed HERE": keeps a prefetch where it was written (the compiler otherwise sinks a load into the branch
// that uses it, i.e. behind the decision -- one more dependent round trip; measured 3.2 -> 0.7 us for the decision).
__device__ __forceinline__ void pin(double v) { asm volatile("" :: "v"(v)); }
__device__ __forceinline__ void pin(int v) { asm volatile("" :: "v"(v)); }
// MODE 0: an active-set iteration.  MODE 1 (INIT): the start of a stretch after an iteration that ran with the direct launches.
// MODE 2 (REG): the start of a stretch that follows a Gram-space stretch -- the regular iteration's x (gathered from the dense
// vector xreg wrote) takes the place of the soft-thresholds: its partials make the NEXT launch's decision the regular
// iteration's, A_U'y follows its recurrence across the regular iteration, and the carried norms are the exact ones the
// previous stretch's tail left.  No n-vector is touched at the regular iteration then (xact / tail / dots<1> are not launched).
template <int MODE>
__global__ void __launch_bounds__(kSbpThreads)
sbp_gs_kernel(SbpParams q, int par, int idx, int first, int nth) {
    constexpr bool INIT = MODE == 1, REG = MODE == 2;
    __shared__ double red[8 * 4];
    __shared__ double xsh[kGsCapMax];
    __shared__ double lx[kGsCapMax], ld[kGsCapMax];
    __shared__ int lc[kGsCapMax], lb[kGsCapMax];
    __shared__ int wcnt[4][4];
    __shared__ double shr[4][kGsSlices][kGsRows];
    const SbpGram& s = q.gs;
    const int cap = s.cap;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (s.test_delay > 0 && (blockIdx.x & 1) == 0) {               // (workgroup 0 among them: its decision is the one that is recorded)
        // Test hook (ADMM_HIP_SBP_TEST_DELAY_US, tests/test_gpu_parbp.py): the workgroups of a launch do not start together -- make
        // that visible.  Whatever one workgroup writes at the end of a launch must not be what another reads at its start.
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < s.test_delay) __builtin_amdgcn_s_sleep(8);
    }
    // ---- requests that depend on nothing
    double xt[4], gyv[4], hrv[4], gam[4], pen[4];
    int bidv[4], ucv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int a = tid + k * kSbpThreads;
        xt[k] = 0.0; gyv[k] = 0.0; hrv[k] = 0.0; gam[k] = 1.0; pen[k] = 0.0; bidv[k] = 0; ucv[k] = 0;
        if (a < cap) {
            xt[k] = s.xs[idx * cap + a];
            bidv[k] = s.ubid[a];
            if (REG) ucv[k] = s.ucol[a];
            if (MODE == 0) {
                gyv[k] = s.gy[idx * cap + a]; hrv[k] = s.hr[idx * cap + a];
                const double2 gp = reinterpret_cast<const double2*>(s.ugp)[a];
                gam[k] = gp.x; pen[k] = gp.y;
            }
        }
    }
    const int a0 = blockIdx.x * kGsRows;
    const int r = tid & (kGsRows - 1);                              // the row this thread finishes (threads 0..7)
    const int rp = tid & 3, cs = tid >> 2;                          // mat-vec: row pair and list slice
    const int arow = a0 + r;                                        // < cap: the grid is cap / 8
    const int2 pbid0 = *reinterpret_cast<const int2*>(s.ubid + a0 + 2 * rp);
    double e_gz = 0.0, e_xo = 0.0, e_hro = 0.0, e_gyo = 0.0, e_sx = 0.0;
    int e_col = 0;
    if (tid < kGsRows) {
        e_gz = s.gz[arow];
        if (!INIT) { e_xo = s.xs[idx * cap + arow]; e_hro = s.hr[idx * cap + arow]; e_gyo = s.gy[idx * cap + arow]; e_sx = s.sx[arow]; e_col = s.ucol[arow]; }
    }
    double pp[7] = {0, 0, 0, 0, 0, 0, 0};
    double cR = 0.0, cY = 0.0, cyz = 0.0;
    if (MODE == 0 && !first) {
        if (tid < cap / kGsRows) {
            const double2* pr = reinterpret_cast<const double2*>(s.Ps + ((size_t)idx * (cap / kGsRows) + tid) * 8);
            const double2 p0 = pr[0], p1 = pr[1], p2 = pr[2], p3 = pr[3];
            pp[0] = p0.x; pp[1] = p0.y; pp[2] = p1.x; pp[3] = p1.y; pp[4] = p2.x; pp[5] = p2.y; pp[6] = p3.x;
        }
        cR = s.sc[idx * 4]; cY = s.sc[idx * 4 + 1]; cyz = s.sc[idx * 4 + 2];
    }
    SbpCtl in{};
    if (INIT || !first) in = load_ctl_vector(q.ctl + par);
    const int smode = load_flag_vector(s.ust + 2);
    const int uc = load_flag_vector(s.ust);
#pragma unroll
    for (int k = 0; k < 4; ++k) { pin(xt[k]); pin(gyv[k]); pin(hrv[k]); pin(gam[k]); pin(pen[k]); pin(bidv[k]); pin(ucv[k]); }
#pragma unroll
    for (int k = 0; k < 7; ++k) pin(pp[k]);
    pin(cR); pin(cY); pin(cyz); pin(pbid0.x); pin(pbid0.y); pin(e_gz); pin(e_xo); pin(e_hro); pin(e_gyo); pin(e_sx); pin(e_col);
    if (smode == 0) return;
    if (INIT || !first) { if (in.done) { if (MODE == 0 && blockIdx.x == 0 && tid == 0) q.ctl[par ^ 1] = in; return; } }
    double xg[4] = {0.0, 0.0, 0.0, 0.0};                            // REG: the x the regular iteration left
    if (REG) {
#pragma unroll
        for (int k = 0; k < 4; ++k) if (tid + k * kSbpThreads < uc) xg[k] = q.x[ucv[k]];
    }
    // ---- the entries the mat-vecs run over (ascending): the pattern of x, known before the decision
    bool nz[4];
    unsigned long long bal[4];
    int slot[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int a = tid + k * kSbpThreads;
        if (a >= uc) xt[k] = 0.0;
        nz[k] = xt[k] != 0.0 || (REG && xg[k] != 0.0);              // an entry the active set has already pruned stays zero (PADMMBP.h:43)
        bal[k] = __ballot(nz[k]);
        if (lane == 0) wcnt[k][wid] = __popcll(bal[k]);
    }
    __syncthreads();
    int nnz = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int before = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int cw = wcnt[k][w]; before += w < wid ? cw : 0; tot += cw; }
        slot[k] = nnz + before + __popcll(bal[k] & ((1ull << lane) - 1ull));
        if (nz[k]) { lc[slot[k]] = tid + k * kSbpThreads; lb[slot[k]] = bidv[k]; }
        nnz += tot;
    }
    __syncthreads();
    const bool valid = arow < uc;
    // ---- rows a0 .. a0 + 7 of G, the listed columns: requested now, used after the decision (64 slices of the list; a thread
    // takes two rows with one 16-byte request -- rows beyond U hold zeros or stale numbers and are never stored)
    double2 gv[kGsFlight];
    {
        const double* Gr = s.G + a0 + 2 * rp;
#pragma unroll
        for (int j = 0; j < kGsFlight; ++j) {
            const int e = cs + kGsSlices * j;
            gv[j] = (e < nnz && a0 < uc) ? *reinterpret_cast<const double2*>(Gr + (size_t)lc[e] * s.ldg) : make_double2(0.0, 0.0);
        }
    }
    SbpCtl out;
    if (INIT || REG) {
        out = in;                                                   // written by this (regular) iteration's xreg launch
        if (blockIdx.x == 0) {                                      // the carried norms: exact, from the tail's partials
            double p[3] = {0.0, 0.0, 0.0};
            for (int w = tid; w < q.nT; w += kSbpThreads) { p[0] += q.Q[w * 8 + 2]; p[1] += q.Q[w * 8 + 3]; p[2] += q.Q[w * 8 + 5]; }
            block_sum<double, 3>(p, red);
            // INIT: for the launch that reads this buffer next (the first iteration decides from the direct tail's partials and
            // forwards them); REG: for the next launch, whose decision is this iteration's
            double* o = s.sc + (REG ? idx ^ 1 : idx) * 4;
            if (tid == 0) { o[0] = p[0]; o[1] = p[1]; o[2] = REG ? p[2] : s.sc[8]; }
        }
    } else if (first) {
        if (!sbp_decide(q, par, out, red)) return;
        if (blockIdx.x == 0 && tid == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) s.sc[(idx ^ 1) * 4 + k] = s.sc[idx * 4 + k];
        }
    } else {
        if (tid >= (uc + kGsRows - 1) / kGsRows) {                  // partial rows of workgroups beyond U are stale
#pragma unroll
            for (int k = 0; k < 7; ++k) pp[k] = 0.0;
        }
        if (!sbp_decide_gram(q, in, par, idx, pp, cR, cY, cyz, out, red)) return;
    }
    if (a0 >= uc) return;                                           // no rows (the decision's stores, workgroup 0's, are done)
    // ---- the new x for every entry of U
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int a = tid + k * kSbpThreads;
        double xn = REG ? xg[k] : xt[k];
        if (MODE == 0 && xt[k] != 0.0) {
#pragma clang fp contract(off)
            const double d = gyv[k] * q.inv_rho + hrv[k];
            xn = sbp_soft(xt[k] - d * gam[k], pen[k]);                // gam: 1 / gamma
        }
        if (a < uc) xsh[a] = xn;
        if (nz[k]) { lx[slot[k]] = xn; ld[slot[k]] = xn - xt[k]; }
    }
    __syncthreads();
    double au[2] = {0.0, 0.0}, ab[2] = {0.0, 0.0}, aw[2] = {0.0, 0.0}, awb[2] = {0.0, 0.0};
#pragma unroll
    for (int j = 0; j < kGsFlight; ++j) {
        const int e = cs + kGsSlices * j;
        if (e < nnz) {
            const double xe = lx[e], de = ld[e];
            const int be = lb[e];
            au[0] = fma(gv[j].x, xe, au[0]); aw[0] = fma(gv[j].x, de, aw[0]);
            au[1] = fma(gv[j].y, xe, au[1]); aw[1] = fma(gv[j].y, de, aw[1]);
            if (be == pbid0.x) { ab[0] = fma(gv[j].x, xe, ab[0]); awb[0] = fma(gv[j].x, de, awb[0]); }
            if (be == pbid0.y) { ab[1] = fma(gv[j].y, xe, ab[1]); awb[1] = fma(gv[j].y, de, awb[1]); }
        }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) { shr[0][cs][2 * rp + h] = au[h]; shr[1][cs][2 * rp + h] = ab[h]; shr[2][cs][2 * rp + h] = aw[h]; shr[3][cs][2 * rp + h] = awb[h]; }
    __syncthreads();
    // ---- the 64 slices' sums: thread (quantity, row, quarter) adds sixteen slices in order, the quarters are added in order
    double part = 0.0;
    if (tid < 128) {
        const int qn = tid >> 5, rr = (tid >> 2) & 7, h = tid & 3;
#pragma unroll
        for (int c = 0; c < kGsSlices / 4; ++c) part += shr[qn][h * (kGsSlices / 4) + c][rr];
    }
    part = part + __shfl_down(part, 1);
    const double part2 = __shfl_down(part, 2);
    part = part + part2;                                            // lanes with (tid & 3) == 0: ((h0 + h1) + (h2 + h3))
    __syncthreads();
    if (tid < 128 && (tid & 3) == 0) shr[tid >> 5][0][(tid >> 2) & 7] = part;
    __syncthreads();
    if (tid >= kGsRows) return;
    const double u = shr[0][0][r], ub = shr[1][0][r], w = shr[2][0][r], wb = shr[3][0][r];
    double p[7] = {0, 0, 0, 0, 0, 0, 0};
    if (valid) {
        const double hrn = u * q.invN - e_gz;
        if (INIT) {
            s.hr[idx * cap + arow] = hrn;
        } else {
            const double xv = xsh[arow], dx = xv - e_xo;
            p[0] = dx * w; p[1] = e_hro * dx; p[2] = xv * hrn; p[3] = xv * ub; p[4] = dx * wb; p[5] = e_gyo * xv; p[6] = e_gz * xv;
            const int o = (idx ^ 1) * cap + arow;
            s.xs[o] = xv; s.hr[o] = hrn; s.gy[o] = e_gyo + q.rho * hrn;
            const double sxn = (REG ? 0.0 : e_sx) + xv;                // REG: the stretch's sum starts with the regular iteration's x
            s.sx[arow] = sxn;
            if (!REG) q.x[e_col] = xv;                              // (REG: that is where xv was read from, by every workgroup)
            s.sxd[e_col] = sxn - (double)nth * xv;
        }
    }
    if (INIT) return;
    // the eight rows' partials, in row order (lanes 0..7 of wave 0)
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < kGsRows; ++i) t += readlane_f64(p[k], i);
        p[k] = t;
    }
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < 7; ++k) s.Ps[((size_t)(idx ^ 1) * (cap / kGsRows) + blockIdx.x) * 8 + k] = p[k];
    }
}

// End of a stretch: the n-vectors from the gather partials (see the section header).  grid: npad / 64 tiles, eight waves.
__global__ void __launch_bounds__(64 * kSbpTailWaves)
sbp_gs_tail_kernel(SbpParams q, int par, int nth) {
    __shared__ double shA[kSbpTailBlocks][64], shT[kSbpTailBlocks][64];
    __shared__ double red[2 * kSbpTailWaves];
    const SbpGram& s = q.gs;
    if (load_flag_vector(s.ust + 2) == 0) return;
    const int uc = load_flag_vector(s.ust);
    const SbpCtl c = load_ctl_vector(q.ctl + par);                  // written by the stretch's last iteration
    if (c.done) return;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int row = blockIdx.x * 64 + lane;
    double D = 0.0, qq = 0.0;
    if (blockIdx.x == 0) {                                          // the two sums that only exist in Gram space
        double p[2] = {0.0, 0.0};
        const int nR = (uc + kGsRows - 1) / kGsRows;
        const double* Pl = s.Ps + (size_t)par * (s.cap / kGsRows) * 8;       // the buffer the stretch's last iteration wrote: it read parity par ^ 1
        for (int w = threadIdx.x; w < nR; w += 64 * kSbpTailWaves) { p[0] += Pl[w * 8]; p[1] += Pl[w * 8 + 4]; }
        block_sum<double, 2>(p, red);
        D = p[0]; qq = p[1];
    }
    double S = 0.0, Tq = 0.0, sax = 0.0;                            // wave 0 only
    for (int b0 = 0; b0 < q.NL; b0 += kSbpTailBlocks) {
        const int b = b0 + grp;
        double av = 0.0, tv = 0.0;
        if (b < q.NL) {
            const double* pp = s.partP + (size_t)b * s.ngroups * s.pstride + row;
            const double* pt = s.partT + (size_t)b * s.ngroups * s.pstride + row;
            for (int gq = 0; gq < s.ngroups; ++gq) { av += pp[(size_t)gq * s.pstride]; tv += pt[(size_t)gq * s.pstride]; }
            q.Axo[(size_t)b * q.npad + row] = av;
        }
        __syncthreads();
        shA[grp][lane] = av; shT[grp][lane] = tv;
        __syncthreads();
        if (grp == 0) {
            const int nb = min(kSbpTailBlocks, q.NL - b0);
            for (int t = 0; t < nb; ++t) { const double a = shA[t][lane]; S += a; sax += a * a; Tq += shT[t][lane]; }
        }
    }
    if (grp != 0) return;
    sax = wave_sum(sax);
    const bool t0 = blockIdx.x == 0;
    if (lane == 0) { q.Qa[blockIdx.x * 2] = sax; q.Qa[blockIdx.x * 2 + 1] = t0 ? qq : 0.0; }
    double r2, y2, abar_r, yzp;
    {
#pragma clang fp contract(off)
        q.Sold[row] = S;
        const double abar = S / q.dN;
        const double rn = abar - q.zbar[row];
        q.r[row] = rn;
        const double yn = q.y[row] + q.rho * ((double)nth * rn + Tq / q.dN);
        q.y[row] = yn;
        q.v[row] = yn / q.rho + rn;
        r2 = rn * rn; y2 = yn * yn; abar_r = abar * rn; yzp = yn * q.zbar[row];
    }
    r2 = wave_sum(r2); y2 = wave_sum(y2); abar_r = wave_sum(abar_r); yzp = wave_sum(yzp);
    if (lane == 0) {
        double* Q = q.Q + (size_t)blockIdx.x * 8;
        Q[0] = t0 ? D / q.dN : 0.0; Q[1] = t0 ? D / (q.dN * q.dN) : 0.0; Q[2] = r2; Q[3] = y2; Q[4] = abar_r;
        Q[5] = yzp;                                                 // y'zbar, exact: the next stretch carries it on (MODE 2 above)
    }
}

// ------------------------------------------------------------------------------------------------ setup
static int env_int(const char* name, int dflt) {          // a positive option value, or the default
    const int v = option_int(name, 0);
    return v > 0 ? v : dflt;
}


// Largest eigenvalue of a symmetric tridiagonal matrix (diagonal a[0..m), off-diagonal e[0..m-1)) and the LAST component of
// its unit eigenvector: implicit QL with the rotations applied to one row only.
static void tridiag_top(std::vector<double> d, std::vector<double> e, double* theta, double* last) {
    const int m = (int)d.size();
    std::vector<double> z(m, 0.0);
    z[m - 1] = 1.0;
    e.resize(m, 0.0);
    for (int l = 0; l < m; ++l) {
        int iter = 0, mm;
        do {
            for (mm = l; mm < m - 1; ++mm) {
                const double dd = std::fabs(d[mm]) + std::fabs(d[mm + 1]);
                if (std::fabs(e[mm]) <= 2.3e-16 * dd) break;
            }
            if (mm != l) {
                if (iter++ == 200) throw Error(ADMM_ERR_EIGS, "tridiagonal QL: no convergence");
                double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
                double r = std::hypot(g, 1.0);
                g = d[mm] - d[l] + e[l] / (g + (g >= 0 ? std::fabs(r) : -std::fabs(r)));
                double s = 1.0, c = 1.0, p = 0.0;
                int i;
                for (i = mm - 1; i >= l; --i) {
                    double f = s * e[i], b = c * e[i];
                    r = std::hypot(f, g);
                    e[i + 1] = r;
                    if (r == 0.0) { d[i + 1] -= p; e[mm] = 0.0; break; }
                    s = f / r; c = g / r;
                    g = d[i + 1] - p;
                    r = (d[i] - g) * s + 2.0 * c * b;
                    p = s * r;
                    d[i + 1] = g + p;
                    g = c * r - b;
                    f = z[i + 1];
                    z[i + 1] = s * z[i] + c * f;
                    z[i] = c * z[i] - s * f;
                }
                if (r == 0.0 && i >= l) continue;
                d[l] -= p; e[l] = g; e[mm] = 0.0;
            }
        } while (mm != l);
    }
    int k = 0;
    for (int i = 1; i < m; ++i) if (d[i] > d[k]) k = i;
    *theta = d[k]; *last = z[k];
}

// lambda_max(A_b'A_b) = lambda_max(A_b A_b') : Lanczos with full re-orthogonalisation on the n x n matrix A_b A_b' (fp64
// matrix-core Gram), products on the device, the short recurrences on the host; run until the Ritz pair's residual bound
// |beta_m s_m| is below 1e-14 of the value (the reference asks an R function that does not exist: PADMMBP.h:64-71).
// Round 4: the Krylov basis stays on the DEVICE and so does the full re-orthogonalisation (two passes of c = V'w, w -= V c): round 3
// kept the basis on the host and orthogonalised there in scalar loops -- ~150 steps per block at the C5 shape, 4.7e8 host flops per
// block, 0.71 s of the solver's 1.09 s.  Per step the host now receives two numbers (alpha_j = v_j'A v_j, ||w||^2), eight steps at a time.
// All kernels below take the column block from blockIdx.y: the Lanczos runs of several blocks advance in lockstep (same n, same
// step), one launch per operation for all of them -- at the C5 shape 8 blocks x ~155 steps x 8 small launches were 0.12 s of
// launches that each kept the device busy for 7 us.
struct LzBatch {
    double* V; long long ldv, vblk;                                 // basis of block b: V + b * vblk, columns ldv apart
    double* w; double* vcur; long long wblk;                        // w and the current basis vector (the mat-vec's fixed input) of block b: + b * wblk
    double* c; long long cblk;                                      // coefficients of block b: c + b * cblk
    double* al; double* b2; long long sblk;                         // per-step alpha and ||w||^2 of block b: + b * sblk
    const double* part; long long pstride, pblk; int nseg;          // the mat-vec's partial rows of block b: part + b * pblk
    int n;
};
__global__ void __launch_bounds__(256) sbp_lz_reduce_kernel(LzBatch q) {
    const int i = blockIdx.x * 256 + threadIdx.x;                   // w = the mat-vec's partial rows summed in order
    if (i >= q.n) return;
    const double* p = q.part + (size_t)blockIdx.y * q.pblk + i;
    double t = 0.0;
    for (int k = 0; k < q.nseg; ++k) t += p[(size_t)k * q.pstride];
    q.w[(size_t)blockIdx.y * q.wblk + i] = t;
}
// c[k] = V[:, k]'w, one workgroup per column k (`self`: w'w into b2[step] instead); the LAST coefficient of the first pass is alpha
__global__ void __launch_bounds__(256) sbp_lz_dots_kernel(LzBatch q, int step, int self, int keep) {
    __shared__ double red[4];
    const double* w = q.w + (size_t)blockIdx.y * q.wblk;
    const double* col = self ? w : q.V + (size_t)blockIdx.y * q.vblk + (size_t)blockIdx.x * q.ldv;
    double s = 0.0;
    for (int i = threadIdx.x; i < q.n; i += 256) s += col[i] * w[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double v = (red[0] + red[1]) + (red[2] + red[3]);
        if (self) { q.b2[(size_t)blockIdx.y * q.sblk + step] = v; return; }
        q.c[(size_t)blockIdx.y * q.cblk + blockIdx.x] = v;
        if (keep && blockIdx.x == gridDim.x - 1) q.al[(size_t)blockIdx.y * q.sblk + step] = v;
    }
}
// w[i] -= sum_k V[i, k] c[k]: a workgroup = 32 rows x 8 slices of k (k = slice, slice + 8, ... ascending; the slices added in order)
__global__ void __launch_bounds__(256) sbp_lz_update_kernel(LzBatch q, int m) {
    __shared__ double sh[8][32];
    const int r = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + r;
    const double* V = q.V + (size_t)blockIdx.y * q.vblk;
    const double* c = q.c + (size_t)blockIdx.y * q.cblk;
    const long long ldv = q.ldv;
    double acc = 0.0;
    if (i < q.n) {
        int k = sl;
        for (; k + 24 < m; k += 32) {
            const double v0 = V[(size_t)k * ldv + i], v1 = V[(size_t)(k + 8) * ldv + i], v2 = V[(size_t)(k + 16) * ldv + i], v3 = V[(size_t)(k + 24) * ldv + i];
            acc += v0 * c[k]; acc += v1 * c[k + 8]; acc += v2 * c[k + 16]; acc += v3 * c[k + 24];
        }
        for (; k < m; k += 8) acc += V[(size_t)k * ldv + i] * c[k];
    }
    sh[sl][r] = acc;
    __syncthreads();
    if (sl == 0 && i < q.n) {
        double t = sh[0][r];
#pragma unroll
        for (int q8 = 1; q8 < 8; ++q8) t += sh[q8][r];
        q.w[(size_t)blockIdx.y * q.wblk + i] -= t;
    }
}
// v_next = w / ||w||, the norm's square read from the device (the host sees it a few steps later); also the mat-vec's input
__global__ void __launch_bounds__(256) sbp_lz_scale_kernel(LzBatch q, int step) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const double inv = 1.0 / sqrt(q.b2[(size_t)blockIdx.y * q.sblk + step]);
    if (i >= q.n) return;
    const double v = q.w[(size_t)blockIdx.y * q.wblk + i] * inv;
    q.V[(size_t)blockIdx.y * q.vblk + (size_t)(step + 1) * q.ldv + i] = v;
    q.vcur[(size_t)blockIdx.y * q.wblk + i] = v;
}

// The spectral radii of the column blocks [b0, b0 + B) (local indices): their Gram matrices, then B Lanczos runs in lockstep.
static void sbp_sprad_batch(const double* A, long long lda, int n, const std::vector<int>& c0, int b0, int B, hipStream_t st, double* out, int* nsteps) {
    const long long ldg = round_up(n, 32);
    DevBuf<double> Gm((size_t)ldg * ldg * B);
    Gm.zero(st);
    for (int b = 0; b < B; ++b)
        gram_full<double>(A + (size_t)c0[b0 + b] * lda, lda, n, c0[b0 + b + 1] - c0[b0 + b], false, Gm.get() + (size_t)b * ldg * ldg, ldg, st);
    GemvTPlan pl = plan_gemv_t<double>(n, n, 1, 4);
    pl.nt = (size_t)B * (size_t)ldg * (size_t)ldg * sizeof(double) > kGemvNtBytes;      // what the launch streams, not one block of it
    const long long pstride = ldg;
    const int mmax = std::min(n, 600);
    const long long ldv = ldg;
    DevBuf<double> V((size_t)ldv * (mmax + 1) * B), w((size_t)ldg * B), vcur((size_t)ldg * B), c((size_t)(mmax + 2) * B), part((size_t)pl.nseg * pstride * B);
    DevBuf<double> d_al((size_t)mmax * B), d_b2((size_t)mmax * B);
    DevBuf<GemvTArgs<double>> d_args(B);
    V.zero(st); w.zero(st); vcur.zero(st); part.zero(st);
    std::vector<double> v0(n);
    double nrm = 0;
    for (int i = 0; i < n; ++i) { v0[i] = 1.0 + 0.5 * std::sin(0.7 * (i + 1)); nrm += v0[i] * v0[i]; }
    nrm = std::sqrt(nrm);
    for (int i = 0; i < n; ++i) v0[i] /= nrm;
    std::vector<GemvTArgs<double>> ha(B);
    for (int b = 0; b < B; ++b) {
        ADMM_HIP_CHECK(hipMemcpyAsync(V.get() + (size_t)b * ldv * (mmax + 1), v0.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
        ADMM_HIP_CHECK(hipMemcpyAsync(vcur.get() + (size_t)b * ldg, v0.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
        GemvTArgs<double>& a = ha[b];
        a.A = Gm.get() + (size_t)b * ldg * ldg; a.lda = ldg; a.m = n; a.k = n;
        a.v[0] = vcur.get() + (size_t)b * ldg; a.v[1] = nullptr; a.vparts = 1; a.vstride = 0;
        a.out[0] = part.get() + (size_t)b * pl.nseg * pstride; a.out[1] = nullptr; a.out_stride = pstride;
        a.seg_len = pl.seg_len; a.seg_alloc = pl.seg_alloc; a.nseg = pl.nseg; a.groups_per_wg = pl.groups_per_wg; a.skip = nullptr;
    }
    ADMM_HIP_CHECK(hipMemcpyAsync(d_args.get(), ha.data(), (size_t)B * sizeof(GemvTArgs<double>), hipMemcpyHostToDevice, st));
    LzBatch q{};
    q.V = V.get(); q.ldv = ldv; q.vblk = ldv * (mmax + 1);
    q.w = w.get(); q.vcur = vcur.get(); q.wblk = ldg;
    q.c = c.get(); q.cblk = mmax + 2;
    q.al = d_al.get(); q.b2 = d_b2.get(); q.sblk = mmax;
    q.part = part.get(); q.pstride = pstride; q.pblk = (long long)pl.nseg * pstride; q.nseg = pl.nseg;
    q.n = n;
    const int rows = (n + 255) / 256, rows32 = (n + 31) / 32;
    // The host needs two numbers per step and block (alpha_j, ||w||^2) but only to DECIDE, every fourth step: they are kept on the
    // device and fetched eight steps at a time (a fetch is a stream synchronisation); the steps are then judged one by one in
    // order, so the value returned is the one a step-by-step loop returns -- up to seven steps run for nothing, and a block that
    // is finished keeps stepping (its numbers are not looked at again) until the last one is.
    // The host needs two numbers per step and block (alpha_j, ||w||^2) but only to DECIDE, every fourth step: they are kept on the
    // device and fetched a group of eight steps at a time into pinned memory; the steps are then judged one by one in order, so
    // the value returned is the one a step-by-step loop returns.  Judging is host work (an implicit-QL sweep of the tridiagonal
    // matrix per block, 55 ms in all at the C5 shape -- as long as the device needs for the steps): the NEXT group is enqueued
    // before a group is judged, and the blocks are judged by a thread each.  Up to two groups run for nothing, and a block that is
    // finished keeps stepping (its numbers are not looked at again) until the last one is.
    struct Pinned {
        double* p = nullptr;
        explicit Pinned(size_t n_) { ADMM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&p), n_ * sizeof(double), hipHostMallocDefault)); }
        ~Pinned() { if (p) (void)hipHostFree(p); }
        Pinned(const Pinned&) = delete;
        Pinned& operator=(const Pinned&) = delete;
    } hal((size_t)mmax * B), hb2((size_t)mmax * B);
    std::vector<std::vector<double>> al(B), be(B);
    std::vector<char> fin(B, 0);
    std::vector<double> theta(B, 0.0);
    // judge steps [s0, s1) of block b; returns true when the block is finished (out[b] set).  Throws like the serial loop.
    auto judge = [&](int b, int s0, int s1) -> bool {
        for (int jj = s0; jj < s1; ++jj) {
            const double a = hal.p[(size_t)b * mmax + jj], bb = std::sqrt(hb2.p[(size_t)b * mmax + jj]);
            al[b].push_back(a);
            nsteps[b] = jj + 1;
            if (jj + 1 >= 2 && ((jj + 1) % 4 == 0 || jj + 1 == mmax || bb <= 1e-300)) {
                double last = 0;
                tridiag_top(al[b], be[b], &theta[b], &last);
                if (std::fabs(bb * last) <= 1e-14 * std::fabs(theta[b]) || bb <= 1e-300) { out[b] = theta[b]; return true; }
            } else if (jj == 0) {
                theta[b] = a;
                if (bb <= 1e-300 || n == 1) { out[b] = theta[b]; return true; }
            }
            if (jj + 1 == mmax) {
                // n steps span the whole space (exact up to rounding).  Fewer, without the 1e-14 residual bound met (clustered top
                // eigenvalues; that bound is close to the rounding floor of the device Gram): rho and gamma_i = 2 rho + sprad_i hang on the
                // value and an UNDER-estimate breaks the majorisation of the linearised x-update, so the safe side is returned -- an
                // eigenvalue lies within |b last| of theta, theta + |b last| bounds it from above -- and only a value that is
                // not even accurate to 1e-8 is an error.
                if (mmax < n) {
                    double last = 0;
                    tridiag_top(al[b], be[b], &theta[b], &last);
                    const double r = std::fabs(bb * last);
                    if (r <= 1e-8 * std::fabs(theta[b])) { out[b] = theta[b] + r; return true; }
                    throw Error(ADMM_ERR_EIGS, "admm_parbp: the spectral radius of a column block did not converge in 600 Lanczos steps");
                }
                out[b] = theta[b];                                  // n steps: exact up to rounding
                return true;
            }
            be[b].push_back(bb);
        }
        return false;
    };
    // groups of steps between two fetches: [0, 1), [1, 8), [8, 16), ...
    std::vector<int> gend;
    for (int j = 0; j < mmax; ++j) if (j == 0 || (j + 1) % 8 == 0 || j + 1 == mmax) gend.push_back(j + 1);
    std::vector<Event> ev(gend.size());
    auto enqueue_group = [&](size_t g) {
        const int s0 = g == 0 ? 0 : gend[g - 1], s1 = gend[g];
        for (int j = s0; j < s1; ++j) {
            launch_gemv_t_batch<double>(d_args.get(), B, pl.grid, pl.lds_bytes, pl.nt, st);
            hipLaunchKernelGGL(sbp_lz_reduce_kernel, dim3(rows, B), dim3(256), 0, st, q);
            // full re-orthogonalisation, twice; the first pass's coefficient of v_j is alpha_j = v_j'A v_j
            hipLaunchKernelGGL(sbp_lz_dots_kernel, dim3(j + 1, B), dim3(256), 0, st, q, j, 0, 1);
            hipLaunchKernelGGL(sbp_lz_update_kernel, dim3(rows32, B), dim3(256), 0, st, q, j + 1);
            hipLaunchKernelGGL(sbp_lz_dots_kernel, dim3(j + 1, B), dim3(256), 0, st, q, j, 0, 0);
            hipLaunchKernelGGL(sbp_lz_update_kernel, dim3(rows32, B), dim3(256), 0, st, q, j + 1);
            hipLaunchKernelGGL(sbp_lz_dots_kernel, dim3(1, B), dim3(256), 0, st, q, j, 1, 0);                                   // ||w||^2
            if (j + 1 < mmax) hipLaunchKernelGGL(sbp_lz_scale_kernel, dim3(rows, B), dim3(256), 0, st, q, j);
        }
        const size_t wbytes = (size_t)(s1 - s0) * sizeof(double), pitch = (size_t)mmax * sizeof(double);
        ADMM_HIP_CHECK(hipMemcpy2DAsync(hal.p + s0, pitch, d_al.get() + s0, pitch, wbytes, B, hipMemcpyDeviceToHost, st));
        ADMM_HIP_CHECK(hipMemcpy2DAsync(hb2.p + s0, pitch, d_b2.get() + s0, pitch, wbytes, B, hipMemcpyDeviceToHost, st));
        ADMM_HIP_CHECK(hipEventRecord(ev[g].e, st));
    };
    int left = B;
    enqueue_group(0);
    for (size_t g = 0; g < gend.size() && left > 0; ++g) {
        if (g + 1 < gend.size()) enqueue_group(g + 1);
        comm_event_sync(ev[g].e);
        const int s0 = g == 0 ? 0 : gend[g - 1], s1 = gend[g];
        std::vector<std::exception_ptr> err(B);
        std::vector<char> now(B, 0);
        auto run = [&](int b) { try { now[b] = judge(b, s0, s1) ? 1 : 0; } catch (...) { err[b] = std::current_exception(); } };
        std::vector<std::thread> th;
        struct Joiner { std::vector<std::thread>& t; ~Joiner() { for (auto& x : t) if (x.joinable()) x.join(); } } joiner{th};      // joined on every path out
        int first = -1;
        for (int b = 0; b < B; ++b) {
            if (fin[b]) continue;
            if (first < 0) { first = b; continue; }                 // (one block on this thread)
            try { th.emplace_back(run, b); } catch (const std::system_error&) { run(b); }      // no thread to be had: judged here
        }
        if (first >= 0) run(first);
        for (auto& t : th) t.join();
        for (int b = 0; b < B; ++b) {
            if (err[b]) { (void)hipStreamSynchronize(st); std::rethrow_exception(err[b]); }
            if (now[b]) { fin[b] = 1; --left; }
        }
    }
    comm_stream_sync(st);                        // (a group may still be running: its buffers go out of scope here)
    ADMM_HIP_CHECK(hipGetLastError());
}

static int sbp_batch() {
    const char* e = option("BATCH_ITERS");
    const int v = e ? std::atoi(e) : 0;
    return v > 0 ? (v + 1) / 2 * 2 : 20;                            // even: the parity pattern of the control block
}

// d: this rank's columns (n x pl).  nblocks: N (global); blk_first / nloc: the global blocks this rank holds; p_total.
void solve_parbp(const DeviceData<double>& d, const admm_opts& opts, int nblocks, long long p_total, long long col_offset,
                 DenseResult& res, hipStream_t st) {
    const int n = d.n, pl = d.p;
    const int N = nblocks;
    const CommInfo ci = comm_info();
    const bool dist = p_total != (long long)pl;
    ADMM_REQUIRE(!dist || ci.active, "no communicator: call admm_hip_comm_init first");
    ADMM_REQUIRE(n <= 16384, "admm_parbp: at most 16384 rows (v lives in LDS, the x-update keeps a row slice per thread); use admm_bp");
    const long long chunk = p_total / N;
    ADMM_REQUIRE(chunk >= 1, "more column blocks than columns");
    // the global partition (PADMMBP.h:150-167): N - 1 blocks of p div N columns, the last takes the remainder
    ADMM_REQUIRE(col_offset % chunk == 0 && col_offset / chunk < N, "a rank's columns must start at a block boundary");
    const int b_first = (int)(col_offset / chunk);
    std::vector<int> c0;                                            // local block starts (local column indices)
    {
        long long c = col_offset;
        int b = b_first;
        while (c < col_offset + pl) {
            c0.push_back((int)(c - col_offset));
            c = (b == N - 1) ? p_total : c + chunk;
            ++b;
        }
        ADMM_REQUIRE(c == col_offset + pl, "a rank's columns must end at a block boundary");
        c0.push_back(pl);
    }
    const int NL = (int)c0.size() - 1;
    admm_stats& S = res.stats;
    S.branch = 6;

    // ---- spectral radii, rho
    double t0 = now_s();
    std::vector<double> sprad(N, 0.0);
    int lsteps = 0;
    {
        // as many blocks in lockstep as ~3 GB of Gram matrices and Lanczos bases allow (A/B: ADMM_HIP_SBP_LZ_BATCH)
        const double per = (double)round_up(n, 32) * ((double)round_up(n, 32) + (double)std::min(n, 600) + 8.0) * 8.0;
        int Bmax = std::max(1, std::min(NL, (int)(3.0e9 / per)));
        Bmax = std::min(Bmax, env_int("SBP_LZ_BATCH", Bmax));
        for (int b = 0; b < NL; b += Bmax) {
            const int B = std::min(Bmax, NL - b);
            std::vector<int> ns(B, 0);
            sbp_sprad_batch(d.X.get(), d.ldx, n, c0, b, B, st, sprad.data() + b_first + b, ns.data());
            for (int k = 0; k < B; ++k) lsteps = std::max(lsteps, ns[k]);
        }
    }
    if (dist) {
        DevBuf<double> t(N);
        ADMM_HIP_CHECK(hipMemcpyAsync(t.get(), sprad.data(), (size_t)N * sizeof(double), hipMemcpyHostToDevice, st));
        allreduce_sum_f64(t.get(), (size_t)N, st);
        ADMM_HIP_CHECK(hipMemcpyAsync(sprad.data(), t.get(), (size_t)N * sizeof(double), hipMemcpyDeviceToHost, st));
        comm_stream_sync(st);
        comm_check();
    }
    double avg = 0;
    for (int b = 0; b < N; ++b) avg += sprad[b];
    avg /= N;
    const double rho = 1.0 / (opts.rho * avg);                     // opts.rho carries rho_ratio (R/10_admm_bp.R:115)
    S.rho = rho; S.eig_est = avg; S.t_eigs = now_s() - t0;

    // ---- layout
    const int npad = (int)round_up(n, 128);                      // 16-byte loads of the regular launch: 128 rows per wave instruction
    ADMM_REQUIRE(d.ldx >= npad || d.ldx >= n, "internal: leading dimension");
    // rows beyond n must read as zero: DeviceData pads to 32 rows, the kernels to 64 -> own copy when the paddings differ
    DevBuf<double> Aown;
    const double* A = d.X.get();
    long long lda = d.ldx;
    if (d.ldx < npad) {
        lda = npad;
        Aown.alloc((size_t)lda * pl); Aown.zero(st);
        ADMM_HIP_CHECK(hipMemcpy2DAsync(Aown.get(), (size_t)lda * sizeof(double), d.X.get(), (size_t)d.ldx * sizeof(double),
                                        (size_t)n * sizeof(double), pl, hipMemcpyDeviceToDevice, st));
        A = Aown.get();
    }
    int ncu = 256;
    { hipDeviceProp_t prop; int dev = 0; ADMM_HIP_CHECK(hipGetDevice(&dev)); ADMM_HIP_CHECK(hipGetDeviceProperties(&prop, dev)); ncu = prop.multiProcessorCount; }
    // workgroups per block: the same for every local block (block = g / Gb), about two per CU in all, never more than a block has columns
    int Gb = std::max(1, env_int("SBP_WGS", 2 * ncu) / NL);
    for (int b = 0; b < NL; ++b) Gb = std::min(Gb, c0[b + 1] - c0[b]);
    const int G = Gb * NL;
    const int nT = npad / 64;
    std::vector<SbpBlk> hblk(NL);
    for (int b = 0; b < NL; ++b) {
        hblk[b].c0 = c0[b]; hblk[b].pb = c0[b + 1] - c0[b];
        hblk[b].gamma = 2.0 * rho + sprad[b_first + b];
        hblk[b].pen = 1.0 / (rho * hblk[b].gamma);
        hblk[b].pad = 0;
    }

    DevBuf<int> d_list(pl), d_cnt(std::max(NL, 8)), d_wcount(G), d_done(1);
    DevBuf<SbpBlk> d_blk(NL);
    DevBuf<double> x(pl), xl(pl), P((size_t)G * npad), Axo((size_t)NL * npad), ex((size_t)npad + 2 * nT), Sold(npad), y(npad), r(npad), v(npad),
        zbar(npad), Q((size_t)nT * 8), trace;
    DevBuf<SbpCtl> ctl(2);
    auto h2d = [&](void* dst, const void* src, size_t bytes) { ADMM_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st)); };
    h2d(d_blk.get(), hblk.data(), (size_t)NL * sizeof(SbpBlk));
    x.zero(st); xl.zero(st); Axo.zero(st); ex.zero(st); Sold.zero(st); y.zero(st); r.zero(st); v.zero(st); zbar.zero(st); Q.zero(st);
    d_list.zero(st); d_cnt.zero(st); d_wcount.zero(st); d_done.zero(st);
    double zz = 0.0;                                                // zbar'zbar (Gram space)
    {
        std::vector<double> hz(n), hy(n);
        ADMM_HIP_CHECK(hipMemcpyAsync(hy.data(), d.Y.get(), (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
        comm_stream_sync(st);
        for (int i = 0; i < n; ++i) { hz[i] = hy[i] / (double)N; zz += hz[i] * hz[i]; }
        h2d(zbar.get(), hz.data(), (size_t)n * sizeof(double));
        comm_stream_sync(st);
    }
    SbpCtl c0ctl{};
    c0ctl.rp = c0ctl.rd = 9999.0;
    h2d(ctl.get(), &c0ctl, sizeof(SbpCtl)); h2d(ctl.get() + 1, &c0ctl, sizeof(SbpCtl));
    if (res.trace_cap > 0) { trace.alloc((size_t)res.trace_cap * ADMM_TRACE_FIELDS); trace.zero(st); }
    PinnedFlag hflag;

    SbpParams q{};
    q.min_share = std::max(1, env_int("SBP_SHARE", 4));
    q.n = n; q.npad = npad; q.N = N; q.NL = NL; q.maxit = opts.maxit; q.G = G; q.nT = nT;
    q.eps_abs = opts.eps_abs; q.eps_rel = opts.eps_rel; q.rho = rho;
    q.sqrt_nN = std::sqrt((double)n * (double)N); q.sqrtN = std::sqrt((double)N); q.dN = (double)N;
    q.invN = 1.0 / (double)N; q.inv_rho = 1.0 / rho;
    q.A = A; q.lda = lda;
    q.Gb = Gb; q.blk = d_blk.get();
    q.x = x.get(); q.list = d_list.get(); q.cnt = d_cnt.get(); q.wcount = d_wcount.get();
    q.xl = xl.get(); q.P = P.get(); q.Axo = Axo.get(); q.S = ex.get(); q.Qa = ex.get() + npad;
    q.Sold = Sold.get(); q.y = y.get(); q.r = r.get(); q.v = v.get(); q.zbar = zbar.get(); q.Q = Q.get();
    q.ctl = ctl.get(); q.done = d_done.get(); q.hflag = dist ? nullptr : hflag.p;
    q.trace = res.trace_cap > 0 ? trace.get() : nullptr; q.trace_cap = res.trace_cap;

    // ---- Gram space (single process): state, the gather launches that materialise the n-vectors
    bool gram = !dist && NL <= kGsMaxBlocks;
    if (const char* e = option("SBP_GRAM")) gram = gram && std::atoi(e) != 0;      // 0: the direct launches on every iteration (A/B)
    const int gcap = std::min(kGsCapMax, std::max(kGsRows, env_int("SBP_GRAM_CAP", kGsCapMax)) / kGsRows * kGsRows);
    DevBuf<int> g_umap, g_ucol, g_ubid, g_ust;
    DevBuf<double> g_G, g_gz, g_ugp, g_xs, g_hr, g_gy, g_sx, g_sxd, g_Ps, g_sc, g_partP, g_partT;
    DevBuf<GatherArgs<double>> g_args;
    GatherPlan gp;
    if (gram) {
        int maxpb = 0;
        for (int b = 0; b < NL; ++b) maxpb = std::max(maxpb, c0[b + 1] - c0[b]);
        gp = plan_gather<double>(n, maxpb, 2 * NL);
        g_umap.alloc(pl); g_ucol.alloc(gcap); g_ubid.alloc(gcap); g_ust.alloc(8);
        g_G.alloc((size_t)gcap * gcap); g_gz.alloc(gcap); g_ugp.alloc(2 * (size_t)gcap);
        g_xs.alloc(2 * (size_t)gcap); g_hr.alloc(2 * (size_t)gcap); g_gy.alloc(2 * (size_t)gcap); g_sx.alloc(gcap); g_sxd.alloc(pl);
        g_Ps.alloc(2 * (size_t)(gcap / kGsRows) * 8); g_sc.alloc(16);
        g_partP.alloc((size_t)NL * gp.ngroups * npad); g_partT.alloc((size_t)NL * gp.ngroups * npad);
        g_args.alloc(2 * (size_t)NL);
        ADMM_HIP_CHECK(hipMemsetAsync(g_umap.get(), 0xff, (size_t)pl * sizeof(int), st));
        g_ucol.zero(st); g_ubid.zero(st); g_ust.zero(st); g_G.zero(st); g_gz.zero(st); g_ugp.zero(st); g_xs.zero(st); g_hr.zero(st); g_gy.zero(st);
        g_sx.zero(st); g_sxd.zero(st); g_Ps.zero(st); g_sc.zero(st); g_partP.zero(st); g_partT.zero(st);
        SbpGram& gs = q.gs;
        gs.cap = gcap; gs.ldg = gcap;
        gs.umap = g_umap.get(); gs.ucol = g_ucol.get(); gs.ubid = g_ubid.get(); gs.ust = g_ust.get();
        gs.G = g_G.get(); gs.gz = g_gz.get(); gs.ugp = g_ugp.get(); gs.xs = g_xs.get(); gs.hr = g_hr.get(); gs.gy = g_gy.get();
        gs.sx = g_sx.get(); gs.sxd = g_sxd.get(); gs.Ps = g_Ps.get(); gs.sc = g_sc.get();
        gs.partP = g_partP.get(); gs.partT = g_partT.get(); gs.ngroups = gp.ngroups; gs.pstride = npad;
        gs.zz = zz;
        gs.test_delay = 100 * env_int("SBP_TEST_DELAY_US", 0); gs.pad2 = 0;
        std::vector<GatherArgs<double>> ha(2 * (size_t)NL);
        for (int k = 0; k < 2 * NL; ++k) {
            const int b = k % NL;
            GatherArgs<double>& a = ha[k];
            a.A = A + (size_t)c0[b] * lda; a.lda = lda; a.rows = n; a.cols = c0[b + 1] - c0[b];
            a.v = (k < NL ? x.get() : g_sxd.get()) + c0[b];
            a.part = (k < NL ? g_partP.get() : g_partT.get()) + (size_t)b * gp.ngroups * npad;
            a.pstride = npad; a.ngroups = gp.ngroups; a.cols_per_group = gp.cols_per_group;
            a.skip = d_done.get(); a.only_if = g_ust.get() + 2;
        }
        h2d(g_args.get(), ha.data(), ha.size() * sizeof(GatherArgs<double>));
        comm_stream_sync(st);                   // (ha leaves scope)
    }

    const bool big = npad > 8192;                                  // the x-update's row slice per thread: 16 rows up to npad = 8192, 32 beyond
    {
        // v lives in dynamic LDS (npad doubles) NEXT TO the kernels' static arrays: what must fit the default per-workgroup limit is
        // their sum -- at npad = 8192 the dynamic part alone is exactly 64 KB and the launch was refused without a trace (the
        // advisor's finding: hipLaunchKernelGGL does not surface the error and the loop ended "without a decision").  Opt in per
        // kernel whenever static + dynamic exceeds the default, and fail loudly when even the opt-in limit is too small.
        const size_t lds = (size_t)npad * sizeof(double);
        const void* fns[] = {reinterpret_cast<const void*>(&sbp_xreg_kernel<true>), reinterpret_cast<const void*>(&sbp_xreg_kernel<false>),
                             reinterpret_cast<const void*>(&sbp_xreg_screen_kernel),
                             big ? reinterpret_cast<const void*>(&sbp_xact_kernel<false, 32>) : reinterpret_cast<const void*>(&sbp_xact_kernel<false, 16>),
                             reinterpret_cast<const void*>(&sbp_gs_dots_kernel<0>), reinterpret_cast<const void*>(&sbp_gs_dots_kernel<1>)};
        for (const void* fn : fns) {
            hipFuncAttributes fa{};
            ADMM_HIP_CHECK(hipFuncGetAttributes(&fa, fn));
            const size_t need = lds + fa.sharedSizeBytes;
            if (need > device_info().lds_per_block) {
                ADMM_REQUIRE(need <= device_info().lds_optin, "admm_parbp: the rows do not fit the LDS of a workgroup");
                ADMM_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            }
        }
    }
    const bool nt = (double)lda * (double)pl * 8.0 > 220e6;       // as gemv_plan.h: beyond what the 256 MB Infinity Cache keeps
    // the regular iterations' screen (sbp_xreg_screen_kernel): worth its 2 n p bytes when A streams from HBM; SBP_SCREEN = 0 never, 1 always
    bool screen = nt;
    if (const char* e = option("SBP_SCREEN")) screen = std::atoi(e) != 0;
    DevBuf<unsigned short> Ah;
    DevBuf<float> scr_s;
    DevBuf<unsigned long long> scr_stat;
    if (screen) {
        const long long ldh = round_up(n, 8);
        try {
            Ah.alloc((size_t)pl * (size_t)ldh); scr_s.alloc(pl);
        } catch (const Error&) {
            Ah.release(); scr_s.release(); screen = false;          // the copy does not fit: unscreened
        }
        if (screen) {
            hipLaunchKernelGGL(sbp_screen_prep_kernel, dim3((unsigned)((pl + 3) / 4)), dim3(256), 0, st, A, lda, n, pl, Ah.get(), ldh, scr_s.get());
            q.Ah = Ah.get(); q.ldh = ldh; q.scr_s = scr_s.get();
            if (option("SBP_SCREEN_STATS")) { scr_stat.alloc(2); scr_stat.zero(st); q.scr_stat = scr_stat.get(); }
        }
    }
    comm_stream_sync(st);
    const size_t ldsv = (size_t)npad * sizeof(double);
    const bool gram_on = gram;
    bool carry_on = true;                                          // 0: every Gram-space stretch starts from the direct launches' n-vectors (A/B)
    if (const char* e = option("SBP_GRAM_CARRY")) carry_on = std::atoi(e) != 0;
    long long gram_from = 0;                                       // the first regular iteration whose stretch may run in Gram space
    long long last_gram = -100;                                    // the regular iteration of the last stretch enqueued in Gram space
    std::vector<char> carried;                                     // per stretch: it started from the previous stretch's Gram-space state
    bool cur_carried = false;
    auto launch_xact_tail = [&](int par) {                         // a regular iteration's n-vectors (direct launches)
        if (big) hipLaunchKernelGGL((sbp_xact_kernel<true, 32>), dim3(G), dim3(kSbpThreads), 0, st, q, par ^ 1);
        else hipLaunchKernelGGL((sbp_xact_kernel<true, 16>), dim3(G), dim3(kSbpThreads), 0, st, q, par ^ 1);
        hipLaunchKernelGGL((sbp_tail_kernel<true>), dim3(nT), dim3(64 * kSbpTailWaves), 0, st, q, par ^ 1);
    };
    auto enqueue = [&](long long g) {
        const int par = (int)(g & 1);
        const int t = (int)(g % 10);
        const bool gram = gram_on && g - t >= gram_from;            // a stretch runs one way from its regular iteration on
        if (t == 0) {                                                // regular iteration (the counter IS the enqueue index until `done`)
            if (screen) hipLaunchKernelGGL(sbp_xreg_screen_kernel, dim3(G), dim3(kSbpThreads), ldsv, st, q, par);
            else if (nt) hipLaunchKernelGGL((sbp_xreg_kernel<true>), dim3(G), dim3(kSbpThreads), ldsv, st, q, par);
            else hipLaunchKernelGGL((sbp_xreg_kernel<false>), dim3(G), dim3(kSbpThreads), ldsv, st, q, par);
            hipLaunchKernelGGL(sbp_list_kernel, dim3(G), dim3(kSbpThreads), 0, st, q, par ^ 1);
            if (gram) {
                // U, G and the start of the stretch in Gram space.  After a Gram-space stretch everything carries over (MODE 2:
                // no n-vector is touched at this iteration); otherwise the direct launches make the n-vectors first.
                cur_carried = carry_on && last_gram == g - 10;
                if ((size_t)(g / 10) >= carried.size()) carried.resize((size_t)(g / 10) + 1, 0);
                carried[(size_t)(g / 10)] = cur_carried ? 1 : 0;
                last_gram = g;
                if (!cur_carried) launch_xact_tail(par);
                hipLaunchKernelGGL(sbp_gs_merge_kernel, dim3(1), dim3(kSbpThreads), 0, st, q, par ^ 1, (int)g, cur_carried ? 0 : 1);
                hipLaunchKernelGGL((sbp_gs_dots_kernel<0>), dim3(32, 8), dim3(kSbpThreads), ldsv, st, q, par);
                if (cur_carried) {
                    hipLaunchKernelGGL((sbp_gs_kernel<2>), dim3(gcap / kGsRows), dim3(kSbpThreads), 0, st, q, par ^ 1, par, 0, 1);
                } else {
                    hipLaunchKernelGGL((sbp_gs_dots_kernel<1>), dim3(gcap / 4 + 1), dim3(kSbpThreads), ldsv, st, q, par ^ 1);
                    hipLaunchKernelGGL((sbp_gs_kernel<1>), dim3(gcap / kGsRows), dim3(kSbpThreads), 0, st, q, par ^ 1, par ^ 1, 0, 0);
                }
                return;
            }
            if (big) hipLaunchKernelGGL((sbp_xact_kernel<true, 32>), dim3(G), dim3(kSbpThreads), 0, st, q, par ^ 1);
            else hipLaunchKernelGGL((sbp_xact_kernel<true, 16>), dim3(G), dim3(kSbpThreads), 0, st, q, par ^ 1);
        } else if (gram) {
            // (a carried stretch's sum of iterates starts with the regular iteration's x: one more term)
            hipLaunchKernelGGL((sbp_gs_kernel<0>), dim3(gcap / kGsRows), dim3(kSbpThreads), 0, st, q, par, par, (!cur_carried && t == 1) ? 1 : 0, cur_carried ? t + 1 : t);
            if (t == 9) {                                            // the n-vectors for the regular iteration that follows
                hipLaunchKernelGGL((gather_batch_kernel<double>), dim3(gp.tiles, gp.ngroups, 2 * NL), dim3(kGatherThreads), 0, st, g_args.get());
                hipLaunchKernelGGL(sbp_gs_tail_kernel, dim3(nT), dim3(64 * kSbpTailWaves), 0, st, q, par ^ 1, cur_carried ? 10 : 9);
            }
            return;
        } else {
            if (big) hipLaunchKernelGGL((sbp_xact_kernel<false, 32>), dim3(G), dim3(kSbpThreads), ldsv, st, q, par);
            else hipLaunchKernelGGL((sbp_xact_kernel<false, 16>), dim3(G), dim3(kSbpThreads), ldsv, st, q, par);
        }
        if (dist) {
            hipLaunchKernelGGL((sbp_tail_kernel<false>), dim3(nT), dim3(64 * kSbpTailWaves), 0, st, q, par ^ 1);
            allreduce_sum_f64(ex.get(), (size_t)npad + 2 * nT, st);
            hipLaunchKernelGGL(sbp_tail_b_kernel, dim3(nT), dim3(64), 0, st, q, par ^ 1);
        } else {
            hipLaunchKernelGGL((sbp_tail_kernel<true>), dim3(nT), dim3(64 * kSbpTailWaves), 0, st, q, par ^ 1);
        }
    };
    LoopTimes lt;
    int gstat[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int halts = 0;
    for (long long g_start = 0;;) {
        const LoopTimes l1 = run_until_done(st, d_done.get(), sbp_batch(), (long long)opts.maxit + 2, enqueue, dist ? nullptr : hflag.p, g_start);
        lt.wall_s += l1.wall_s; lt.events_ms += l1.events_ms; lt.launched += l1.launched;
        if (!gram_on) break;
        ADMM_HIP_CHECK(hipGetLastError());
        ADMM_HIP_CHECK(hipMemcpy(gstat, g_ust.get(), sizeof(gstat), hipMemcpyDeviceToHost));
        if (gstat[3] == 0) break;
        // More non-zeros than the Gram matrix has room for: the merge launch of (regular) iteration gstat[4] halted the stream
        // right after that iteration.  Lift the halt and go on from the next iteration with the direct launches; Gram space
        // is tried again 100, 200, 400, ... iterations later (supports usually shrink as the solve goes on).
        SbpCtl hc[2];
        ADMM_HIP_CHECK(hipMemcpy(hc, ctl.get(), sizeof(hc), hipMemcpyDeviceToHost));
        hc[0].done = hc[1].done = 0;
        ADMM_HIP_CHECK(hipMemcpy(ctl.get(), hc, sizeof(hc), hipMemcpyHostToDevice));
        ADMM_HIP_CHECK(hipMemset(d_done.get(), 0, sizeof(int)));
        ADMM_HIP_CHECK(hipMemset(g_ust.get() + 2, 0, 2 * sizeof(int)));
        *hflag.p = 0;
        if ((size_t)(gstat[4] / 10) < carried.size() && carried[(size_t)(gstat[4] / 10)]) launch_xact_tail(gstat[4] & 1);   // (not made at a carried start)
        last_gram = -100;
        g_start = (long long)gstat[4] + 1;
        gram_from = (long long)gstat[4] + 100LL * (1LL << std::min(halts, 20));
        ++halts;
    }
    ADMM_HIP_CHECK(hipGetLastError());                             // a refused launch (LDS request, grid) is an error, not a silent no-op
    SbpCtl hc[2];
    ADMM_HIP_CHECK(hipMemcpy(hc, ctl.get(), sizeof(hc), hipMemcpyDeviceToHost));
    const SbpCtl& fin = hc[0].done ? hc[0] : hc[1];
    ADMM_REQUIRE(fin.done, "admm_parbp: the loop ended without a decision");
    res.niter = fin.niter;
    res.beta.assign(pl, 0.0);
    read_back(res.beta.data(), x.get(), (size_t)pl * sizeof(double), st);
    if (res.trace_cap > 0) {
        const long long nrec = std::min<long long>(fin.total, res.trace_cap);
        res.trace.assign((size_t)nrec * ADMM_TRACE_FIELDS, 0.0);
        if (nrec > 0) read_back(res.trace.data(), trace.get(), res.trace.size() * sizeof(double), st);
    }
    S.total_iter = fin.niter > opts.maxit ? opts.maxit : fin.niter;
    S.t_loop = lt.wall_s;
    S.loop_ms_events = lt.events_ms;
    S.exchange_variant = dist ? 1 : 0;
    S.xupdate_samples = lsteps;                                     // Lanczos steps of the longest spectral-radius run
    S.xupdate_variant = (gstat[6] == 0 ? 0 : (halts > 0 ? 2 : 1)) + (screen ? 4 : 0);      // how the active-set iterations ran, + 4: regular iterations screened (include/admm_hip.h)
    if (q.scr_stat != nullptr) {
        unsigned long long hs[2] = {0, 0};
        ADMM_HIP_CHECK(hipMemcpy(hs, scr_stat.get(), sizeof(hs), hipMemcpyDeviceToHost));
        std::fprintf(stderr, "[parbp screen] %llu columns screened on regular iterations, %llu took the exact step (%.3f %%)\n", hs[0], hs[1], hs[0] ? 100.0 * (double)hs[1] / (double)hs[0] : 0.0);
    }
    S.xupdate_launches = gstat[6];                                  // stretches that ran in Gram space
    S.persist_iter = gstat[5];                                      // times U was rebuilt from the current lists
}

}  // namespace admm
