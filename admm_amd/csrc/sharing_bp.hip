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
#include "prep.h"
#include "gemv_kernels.h"
#include "solvers.h"
#include "loop_driver.h"
#include "comm.h"
#include "device_utils.h"

#include <cmath>

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

struct SbpParams {
    int n, npad, N, NL, maxit, G, nT, min_share;
    double eps_abs, eps_rel, rho, sqrt_nN, sqrtN, dN;
    const double* A; long long lda;              // this rank's columns, n x pl column-major, rows padded with zeros to npad
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
    const double drdS = s[0], dr2 = s[1], r2 = s[2], y2 = s[3], abar_r = s[4], sax = s[5], qq = s[6];
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

// ------------------------------------------------------------------------------------------------ setup
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
// block, 0.71 s of the solver's 1.09 s.  Per step the host now receives two numbers (alpha_j = v_j'A v_j, ||w||^2).
__global__ void __launch_bounds__(256) sbp_lz_dots_kernel(const double* __restrict__ V, long long ldv, int n, const double* __restrict__ w, double* __restrict__ c) {
    __shared__ double red[4];                                       // c[k] = V[:, k]'w, one workgroup per column k (k = gridDim.x - 1: w'w when V == nullptr there)
    const double* col = V + (size_t)blockIdx.x * ldv;
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += col[i] * w[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) c[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void __launch_bounds__(256) sbp_lz_update_kernel(const double* __restrict__ V, long long ldv, int n, int m, const double* __restrict__ c, double* __restrict__ w) {
    const int i = blockIdx.x * 256 + threadIdx.x;                   // w[i] -= sum_k V[i, k] c[k], k ascending
    if (i >= n) return;
    double acc = 0.0;
    for (int k = 0; k < m; ++k) acc += V[(size_t)k * ldv + i] * c[k];
    w[i] -= acc;
}
__global__ void __launch_bounds__(256) sbp_lz_scale_kernel(const double* __restrict__ w, double inv, int n, double* __restrict__ vnext) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) vnext[i] = w[i] * inv;
}

static double sbp_sprad(const double* Ab, long long lda, int n, int pb, hipStream_t st, int* nsteps) {
    const long long ldg = round_up(n, 32);
    DevBuf<double> Gm((size_t)ldg * ldg);
    Gm.zero(st);
    gram_full<double>(Ab, lda, n, pb, false, Gm.get(), ldg, st);
    SymMatVec<double> op(Gm.get(), ldg, n, st);                     // (its plan and partial buffer; the products below stay on the device)
    const int mmax = std::min(n, 600);
    const long long ldv = ldg;
    DevBuf<double> V((size_t)ldv * (mmax + 1)), w(ldg), c(mmax + 2);
    V.zero(st); w.zero(st);
    std::vector<double> al, be, v0(n);
    double nrm = 0;
    for (int i = 0; i < n; ++i) { v0[i] = 1.0 + 0.5 * std::sin(0.7 * (i + 1)); nrm += v0[i] * v0[i]; }
    nrm = std::sqrt(nrm);
    for (int i = 0; i < n; ++i) v0[i] /= nrm;
    ADMM_HIP_CHECK(hipMemcpyAsync(V.get(), v0.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
    const dim3 rows((n + 255) / 256);
    double theta = 0;
    for (int j = 0; j < mmax; ++j) {
        const double* vj = V.get() + (size_t)j * ldv;
        launch_gemv_t<double, 1, 4>(op.pl, Gm.get(), ldg, n, n, vj, nullptr, op.part.get(), nullptr, op.stride, nullptr, st);
        hipLaunchKernelGGL((reduce_partials_kernel<double>), rows, dim3(256), 0, st, op.part.get(), op.stride, op.pl.nseg, n, w.get(), (const int*)nullptr);
        // full re-orthogonalisation, twice; the first pass's coefficient of v_j is alpha_j = v_j'A v_j
        hipLaunchKernelGGL(sbp_lz_dots_kernel, dim3(j + 1), dim3(256), 0, st, V.get(), ldv, n, w.get(), c.get());
        double hc[2] = {0, 0};
        ADMM_HIP_CHECK(hipMemcpyAsync(&hc[0], c.get() + j, sizeof(double), hipMemcpyDeviceToHost, st));
        hipLaunchKernelGGL(sbp_lz_update_kernel, rows, dim3(256), 0, st, V.get(), ldv, n, j + 1, c.get(), w.get());
        hipLaunchKernelGGL(sbp_lz_dots_kernel, dim3(j + 1), dim3(256), 0, st, V.get(), ldv, n, w.get(), c.get());
        hipLaunchKernelGGL(sbp_lz_update_kernel, rows, dim3(256), 0, st, V.get(), ldv, n, j + 1, c.get(), w.get());
        hipLaunchKernelGGL(sbp_lz_dots_kernel, dim3(1), dim3(256), 0, st, w.get(), ldv, n, w.get(), c.get() + mmax + 1);      // ||w||^2
        ADMM_HIP_CHECK(hipMemcpyAsync(&hc[1], c.get() + mmax + 1, sizeof(double), hipMemcpyDeviceToHost, st));
        ADMM_HIP_CHECK(hipStreamSynchronize(st));
        const double a = hc[0], b = std::sqrt(hc[1]);
        al.push_back(a);
        *nsteps = j + 1;
        if (j + 1 >= 2 && ((j + 1) % 4 == 0 || j + 1 == mmax || b <= 1e-300)) {
            double last = 0;
            tridiag_top(al, be, &theta, &last);
            if (std::fabs(b * last) <= 1e-14 * std::fabs(theta) || b <= 1e-300) return theta;
        } else if (j == 0) {
            theta = a;
            if (b <= 1e-300 || n == 1) return theta;
        }
        if (j + 1 == mmax) {
            // n steps span the whole space (exact up to rounding).  Fewer, without the 1e-14 residual bound met (clustered top
            // eigenvalues; that bound is close to the rounding floor of the device Gram): rho and gamma_i = 2 rho + sprad_i hang on the
            // value and an UNDER-estimate breaks the majorisation of the linearised x-update, so the safe side is returned -- an
            // eigenvalue lies within |b last| of theta, theta + |b last| bounds it from above -- and only a value that is
            // not even accurate to 1e-8 is an error.
            if (mmax < n) {
                double last = 0;
                tridiag_top(al, be, &theta, &last);
                const double r = std::fabs(b * last);
                if (r <= 1e-8 * std::fabs(theta)) return theta + r;
                throw Error(ADMM_ERR_EIGS, "admm_parbp: the spectral radius of a column block did not converge in 600 Lanczos steps");
            }
            break;
        }
        be.push_back(b);
        hipLaunchKernelGGL(sbp_lz_scale_kernel, rows, dim3(256), 0, st, w.get(), 1.0 / b, n, V.get() + (size_t)(j + 1) * ldv);
    }
    ADMM_HIP_CHECK(hipGetLastError());
    return theta;                                                   // n steps: exact up to rounding
}

static int env_int(const char* name, int dflt) {
    const char* e = std::getenv(name);
    const int v = e ? std::atoi(e) : 0;
    return v > 0 ? v : dflt;
}

static int sbp_batch() {
    const char* e = std::getenv("ADMM_HIP_BATCH_ITERS");
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
    for (int b = 0; b < NL; ++b) {
        int ns = 0;
        sprad[b_first + b] = sbp_sprad(d.X.get() + (size_t)c0[b] * d.ldx, d.ldx, n, c0[b + 1] - c0[b], st, &ns);
        lsteps = std::max(lsteps, ns);
    }
    if (dist) {
        DevBuf<double> t(N);
        ADMM_HIP_CHECK(hipMemcpyAsync(t.get(), sprad.data(), (size_t)N * sizeof(double), hipMemcpyHostToDevice, st));
        allreduce_sum_f64(t.get(), (size_t)N, st);
        ADMM_HIP_CHECK(hipMemcpyAsync(sprad.data(), t.get(), (size_t)N * sizeof(double), hipMemcpyDeviceToHost, st));
        ADMM_HIP_CHECK(hipStreamSynchronize(st));
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
    int Gb = std::max(1, env_int("ADMM_HIP_SBP_WGS", 2 * ncu) / NL);
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
    {
        std::vector<double> hz(n), hy(n);
        ADMM_HIP_CHECK(hipMemcpyAsync(hy.data(), d.Y.get(), (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
        ADMM_HIP_CHECK(hipStreamSynchronize(st));
        for (int i = 0; i < n; ++i) hz[i] = hy[i] / (double)N;
        h2d(zbar.get(), hz.data(), (size_t)n * sizeof(double));
        ADMM_HIP_CHECK(hipStreamSynchronize(st));
    }
    SbpCtl c0ctl{};
    c0ctl.rp = c0ctl.rd = 9999.0;
    h2d(ctl.get(), &c0ctl, sizeof(SbpCtl)); h2d(ctl.get() + 1, &c0ctl, sizeof(SbpCtl));
    if (res.trace_cap > 0) { trace.alloc((size_t)res.trace_cap * ADMM_TRACE_FIELDS); trace.zero(st); }
    PinnedFlag hflag;

    SbpParams q{};
    q.min_share = std::max(1, env_int("ADMM_HIP_SBP_SHARE", 4));
    q.n = n; q.npad = npad; q.N = N; q.NL = NL; q.maxit = opts.maxit; q.G = G; q.nT = nT;
    q.eps_abs = opts.eps_abs; q.eps_rel = opts.eps_rel; q.rho = rho;
    q.sqrt_nN = std::sqrt((double)n * (double)N); q.sqrtN = std::sqrt((double)N); q.dN = (double)N;
    q.A = A; q.lda = lda;
    q.Gb = Gb; q.blk = d_blk.get();
    q.x = x.get(); q.list = d_list.get(); q.cnt = d_cnt.get(); q.wcount = d_wcount.get();
    q.xl = xl.get(); q.P = P.get(); q.Axo = Axo.get(); q.S = ex.get(); q.Qa = ex.get() + npad;
    q.Sold = Sold.get(); q.y = y.get(); q.r = r.get(); q.v = v.get(); q.zbar = zbar.get(); q.Q = Q.get();
    q.ctl = ctl.get(); q.done = d_done.get(); q.hflag = dist ? nullptr : hflag.p;
    q.trace = res.trace_cap > 0 ? trace.get() : nullptr; q.trace_cap = res.trace_cap;

    const bool big = npad > 8192;                                  // the x-update's row slice per thread: 16 rows up to npad = 8192, 32 beyond
    {
        // v lives in dynamic LDS (npad doubles) NEXT TO the kernels' static arrays: what must fit the default per-workgroup limit is
        // their sum -- at npad = 8192 the dynamic part alone is exactly 64 KB and the launch was refused without a trace (the
        // advisor's finding: hipLaunchKernelGGL does not surface the error and the loop ended "without a decision").  Opt in per
        // kernel whenever static + dynamic exceeds the default, and fail loudly when even the opt-in limit is too small.
        const size_t lds = (size_t)npad * sizeof(double);
        const void* fns[] = {reinterpret_cast<const void*>(&sbp_xreg_kernel<true>), reinterpret_cast<const void*>(&sbp_xreg_kernel<false>),
                             big ? reinterpret_cast<const void*>(&sbp_xact_kernel<false, 32>) : reinterpret_cast<const void*>(&sbp_xact_kernel<false, 16>)};
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
    ADMM_HIP_CHECK(hipStreamSynchronize(st));
    LoopTimes lt = run_until_done(st, d_done.get(), sbp_batch(), (long long)opts.maxit + 2,
        [&](long long g) {
            const int par = (int)(g & 1);
            if (g % 10 == 0) {                                       // regular iteration (the counter IS the enqueue index until `done`)
                if (nt) hipLaunchKernelGGL((sbp_xreg_kernel<true>), dim3(G), dim3(kSbpThreads), (size_t)npad * sizeof(double), st, q, par);
                else hipLaunchKernelGGL((sbp_xreg_kernel<false>), dim3(G), dim3(kSbpThreads), (size_t)npad * sizeof(double), st, q, par);
                hipLaunchKernelGGL(sbp_list_kernel, dim3(G), dim3(kSbpThreads), 0, st, q, par ^ 1);
                if (big) hipLaunchKernelGGL((sbp_xact_kernel<true, 32>), dim3(G), dim3(kSbpThreads), 0, st, q, par ^ 1);
                else hipLaunchKernelGGL((sbp_xact_kernel<true, 16>), dim3(G), dim3(kSbpThreads), 0, st, q, par ^ 1);
            } else {
                if (big) hipLaunchKernelGGL((sbp_xact_kernel<false, 32>), dim3(G), dim3(kSbpThreads), (size_t)npad * sizeof(double), st, q, par);
                else hipLaunchKernelGGL((sbp_xact_kernel<false, 16>), dim3(G), dim3(kSbpThreads), (size_t)npad * sizeof(double), st, q, par);
            }
            if (dist) {
                hipLaunchKernelGGL((sbp_tail_kernel<false>), dim3(nT), dim3(64 * kSbpTailWaves), 0, st, q, par ^ 1);
                allreduce_sum_f64(ex.get(), (size_t)npad + 2 * nT, st);
                hipLaunchKernelGGL(sbp_tail_b_kernel, dim3(nT), dim3(64), 0, st, q, par ^ 1);
            } else {
                hipLaunchKernelGGL((sbp_tail_kernel<true>), dim3(nT), dim3(64 * kSbpTailWaves), 0, st, q, par ^ 1);
            }
        }, dist ? nullptr : hflag.p);
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
}

}  // namespace admm
