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
// Per iteration: `step` (the decision of the previous iteration, evaluated identically by every workgroup from the norm
// partials; then the x-update: a workgroup owns a range of columns of one block -- or of the block's non-zero list --, its
// 256 threads own the rows: a batch of column dots is reduced over the workgroup, the new x_j is known to every thread, and
// x_j A_j is added to the workgroup's own partial of A_i x_i in registers, summation order fixed), on regular iterations
// `list` (the blocks' non-zero lists, ascending, from per-workgroup counts), `tail_a` (block sums of the partials,
// S = sum_i A_i x_i, sum ||A_i x_i||^2, sum ||A_i dx_i||^2) and `tail_b` (r, y, v, norm partials).  With a communicator
// attached the column blocks are spread over the ranks and ONE sum all-reduce of n + 2 nT doubles sits between the two
// tails: S and the two block sums -- the dual residual is evaluated as  sum_i ||A_i dx_i - dr||^2 = sum_i ||A_i dx_i||^2
// - 2 dr'dS + N ||dr||^2  so that it needs nothing else.  The host enqueues iterations in batches and polls a sticky flag.
#include "prep.h"
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

struct SbpParams {
    int n, npad, N, NL, maxit, G, nT, pad;
    double eps_abs, eps_rel, rho, sqrt_nN, sqrtN, dN;
    const double* A; long long lda;              // this rank's columns, n x pl column-major, rows padded with zeros to npad
    const int* wg_block; const int* wg_sub;      // [G]
    const int* blk_g0; const int* blk_c0;        // [NL + 1]
    const double* blk_gamma; const double* blk_pen;   // [NL]
    double* x;                                   // [pl]
    int* list; int* cnt;                         // [pl] block-relative indices of the non-zeros, ascending; [NL]
    int* wcount; int* pnz;                       // [G]
    double* P;                                   // [G][npad]
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

template <int RPT> struct SbpBatch { static constexpr int value = RPT <= 4 ? 8 : (RPT <= 8 ? 4 : 2); };

template <int RPT>
__global__ void __launch_bounds__(kSbpThreads)
sbp_step_kernel(SbpParams q, int par) {
    constexpr int CB = SbpBatch<RPT>::value;
    __shared__ double red[8 * 4];
    const SbpCtl in = load_ctl_vector(q.ctl + par);
    SbpCtl* outp = &q.ctl[par ^ 1];
    if (in.done) {
        if (blockIdx.x == 0 && threadIdx.x == 0) *outp = in;
        return;
    }
    // ---- the iteration just finished: residuals against the thresholds it ran with; then the thresholds of this one
    double s[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int w = threadIdx.x; w < q.nT; w += kSbpThreads) {
#pragma unroll
        for (int k = 0; k < 5; ++k) s[k] += q.Q[w * 8 + k];
        s[5] += q.Qa[w * 2]; s[6] += q.Qa[w * 2 + 1];
    }
    block_sum<double, 7>(s, red);
    const double drdS = s[0], dr2 = s[1], r2 = s[2], y2 = s[3], abar_r = s[4], sax = s[5], qq = s[6];
    SbpCtl out = in;
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
    if (out.done) return;

    // ---- x-update of this workgroup's share of its block
    const int g = blockIdx.x;
    const int b = q.wg_block[g], sub = q.wg_sub[g];
    const int Gb = q.blk_g0[b + 1] - q.blk_g0[b];
    const int c0 = q.blk_c0[b], pb = q.blk_c0[b + 1] - c0;
    const double gamma = q.blk_gamma[b], pen = q.blk_pen[b];
    const bool regular = (in.iter % 10) == 0;
    const int total = regular ? pb : load_flag_vector(q.cnt + b);
    const int per = (total + Gb - 1) / Gb;
    const int lo = sub * per, hi = min(total, lo + per);
    double v[RPT], axp[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int row = threadIdx.x + kSbpThreads * k;
        v[k] = row < q.npad ? q.v[row] : 0.0;
        axp[k] = 0.0;
    }
    int nzc = 0;
    for (int e = lo; e < hi; e += CB) {
        int col[CB]; double xj[CB], d[CB];
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            col[c] = -1; xj[c] = 0.0; d[c] = 0.0;
            if (e + c < hi) {
                const int j = c0 + (regular ? e + c : q.list[c0 + e + c]);
                const double xv = q.x[j];
                if (regular || xv != 0.0) { col[c] = j; xj[c] = xv; }      // an entry the active set has already pruned stays zero (PADMMBP.h:43)
            }
        }
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            if (col[c] < 0) continue;                                       // uniform
            const double* a = q.A + (size_t)col[c] * q.lda;
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const int row = threadIdx.x + kSbpThreads * k;
                if (row < q.npad) d[c] = fma(a[row], v[k], d[c]);
            }
        }
        block_sum<double, CB>(d, red);
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            if (col[c] < 0) continue;
            double xn;
            {
#pragma clang fp contract(off)
                const double val = xj[c] - d[c] / gamma;
                xn = sbp_soft(val, pen);
            }
            if (threadIdx.x == 0) q.x[col[c]] = xn;
            if (xn != 0.0) {
                ++nzc;
                const double* a = q.A + (size_t)col[c] * q.lda;
#pragma unroll
                for (int k = 0; k < RPT; ++k) {
                    const int row = threadIdx.x + kSbpThreads * k;
                    if (row < q.npad) axp[k] = fma(xn, a[row], axp[k]);
                }
            }
        }
    }
    if (nzc > 0) {
        double* P = q.P + (size_t)g * q.npad;
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int row = threadIdx.x + kSbpThreads * k;
            if (row < q.npad) P[row] = axp[k];
        }
    }
    if (threadIdx.x == 0) {
        q.pnz[g] = nzc > 0 ? 1 : 0;
        if (regular) q.wcount[g] = nzc;
    }
}

// Regular iterations only: the non-zero lists of the blocks, ascending, from the counts the step left per workgroup.
__global__ void __launch_bounds__(kSbpThreads)
sbp_list_kernel(SbpParams q, int par) {
    __shared__ int sh[4];
    __shared__ int sbase;
    const SbpCtl c = load_ctl_vector(q.ctl + par);                  // written by this iteration's step
    if (c.done) return;
    const int g = blockIdx.x;
    const int b = q.wg_block[g], sub = q.wg_sub[g];
    const int g0 = q.blk_g0[b], Gb = q.blk_g0[b + 1] - g0;
    const int c0 = q.blk_c0[b], pb = q.blk_c0[b + 1] - c0;
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
        if (nz) q.list[c0 + base + woff + before] = e;
        base += sh[0] + sh[1] + sh[2] + sh[3];
    }
    if (sub == Gb - 1 && threadIdx.x == 0) q.cnt[b] = base;
}

// tail_a: one thread per row.  A_i x_i of every local block from the workgroup partials (ascending), what changed, the sums.
__global__ void __launch_bounds__(64)
sbp_tail_a_kernel(SbpParams q, int par) {
    const SbpCtl c = load_ctl_vector(q.ctl + par);
    if (c.done) return;
    const int row = blockIdx.x * 64 + threadIdx.x;                   // < npad (npad is a multiple of 64)
    double S = 0.0, sax = 0.0, qq = 0.0;
    for (int b = 0; b < q.NL; ++b) {
        double a = 0.0;
        const int g1 = q.blk_g0[b + 1];
        for (int g = q.blk_g0[b]; g < g1; ++g)
            if (q.pnz[g]) a += q.P[(size_t)g * q.npad + row];        // uniform
        double* ao = q.Axo + (size_t)b * q.npad + row;
        const double d = a - *ao;
        *ao = a;
        S += a;
        sax = fma(a, a, sax);
        qq = fma(d, d, qq);
    }
    q.S[row] = S;
    sax = wave_sum(sax); qq = wave_sum(qq);
    if (threadIdx.x == 0) { q.Qa[blockIdx.x * 2] = sax; q.Qa[blockIdx.x * 2 + 1] = qq; }
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
static double sbp_sprad(const double* Ab, long long lda, int n, int pb, hipStream_t st, int* nsteps) {
    const long long ldg = round_up(n, 32);
    DevBuf<double> Gm((size_t)ldg * ldg);
    Gm.zero(st);
    gram_full<double>(Ab, lda, n, pb, false, Gm.get(), ldg, st);
    SymMatVec<double> op(Gm.get(), ldg, n, st);
    const int mmax = std::min(n, 600);
    std::vector<std::vector<double>> V;
    std::vector<double> al, be;
    std::vector<double> vj(n), w(n);
    double nrm = 0;
    for (int i = 0; i < n; ++i) { vj[i] = 1.0 + 0.5 * std::sin(0.7 * (i + 1)); nrm += vj[i] * vj[i]; }
    nrm = std::sqrt(nrm);
    for (int i = 0; i < n; ++i) vj[i] /= nrm;
    double theta = 0;
    for (int j = 0; j < mmax; ++j) {
        V.push_back(vj);
        op(vj.data(), w.data());
        double a = 0;
        for (int i = 0; i < n; ++i) a += w[i] * vj[i];
        al.push_back(a);
        for (int pass = 0; pass < 2; ++pass)
            for (size_t k = 0; k < V.size(); ++k) {
                double dot = 0;
                const double* vk = V[k].data();
                for (int i = 0; i < n; ++i) dot += w[i] * vk[i];
                for (int i = 0; i < n; ++i) w[i] -= dot * vk[i];
            }
        double b2 = 0;
        for (int i = 0; i < n; ++i) b2 += w[i] * w[i];
        const double b = std::sqrt(b2);
        *nsteps = j + 1;
        if (j + 1 >= 2 && ((j + 1) % 4 == 0 || j + 1 == mmax || b <= 1e-300)) {
            double last = 0;
            tridiag_top(al, be, &theta, &last);
            if (std::fabs(b * last) <= 1e-14 * std::fabs(theta) || b <= 1e-300) return theta;
        } else if (j == 0) {
            theta = a;
            if (b <= 1e-300 || n == 1) return theta;
        }
        if (j + 1 == mmax) break;
        be.push_back(b);
        for (int i = 0; i < n; ++i) vj[i] = w[i] / b;
    }
    return theta;                                                   // n steps: exact up to rounding
}

static int sbp_batch() {
    const char* e = std::getenv("ADMM_HIP_BATCH_ITERS");
    const int v = e ? std::atoi(e) : 0;
    return v > 0 ? (v + 1) / 2 * 2 : 20;                            // even: the parity pattern of the control block
}

template <int RPT>
static void sbp_launch_step(const SbpParams& q, int par, hipStream_t st) {
    hipLaunchKernelGGL((sbp_step_kernel<RPT>), dim3(q.G), dim3(kSbpThreads), 0, st, q, par);
}

// d: this rank's columns (n x pl).  nblocks: N (global); blk_first / nloc: the global blocks this rank holds; p_total.
void solve_parbp(const DeviceData<double>& d, const admm_opts& opts, int nblocks, long long p_total, long long col_offset,
                 DenseResult& res, hipStream_t st) {
    const int n = d.n, pl = d.p;
    const int N = nblocks;
    const CommInfo ci = comm_info();
    const bool dist = p_total != (long long)pl;
    ADMM_REQUIRE(!dist || ci.active, "no communicator: call admm_hip_comm_init first");
    ADMM_REQUIRE(n <= 8192, "admm_parbp: at most 8192 rows (the x-update keeps a row slice per thread); use admm_bp");
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
    const int npad = (int)round_up(n, 64);
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
    const int Gwant = std::max(NL, std::min(2 * ncu, pl));
    std::vector<int> wg_block, wg_sub, blk_g0(NL + 1, 0);
    for (int b = 0; b < NL; ++b) {
        const int pb = c0[b + 1] - c0[b];
        int Gb = (int)std::max<long long>(1, (long long)Gwant * pb / pl);
        Gb = std::min(Gb, pb);
        blk_g0[b + 1] = blk_g0[b] + Gb;
        for (int k = 0; k < Gb; ++k) { wg_block.push_back(b); wg_sub.push_back(k); }
    }
    const int G = blk_g0[NL];
    const int nT = npad / 64;
    std::vector<double> bg(NL), bp(NL);
    for (int b = 0; b < NL; ++b) { bg[b] = 2.0 * rho + sprad[b_first + b]; bp[b] = 1.0 / (rho * bg[b]); }

    DevBuf<int> d_wg_block(G), d_wg_sub(G), d_blk_g0(NL + 1), d_blk_c0(NL + 1), d_list(pl), d_cnt(NL), d_wcount(G), d_pnz(G), d_done(1);
    DevBuf<double> d_bg(NL), d_bp(NL), x(pl), P((size_t)G * npad), Axo((size_t)NL * npad), ex((size_t)npad + 2 * nT), Sold(npad), y(npad), r(npad), v(npad),
        zbar(npad), Q((size_t)nT * 8), trace;
    DevBuf<SbpCtl> ctl(2);
    auto h2d = [&](void* dst, const void* src, size_t bytes) { ADMM_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st)); };
    h2d(d_wg_block.get(), wg_block.data(), (size_t)G * sizeof(int)); h2d(d_wg_sub.get(), wg_sub.data(), (size_t)G * sizeof(int));
    h2d(d_blk_g0.get(), blk_g0.data(), (size_t)(NL + 1) * sizeof(int)); h2d(d_blk_c0.get(), c0.data(), (size_t)(NL + 1) * sizeof(int));
    h2d(d_bg.get(), bg.data(), (size_t)NL * sizeof(double)); h2d(d_bp.get(), bp.data(), (size_t)NL * sizeof(double));
    x.zero(st); P.zero(st); Axo.zero(st); ex.zero(st); Sold.zero(st); y.zero(st); r.zero(st); v.zero(st); zbar.zero(st); Q.zero(st);
    d_list.zero(st); d_cnt.zero(st); d_wcount.zero(st); d_pnz.zero(st); d_done.zero(st);
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
    q.n = n; q.npad = npad; q.N = N; q.NL = NL; q.maxit = opts.maxit; q.G = G; q.nT = nT;
    q.eps_abs = opts.eps_abs; q.eps_rel = opts.eps_rel; q.rho = rho;
    q.sqrt_nN = std::sqrt((double)n * (double)N); q.sqrtN = std::sqrt((double)N); q.dN = (double)N;
    q.A = A; q.lda = lda;
    q.wg_block = d_wg_block.get(); q.wg_sub = d_wg_sub.get(); q.blk_g0 = d_blk_g0.get(); q.blk_c0 = d_blk_c0.get();
    q.blk_gamma = d_bg.get(); q.blk_pen = d_bp.get();
    q.x = x.get(); q.list = d_list.get(); q.cnt = d_cnt.get(); q.wcount = d_wcount.get(); q.pnz = d_pnz.get();
    q.P = P.get(); q.Axo = Axo.get(); q.S = ex.get(); q.Qa = ex.get() + npad;
    q.Sold = Sold.get(); q.y = y.get(); q.r = r.get(); q.v = v.get(); q.zbar = zbar.get(); q.Q = Q.get();
    q.ctl = ctl.get(); q.done = d_done.get(); q.hflag = dist ? nullptr : hflag.p;
    q.trace = res.trace_cap > 0 ? trace.get() : nullptr; q.trace_cap = res.trace_cap;

    const int rpt = (npad + kSbpThreads - 1) / kSbpThreads;
    auto step = [&](int par) {
        if (rpt <= 1) sbp_launch_step<1>(q, par, st);
        else if (rpt <= 2) sbp_launch_step<2>(q, par, st);
        else if (rpt <= 4) sbp_launch_step<4>(q, par, st);
        else if (rpt <= 8) sbp_launch_step<8>(q, par, st);
        else if (rpt <= 16) sbp_launch_step<16>(q, par, st);
        else sbp_launch_step<32>(q, par, st);
    };
    ADMM_HIP_CHECK(hipStreamSynchronize(st));
    LoopTimes lt = run_until_done(st, d_done.get(), sbp_batch(), (long long)opts.maxit + 2,
        [&](long long g) {
            const int par = (int)(g & 1);
            step(par);
            if (g % 10 == 0) hipLaunchKernelGGL(sbp_list_kernel, dim3(G), dim3(kSbpThreads), 0, st, q, par ^ 1);
            hipLaunchKernelGGL(sbp_tail_a_kernel, dim3(nT), dim3(64), 0, st, q, par ^ 1);
            if (dist) allreduce_sum_f64(ex.get(), (size_t)npad + 2 * nT, st);
            hipLaunchKernelGGL(sbp_tail_b_kernel, dim3(nT), dim3(64), 0, st, q, par ^ 1);
        }, dist ? nullptr : hflag.p);
    SbpCtl hc[2];
    ADMM_HIP_CHECK(hipMemcpy(hc, ctl.get(), sizeof(hc), hipMemcpyDeviceToHost));
    const SbpCtl& fin = hc[0].done ? hc[0] : hc[1];
    ADMM_REQUIRE(fin.done, "admm_parbp: the loop ended without a decision");
    res.niter = fin.niter;
    res.beta.assign(pl, 0.0);
    ADMM_HIP_CHECK(hipMemcpy(res.beta.data(), x.get(), (size_t)pl * sizeof(double), hipMemcpyDeviceToHost));
    if (res.trace_cap > 0) {
        const long long nrec = std::min<long long>(fin.total, res.trace_cap);
        res.trace.assign((size_t)nrec * ADMM_TRACE_FIELDS, 0.0);
        if (nrec > 0) ADMM_HIP_CHECK(hipMemcpy(res.trace.data(), trace.get(), res.trace.size() * sizeof(double), hipMemcpyDeviceToHost));
    }
    S.total_iter = fin.niter > opts.maxit ? opts.maxit : fin.niter;
    S.t_loop = lt.wall_s;
    S.loop_ms_events = lt.events_ms;
    S.exchange_variant = dist ? 1 : 0;
    S.xupdate_samples = lsteps;                                     // Lanczos steps of the longest spectral-radius run
}

}  // namespace admm
