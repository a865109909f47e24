// Least-absolute-deviation and basis-pursuit solvers (fp64), device resident.
//
// Replaces ADMMLAD / ADMMBP driven by FADMMBase::solve:
//   /root/reference/src/FADMMBase.h:109-133 (update_rho), :185-265 (solve)
//   /root/reference/src/ADMMLAD.h:62-107 (next_x/next_z/next_residual), :152-169 (eps/resid), :172-225 (setup, get_x)
//   /root/reference/src/ADMMBP.h:48-93, :138-153, :157-197
//
// Per iteration: `head` (decision of the previous iteration evaluated redundantly by every
// workgroup from the norm partials -> acceleration/restart scalars, rho adaptation, convergence;
// then adj_z/adj_y and the vector the projection is applied to), 2-3 streaming mat-vecs
// (gemv_t on the matrix and on its stored transpose, so every product reads contiguous columns),
// `tail` (x, soft-threshold z, dual y, six squared norms).  The host enqueues iterations in
// batches and polls a sticky `done` word; no per-iteration synchronisation.
#include "prep.h"
#include "gemv_kernels.h"
#include "gather_kernels.h"
#include "solvers.h"
#include <cstring>
#include <cstdio>
#include "loop_driver.h"

namespace admm {

struct DenseCtl {
    double rho, eps_primal, eps_dual, adj_a, adj_c, tau;
    int restart, iter, done, first, total, niter, conv, pad;
};

struct DenseParams {
    int dim, prob, maxit, nwg_tail;          // prob: 0 = LAD, 1 = BP
    double eps_abs, eps_rel, sqrt_dim, extra_norm;
    const double* data_vec;                   // LAD: y (n);  BP: A'(AA')^-1 b (p)
    double *x, *z0, *z1, *y0, *y1, *adj_z, *adj_y, *vec;
    const double* gout; int gout_nseg; long long gout_stride;
    DenseCtl* ctl;                            // [2]
    double* P;                                // [nwg_tail][8]
    int* done;
    double* trace; long long trace_cap;       // optional decision records (admm_hip_lad_traced / admm_hip_bp_traced), or NULL
    double* state; long long state_cap;       // optional [state_cap][5][dim] iterates x, z, y, adj_z, adj_y of every iteration (admm_hip_lad_state / admm_hip_bp_state), or NULL
    // BP, one-pass form (round 5; bpn = 0: the two-pass form).  B = L^-1 A has B B' = I, so B x = B vec + B A'(AA')^-1 b - B B' B vec
    // = L^-1 b =: w0 for EVERY iterate; with y = adj_y + rho (x - z) the n-vector B y follows from B adj_y and B z, B adj_z / B adj_y
    // from the same accelerate / restart combinations as adj_z / adj_y, and w = B vec = B adj_z - B adj_y / rho needs no pass over B:
    // only B z (z = prox output, sparse: gather_kernels.h) and the one streaming product B'w remain.
    // The recurrence is dead-beat, not an integrator: if the held B adj_y is off by e, w is off by -e / rho, then B x is off by
    // -e / rho as well and the true B y_new = B adj_y + e + rho (w0 - e / rho - B z_new) equals the held one.  What is left per iteration
    // is (I - B B') B vec, i.e. the rounding of the factorisation, and the rounding of the elementwise updates.
    int bpn;
    const double* w0;
    double *Bz0, *Bz1, *By0, *By1, *Badjy, *w;
    const double* gpart; int gngroups; long long gpstride;
    // LAD, one-pass form (round 6; lp = 0: the two-pass form).  The projection needs X'vec with vec = d - adj_y / rho + adj_z, and adj_z /
    // adj_y are the accelerate / restart combinations of the two latest z / y: X'vec is the same combination of the p-vectors X'd
    // (fixed), X'z and X'y of the two latest iterates -- and X'z_new, X'y_new can be formed by the launch that produces z_new, y_new:
    // lad_rows_kernel streams the ROWS of X once, forms x_i = row_i . s, the prox and the dual for that row, and adds row_i z_i,
    // row_i y_i to its partials of X'z, X'y.  One pass over X per iteration instead of two (X'vec, then X s); the p-vectors are direct
    // products of the current iterates, not recurrences: nothing accumulates.
    int lp;
    const double* Xd;                         // X'd [lp]
    double *Xz0, *Xz1, *Xy0, *Xy1;            // X'z, X'y in the slots of z0 / z1, y0 / y1
    double* u;                                // X'vec of this iteration
    const double* cpart; int cnwg; long long cstride;      // the rows launch's partials: [cnwg][2][cstride] (X'z_new | X'y_new)
};

constexpr int kDenseThreads = 256;

__device__ __forceinline__ double soft1(double v, double pen) {
    return v > pen ? v - pen : (v < -pen ? v + pen : 0.0);
}

__global__ void __launch_bounds__(kDenseThreads)
dense_head_kernel(DenseParams q, int par) {
    __shared__ double sums[8];
    extern __shared__ __attribute__((aligned(16))) double pstage[];
    const DenseCtl in = load_ctl_vector(q.ctl + par);
    DenseCtl* outp = &q.ctl[par ^ 1];
    if (in.done) {
        if (blockIdx.x == 0 && threadIdx.x == 0) *outp = in;
        return;
    }
    const int np = q.nwg_tail * 8;
    for (int k = threadIdx.x; k < np; k += kDenseThreads) pstage[k] = q.P[k];
    __syncthreads();
    if (threadIdx.x < 6) {
        double s = 0.0;
        for (int w = 0; w < q.nwg_tail; ++w) s += pstage[w * 8 + threadIdx.x];
        sums[threadIdx.x] = s;
    }
    __syncthreads();
    const double r2 = sums[0], dz2 = sums[1], daz2 = sums[2], x2 = sums[3], z2 = sums[4], y2 = sums[5];
    DenseCtl out = in;
    out.first = 0;
    bool write_adj = true;
    double tr_rp = 0, tr_rd = 0, tr_c = 0, tr_code = ADMM_TRACE_COLD;
    if (!in.first) {
        const double rp = sqrt(r2), rd = in.rho * sqrt(dz2);
        tr_rp = rp; tr_rd = rd;
        if (rp < in.eps_primal && rd < in.eps_dual) {          // converged(): adj_z/adj_y/rho stay as they are
            out.done = 1; out.conv = 1; out.niter = in.iter + 1;
            write_adj = false;
            tr_code = ADMM_TRACE_CONVERGED;
        } else {
            const double old_c = in.adj_c;
            const double c = in.rho * rp * rp + in.rho * daz2;
            tr_c = c; tr_code = c < 0.999 * old_c ? ADMM_TRACE_ACCELERATE : ADMM_TRACE_RESTART;
            if (c < 0.999 * old_c) {
                const double old_a = in.adj_a;
                const double a = 0.5 + 0.5 * sqrt(1.0 + 4.0 * old_a * old_a);
                out.adj_a = a; out.adj_c = c; out.tau = (old_a - 1.0) / a; out.restart = 0;
            } else {
                out.adj_a = 1.0; out.adj_c = old_c / 0.999; out.tau = -1.0; out.restart = 1;
            }
            if (in.iter > 5) {                                 // update_rho(), FADMMBase.h:109-133,258-259
                double rho = in.rho;
                if (rp / in.eps_primal > 10 * rd / in.eps_dual) rho *= 2;
                else if (rd / in.eps_dual > 10 * rp / in.eps_primal) rho /= 2;
                if (rp < in.eps_primal) rho /= 1.2;
                if (rd < in.eps_dual) rho *= 1.2;
                out.rho = rho;
            }
            out.iter = in.iter + 1;
            if (in.iter + 1 >= q.maxit) { out.done = 1; out.conv = 0; out.niter = q.maxit + 1; }
        }
    } else {
        out.tau = 0.0; out.restart = 0;
    }
    out.eps_primal = fmax(fmax(sqrt(x2), sqrt(z2)), q.extra_norm) * q.eps_rel + q.sqrt_dim * q.eps_abs;
    out.eps_dual = sqrt(y2) * q.eps_rel + q.sqrt_dim * q.eps_abs;
    out.total = out.done ? in.total : in.total + 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *outp = out;
        if (out.done) *q.done = 1;
        if (q.trace != nullptr && in.total < q.trace_cap) {      // what FADMMBase.h:135-170 (print_row, commented out there) would print
            double* t = q.trace + (size_t)in.total * ADMM_TRACE_FIELDS;
            t[0] = 0.0; t[1] = in.iter; t[2] = in.eps_primal; t[3] = in.eps_dual; t[4] = tr_rp; t[5] = tr_rd;
            t[6] = tr_c; t[7] = in.adj_c; t[8] = tr_code; t[9] = in.rho; t[10] = out.rho; t[11] = 0.0;
        }
    }
    if (!write_adj) return;

    const int cur = in.total & 1;
    const double* zc_ = cur ? q.z1 : q.z0; const double* yc_ = cur ? q.y1 : q.y0;
    const double* zo_ = cur ? q.z0 : q.z1; const double* yo_ = cur ? q.y0 : q.y1;
    const double t = out.tau, t1 = 1.0 + out.tau, rho = out.rho;
    for (int i = blockIdx.x * kDenseThreads + threadIdx.x; i < q.dim; i += gridDim.x * kDenseThreads) {
        // no fused multiply-adds in the elementwise arithmetic: the reference is built without them (lasso_tall.hip, tall_update_elem)
#pragma clang fp contract(off)
        double adjz, adjy;
        if (out.restart) { adjz = zo_[i]; adjy = yo_[i]; }
        else { adjz = t1 * zc_[i] - t * zo_[i]; adjy = t1 * yc_[i] - t * yo_[i]; }
        q.adj_z[i] = adjz; q.adj_y[i] = adjy;
        // LAD: vec = y - adj_y / rho + adj_z (ADMMLAD.h:64-65);  BP: vec = -adj_y / rho + adj_z (ADMMBP.h:50-55)
        q.vec[i] = (q.prob == 0 ? q.data_vec[i] : 0.0) - adjy / rho + adjz;
    }
    if (q.bpn > 0) {
        // the same step for the n-vectors B z, B y, B adj_z, B adj_y (see DenseParams): B z_new = the gather launch's partial rows
        // summed in group order; B y_new from the adj_y and the rho the tail used (in.rho); then w = B vec for this iteration
        double* Bzc = cur ? q.Bz1 : q.Bz0; double* Byc = cur ? q.By1 : q.By0;
        const double* Bzo = cur ? q.Bz0 : q.Bz1; const double* Byo = cur ? q.By0 : q.By1;
        for (int i = blockIdx.x * kDenseThreads + threadIdx.x; i < q.bpn; i += gridDim.x * kDenseThreads) {
#pragma clang fp contract(off)
            double bz = 0.0;
            for (int c0 = 0; c0 < q.gngroups; c0 += 8) {
                double tv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) tv[u] = q.gpart[(size_t)min(c0 + u, q.gngroups - 1) * q.gpstride + i];
#pragma unroll
                for (int u = 0; u < 8; ++u) bz += c0 + u < q.gngroups ? tv[u] : 0.0;
            }
            const double by = in.first ? 0.0 : q.Badjy[i] + in.rho * (q.w0[i] - bz);
            const double bzo = Bzo[i], byo = Byo[i];
            Bzc[i] = bz; Byc[i] = by;
            double badjz, badjy;
            if (out.restart) { badjz = bzo; badjy = byo; }
            else { badjz = t1 * bz - t * bzo; badjy = t1 * by - t * byo; }
            q.Badjy[i] = badjy;
            q.w[i] = badjz - badjy / rho;
        }
    }
    if (q.lp > 0) {
        // X'z, X'y of the iterate the rows launch has just produced (its partials summed in workgroup order) into the `cur` slots, then
        // u = X'vec as the combination the adj vectors above are of z / y
        double* Xzc = cur ? q.Xz1 : q.Xz0; double* Xyc = cur ? q.Xy1 : q.Xy0;
        const double* Xzo = cur ? q.Xz0 : q.Xz1; const double* Xyo = cur ? q.Xy0 : q.Xy1;
        // eight lanes share an element: lane `sub` adds the partial rows sub, sub + 8, ... (eight requests of each vector in flight), the eight
        // sums are added in a fixed order -- one memory round trip per 64 partial rows instead of one per 8
        const int sub = threadIdx.x & 7;
        const int epb = kDenseThreads / 8;                                // elements per block and pass
        for (int i0 = blockIdx.x * epb; i0 < q.lp; i0 += gridDim.x * epb) {
#pragma clang fp contract(off)
            const int i = min(i0 + (int)(threadIdx.x >> 3), q.lp - 1);    // (clamped: whole groups of eight lanes take part in the exchange)
            double xz = 0.0, xy = 0.0;
            for (int w0 = sub; w0 < q.cnwg; w0 += 64) {
                double tz[8], ty[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const size_t w = (size_t)min(w0 + 8 * k, q.cnwg - 1);
                    tz[k] = q.cpart[(w * 2) * q.cstride + i]; ty[k] = q.cpart[(w * 2 + 1) * q.cstride + i];
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) { xz += w0 + 8 * k < q.cnwg ? tz[k] : 0.0; xy += w0 + 8 * k < q.cnwg ? ty[k] : 0.0; }
            }
#pragma unroll
            for (int m = 1; m < 8; m <<= 1) { xz += __shfl_xor(xz, m, 64); xy += __shfl_xor(xy, m, 64); }
            if (sub != 0 || i0 + (int)(threadIdx.x >> 3) >= q.lp) continue;
            if (in.first) { xz = 0.0; xy = 0.0; }
            const double xzo = Xzo[i], xyo = Xyo[i];
            Xzc[i] = xz; Xyc[i] = xy;
            double xadjz, xadjy;
            if (out.restart) { xadjz = xzo; xadjy = xyo; }
            else { xadjz = t1 * xz - t * xzo; xadjy = t1 * xy - t * xyo; }
            q.u[i] = q.Xd[i] - xadjy / rho + xadjz;
        }
    }
}

// LAD one-pass form: the rows of X (the stored transpose: row i is ld contiguous doubles) streamed ONCE.  A workgroup of eight waves
// owns a run of rows and takes them R at a time: every thread holds its 2 NPT columns of the R rows, the lanes' partial dots go through
// one halving butterfly per wave and the eight waves' sums through LDS (waves in order: a fixed order), every thread then knows
// x_i = row_i . s and forms z_i, y_i as the tail kernel does (same expressions, no contraction: the stepwise instrument replays them bit
// for bit), thread 0 stores them and adds up the six norms, and every thread adds row_i z_i, row_i y_i to its columns of X'z, X'y.
// The next R rows are requested before the current ones are reduced.
constexpr int kLadThreads = 512;
constexpr int kLadRows = 4;
template <int NPT>
__global__ void __launch_bounds__(kLadThreads)
lad_rows_kernel(DenseParams q, int par, const double* __restrict__ Xt, long long ld, const double* __restrict__ spart, int snseg, long long sstride,
                double* __restrict__ cpart, int rows_per_wg) {
    constexpr int R = NPT <= 4 ? kLadRows : 2, NWV = kLadThreads / 64;      // (beyond 4096 columns two rows at a time: the registers hold two groups of rows)
    __shared__ double wsum[2][NWV][R];
    const DenseCtl c = load_ctl_vector(q.ctl + (par ^ 1));           // written by this iteration's head
    if (c.done) return;
    const int cur = (c.total - 1) & 1;
    const double* zc_ = cur ? q.z1 : q.z0;
    double* zn_ = cur ? q.z0 : q.z1; double* yn_ = cur ? q.y0 : q.y1;
    const double rho = c.rho, pen = 1.0 / rho;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int row_lo = blockIdx.x * rows_per_wg, row_hi = min(q.dim, row_lo + rows_per_wg);
    double2 sv[NPT], az[NPT], ay[NPT];
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const long long col = ((long long)k * kLadThreads + tid) * 2;
        double2 a = make_double2(0.0, 0.0);
        if (col < ld) for (int g = 0; g < snseg; ++g) { const double2 v = *reinterpret_cast<const double2*>(spart + (size_t)g * sstride + col); a.x += v.x; a.y += v.y; }
        if (col >= q.lp) a.x = 0.0;
        if (col + 1 >= q.lp) a.y = 0.0;
        sv[k] = a; az[k] = make_double2(0.0, 0.0); ay[k] = make_double2(0.0, 0.0);
    }
    auto load_rows = [&](int i0, double2 (&rv)[R][NPT]) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double* row = Xt + (size_t)min(i0 + r, q.dim - 1) * ld;       // clamped: rows beyond the run are weighted with zero below
#pragma unroll
            for (int k = 0; k < NPT; ++k) {
                const long long col = ((long long)k * kLadThreads + tid) * 2;
                rv[r][k] = col < ld ? load16_nt<double2>(row + col) : make_double2(0.0, 0.0);
            }
        }
    };
    double nacc[6] = {0, 0, 0, 0, 0, 0};
    double2 ra[R][NPT], rb[R][NPT];
    int buf = 0;
    auto group = [&](int i0, const double2 (&rv)[R][NPT]) {
        // the four vectors' entries of the R rows (the same addresses in every lane), requested before the reduction
        double dv[R], ajy[R], ajz[R], zc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = min(i0 + r, q.dim - 1);
            dv[r] = q.data_vec[i]; ajy[r] = q.adj_y[i]; ajz[r] = q.adj_z[i]; zc[r] = zc_[i];
        }
        double v8[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v8[r] = 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            double d = 0.0;
#pragma unroll
            for (int k = 0; k < NPT; ++k) { d = fma(rv[r][k].x, sv[k].x, d); d = fma(rv[r][k].y, sv[k].y, d); }
            v8[r] = d;
        }
        const double tot = halving_sum8(v8, lane);                   // lane 8 r: the wave's sum of row r
        if ((lane & 7) == 0 && (lane >> 3) < R) wsum[buf][wid][lane >> 3] = tot;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = i0 + r;
            if (i >= row_hi) break;                                  // uniform
            double x = wsum[buf][0][r];
#pragma unroll
            for (int w = 1; w < NWV; ++w) x += wsum[buf][w][r];
            double zn, yn;
            {
#pragma clang fp contract(off)
                zn = soft1(x - dv[r] + ajy[r] / rho, pen);           // dense_tail_kernel, prob 0 (ADMMLAD.h:94-107)
                const double rr = x - dv[r] - zn;
                yn = ajy[r] + rho * rr;
                if (tid == 0) {
                    const double dz = zn - zc[r], daz = zn - ajz[r];
                    nacc[0] += rr * rr; nacc[1] += dz * dz; nacc[2] += daz * daz; nacc[3] += x * x; nacc[4] += zn * zn; nacc[5] += yn * yn;
                    q.x[i] = x; zn_[i] = zn; yn_[i] = yn;
                    if (q.state != nullptr && c.total < q.state_cap) {
                        double* s = q.state + (size_t)c.total * 5 * q.dim;
                        s[i] = x; s[(size_t)q.dim + i] = zn; s[2 * (size_t)q.dim + i] = yn; s[3 * (size_t)q.dim + i] = ajz[r]; s[4 * (size_t)q.dim + i] = ajy[r];
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < NPT; ++k) {
                az[k].x = fma(rv[r][k].x, zn, az[k].x); az[k].y = fma(rv[r][k].y, zn, az[k].y);
                ay[k].x = fma(rv[r][k].x, yn, ay[k].x); ay[k].y = fma(rv[r][k].y, yn, ay[k].y);
            }
        }
        buf ^= 1;                                                    // (the next group writes the other half of wsum: one barrier per group)
    };
    if (row_lo < row_hi) {
        load_rows(row_lo, ra);
        for (int i0 = row_lo; i0 < row_hi; i0 += 2 * R) {
            if (i0 + R < row_hi) load_rows(i0 + R, rb);
            group(i0, ra);
            if (i0 + R >= row_hi) break;
            if (i0 + 2 * R < row_hi) load_rows(i0 + 2 * R, ra);
            group(i0 + R, rb);
        }
    }
    double* cz = cpart + ((size_t)blockIdx.x * 2) * q.cstride;
    double* cy = cz + q.cstride;
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const long long col = ((long long)k * kLadThreads + tid) * 2;
        if (col < q.cstride) { *reinterpret_cast<double2*>(cz + col) = az[k]; *reinterpret_cast<double2*>(cy + col) = ay[k]; }
    }
    if (tid == 0) {
        double* Pout = q.P + (size_t)blockIdx.x * 8;
#pragma unroll
        for (int k = 0; k < 6; ++k) Pout[k] = nacc[k];
    }
}

// BP one-pass form: B z_new for the z the tail of this iteration just wrote (gather over its non-zeros)
__global__ void __launch_bounds__(kGatherThreads)
bp_gather_kernel(DenseParams q, int par, GatherArgs<double> a) {
    const DenseCtl c = load_ctl_vector(q.ctl + (par ^ 1));
    if (c.done) return;
    const int cur = (c.total - 1) & 1;
    a.v = cur ? q.z0 : q.z1;                     // the tail wrote z_new into the buffer that is not `cur`
    gather_body<double>(a, (int)blockIdx.x, (int)blockIdx.y);
}

__global__ void __launch_bounds__(kDenseThreads)
dense_tail_kernel(DenseParams q, int par) {
    __shared__ double scratch[6 * (kDenseThreads / 64)];
    const DenseCtl c = load_ctl_vector(q.ctl + (par ^ 1));           // written by this iteration's head
    if (c.done) return;
    const int cur = (c.total - 1) & 1;           // head already advanced `total`
    const double* zc_ = cur ? q.z1 : q.z0;
    double* zn_ = cur ? q.z0 : q.z1; double* yn_ = cur ? q.y0 : q.y1;
    const double rho = c.rho, pen = 1.0 / rho;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int i = blockIdx.x * kDenseThreads + threadIdx.x; i < q.dim; i += gridDim.x * kDenseThreads) {
#pragma clang fp contract(off)
        double g = 0.0;
        for (int s = 0; s < q.gout_nseg; ++s) g += q.gout[(size_t)s * q.gout_stride + i];
        const double adjy = q.adj_y[i], adjz = q.adj_z[i], zc = zc_[i];
        double x, zn, r;
        if (q.prob == 0) {                        // LAD: x = P_X(vec); z = soft(x - y + adj_y/rho, 1/rho); r = x - y - z
            x = g;
            const double d = q.data_vec[i];
            zn = soft1(x - d + adjy / rho, pen);
            r = x - d - zn;
        } else {                                  // BP: x = vec + A'(AA')^-1 b - B'(B vec); z = soft(x + adj_y/rho, 1/rho); r = x - z
            x = q.vec[i] + q.data_vec[i] - g;
            zn = soft1(x + adjy / rho, pen);
            r = x - zn;
        }
        const double yn = adjy + rho * r;
        const double dz = zn - zc, daz = zn - adjz;
        acc[0] += r * r; acc[1] += dz * dz; acc[2] += daz * daz; acc[3] += x * x; acc[4] += zn * zn; acc[5] += yn * yn;
        q.x[i] = x; zn_[i] = zn; yn_[i] = yn;
        if (q.state != nullptr && c.total < q.state_cap) {     // record c.total = the trace record that will judge this iteration
            double* s = q.state + (size_t)c.total * 5 * q.dim;
            s[i] = x; s[(size_t)q.dim + i] = zn; s[2 * (size_t)q.dim + i] = yn; s[3 * (size_t)q.dim + i] = adjz; s[4 * (size_t)q.dim + i] = adjy;
        }
    }
    block_sum<double, 6>(acc, scratch);
    if (threadIdx.x == 0) {
        double* Pout = q.P + (size_t)blockIdx.x * 8;
#pragma unroll
        for (int k = 0; k < 6; ++k) Pout[k] = acc[k];
    }
}

__global__ void dense_init_kernel(DenseParams q, double rho) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < q.dim) { q.x[i] = 0; q.z0[i] = 0; q.z1[i] = 0; q.y0[i] = 0; q.y1[i] = 0; q.adj_z[i] = 0; q.adj_y[i] = 0; q.vec[i] = 0; }
    if (i < q.nwg_tail * 8) q.P[i] = 0.0;
    if (i == 0) {
        DenseCtl c;
        c.rho = rho; c.eps_primal = 0; c.eps_dual = 0; c.adj_a = 1.0; c.adj_c = 9999.0; c.tau = 0.0;
        c.restart = 0; c.iter = 0; c.done = 0; c.first = 1; c.total = 0; c.niter = 0; c.conv = 0; c.pad = 0;
        q.ctl[0] = c; q.ctl[1] = c;
        *q.done = 0;
    }
}

// vec = y - adj_y / rho + adj_z for LAD's get_x (ADMMLAD.h:220-225)
__global__ void lad_final_vec_kernel(const double* y, const double* adj_y, const double* adj_z, double rho, int n, double* vec) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) vec[i] = y[i] - adj_y[i] / rho + adj_z[i];
}

__global__ void __launch_bounds__(1024)
norm2_kernel(const double* v, int n, double* out) {
    __shared__ double scratch[16];
    double s[1] = {0.0};
    for (int i = threadIdx.x; i < n; i += 1024) s[0] += v[i] * v[i];
    block_sum<double, 1>(s, scratch);
    if (threadIdx.x == 0) out[0] = sqrt(s[0]);
}

namespace {

struct DenseLoop {
    DevBuf<double> x, z0, z1, y0, y1, adj_z, adj_y, vec, P;
    DevBuf<DenseCtl> ctl;
    DevBuf<int> done;
    DenseParams q{};
    int nwg_head = 0;
    DevBuf<double> trace, state;

    void init(int dim, int prob, const admm_opts& o, const double* data_vec, double extra_norm, hipStream_t st, long long trace_cap = 0, long long state_cap = 0,
              int norm_rows = 0) {      // norm_rows > 0: that many rows of norm partials (the launch that forms them is not dense_tail_kernel)
        const long long ld = round_up(dim, 32);
        for (DevBuf<double>* b : {&x, &z0, &z1, &y0, &y1, &adj_z, &adj_y, &vec}) { b->alloc(ld); b->zero(st); }
        const int nwg_tail = norm_rows > 0 ? norm_rows : std::max(1, std::min(64, (dim + kDenseThreads - 1) / kDenseThreads));
        nwg_head = std::max(1, std::min(device_info().num_cu, (dim + kDenseThreads - 1) / kDenseThreads));
        P.alloc((size_t)nwg_tail * 8); ctl.alloc(2); done.alloc(1);
        q.dim = dim; q.prob = prob; q.maxit = o.maxit; q.nwg_tail = nwg_tail;
        q.eps_abs = o.eps_abs; q.eps_rel = o.eps_rel; q.sqrt_dim = std::sqrt((double)dim); q.extra_norm = extra_norm;
        q.data_vec = data_vec;
        q.x = x.get(); q.z0 = z0.get(); q.z1 = z1.get(); q.y0 = y0.get(); q.y1 = y1.get();
        q.adj_z = adj_z.get(); q.adj_y = adj_y.get(); q.vec = vec.get();
        q.ctl = ctl.get(); q.P = P.get(); q.done = done.get();
        if (trace_cap > 0) { trace.alloc((size_t)trace_cap * ADMM_TRACE_FIELDS); q.trace = trace.get(); q.trace_cap = trace_cap; }
        if (state_cap > 0) {     // iterate dump; record 0 (the cold start has no iterates) carries data_vec as this solver holds it, in the x slot
            state.alloc((size_t)state_cap * 5 * dim);
            ADMM_HIP_CHECK(hipMemsetAsync(state.get(), 0, (size_t)state_cap * 5 * dim * sizeof(double), st));
            ADMM_HIP_CHECK(hipMemcpyAsync(state.get(), data_vec, (size_t)dim * sizeof(double), hipMemcpyDeviceToDevice, st));
            q.state = state.get(); q.state_cap = state_cap;
        }
        const int init_n = std::max(dim, nwg_tail * 8);
        hipLaunchKernelGGL(dense_init_kernel, dim3((init_n + 255) / 256), dim3(256), 0, st, q, o.rho);
    }
    void head(long long g, hipStream_t st) {
        hipLaunchKernelGGL(dense_head_kernel, dim3(nwg_head), dim3(kDenseThreads), (size_t)q.nwg_tail * 8 * sizeof(double), st, q, (int)(g & 1));
    }
    void tail(long long g, hipStream_t st) {
        hipLaunchKernelGGL(dense_tail_kernel, dim3(q.nwg_tail), dim3(kDenseThreads), 0, st, q, (int)(g & 1));
    }
    DenseCtl final_ctl(hipStream_t st) {
        DenseCtl h[2];
        ADMM_HIP_CHECK(hipMemcpyAsync(h, ctl.get(), sizeof(h), hipMemcpyDeviceToHost, st));
        ADMM_HIP_CHECK(hipStreamSynchronize(st));
        return h[0].done ? h[0] : h[1];
    }
};

int env_batch(int dflt) {
    const char* v = option("BATCH_ITERS");
    int b = v ? std::atoi(v) : dflt;
    if (b <= 0) b = dflt;
    return (b + 1) / 2 * 2;
}

}  // namespace

__global__ void __launch_bounds__(256) copy_cols_f64_kernel(const double* __restrict__ in, long long ldi, int rows, double* __restrict__ out, long long ldo) {
    const int c = blockIdx.y;
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < rows) out[(size_t)c * ldo + r] = in[(size_t)c * ldi + r];
}

// ---------------------------------------------------------------------------------------------- LAD
// copies the decision records of a finished loop into the result
static void dense_collect_trace(DenseLoop& L, const DenseCtl& fc, DenseResult& res, hipStream_t st) {
    if (res.trace_cap <= 0) return;
    const long long nrec = std::min<long long>(fc.total + (fc.done ? 1 : 0), res.trace_cap);      // the finishing decision does not advance `total`
    res.trace.resize((size_t)nrec * ADMM_TRACE_FIELDS);
    if (nrec > 0) read_back(res.trace.data(), L.trace.get(), res.trace.size() * sizeof(double), st);
    if (res.state_cap > 0) {                                   // one record per decision, same numbering as the trace
        const long long ns = std::min<long long>(fc.total + (fc.done ? 1 : 0), res.state_cap);
        res.state_dim = L.q.dim;
        res.state.resize((size_t)ns * 5 * L.q.dim);
        if (ns > 0) read_back(res.state.data(), L.state.get(), res.state.size() * sizeof(double), st);
        if (ns > 0 && option("DEBUG_REREAD")) {
            // diagnosis of profiles/r04_transient_stale_lines.md: the dump above came through read_back()'s pinned bounce buffer.  Read the
            // same device memory again through the pinned path and through the runtime's pageable hipMemcpy, and report which differs.
            std::vector<double> pinned2(res.state.size()), pageable(res.state.size());
            read_back(pinned2.data(), L.state.get(), pinned2.size() * sizeof(double), st);
            ADMM_HIP_CHECK(hipMemcpy(pageable.data(), L.state.get(), pageable.size() * sizeof(double), hipMemcpyDeviceToHost));
            size_t d12 = 0, d13 = 0, first12 = 0, first13 = 0;
            for (size_t i = 0; i < pinned2.size(); ++i) {
                if (std::memcmp(&pinned2[i], &res.state[i], 8) != 0) { if (!d12) first12 = i; ++d12; }
                if (std::memcmp(&pageable[i], &res.state[i], 8) != 0) { if (!d13) first13 = i; ++d13; }
            }
            if (d12 || d13)
                std::fprintf(stderr, "[admm_hip reread] dump of %zu doubles (dim %d): second pinned read differs in %zu entries (first at %zu), pageable hipMemcpy differs in %zu (first at %zu)\n",
                             pinned2.size(), L.q.dim, d12, first12, d13, first13);
        }
    }
}

void solve_lad(const DeviceData<double>& d, const admm_opts& opts, DenseResult& res, hipStream_t st) {
    const int n = d.n, p = d.p;
    admm_stats& S = res.stats;
    const long long ldp = round_up(p, 128);                // whole 128-blocks for the matrix-core inverse

    // X'X, its inverse (LLT of X'X in the reference, ADMMLAD.h:186-189), X' stored for the X*s product
    double t0 = now_s();
    DevBuf<double> M((size_t)ldp * ldp); M.zero(st);
    gram_full<double>(d.X.get(), d.ldx, n, p, true, M.get(), ldp, st);
    ADMM_HIP_CHECK(hipStreamSynchronize(st));
    S.t_gram = now_s() - t0;
    t0 = now_s();
    // n <= 2000: the hat-matrix branch below also needs the Cholesky factor of the same Gram matrix: keep a copy
    bool hat = n <= 2000;
    if (const char* e = option("LAD_HAT")) hat = hat && std::string(e) != "0";
    DevBuf<double> G2;
    if (hat) {
        G2.alloc((size_t)ldp * ldp);
        ADMM_HIP_CHECK(hipMemcpyAsync(G2.get(), M.get(), (size_t)ldp * ldp * sizeof(double), hipMemcpyDeviceToDevice, st));
    }
    spd_inverse_f64(M.get(), ldp, p, st);
    const long long ldxt = round_up(p, 32);
    DevBuf<double> Xt((size_t)ldxt * n); Xt.zero(st);
    transpose<double>(d.X.get(), d.ldx, n, p, Xt.get(), ldxt, st);
    ADMM_HIP_CHECK(hipStreamSynchronize(st));
    S.t_factor = now_s() - t0;

    DevBuf<double> ynorm_d(1);
    hipLaunchKernelGGL(norm2_kernel, dim3(1), dim3(1024), 0, st, d.Y.get(), n, ynorm_d.get());
    double ynorm = 0;
    ADMM_HIP_CHECK(hipMemcpyAsync(&ynorm, ynorm_d.get(), sizeof(double), hipMemcpyDeviceToHost, st));
    ADMM_HIP_CHECK(hipStreamSynchronize(st));

    // general branch, one-pass form (DenseParams::lp): X streamed once per iteration by rows.  LAD_ONEPASS=0: the reference's two products.
    constexpr int kLadMaxCols = 2 * kLadThreads * 6;             // 6144 columns: six double2 per thread and row (more would spill)
    bool onepass = !hat && p <= kLadMaxCols;
    if (const char* e = option("LAD_ONEPASS")) onepass = onepass && std::string(e) != "0";
    const int lad_nwg = std::max(1, std::min(device_info().num_cu, (n + 2 * kLadRows - 1) / (2 * kLadRows)));
    const int lad_rows = (int)round_up((n + lad_nwg - 1) / lad_nwg, kLadRows);

    DenseLoop L;
    L.init(n, 0, opts, d.Y.get(), ynorm, st, res.trace_cap, res.state_cap, onepass ? lad_nwg : 0);
    GemvT<double> g1, g2, g3, gH;                // t = X' vec ; s = (X'X)^-1 t ; xs = X s ;  or xs = H vec
    g1.init(d.X.get(), d.ldx, n, p);
    g2.init(M.get(), ldp, p, p);
    DevBuf<double> tvec(ldp), svec(ldp);
    tvec.zero(st); svec.zero(st);

    // n <= 2000: the reference caches the hat matrix H = X (X'X)^-1 X' = T T', T = X L^-T, and projects with one
    // symmetric product (ADMMLAD.h:67-73,191-203).  Same here: T = X U with U = L^-T from the blocked factorisation,
    // H = T T' on the fp64 matrix cores, then ONE mat-vec per iteration.  ADMM_HIP_LAD_HAT=0 keeps the general form.
    DevBuf<double> H;
    long long ldh = 0;
    if (hat) {
        t0 = now_s();
        const long long ldn = round_up(n, 128);
        const int pk = (int)round_up(p, 8);
        DevBuf<double> U = cholesky_linvt_mfma_f64(G2.get(), ldp, p, st);          // U = L^-T (p x p, upper)
        DevBuf<double> W((size_t)ldp * ldp), Xp((size_t)ldn * pk), T((size_t)ldn * ldp);
        Xp.zero(st); T.zero(st);
        transpose<double>(U.get(), ldp, (int)ldp, (int)ldp, W.get(), ldp, st);      // W = L^-1: W[j, k] = U[k, j]
        hipLaunchKernelGGL(copy_cols_f64_kernel, dim3((n + 255) / 256, p), dim3(256), 0, st, d.X.get(), d.ldx, n, Xp.get(), ldn);
        gemm_nt_f64(Xp.get(), ldn, W.get(), ldp, T.get(), ldn, n, p, pk, st, true); // T[i, j] = sum_k X[i, k] U[k, j]   (W = U' = L^-1: lower triangular)
        ldh = ldn;
        H.alloc((size_t)ldh * ldh); H.zero(st);
        gram_full<double>(T.get(), ldn, n, p, false, H.get(), ldh, st);             // H = T T' (tcross_prod_lower)
        ADMM_HIP_CHECK(hipStreamSynchronize(st));
        S.t_factor += now_s() - t0;
        gH.init(H.get(), ldh, n, n);
        gH.set_nt(gemv_stream_nt(gH.bytes()));
        L.q.gout = gH.part.get(); L.q.gout_nseg = gH.pl.nseg; L.q.gout_stride = gH.stride;
    } else if (!onepass) {
        g3.init(Xt.get(), ldxt, p, n);
        const bool nt = gemv_stream_nt(g1.bytes() + g2.bytes() + g3.bytes());        // per iteration: X', the inverse, X
        g1.set_nt(nt); g2.set_nt(nt); g3.set_nt(nt);
        L.q.gout = g3.part.get(); L.q.gout_nseg = g3.pl.nseg; L.q.gout_stride = g3.stride;
    }
    DevBuf<double> Xd, Xz0, Xz1, Xy0, Xy1, uvec, cpart;
    if (onepass) {
        g2.set_nt(gemv_stream_nt(g1.bytes() + g2.bytes()));
        for (DevBuf<double>* b : {&Xd, &Xz0, &Xz1, &Xy0, &Xy1, &uvec}) { b->alloc(ldp); b->zero(st); }
        cpart.alloc((size_t)lad_nwg * 2 * ldp); cpart.zero(st);
        g1.run(d.Y.get(), Xd.get(), nullptr, st);                // X'd, once
        L.q.lp = p; L.q.Xd = Xd.get(); L.q.Xz0 = Xz0.get(); L.q.Xz1 = Xz1.get(); L.q.Xy0 = Xy0.get(); L.q.Xy1 = Xy1.get(); L.q.u = uvec.get();
        L.q.cpart = cpart.get(); L.q.cnwg = lad_nwg; L.q.cstride = ldp;
    }
    auto launch_rows = [&](long long g) {
        const int par = (int)(g & 1);
        const int npt = (int)((ldxt / 2 + kLadThreads - 1) / kLadThreads);
#define ADMM_LAD_ROWS(N) hipLaunchKernelGGL((lad_rows_kernel<N>), dim3(lad_nwg), dim3(kLadThreads), 0, st, L.q, par, Xt.get(), ldxt, g2.part.get(), g2.pl.nseg, g2.stride, cpart.get(), lad_rows)
        switch (npt) {
            case 1: ADMM_LAD_ROWS(1); break;
            case 2: ADMM_LAD_ROWS(2); break;
            case 3: ADMM_LAD_ROWS(3); break;
            case 4: ADMM_LAD_ROWS(4); break;
            case 5: ADMM_LAD_ROWS(5); break;
            default: ADMM_LAD_ROWS(6); break;
        }
#undef ADMM_LAD_ROWS
    };

    const int* skip = L.done.get();
    LoopTimes lt = run_until_done(st, skip, env_batch(8), (long long)opts.maxit + 2, [&](long long g) {
        L.head(g, st);
        if (hat) {
            gH.run_partials(L.vec.get(), skip, st);          // dsymv(H, vec)
        } else if (onepass) {
            g2.run_partials(uvec.get(), skip, st);           // s = (X'X)^-1 u, u = X'vec formed by the head from X'd, X'z, X'y
            launch_rows(g);                                   // x = X s, z, y, norms, X'z_new, X'y_new: one pass over the rows of X
            return;
        } else {
            g1.run_partials(L.vec.get(), skip, st);          // chained: the next product sums these partial rows while staging
            g2.run_partials_from(g1, skip, st);
            g3.run_partials_from(g2, skip, st);
        }
        L.tail(g, st);
    });
    S.t_loop = lt.wall_s; S.loop_ms_events = lt.events_ms; S.xupdate_launches = lt.launched;

    const DenseCtl fc = L.final_ctl(st);
    res.niter = fc.niter;
    S.total_iter = fc.niter; S.rho = fc.rho;
    S.xupdate_variant = onepass ? 1 : 0;
    dense_collect_trace(L, fc, res, st);
    // get_x(): beta = (X'X)^-1 X' (y - adj_y/rho + adj_z) with the final adj and rho (ADMMLAD.h:220-225)
    hipLaunchKernelGGL(lad_final_vec_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d.Y.get(), L.adj_y.get(), L.adj_z.get(), fc.rho, n, L.vec.get());
    g1.run(L.vec.get(), tvec.get(), nullptr, st);
    g2.run(tvec.get(), svec.get(), nullptr, st);
    std::vector<double> coef(p), out(p);
    ADMM_HIP_CHECK(hipMemcpyAsync(coef.data(), svec.get(), (size_t)p * sizeof(double), hipMemcpyDeviceToHost, st));
    ADMM_HIP_CHECK(hipStreamSynchronize(st));
    double b0 = 0;
    recover_coef<double>(d, coef.data(), &b0, out.data());      // LAD.cpp:41
    res.beta.assign(p + 1, 0.0);
    res.beta[0] = b0;
    for (int j = 0; j < p; ++j) res.beta[j + 1] = out[j];
}

// ---------------------------------------------------------------------------------------------- BP
void solve_bp(const DeviceData<double>& d, const admm_opts& opts, DenseResult& res, hipStream_t st) {
    const int n = d.n, p = d.p;
    admm_stats& S = res.stats;
    const long long ldn = round_up(n, 128);                // whole 128-blocks for the matrix-core factorisation

    // AA' = LL' (ADMMBP.h:167-169)
    double t0 = now_s();
    DevBuf<double> G((size_t)ldn * ldn); G.zero(st);
    gram_full<double>(d.X.get(), d.ldx, n, p, false, G.get(), ldn, st);
    ADMM_HIP_CHECK(hipStreamSynchronize(st));
    S.t_gram = now_s() - t0;
    t0 = now_s();
    // B = L^-1 A (ADMMBP.h:173-182) and its transpose; w0 = L^-1 b; cache_AAAb = B' w0 = A'(AA')^-1 b (:170)
    DevBuf<double> B((size_t)d.ldx * p);
    DevBuf<double> w0(d.ldx); w0.zero(st);
    const long long ldbt = round_up(p, 32);
    DevBuf<double> Bt((size_t)ldbt * n); Bt.zero(st);
    const char* efac = option("FACTOR");
    if (efac && std::string(efac) == "rocsolver") {
        cholesky_lower<double>(G.get(), ldn, n, st);
        ADMM_HIP_CHECK(hipMemcpyAsync(B.get(), d.X.get(), (size_t)d.ldx * p * sizeof(double), hipMemcpyDeviceToDevice, st));
        trsm_left_lower<double>(G.get(), ldn, n, B.get(), d.ldx, p, st);
        ADMM_HIP_CHECK(hipMemcpyAsync(w0.get(), d.Y.get(), (size_t)d.ldx * sizeof(double), hipMemcpyDeviceToDevice, st));
        trsm_left_lower<double>(G.get(), ldn, n, w0.get(), d.ldx, 1, st);
        transpose<double>(B.get(), d.ldx, n, p, Bt.get(), ldbt, st);
    } else {
        // hand-written matrix-core path: blocked Cholesky fused with U = L^-T; then B' = A' U is one NT product
        // (B'[j, i] = sum_k A'[j, k] W[i, k] with W = U' = L^-1), B its transpose, w0 = U' b
        DevBuf<double> U = cholesky_linvt_mfma_f64(G.get(), ldn, n, st);
        const int nk = (int)round_up(n, 8);
        const long long ldxt = round_up(p, 128);
        DevBuf<double> Xt((size_t)ldxt * nk), W((size_t)ldn * ldn);
        Xt.zero(st);
        transpose<double>(d.X.get(), d.ldx, n, p, Xt.get(), ldxt, st);           // A' (p x n), output index contiguous
        transpose<double>(U.get(), ldn, (int)ldn, (int)ldn, W.get(), ldn, st);    // W = L^-1, zero padded
        DevBuf<double> Btp((size_t)ldxt * n);                                     // B' with whole 128-row blocks
        gemm_nt_f64(Xt.get(), ldxt, W.get(), ldn, Btp.get(), ldxt, p, n, nk, st, true);      // (W = L^-1: lower triangular, the K loop of column block j ends at its last column)
        hipLaunchKernelGGL(copy_cols_f64_kernel, dim3((p + 255) / 256, n), dim3(256), 0, st, Btp.get(), ldxt, p, Bt.get(), ldbt);
        transpose<double>(Bt.get(), ldbt, p, n, B.get(), d.ldx, st);
        GemvT<double> gU;                                                         // w0 = U' b
        gU.init(U.get(), ldn, n, n);
        gU.run(d.Y.get(), w0.get(), nullptr, st);
        ADMM_HIP_CHECK(hipStreamSynchronize(st));                                 // temporaries are released here
    }
    const long long ldp = round_up(p, 32);
    DevBuf<double> AAAb(ldp); AAAb.zero(st);
    // One-pass form (default; ADMM_HIP_BP_ONEPASS=0 keeps the two streaming products of ADMMBP.h:65-66): only B' w streams per
    // iteration, B vec comes from n-sized recurrences + a gather over the non-zeros of z (DenseParams), and the second stored
    // layout of B (its transpose, another 8np bytes) is a setup temporary.
    bool onepass = true;
    if (const char* e = option("BP_ONEPASS")) onepass = std::string(e) != "0";
    GemvT<double> gB, gBt;                       // t = B' w (p outputs) ; w = B vec (n outputs, via the stored transpose)
    gB.init(B.get(), d.ldx, n, p);
    if (!onepass) gBt.init(Bt.get(), ldbt, p, n);
    else Bt.release();
    { const bool nt = gemv_stream_nt(gB.bytes() + (onepass ? 0 : gBt.bytes())); gB.set_nt(nt); if (!onepass) gBt.set_nt(nt); }
    gB.run(w0.get(), AAAb.get(), nullptr, st);
    ADMM_HIP_CHECK(hipStreamSynchronize(st));
    S.t_factor = now_s() - t0;

    DenseLoop L;
    L.init(p, 1, opts, AAAb.get(), 0.0, st, res.trace_cap, res.state_cap);
    L.q.gout = gB.part.get(); L.q.gout_nseg = gB.pl.nseg; L.q.gout_stride = gB.stride;

    DevBuf<double> nvec, gpart;                  // one-pass form: B z (2), B y (2), B adj_y, w ; partial rows of the gather
    GatherPlan gp;
    GatherArgs<double> ga{};
    if (onepass) {
        const long long ldw = round_up(n, 32);
        nvec.alloc((size_t)6 * ldw); nvec.zero(st);
        gp = plan_gather<double>(n, p);
        gpart.alloc((size_t)gp.ngroups * gp.pstride); gpart.zero(st);
        ga = gather_args<double>(gp, B.get(), d.ldx, n, p, nullptr, gpart.get(), nullptr);
        L.q.bpn = n; L.q.w0 = w0.get();
        L.q.Bz0 = nvec.get(); L.q.Bz1 = nvec.get() + ldw; L.q.By0 = nvec.get() + 2 * ldw; L.q.By1 = nvec.get() + 3 * ldw;
        L.q.Badjy = nvec.get() + 4 * ldw; L.q.w = nvec.get() + 5 * ldw;
        L.q.gpart = gpart.get(); L.q.gngroups = gp.ngroups; L.q.gpstride = gp.pstride;
    }

    const int* skip = L.done.get();
    LoopTimes lt = run_until_done(st, skip, env_batch(8), (long long)opts.maxit + 2, [&](long long g) {
        L.head(g, st);
        if (onepass) {
            gB.run_partials(L.q.w, skip, st);               // B' w, w = B vec from the recurrences   (mat_vec_tprod, ADMMBP.h:66)
            L.tail(g, st);
            hipLaunchKernelGGL(bp_gather_kernel, dim3(gp.tiles, gp.ngroups), dim3(kGatherThreads), 0, st, L.q, (int)(g & 1), ga);      // B z_new
            return;
        }
        gBt.run_partials(L.vec.get(), skip, st);        // workspace = B vec   (mat_vec_prod,  ADMMBP.h:65)
        gB.run_partials_from(gBt, skip, st);            // B' workspace        (mat_vec_tprod, ADMMBP.h:66)
        L.tail(g, st);
    });
    S.t_loop = lt.wall_s; S.loop_ms_events = lt.events_ms; S.xupdate_launches = lt.launched;

    const DenseCtl fc = L.final_ctl(st);
    res.niter = fc.niter;
    S.total_iter = fc.niter; S.rho = fc.rho;
    dense_collect_trace(L, fc, res, st);
    const double* zfin = (fc.total & 1) ? L.z1.get() : L.z0.get();      // get_z() (BP.cpp:40)
    res.beta.assign(p, 0.0);
    read_back(res.beta.data(), zfin, (size_t)p * sizeof(double), st);
}

}  // namespace admm
