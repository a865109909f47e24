// Wide (n <= p) Lasso / Elastic-net lambda path: linearised ADMM with active-set iterations.
//
// Replaces ADMMLassoWide / ADMMEnetWide driven by ADMMBase::solve:
//   /root/reference/src/ADMMBase.h:85-109 (update_rho), :158-216 (update_x/z/y, solve)
//   /root/reference/src/ADMMLassoWide.h:70-84 (soft_threshold), :86-118 (active_set_update),
//   :121-127 (is_regular_update), :129-155 (next_x), :156-170 (next_z, next_residual),
//   :174-186 (eps / resid), :189-251 (ctor, init, init_warm)
//   /root/reference/src/ADMMEnet.h:62-154, and the lambda loop of Lasso.cpp:97-124.
//
// Device design:
//  * "Regular" iterations (counter 0, 3, 15, 63, ... = 4^k - 1) stream all of X once and apply the
//    prox to every coordinate; all other iterations touch only the current support.
//  * No compaction and no index lists: column j belongs to wave (j mod NW).  A wave loads the x
//    values of its columns, ballots the non-zeros and processes exactly those, so a zero
//    coordinate stays zero until the next regular iteration (== SparseVector::prune) and a
//    clustered support is spread over many waves.  Deterministic, no atomics.
//  * Ax = sum_{j in supp} x_j X_j is a gather mat-vec with the same column->wave map; per-workgroup
//    partials are summed by the z/y kernel.
//  * Convergence test, rho adaptation (from iteration 5), the regular/active schedule and the
//    lambda schedule (init_warm resets the counter, keeps x, z, y, rho) run on the device, evaluated
//    identically by every workgroup of the x-update launch; the host enqueues batches and polls a
//    sticky done word.
//  * n <= 8192 (a column fits the registers of one wave: 4 / 8 / 16 / 24 / 32 float4 per lane): the x-update keeps the column it has just
//    dotted with t and adds x_j X_j to its own partial of Ax right away -- TWO launches per iteration
//    (x-update + gather, z/y + norms), X_j read once.  On active-set iterations only the first 128
//    workgroups take part, so the z/y kernel sums 128 partials (all of them on regular iterations).
//    Larger n: three launches (x-update, gather mat-vec over row tiles of 4096, z/y + norms).  (Measured at n = 5000 .. 8000:
//    the two-launch form takes 32-40 us per iteration, the three-launch form 44-70 us.)
#include "prep.h"
#include "gemv_kernels.h"
#include "solvers.h"
#include "loop_driver.h"
#include "comm.h"
#include "probe.h"
#include "peer_device.h"

namespace admm {

enum { W_ZERO = 0, W_REG = 1, W_ACT = 2 };

struct WideCtl {
    double rho, eps_primal, eps_dual;
    float lam; int type;
    int iter, counter, lam_idx, done, first, total, pad1, pad2;      // total: decisions taken so far (index of the trace record)
};

static_assert(sizeof(WideCtl) == 64, "WideCtl is loaded as four 16-byte words");

// The control block through the VECTOR memory path, issued together with the other prologue loads.  (A scalar s_load of
// it shares its wait counter with the kernel-argument loads: the compiler waits for all of them before it can form the
// first vector address, which puts the control block's miss in front of every other load -- one more dependent round
// trip per launch.)  The zero offset is produced by inline asm so that the address is not provably wave-uniform.
struct WideCtlRaw { uint4 w[4]; };
__device__ __forceinline__ WideCtlRaw wide_ctl_request(const WideCtl* c) {
    int vz;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
    const uint4* cp = reinterpret_cast<const uint4*>(c) + vz;
    WideCtlRaw r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r.w[k] = cp[k];
    return r;
}
__device__ __forceinline__ double uniform_f64(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ WideCtl wide_ctl_uniform(WideCtl c) {          // identical in every lane: tell the compiler
    c.rho = uniform_f64(c.rho); c.eps_primal = uniform_f64(c.eps_primal); c.eps_dual = uniform_f64(c.eps_dual);
    c.lam = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(c.lam)));
    c.type = __builtin_amdgcn_readfirstlane(c.type); c.iter = __builtin_amdgcn_readfirstlane(c.iter);
    c.counter = __builtin_amdgcn_readfirstlane(c.counter); c.lam_idx = __builtin_amdgcn_readfirstlane(c.lam_idx);
    c.done = __builtin_amdgcn_readfirstlane(c.done); c.first = __builtin_amdgcn_readfirstlane(c.first);
    c.total = __builtin_amdgcn_readfirstlane(c.total);
    return c;
}
__device__ __forceinline__ WideCtl wide_ctl_unpack(const WideCtlRaw& r) {
    WideCtl c;
    __builtin_memcpy(&c, &r, sizeof(c));
    return wide_ctl_uniform(c);
}

constexpr int kWideThreads = 256;
constexpr int kAxWG = 128;                // workgroups of the gather mat-vec (partials per output)
constexpr int kActWG = 256;               // fused mode: workgroups taking part in an active-set iteration
constexpr int kAxRT = 16;                 // float4 row accumulators per lane -> 4096 rows per row tile

struct WideParams {
    int n, p, maxit, nlam, enet, nwg_tail;
    int fused, nwg_x;                     // fused: the x-update launch also writes the Ax partials (n <= 8192)
    int x_nt;                             // regular steps stream X with non-temporal loads (X larger than the Infinity Cache keeps, gemv_plan.h)
    long long ldx;
    const float* X; const float* Y;
    float gamma, lambda0, alpha;          // sprad, lambda_0, enet alpha
    double eps_abs, eps_rel, sqrt_n, sqrt_p, sqrt_gamma;
    const float* lambdas;                 // device [nlam], internal lambdas as float (Scalar lambda)
    float* x;                             // p, dense storage of the sparse main_x
    float* Ax; float* z; float* y;        // n
    float* axpart;                        // [kAxWG][ldn]
    float* tbuf;                          // [ldn] t (or t / gamma) of the current iteration in global memory (large-n mode only)
    const float* ax_given;                // column-sharded mode: Ax already summed over this rank's partials AND over the ranks (else NULL)
    long long ldn;
    WideCtl* ctl;                         // [2]
    double* P;                            // [nwg_tail][8]: |r|^2, |z_new - z|^2, |Ax|^2, |z_new|^2, |y_new|^2
    float* beta; int* niter; int* done;
    int* done_host;                       // pinned host word, set together with *done (loop_driver.h: PinnedFlag)
    double* trace; long long trace_cap;   // optional decision records (admm_hip_lasso_plan_trace_*), or NULL
    float* state; long long state_cap;    // optional [state_cap][p + 3 n] iterates x | Ax | z | y of every iteration (admm_hip_lasso_plan_state_*), or NULL
    // safe screening of the regular steps (wide_x_kernel, "screen"): a 2-byte copy of X and a per-column error bound, or NULL
    const void* Xh; long long ldh;                // [p][ldh] the copy: fp16 roundings (scr_fmt 16; non-finite ones stored as 0) or signed bytes q with
                                                  // X ~ scale_j q (scr_fmt 8); ldh a multiple of the 8 / 16 elements of a 16-byte piece, rows [n, ldh) zero
    int scr_fmt;
    const float* scr_s; int scr_S;                // s_j in the x-update launch's own order: column (k, w) = k * NW + w at scr_s[w * scr_S + k]
    const float* scr_scale;                       // scr_fmt 8: scale_j, same order
    unsigned long long* scr_stat;                 // optional [2]: columns screened, columns that took the exact path (WIDE_SCREEN_STATS)
#ifdef ADMM_HIP_PROBE
    long long* probe;                     // dev build only: in-kernel timestamps [4096 iterations][4 observers][8]
#endif
};



__device__ __forceinline__ bool is_regular_update(unsigned int x) {      // 4^k - 1   ADMMLassoWide.h:121-127
    if (x == 0 || x == 3 || x == 15 || x == 63) return true;
    x++;
    if (x & (x - 1)) return false;
    return (x & 0x55555555u) != 0;
}

__device__ __forceinline__ float prox_f(float val, float thresh, float denom, bool enet) {
    // active_set_update thresholding in float (ADMMLassoWide.h:108-113, ADMMEnet.h:111-116)
    if (val > thresh) return enet ? (val - thresh) / denom : val - thresh;
    if (val < -thresh) return enet ? (val + thresh) / denom : val + thresh;
    return 0.f;
}

// The decision for the iteration that just finished and the kind of x-update that runs now: convergence, rho adaptation
// (ADMMBase.h:85-109), lambda schedule (init_warm), regular / active-set schedule (ADMMLassoWide.h:121-155).  Every wave
// that calls it reduces the norm partials itself in a fixed order (no LDS, no barrier) and gets the identical result.
struct WideDecision { WideCtl out; int lam_finished; int niter_val; double rp, rd; int code; };
__device__ __forceinline__ double wide_halving_dpp(const double (&v)[8], int lane);      // = halving_sum8_dpp (defined with the 2-D stretch below)
constexpr int kWideNormRows = 128;        // rows of P that every lane requests unconditionally (P is allocated and zeroed to at least this)
// The lane's share of the norm partials of the previous iteration (rows lane, lane + 64, ...).  Does not depend on the
// control block: callers request it in the same memory round trip as the control block itself.
struct WideNormRaw { double a[5], b[5]; };
__device__ __forceinline__ WideNormRaw wide_norms_request(const WideParams& q, int lane) {
    WideNormRaw r;
#pragma unroll
    for (int k = 0; k < 5; ++k) { r.a[k] = q.P[(size_t)lane * 8 + k]; r.b[k] = q.P[(size_t)(lane + 64) * 8 + k]; }
    return r;
}
__device__ __forceinline__ void wide_norms_finish(const WideParams& q, int lane, const WideNormRaw& r, double (&sums)[5]) {
#pragma unroll
    for (int k = 0; k < 5; ++k) sums[k] = r.a[k] + r.b[k];
    for (int row = lane + kWideNormRows; row < q.nwg_tail; row += 64) {
#pragma unroll
        for (int k = 0; k < 5; ++k) sums[k] += q.P[(size_t)row * 8 + k];
    }
}
template <bool DPP = false>
__device__ __forceinline__ WideDecision wide_decide(const WideParams& q, const WideCtl& in, const double (&sums)[5], int lane) {
    int lam_finished = -1, niter_val = 0;
    // Every wave reduces the norm partials itself (fixed order, no LDS, no barrier): the five sums through one halving
    // butterfly (bit-identical to five wave_sum calls), then ONE square-root sequence with lane 8 k working on sum k,
    // and the five results read back as wave-uniform scalars.
    const double v8[8] = {sums[0], sums[1], sums[2], sums[3], sums[4], 0.0, 0.0, 0.0};
    double tot8;
    if constexpr (DPP) tot8 = wide_halving_dpp(v8, lane);      // (wide_rows_persist_kernel: the cross-lane hardware of gfx950 instead of LDS permutes)
    else tot8 = halving_sum8(v8, lane);
    const double root = sqrt(tot8);
    const double sq_r2 = readlane_f64(root, 0), sq_dz2 = readlane_f64(root, 8), sq_ax2 = readlane_f64(root, 16);
    const double sq_z2 = readlane_f64(root, 24), sq_y2 = readlane_f64(root, 32);
    WideCtl out = in;
    out.first = 0;
    double tr_rp = 0, tr_rd = 0; int tr_code = ADMM_TRACE_COLD;
    if (!in.first) {
        const double rp = sq_r2;                                      // resid_primal = ||Ax + z||       ADMMBase.h:181
        const double rd = in.rho * q.sqrt_gamma * sq_dz2;             // rho sqrt(sprad) ||z_new - z||   ADMMLassoWide.h:183-186
        tr_rp = rp; tr_rd = rd; tr_code = (rp < in.eps_primal && rd < in.eps_dual) ? ADMM_TRACE_CONVERGED : ADMM_TRACE_CONTINUE;
        if (rp < in.eps_primal && rd < in.eps_dual) { lam_finished = in.lam_idx; niter_val = in.iter + 1; }
        else {
            if (in.iter > 3) {                                        // update_rho()  ADMMBase.h:85-109,209-210
                double rho = in.rho;
                if (rp / in.eps_primal > 10 * rd / in.eps_dual) rho *= 2;
                else if (rd / in.eps_dual > 10 * rp / in.eps_primal) rho /= 2;
                if (rp < in.eps_primal) rho /= 1.2;
                if (rd < in.eps_dual) rho *= 1.2;
                out.rho = rho;
            }
            out.iter = in.iter + 1;
            if (in.iter + 1 >= q.maxit) { lam_finished = in.lam_idx; niter_val = q.maxit + 1; }
        }
        if (lam_finished >= 0) {                                      // init_warm: counter = 0, x/z/y/rho kept (:241-251)
            out.lam_idx = in.lam_idx + 1; out.iter = 0; out.counter = 0;
            if (out.lam_idx >= q.nlam) out.done = 1;
            else out.lam = q.lambdas[out.lam_idx];
        }
    }
    // eps for this iteration from the current Ax, z, y (ADMMLassoWide.h:174-182)
    out.eps_primal = fmax(sq_ax2, sq_z2) * q.eps_rel + q.sqrt_n * q.eps_abs;
    out.eps_dual = q.sqrt_gamma * sq_y2 * q.eps_rel + q.sqrt_p * q.eps_abs;
    // which x-update runs now (ADMMLassoWide.h:129-155 / ADMMEnet.h:124-141)
    if (!q.enet) {
        if ((double)out.lam > (double)q.lambda0 - 1e-5) out.type = W_ZERO;        // counter not advanced
        else { out.type = is_regular_update((unsigned)out.counter) ? W_REG : W_ACT; out.counter++; }
    } else {
        out.type = (is_regular_update((unsigned)out.counter) && out.lam < q.lambda0) ? W_REG : W_ACT;
        out.counter++;
    }
    out.total = in.total + 1;
    WideDecision dec;
    dec.out = out; dec.lam_finished = lam_finished; dec.niter_val = niter_val;
    dec.rp = tr_rp; dec.rd = tr_rd; dec.code = tr_code;
    return dec;
}

// x(g): every workgroup evaluates (identically) the decision for iteration g-1 -- convergence, rho
// adaptation, lambda schedule, which x-update runs now -- then builds t = Ax + z + y / rho in LDS and
// updates the columns its waves own.  A regular step visits every column (x = prox(x - X_j't / gamma)),
// an active-set step only the current non-zeros; both stream X_j once with 16-byte loads.
// TG (n too large for the LDS): t is not staged here but read from q.tbuf, written by wide_t_kernel just before.
template <int RT, bool TG = false>     // RT > 0: fused gather, a column is RT float4 per lane (n <= RT * 256); RT == 0: x-update only
__global__ void __launch_bounds__(kWideThreads)
wide_x_kernel(WideParams q, int par) {
    static_assert(!(TG && RT > 0), "the global-t mode is the unfused x-update");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int npad = (q.n + 255) / 256 * 256;
    float* tl = reinterpret_cast<float*>(smem_raw);            // t        [npad]
    float* tdl = tl + npad;                                    // t/gamma  [npad]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int w = blockIdx.x * (kWideThreads / 64) + wid;
    // ---- One memory round trip for everything the prologue needs.  The control block, the norm partials of the
    // decision and -- in workgroups that take part in every kind of step (all of them when not fused) -- the operands of
    // t = Ax + z + y / rho and this wave's first 512 slot values of an active-set step are all requested BEFORE the
    // first use of any of them.  (Round 2 found the earlier form -- control block, then slot values, then a staging
    // loop that waited on each of its 8 passes, then the norm partials -- to be a chain of ~12 dependent round trips:
    // the active-set launch took 12 us.)  The empty asm with a memory clobber keeps the compiler from sinking the
    // loads below the branches that follow.
    WIDE_PROBE_DECL
    WIDE_PROBE(0);
    const int pobs = blockIdx.x == 0 ? 0 : ((int)blockIdx.x == kActWG - 1 ? 1 : (blockIdx.x == gridDim.x - 1 ? 2 : -1));
    (void)pobs;
    const WideCtlRaw in_raw = wide_ctl_request(q.ctl + par);
    WideCtl* outp = &q.ctl[par ^ 1];
    const bool always = RT == 0 || (int)blockIdx.x < kActWG;
    constexpr int NT = RT > 0 ? RT : 1;
    float xs0[8], ta[NT], tz[NT], tb[NT];
#pragma unroll
    for (int u = 0; u < 8; ++u) xs0[u] = 0.f;
    const int NWa = min((int)gridDim.x, kActWG) * (kWideThreads / 64);
    if (RT > 0 && always) {
#pragma unroll
        for (int u = 0; u < 8; ++u)                                    // clamped index (select below): no branch between the loads
            xs0[u] = q.x[min((long long)(u * 64 + lane) * NWa + w, (long long)q.p - 1)];
    }
    auto load_t = [&]() {                                             // Ax, z, y are allocated and zeroed to 4096 entries at least
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const int i = k * kWideThreads + threadIdx.x;
            ta[k] = q.Ax[i]; tz[k] = q.z[i]; tb[k] = q.y[i];
        }
    };
    if (RT > 0 && always) load_t();
    const WideNormRaw nraw = wide_norms_request(q, lane);
    __builtin_amdgcn_sched_barrier(0);                                 // every request above is issued before the first use below
    asm volatile("" ::: "memory");
    constexpr int NRT = RT > 0 ? RT : 1;
    const int nv = (q.n + 3) / 4 * 4;
    // fused mode: col_request puts a whole column in flight (RT 16-byte loads per lane, reused by the gather)
    // A regular step streams all of X once (non-temporal: it cannot stay cache-resident and should not evict the active
    // columns, which are re-read every iteration with plain loads).
    auto col_request = [&](long long jj, float4 (&cv)[NRT], bool nt) {
        const float* col = q.X + (size_t)jj * q.ldx;
#pragma unroll
        for (int k = 0; k < NRT; ++k) {
            const int r = k * 256 + lane * 4;
            cv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < nv) cv[k] = nt ? load16_nt<float4>(col + r) : *reinterpret_cast<const float4*>(col + r);
        }
    };
    // Speculative request of this wave's first non-zero column: almost every step is an active-set step, and what it reads
    // first does not depend on the decision -- so the column's round trip overlaps the decision and the staging of t
    // (a regular / zero / final step simply drops it).
    // RT <= 8 runs two waves per SIMD and both fit; RT >= 16 runs one wave per SIMD anyway (512 registers to spend), RT = 32
    // has no room left for a second column
    constexpr bool kPair = RT > 0 && RT <= 16;                         // regular steps take the wave's columns two at a time
    constexpr bool kSpec = RT > 0 && RT <= 24;                         // the first non-zero column is requested before the decision
    float4 cv0[kSpec ? NRT : 1];
    long long pj = -1;
    if (RT > 0 && always) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if ((long long)(u * 64 + lane) * NWa + w >= q.p) xs0[u] = 0.f;
        if constexpr (kSpec) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const unsigned long long m = __ballot(xs0[u] != 0.f);
                if (pj < 0 && m != 0) pj = (long long)(u * 64 + __ffsll((long long)m) - 1) * NWa + w;
            }
            if (pj >= 0) col_request(pj, cv0, false);
        }
    }
    const WideCtl in = wide_ctl_unpack(in_raw);
    double sums[5];
    wide_norms_finish(q, lane, nraw, sums);
    if (in.done) {
        if (blockIdx.x == 0 && threadIdx.x == 0) *outp = in;
        return;
    }
    WIDE_PROBE(1);
    const WideDecision dec = wide_decide(q, in, sums, lane);
    const WideCtl out = wide_ctl_uniform(dec.out);
    WIDE_PROBE(2);
    const int lam_finished = __builtin_amdgcn_readfirstlane(dec.lam_finished), niter_val = __builtin_amdgcn_readfirstlane(dec.niter_val);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (lam_finished >= 0) q.niter[lam_finished] = niter_val;
        *outp = out;
        if (out.done) { *q.done = 1; *q.done_host = 1; }
        if (q.trace != nullptr && in.total < q.trace_cap) {          // what ADMMBase.h:111-146 (print_row, commented out there) would print
            double* t = q.trace + (size_t)in.total * ADMM_TRACE_FIELDS;
            t[0] = in.lam_idx; t[1] = in.iter; t[2] = in.eps_primal; t[3] = in.eps_dual; t[4] = dec.rp; t[5] = dec.rd;
            t[6] = out.rho; t[7] = out.type; t[8] = dec.code; t[9] = in.rho; t[10] = out.rho; t[11] = in.lam;
        }
    }
    const bool snap = lam_finished >= 0;                               // get_x() snapshot of the OLD x (Lasso.cpp:119)
    float* bsnap = snap ? q.beta + (size_t)lam_finished * q.p : nullptr;
    const int gid = blockIdx.x * kWideThreads + threadIdx.x, gsz = gridDim.x * kWideThreads;
    if (out.done || out.type == W_ZERO) {
        for (int j = gid; j < q.p; j += gsz) {
            if (snap) bsnap[j] = q.x[j];
            if (!out.done) q.x[j] = 0.f;
        }
        return;
    }
    const bool reg = out.type == W_REG;
    // fused: an active-set iteration is done by the first kActWG workgroups only (few columns, few partials);
    // the others leave here, before any barrier or LDS traffic
    if (RT > 0 && !reg && !always) { WIDE_PROBE_FLUSH(pobs, in.total); return; }
    if (!TG) {
        // t = cache_Ax + aux_z + dual_y / Scalar(rho); the active-set form divides by gamma first (:90, :141)
        const float rho_f = (float)out.rho;
        if (RT > 0) {
            if (!always) load_t();                                     // regular step of a workgroup beyond kActWG
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                const int i = k * kWideThreads + threadIdx.x;
                if (i < npad) {
                    const float t = i < q.n ? (ta[k] + tz[k]) + tb[k] / rho_f : 0.f;
                    tl[i] = t;
                    tdl[i] = t / q.gamma;
                }
            }
        } else {
            // any n that fits the LDS: passes of 4 x 256 elements, the 12 loads of a pass in flight together
            for (int i0 = threadIdx.x; i0 < npad; i0 += 4 * kWideThreads) {
                float a4[4], b4[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = min(i0 + k * kWideThreads, npad - 1);       // npad <= ldn: in bounds
                    a4[k] = q.Ax[i] + q.z[i];
                    b4[k] = q.y[i];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = i0 + k * kWideThreads;
                    if (i < npad) {
                        const float t = i < q.n ? a4[k] + b4[k] / rho_f : 0.f;
                        tl[i] = t;
                        tdl[i] = t / q.gamma;
                    }
                }
            }
        }
        __syncthreads();
    }
    WIDE_PROBE(3);
    const double pen_d = (double)out.lam / (out.rho * (double)q.gamma);
    const float penalty = (float)pen_d;                               // `const Scalar penalty` (:89)
    const float thresh_a = q.enet ? q.alpha * penalty : penalty;
    const float denom_a = q.enet ? (float)(1.0 + (double)penalty * (1.0 - (double)q.alpha)) : 1.f;
    const float thresh_r = (float)((double)q.alpha * pen_d);
    const float denom_r = (float)(1.0 + pen_d * (1.0 - (double)q.alpha));
    const float* tv = TG ? q.tbuf : (reg ? tl : tdl);
    const int nblk = (RT > 0 && !reg) ? min((int)gridDim.x, kActWG) : (int)gridDim.x;
    const int NW = nblk * (kWideThreads / 64);
    float4 acc[RT > 0 ? RT : 1];
#pragma unroll
    for (int k = 0; k < (RT > 0 ? RT : 1); ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);

    // One column: d = X_j't (or X_j't/gamma), the prox, and (fused) acc += x_j X_j.  Returns the new x_j (wave uniform).
    auto finish = [&](float d, float xv) -> float {
        float xn;
        if (reg) {
            const float vec = (-d) / q.gamma + xv;                    // vec = -X't / gamma; vec += main_x   (:147-149)
            if (!q.enet) {                                            // soft_threshold, double compare (:70-84)
                const double v = (double)vec;
                xn = v > pen_d ? (float)(v - pen_d) : (v < -pen_d ? (float)(v + pen_d) : 0.f);
            } else {
                xn = vec > thresh_r ? (vec - thresh_r) / denom_r : (vec < -thresh_r ? (vec + thresh_r) / denom_r : 0.f);
            }
        } else {
            xn = prox_f(xv - d, thresh_a, denom_a, q.enet != 0);
        }
        return xn;
    };
    // fused mode: consume a column that col_request put in flight
    auto col_finish = [&](float xv, const float4 (&cv)[NRT]) -> float {
        float d0 = 0.f, d1 = 0.f;
#pragma unroll
        for (int k = 0; k < NRT; ++k) {
            const int r = k * 256 + lane * 4;
            if (r < nv) {
                const float4 b = *reinterpret_cast<const float4*>(tv + r);
                float& dd = (k & 1) ? d1 : d0;
                dd = fmaf(cv[k].x, b.x, dd); dd = fmaf(cv[k].y, b.y, dd); dd = fmaf(cv[k].z, b.z, dd); dd = fmaf(cv[k].w, b.w, dd);
            }
        }
        const float xn = finish(wave_sum(d0 + d1), xv);
        if (xn != 0.f) {                                               // gather: Ax partial += x_j X_j
#pragma unroll
            for (int k = 0; k < NRT; ++k) {
                acc[k].x = fmaf(xn, cv[k].x, acc[k].x); acc[k].y = fmaf(xn, cv[k].y, acc[k].y);
                acc[k].z = fmaf(xn, cv[k].z, acc[k].z); acc[k].w = fmaf(xn, cv[k].w, acc[k].w);
            }
        }
        return xn;
    };
    auto column = [&](long long jj, float xv) -> float {
        if (RT > 0) {
            if constexpr (kSpec) { if (!reg && jj == pj) return col_finish(xv, cv0); }     // requested before the decision
            float4 cv[NRT];
            col_request(jj, cv, reg && q.x_nt);
            return col_finish(xv, cv);
        }
        const float* col = q.X + (size_t)jj * q.ldx;
        float d0 = 0.f, d1 = 0.f;
        int r = lane * 4;
        for (; r + 256 < nv; r += 512) {
            const bool nt = reg && q.x_nt;
            const float4 a0 = nt ? load16_nt<float4>(col + r) : *reinterpret_cast<const float4*>(col + r);
            const float4 a1 = nt ? load16_nt<float4>(col + r + 256) : *reinterpret_cast<const float4*>(col + r + 256);
            const float4 b0 = *reinterpret_cast<const float4*>(tv + r);
            const float4 b1 = *reinterpret_cast<const float4*>(tv + r + 256);
            d0 = fmaf(a0.x, b0.x, d0); d0 = fmaf(a0.y, b0.y, d0); d0 = fmaf(a0.z, b0.z, d0); d0 = fmaf(a0.w, b0.w, d0);
            d1 = fmaf(a1.x, b1.x, d1); d1 = fmaf(a1.y, b1.y, d1); d1 = fmaf(a1.z, b1.z, d1); d1 = fmaf(a1.w, b1.w, d1);
        }
        if (r < nv) {
            const float4 a0 = *reinterpret_cast<const float4*>(col + r);
            const float4 b0 = *reinterpret_cast<const float4*>(tv + r);
            d0 = fmaf(a0.x, b0.x, d0); d0 = fmaf(a0.y, b0.y, d0); d0 = fmaf(a0.z, b0.z, d0); d0 = fmaf(a0.w, b0.w, d0);
        }
        return finish(wave_sum(d0 + d1), xv);
    };

    // ---- safe screening of a regular step (fused mode).  A regular step applies the prox to EVERY column, and for all but a per cent
    // of them the outcome is "stays zero": x_j = 0 and |fl(X_j't)| / gamma below the threshold.  Which ones cannot be known without a
    // product with every column -- but a product with a ROUNDED column is enough to prove it: with Xh = fp16(X),
    //     |fl32(X_j't)| <= |fl32(Xh_j't)| + s_j ||t||_2 ,     s_j = ||X_j - Xh_j||_2 + gamma_n (||X_j||_2 + ||Xh_j||_2)
    // (Cauchy-Schwarz on the rounding of the column, the standard bound gamma_n = n u / (1 - n u) on an n-term float inner product in
    // ANY order for each of the two computed sums; s_j is formed once at setup in double and rounded UP, wide_screen_prep_kernel).  A
    // column with x_j = 0 whose bound stays below gamma * threshold * (1 - 2^-21) is left alone -- the exact step would have stored the
    // zero it already holds (the float division by gamma rounds by at most 2^-24, the comparison is monotone) -- and every other
    // column (x_j != 0, bound not met, anything non-finite) takes the exact path below on the float column: the iterates are
    // BIT-IDENTICAL to the unscreened step's, and a regular step streams 2 n p bytes instead of 4 n p.
    const bool screen = RT > 0 && reg && q.Xh != nullptr;
    double scrT = 0.0, scrG = 0.0;
    if (RT > 0 && screen) {
        double tsq = 0.0;
        for (int i = lane; i < npad; i += 64) { const double v = (double)tl[i]; tsq = fma(v, v, tsq); }
        scrT = sqrt(wave_sum(tsq)) * (1.0 + 1e-12);                    // >= ||t||_2 (every wave forms it itself: 32 LDS reads)
        scrG = (double)q.gamma * (q.enet ? (double)thresh_r : pen_d) * (1.0 - 4.8e-7);
    }
    unsigned long long scr_seen = 0, scr_exact = 0;

    if constexpr (RT > 0) {
        if (screen) {
            // The copy is fp16 (8 elements per 16-byte piece) or, where a linear 8-bit code of the column is fine enough (setup decides),
            // signed bytes with a per-column scale (16 per piece: n p bytes per regular step); the bound has the same form for both.
            // A round = CR columns of the wave; while NP * CR * 2 <= 16 pieces two rounds' requests are in flight at any time (the next
            // round is requested before the current one is reduced); the next group's x / bounds are requested a group ahead.
            auto screened_step = [&](auto fmt_tag) {
                constexpr int FMT = decltype(fmt_tag)::value;
                constexpr int EPL = FMT == 16 ? 8 : 16;                               // elements of a piece
                constexpr int NP = (RT * 4 + EPL - 1) / EPL;                          // pieces of a column per lane (64 EPL rows per wave load)
                constexpr int CR = (FMT == 8 && NP <= 2) ? 4 : 2;                     // columns per round
                constexpr bool kPipe = NP * CR * 2 <= 16;
                const unsigned char* base = reinterpret_cast<const unsigned char*>(q.Xh);
                const size_t colbytes = (size_t)q.ldh * (FMT / 8);
                auto scr_request = [&](unsigned long long& mask, int sc, int (&lc)[CR], long long (&jc)[CR], uint4 (&h)[CR][NP]) {
#pragma unroll
                    for (int c = 0; c < CR; ++c) {
                        lc[c] = -1; jc[c] = 0;
                        if (mask) { lc[c] = __ffsll((long long)mask) - 1; mask &= mask - 1; jc[c] = (long long)(sc + lc[c]) * NW + w; }
                        const unsigned char* hcol = base + (size_t)jc[c] * colbytes;
#pragma unroll
                        for (int k = 0; k < NP; ++k) {
                            const int r = (k * 64 + lane) * EPL;
                            h[c][k] = make_uint4(0u, 0u, 0u, 0u);
                            if (lc[c] >= 0 && r < q.ldh) h[c][k] = load16_nt<uint4>(hcol + (size_t)(k * 64 + lane) * 16);
                        }
                    }
                };
                auto scr_consume = [&](const int (&lc)[CR], const long long (&jc)[CR], const uint4 (&h)[CR][NP], float xj, float sj, float scj) {
                    float dd[CR];
#pragma unroll
                    for (int c = 0; c < CR; ++c) dd[c] = 0.f;
#pragma unroll
                    for (int k = 0; k < NP; ++k) {
                        const int r = (k * 64 + lane) * EPL;
                        if (r < q.ldh) {
                            float tv8[EPL];
#pragma unroll
                            for (int e4 = 0; e4 < EPL / 4; ++e4) {
                                const float4 b = *reinterpret_cast<const float4*>(tl + r + 4 * e4);
                                tv8[4 * e4] = b.x; tv8[4 * e4 + 1] = b.y; tv8[4 * e4 + 2] = b.z; tv8[4 * e4 + 3] = b.w;
                            }
#pragma unroll
                            for (int c = 0; c < CR; ++c) {
                                float d0 = dd[c], d1 = 0.f;
                                if constexpr (FMT == 16) {
                                    typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
                                    const half8_t hv = __builtin_bit_cast(half8_t, h[c][k]);
#pragma unroll
                                    for (int e = 0; e < 4; ++e) { d0 = fmaf((float)hv[e], tv8[e], d0); d1 = fmaf((float)hv[4 + e], tv8[4 + e], d1); }
                                } else {
                                    const unsigned wq[4] = {h[c][k].x, h[c][k].y, h[c][k].z, h[c][k].w};
#pragma unroll
                                    for (int e = 0; e < 16; ++e) {
                                        const float qv = (float)(int)(signed char)(unsigned char)(wq[e >> 2] >> (8 * (e & 3)));
                                        if (e & 1) d1 = fmaf(qv, tv8[e], d1); else d0 = fmaf(qv, tv8[e], d0);
                                    }
                                }
                                dd[c] = d0 + d1;
                            }
                        }
                    }
                    float v8[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) v8[c] = c < CR ? dd[c < CR ? c : 0] : 0.f;
                    const float tot = halving_sum8(v8, lane);          // lanes 8 c .. 8 c + 7: the wave total of column c
#pragma unroll
                    for (int c = 0; c < CR; ++c) {
                        if (lc[c] < 0) break;                          // uniform
                        const float dc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tot), 8 * c));
                        const float sc_j = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sj), lc[c]));
                        const float xc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xj), lc[c]));
                        double dabs = (double)fabsf(dc);
                        if constexpr (FMT == 8) dabs *= (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(scj), lc[c]));
                        const bool stays_zero = xc == 0.f && dabs + (double)sc_j * scrT <= scrG;          // (a NaN anywhere: false)
                        if (!stays_zero) {                             // the exact step on the float column
                            float4 cv[NRT];
                            col_request(jc[c], cv, false);
                            const float xn = col_finish(xc, cv);
                            if (lane == lc[c]) q.x[jc[c]] = xn;
                            ++scr_exact;
                        }
                    }
                };
                auto group_load = [&](int sc, float& xj, float& sj, float& scj) {
                    const long long jl = (long long)(sc + lane) * NW + w;
                    xj = 0.f; sj = 0.f; scj = 0.f;
                    if ((long long)sc * NW < q.p) {
                        xj = jl < q.p ? q.x[jl] : 0.f;
                        sj = q.scr_s[(size_t)w * q.scr_S + sc + lane];                 // the bounds of the group's 64 columns: one 256-byte line
                        if constexpr (FMT == 8) scj = q.scr_scale[(size_t)w * q.scr_S + sc + lane];
                    }
                };
                float xj_n, sj_n, scj_n;
                group_load(0, xj_n, sj_n, scj_n);
                for (int sc = 0; (long long)sc * NW < q.p; sc += 64) {
                    const long long jl = (long long)(sc + lane) * NW + w;
                    const float xj = xj_n, sj = sj_n, scj = scj_n;
                    group_load(sc + 64, xj_n, sj_n, scj_n);
                    if (snap && jl < q.p) bsnap[jl] = xj;
                    unsigned long long mask = __ballot(jl < q.p);
                    scr_seen += __popcll(mask);
                    int lA[CR];
                    long long jA[CR];
                    uint4 hA[CR][NP];
                    if constexpr (kPipe) {
                        int lB[CR];
                        long long jB[CR];
                        uint4 hB[CR][NP];
                        scr_request(mask, sc, lA, jA, hA);
                        for (;;) {
                            scr_request(mask, sc, lB, jB, hB);
                            scr_consume(lA, jA, hA, xj, sj, scj);
                            if (lB[0] < 0) break;
                            scr_request(mask, sc, lA, jA, hA);
                            scr_consume(lB, jB, hB, xj, sj, scj);
                            if (lA[0] < 0) break;
                        }
                    } else {
                        while (mask) {
                            scr_request(mask, sc, lA, jA, hA);
                            scr_consume(lA, jA, hA, xj, sj, scj);
                        }
                    }
                }
            };
            if (q.scr_fmt == 8) screened_step(std::integral_constant<int, 8>{});
            else screened_step(std::integral_constant<int, 16>{});
        }
    }
    if (!screen)
    for (int sc = 0; (long long)sc * NW < q.p; sc += 64 * 8) {
        float xs[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {                                  // the wave's next 512 columns: 8 independent loads per lane
            const long long jl = (long long)(sc + u * 64 + lane) * NW + w;
            if (RT > 0 && !reg && sc == 0) xs[u] = xs0[u];             // fetched before the decision
            else xs[u] = (jl < q.p) ? q.x[jl] : 0.f;
            if (snap && jl < q.p) bsnap[jl] = xs[u];
        }
#pragma unroll 1
        for (int u = 0; u < 8; ++u) {
            const int s0 = sc + u * 64;
            if ((long long)s0 * NW >= q.p) break;
            const long long jl = (long long)(s0 + lane) * NW + w;
            const float xj = u == 0 ? xs[0] : (u == 1 ? xs[1] : (u == 2 ? xs[2] : (u == 3 ? xs[3] : (u == 4 ? xs[4] : (u == 5 ? xs[5] : (u == 6 ? xs[6] : xs[7]))))));
            unsigned long long mask = reg ? __ballot(jl < q.p) : __ballot(xj != 0.f);
            if constexpr (kPair) {
                // regular step, fused mode: two columns in flight per wave AT ALL TIMES -- the column after next is requested into
                // the registers a column has just left (round 3 requested a pair, consumed the pair, requested the next pair: the
                // wave's memory pipe ran empty once per pair).  Same order of columns: bit-identical partial sums.
                if (reg && mask) {
                    float4 c0[NRT], c1[NRT];
                    auto next_col = [&](int& l, long long& j) {
                        l = -1; j = 0;
                        if (mask) { l = __ffsll((long long)mask) - 1; mask &= mask - 1; j = (long long)(s0 + l) * NW + w; }
                    };
                    int l0, l1;
                    long long j0, j1;
                    next_col(l0, j0);
                    col_request(j0, c0, q.x_nt != 0);
                    next_col(l1, j1);
                    if (l1 >= 0) col_request(j1, c1, q.x_nt != 0);
                    for (;;) {
                        const float xn0 = col_finish(__shfl(xj, l0, 64), c0);
                        if (lane == l0) q.x[j0] = xn0;
                        int l2, l3;
                        long long j2, j3;
                        next_col(l2, j2);
                        if (l2 >= 0) col_request(j2, c0, q.x_nt != 0);
                        if (l1 < 0) break;
                        const float xn1 = col_finish(__shfl(xj, l1, 64), c1);
                        if (lane == l1) q.x[j1] = xn1;
                        next_col(l3, j3);
                        if (l3 >= 0) col_request(j3, c1, q.x_nt != 0);
                        if (l2 < 0) break;
                        l0 = l2; j0 = j2; l1 = l3; j1 = j3;
                    }
                }
            }
            while (mask) {
                const int l = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const long long jj = (long long)(s0 + l) * NW + w;
                const float xn = column(jj, __shfl(xj, l, 64));
                if (lane == l) q.x[jj] = xn;
            }
        }
    }
    WIDE_PROBE(4);
    if (RT > 0 && screen && q.scr_stat != nullptr && lane == 0) { atomicAdd(q.scr_stat, scr_seen); atomicAdd(q.scr_stat + 1, scr_exact); }
    if (RT > 0) {
        // combine the 4 waves of the workgroup through the (now free) t buffers, up to 8 row slices per round (one round
        // for n <= 2048), then write this workgroup's partial row
        constexpr int RS = RT < 8 ? RT : 8;
        float4* red = reinterpret_cast<float4*>(smem_raw);             // >= RS * kWideThreads float4 (launch)
#pragma unroll
        for (int k0 = 0; k0 < RT; k0 += RS) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < RS; ++k) red[k * kWideThreads + threadIdx.x] = acc[k0 + k];
            __syncthreads();
#pragma unroll
            for (int h = 0; h < RS / 4; ++h) {                         // RS slices x 64 lanes outputs, RS / 4 per thread
                const int k = h * 4 + (threadIdx.x >> 6), ln = threadIdx.x & 63;
                float4 sacc = red[k * kWideThreads + ln];
#pragma unroll
                for (int ww = 1; ww < kWideThreads / 64; ++ww) {
                    const float4 v = red[k * kWideThreads + ww * 64 + ln];
                    sacc.x += v.x; sacc.y += v.y; sacc.z += v.z; sacc.w += v.w;
                }
                const int rr = (k0 + k) * 256 + ln * 4;
                if (rr < q.ldn) *reinterpret_cast<float4*>(q.axpart + (size_t)blockIdx.x * q.ldn + rr) = sacc;
            }
        }
    }
    WIDE_PROBE(5);
    WIDE_PROBE_FLUSH(pobs, in.total);
}

// Large-n mode (2 n floats exceed the LDS of a workgroup): t = Ax + z + y / rho of the iteration about to run (divided by
// gamma on an active-set step) into global memory, by its own small launch just before the x-update.  Every workgroup
// evaluates the same decision as the x-update will (same inputs, same code); the control block is published by the
// x-update alone.
__global__ void __launch_bounds__(kWideThreads)
wide_t_kernel(WideParams q, int par) {
    const WideCtl in = q.ctl[par];
    double sums[5];
    wide_norms_finish(q, threadIdx.x & 63, wide_norms_request(q, threadIdx.x & 63), sums);
    if (in.done) return;
    const WideCtl out = wide_decide(q, in, sums, threadIdx.x & 63).out;
    if (out.done || out.type == W_ZERO) return;
    const int i = blockIdx.x * kWideThreads + threadIdx.x;
    if (i >= q.ldn) return;
    const float rho_f = (float)out.rho;
    float t = 0.f;
    if (i < q.n) {
        t = (q.Ax[i] + q.z[i]) + q.y[i] / rho_f;
        if (out.type != W_REG) t = t / q.gamma;
    }
    q.tbuf[i] = t;
}

// Gather mat-vec: axpart[b][i] = sum over the non-zero columns j owned by workgroup b of x_j X[i, j].
__global__ void __launch_bounds__(kWideThreads)
wide_ax_kernel(WideParams q) {
    __shared__ float4 red[kWideThreads];
    if (*q.done) return;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int NW = gridDim.x * (kWideThreads / 64);
    const int w = blockIdx.x * (kWideThreads / 64) + wid;
    for (int r0 = 0; r0 < q.n; r0 += 256 * kAxRT) {
        float4 acc[kAxRT];
#pragma unroll
        for (int k = 0; k < kAxRT; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int rows_here = min(q.n - r0, 256 * kAxRT);
        const int npass = (rows_here + 255) / 256;
        for (int sc = 0; (long long)sc * NW < q.p; sc += 64 * 8) {
            float xs[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {                          // the wave's next 512 columns: 8 independent loads per lane
                const long long jl = (long long)(sc + u * 64 + lane) * NW + w;
                xs[u] = (jl < q.p) ? q.x[jl] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int s0 = sc + u * 64;
                const float xj = xs[u];
                unsigned long long mask = __ballot(xj != 0.f);
                while (mask) {
                    const int l = __ffsll((long long)mask) - 1;
                    mask &= mask - 1;
                    const long long jj = (long long)(s0 + l) * NW + w;
                    const float xv = __shfl(xj, l, 64);
                    const float* col = q.X + (size_t)jj * q.ldx + r0;
#pragma unroll
                    for (int k = 0; k < kAxRT; ++k) {
                        if (k < npass) {
                            const int r = k * 256 + lane * 4;
                            if (r < rows_here) {                      // ldx padding rows are zero, vector load is in-bounds
                                const float4 a = *reinterpret_cast<const float4*>(col + r);
                                acc[k].x = fmaf(xv, a.x, acc[k].x); acc[k].y = fmaf(xv, a.y, acc[k].y);
                                acc[k].z = fmaf(xv, a.z, acc[k].z); acc[k].w = fmaf(xv, a.w, acc[k].w);
                            }
                        }
                    }
                }
            }
        }
        // combine the 4 waves of the workgroup, then write this workgroup's partial
#pragma unroll
        for (int k = 0; k < kAxRT; ++k) {
            if (k < npass) {
                __syncthreads();
                red[threadIdx.x] = acc[k];
                __syncthreads();
                if (wid == 0) {
                    float4 s = red[lane];
#pragma unroll
                    for (int ww = 1; ww < kWideThreads / 64; ++ww) {
                        const float4 o = red[ww * 64 + lane];
                        s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
                    }
                    const int r = r0 + k * 256 + lane * 4;
                    float* dst = q.axpart + (size_t)blockIdx.x * q.ldn + r;
                    if (r < q.ldn) *reinterpret_cast<float4*>(dst) = s;
                }
            }
        }
    }
}

// Geometry of the z/y kernels: 8 lanes share one element and issue their 16 partial loads at once (one memory round trip).
constexpr int kWtLanes = 8;
constexpr int kWtElems = kWideThreads / kWtLanes;
static_assert(kAxWG == 16 * kWtLanes && kActWG == 2 * kAxWG, "tail reduction assumes 16 (+16 when fused) partials per lane");

// Ax_i = sum of the x-update's per-workgroup partials: 8 lanes share one element and issue their 16 (+16) partial loads
// at once, BEFORE anything that depends on the control block (a version that took `c` as a function argument made the
// compiler wait for the control block first: one more memory round trip, 9.7 instead of 6.2 us per launch, C3 -15 %).
#define WIDE_SUM_AXPART(ax)                                                                                              \
    const int ic = min(i, q.n - 1);                      /* clamped: every load below is unconditional and in bounds */  \
    const float* ap = q.axpart + (size_t)sub * q.ldn + ic;                                                               \
    const size_t rstep = (size_t)kWtLanes * q.ldn;       /* partial rows k * 8 + sub */                                  \
    float v[16], v2[16];                                                                                                 \
    _Pragma("unroll") for (int k = 0; k < 16; ++k) v[k] = ap[k * rstep];                                                 \
    _Pragma("unroll") for (int k = 0; k < 16; ++k) v2[k] = 0.f;                                                          \
    if (q.fused) { _Pragma("unroll") for (int k = 0; k < 16; ++k) v2[k] = ap[(16 + k) * rstep]; }   /* rows 128..255 */  \
    float zo = q.z[ic], yo = q.y[ic], yd = q.Y[ic];                                                                      \
    __builtin_amdgcn_sched_barrier(0);                   /* all requests issued before the first use */                  \
    if (!valid) { zo = 0.f; yo = 0.f; yd = 0.f; }                                                                        \
    float ax = 0.f;                                                                                                      \
    _Pragma("unroll") for (int k = 0; k < 16; ++k) ax += v[k];                                                           \
    _Pragma("unroll") for (int k = 0; k < 16; ++k) ax += v2[k];                                                          \
    if (q.fused && c.type == W_REG) {      /* a regular step: every x-update workgroup wrote a partial */                \
        for (int b0 = kActWG; b0 < q.nwg_x; b0 += 16 * kWtLanes) {                                                       \
            _Pragma("unroll") for (int k = 0; k < 16; ++k) {                                                             \
                const int b = b0 + k * kWtLanes + sub;                                                                   \
                v[k] = (b < q.nwg_x) ? q.axpart[(size_t)b * q.ldn + ic] : 0.f;                                           \
            }                                                                                                            \
            _Pragma("unroll") for (int k = 0; k < 16; ++k) ax += v[k];                                                   \
        }                                                                                                                \
    }                                                                                                                    \
    _Pragma("unroll") for (int m = 1; m < kWtLanes; m <<= 1) ax += __shfl_xor(ax, m, 64);                                \
    if (q.fused && c.type == W_ZERO) ax = 0.f;   /* x = 0: nobody wrote partials */

// Column-sharded mode: this rank's share of Ax (its column block's partials summed) into `out`, which the ranks then
// all-reduce; the tail reads the global Ax from there (WideParams::ax_given).
__global__ void __launch_bounds__(kWideThreads)
wide_ax_local_kernel(WideParams q, int par, float* out) {
    const WideCtl c = q.ctl[par ^ 1];
    const int sub = threadIdx.x & (kWtLanes - 1);
    const int i = blockIdx.x * kWtElems + threadIdx.x / kWtLanes;
    const bool valid = i < q.n;
    WIDE_SUM_AXPART(ax)
    (void)zo; (void)yo; (void)yd;
    if (!q.fused && c.type == W_ZERO) ax = 0.f;
    if (i < q.ldn && sub == 0) out[i] = valid ? ax : 0.f;
}

// The same over the PEER exchange without launches of the exchange layer: every workgroup writes its elements of this
// rank's share of A x straight into this rank's slot of EVERY rank's buffer (pairs of elements as one 8-byte write-through
// store; lane `sub` of an element's group serves ranks sub, sub + 8, ...), and the last workgroup to finish raises the flags
// (peer_device.h).  Skipped, like the consumer's wait, once the replicated control block says the solve has finished.
__global__ void __launch_bounds__(kWideThreads)
wide_ax_push_kernel(WideParams q, int par, PeerExchange ex) {
    const WideCtl c = q.ctl[par ^ 1];
    const int sub = threadIdx.x & (kWtLanes - 1);
    const int i = blockIdx.x * kWtElems + threadIdx.x / kWtLanes;
    const bool valid = i < q.n;
    WIDE_SUM_AXPART(ax)
    (void)zo; (void)yo; (void)yd;
    if (c.done) return;                                               // uniform over the launch and over the ranks
    if (!q.fused && c.type == W_ZERO) ax = 0.f;
    if (!valid) ax = 0.f;
    const float ax_next = __shfl_down(ax, kWtLanes, 64);              // the element owned by the next group of 8 lanes
    if (((threadIdx.x / kWtLanes) & 1) == 0 && i < q.ldn) {           // even elements store the pair (i, i + 1); ldn is even
        for (int dst = sub; dst < ex.nranks; dst += kWtLanes)
            peer_store_f32x2(reinterpret_cast<float*>(peer_dst_slot(ex, dst)) + i, ax, ax_next);
    }
    peer_publish(ex, gridDim.x);
}

// z/y update: Ax = sum of partials; z_new = -(y_data + y + rho Ax) / (1 + rho); r = Ax + z_new; y += rho r; norms.
// PEER 1: A x arrives in the K exchange slots (wide_ax_push_kernel of every rank): wait for the flags, sum in lane order.
// PEER 2: producer and consumer in ONE launch -- this launch sums the rank's own partials anyway, so it writes the share into
// every rank's slot itself, counts itself in, and then waits (its workgroups wait for one another: the host chooses it only
// when the grid is resident with room to spare, which for n / 32 workgroups it practically always is).
template <int PEER = 0>
__global__ void __launch_bounds__(kWideThreads)
wide_tail_kernel(WideParams q, int par, PeerExchange ex) {
    __shared__ double scratch[5 * (kWideThreads / 64)];
    WIDE_PROBE_DECL
    WIDE_PROBE(0);
    const WideCtl c = q.ctl[par ^ 1];
    const int sub = threadIdx.x & (kWtLanes - 1);
    const int i = blockIdx.x * kWtElems + threadIdx.x / kWtLanes;
    const bool valid = i < q.n;
    WIDE_SUM_AXPART(ax)
    if (!PEER && q.ax_given != nullptr) ax = valid ? q.ax_given[i] : 0.f;      // column-sharded mode: already summed over partials and ranks
    if (c.done) return;
    if (PEER == 2) {
        float mine = valid ? ax : 0.f;
        if (!q.fused && c.type == W_ZERO) mine = 0.f;
        const float next = __shfl_down(mine, kWtLanes, 64);               // the element owned by the next group of 8 lanes
        if (((threadIdx.x / kWtLanes) & 1) == 0 && i < q.ldn) {
            for (int dst = sub; dst < ex.nranks; dst += kWtLanes)
                peer_store_f32x2(reinterpret_cast<float*>(peer_dst_slot(ex, dst)) + i, mine, next);
        }
        peer_publish(ex, gridDim.x);
    }
    if (PEER) {
        const bool ok = peer_wait_relaxed(ex);
        float a = 0.f;
        if (ok && valid) {
            for (int r = sub; r < ex.nranks; r += kWtLanes) {           // rank order fixed by the lane pattern: identical on every rank
                const float2 v = peer_load_f32x2(reinterpret_cast<const float*>(peer_src_slot(ex, r)) + (i & ~1));
                a += (i & 1) ? v.y : v.x;
            }
        }
#pragma unroll
        for (int m = 1; m < kWtLanes; m <<= 1) a += __shfl_xor(a, m, 64);
        ax = a;
    }
    WIDE_PROBE(1);
    double acc[5] = {0, 0, 0, 0, 0};
    if (valid && sub == 0) {
        // no fused multiply-adds: the reference is built without them (R's default flags, /root/reference/src/Makevars), so
        // rho * Ax and rho * r round before they are added (lasso_tall.hip, tall_update_elem)
#pragma clang fp contract(off)
        const float rho_f = (float)c.rho;
        const float den = (float)(-1.0 - c.rho);                      // Scalar(-1 - rho)   (:164)
        const float zn = (yd + yo + rho_f * ax) / den;                // next_z (:156-165)
        const float dz = zn - zo;
        const float r = ax + zn;                                       // next_residual (:166-170)
        const float yn = yo + rho_f * r;                               // dual_y += rho * newr   ADMMBase.h:183
        q.Ax[i] = ax; q.z[i] = zn; q.y[i] = yn;
        acc[0] = (double)r * r; acc[1] = (double)dz * dz; acc[2] = (double)ax * ax;
        acc[3] = (double)zn * zn; acc[4] = (double)yn * yn;
    }
    // Block sum of the five norms.  Only the lanes with sub == 0 hold values, so wave_sum's xor-4 / 2 / 1 steps would add
    // exact zeros: the halving butterfly's top half (xor 32 / 16 / 8) leaves the wave total of value k in lane 8 k,
    // bit-identical to block_sum<double, 5> at 7 exchanges instead of 30.
    {
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
        const double v8[8] = {acc[0], acc[1], acc[2], acc[3], acc[4], 0.0, 0.0, 0.0};
        const double tot = halving_sum8_top(v8, lane);
        if ((lane & 7) == 0 && lane < 40) scratch[(lane >> 3) * (kWideThreads / 64) + wid] = tot;
        __syncthreads();
        if (threadIdx.x < 5) {
            double sum = 0;
            for (int ww = 0; ww < kWideThreads / 64; ++ww) sum += scratch[threadIdx.x * (kWideThreads / 64) + ww];
            q.P[(size_t)blockIdx.x * 8 + threadIdx.x] = sum;
        }
    }
    WIDE_PROBE(2);
    WIDE_PROBE_FLUSH(blockIdx.x == 0 ? 3 : -1, c.total - 1);
}

// Iterate dump (admm_hip_lasso_plan_state_*; test / diagnosis facility, launched only when enabled): the vectors the iteration
// that just finished leaves behind -- x | A x | z | y -- into the record with the number of the trace record that will judge
// them (the decision taken by the NEXT x-update launch; `total` of the control block this iteration's x-update published).
__global__ void __launch_bounds__(kWideThreads)
wide_state_kernel(WideParams q, int par) {
    const WideCtl c = q.ctl[par ^ 1];
    if (c.done || c.total >= q.state_cap) return;
    float* s = q.state + (size_t)c.total * ((size_t)q.p + 3 * (size_t)q.n);
    for (long long i = (long long)blockIdx.x * kWideThreads + threadIdx.x; i < q.p; i += (long long)gridDim.x * kWideThreads) s[i] = q.x[i];
    float* v = s + q.p;
    for (int i = blockIdx.x * kWideThreads + threadIdx.x; i < q.n; i += gridDim.x * kWideThreads) {
        v[i] = q.Ax[i]; v[(size_t)q.n + i] = q.z[i]; v[2 * (size_t)q.n + i] = q.y[i];
    }
}

// ------------------------------------------------------------------------------------------ persistent active-set stretch
// Almost every iteration of a wide path is an ACTIVE-SET step (ADMMLassoWide.h:86-118: only the current non-zeros are updated;
// 17 191 of 17 613 iterations at BASELINE configs[2]) on a few dozen to ~1500 columns -- as two latency-bound launches 16.6 us per
// iteration however little there is to do (profiles/r02_probe_timelines.md).  A persistent kernel runs a whole STRETCH of them --
// from wherever the two-launch path stands until the next regular step, a finished lambda, or an active set too large for it --
// inside ONE launch, its workgroups handing their results to one another through global memory (write-through stores or plain
// stores into a shared L2, cache-bypassing loads, monotonic flag words).  When the stretch ends the kernel leaves x, A x, z, y, the
// norm partials and the control block exactly as a tail launch would, and the next x-update launch simply continues.
// Round 3's kernel split the COLUMNS over 32 workgroups (two all-to-all hand-overs per iteration: partials of A x, then z / y /
// norm shares of the row owners; n <= 2048, <= 512 active columns): 10.3-10.5 us per iteration inside, 84 % of configs[2]'s
// iterations covered, 53.6-56.5 k it/s.  Measured and rejected there: one hand-over per iteration with 16 fat workgroups that each
// sum all partials (12.1 us).  It is replaced by the 2-D kernel below (round 4); helpers shared by both designs follow.
__device__ __forceinline__ void wp_store_f64(double* p, double v, bool wt) {
    if (wt) __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
__device__ __forceinline__ unsigned wp_xcc_id() {                                    // which XCD this wave runs on (0..7)
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}
// Wait (bounded) for the `npart` words that carry (launch << 32) | (XCD of the workgroup + 1); *s_same = 1 if all of them sit on
// one XCD: then payloads may be plain stores, which stay in that XCD's shared L2 where the readers' cache-bypassing loads find
// them at L2 latency; otherwise they are written through.  Lanes < npart of wave 0 poll one word each; everybody leaves together.
__device__ __forceinline__ bool wp_wait_xcc(const unsigned long long* flags, unsigned long long seq, int* err, int* s_ok, int* s_same, int npart) {
    if (threadIdx.x < 64) {
        bool ok = true;
        unsigned long long v = 0;
        if ((int)threadIdx.x < npart) {
            const unsigned long long* f = flags + (size_t)threadIdx.x * 8;
            const long long t0 = wall_clock64();
            while ((v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < ((seq << 32) | 1ull)) {
                __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > 200000000ll) { ok = false; break; }
            }
        }
        ok = __all(ok) != 0;
        const unsigned mine = (unsigned)v, first = (unsigned)__shfl(mine, 0, 64);
        const bool same = __all((int)threadIdx.x >= npart || mine == first) != 0;
        if (threadIdx.x == 0) { *s_ok = ok ? 1 : 0; *s_same = same ? 1 : 0; if (!ok) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
    __syncthreads();
    return *s_ok != 0;
}

// Round 4.  The stretch above splits the COLUMNS over its workgroups, so every iteration needs two all-to-all hand-overs: the
// partials of A x (every row needs every workgroup's share) and then the norm shares + z / y of the row owners (every
// workgroup needs them for the decision and the next t): 10.5 us per iteration, of which ~6 are the two hand-overs, and it
// re-reads every active column from the L2 in every iteration.
// This kernel splits the ROWS instead: workgroup g owns RS = 64 M rows of EVERYTHING -- of A x, z, y, t and of every active
// column of X.  Then
//   * A x, z, y, the residuals and t of its rows are local (no partials of A x at all);
//   * what needs all rows is the dot product X_j't of an active column: every workgroup forms the PARTIAL dots of its row slice
//     for all active columns (nS floats), and every workgroup sums the G partial dots of every column itself, in workgroup
//     order, and applies the prox -- redundantly and identically (x is replicated, like the decision);
//   * the only other global quantity, the five norms of the stopping test, travels WITH the partial dots of the NEXT iteration:
//     a workgroup that has finished its rows of iteration k forms t_{k+1} and the partial dots of iteration k+1 right away,
//     SPECULATING that the decision on iteration k will be "go on, same rho" (it is in ~95 % of the iterations: rho changes 49
//     times in 1120 iterations of the n = 300 test path), and publishes both together.  ONE hand-over per iteration.  When the
//     decision changes rho, t_{k+1} = (A x + z) + y / rho is formed again with the new rho and the partial dots are exchanged
//     once more (an extra hand-over for that iteration); when it ends the lambda / asks for a regular step / finishes the path,
//     the speculative dots are dropped and the stretch ends;
//   * a workgroup's slice of an active column is 64 M floats -- M per lane -- and the active set only shrinks inside a stretch, so
//     the slices of the first KRES columns of every wave stay in REGISTERS for the whole stretch (8 waves x 96 / M columns:
//     768 columns at n <= 2048): no matrix traffic at all in the steady state.  (First cut of this kernel: 8 workgroups of 256
//     rows re-reading their slices from the L2 twice per iteration -- 14.7 us per iteration at BASELINE configs[2], bound by what
//     EIGHT compute units can pull from the L2.)
// The arithmetic of an iteration is the reference's (ADMMLassoWide.h:86-118,156-170; same prox, same float formulas, FMA
// contraction off in the elementwise part); what differs from the two-launch path is only the ORDER of the sums inside the two
// mat-vecs (dot: M rows per lane, lanes, then row slices in order; A x: a wave's columns in list order, then the waves in
// order), which the stepwise rule holds to the float dot-product yardstick (oracle/stepcheck.py check_wide) like every other
// variant.  The active columns (the non-zeros of x when the stretch starts; a zero never comes back before the next regular
// step) are listed once per stretch by a deterministic two-pass ballot compaction that every workgroup performs identically.
constexpr int kRCMAX = 2048;              // active columns a stretch can carry
constexpr int kRG = 32;                   // most workgroups
constexpr int kRNW = 8;                   // waves per workgroup
struct WideRows {
    unsigned long long* flag;             // [kRG][8] hand-over words (launch << 32) | hand-over number, one 64-byte line each
    unsigned long long* flagX;            // [kRG][8] (launch << 32) | (XCD + 1)
    unsigned long long* flagS;            // [kRG][8] (launch << 32) | iteration whose norm share row group r has published
    float* pd;                            // [2][G][kRCMAX] partial dots, by parity of the hand-over number
    double* np;                           // [2][kRG][8] norm shares of the row groups, by parity of the iteration number
    int* err;
    unsigned long long seq;               // launch number
    unsigned long long* stat;             // [16] iterations, stretches, ticks, not-one-XCD launches, redo rounds, phase ticks
    double* hint;                         // [2][2] as WidePersist::hint
    int* lst_idx; float* lst_x;           // [G][kRCMAX] the workgroups' lists of non-zeros of their slice of x (column, value)
    int* lcount;                          // [kRG] their lengths
    float* pa;                            // [2][G][256] partials of A x (workgroup (r, c): its columns, its rows)
    int G, R, C;                          // G = R C workgroups: R row groups of 256 rows x C column groups
    int diag;                             // per-phase timestamps (ADMM_HIP_WIDE_PERSIST_STATS)
};

// halving_sum8 (device_utils.h) for floats on the cross-lane hardware of gfx950 instead of eight LDS permutes + three: the
// distance-32 / 16 steps as v_permlane32_swap / v_permlane16_swap (one swap + one add per pair of values), distance 8 as a DPP
// row rotation, the last three as DPP quad permutes and a half-row mirror.  Lane l ends up with the wave total of value (l >> 3).
// (Other ORDER of the last three additions than halving_sum8: a fixed one, used by wide_rows_persist_kernel only.)
template <int CTRL> __device__ __forceinline__ float dpp_mov_f32(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float halving_sum8_dpp(const float (&v)[8], int lane) {
    float a4[4], a2[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[j]), __float_as_uint(v[j + 4]), false, false);
        a4[j] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a4[j]), __float_as_uint(a4[j + 2]), false, false);
        a2[j] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    const int b = (lane >> 3) & 1;
    const float keep = b ? a2[1] : a2[0], send = b ? a2[0] : a2[1];
    float r = keep + dpp_mov_f32<0x128>(send);              // row_ror:8
    r += dpp_mov_f32<0xB1>(r);                              // quad_perm [1, 0, 3, 2]
    r += dpp_mov_f32<0x4E>(r);                              // quad_perm [2, 3, 0, 1]
    r += dpp_mov_f32<0x141>(r);                             // row_half_mirror
    return r;
}

// The same for eight doubles per lane (the five squared norms of the stopping test): every exchange moves the two 32-bit halves.
__device__ __forceinline__ double dpp_swap_add_f64(double a, double b, bool s32) {
    const unsigned long long ua = (unsigned long long)__double_as_longlong(a), ub = (unsigned long long)__double_as_longlong(b);
    unsigned alo = (unsigned)ua, ahi = (unsigned)(ua >> 32), blo = (unsigned)ub, bhi = (unsigned)(ub >> 32);
    if (s32) {
        const auto r0 = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
        alo = r0[0]; blo = r0[1]; ahi = r1[0]; bhi = r1[1];
    } else {
        const auto r0 = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
        const auto r1 = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
        alo = r0[0]; blo = r0[1]; ahi = r1[0]; bhi = r1[1];
    }
    return __longlong_as_double((long long)(((unsigned long long)ahi << 32) | alo)) + __longlong_as_double((long long)(((unsigned long long)bhi << 32) | blo));
}
template <int CTRL> __device__ __forceinline__ double dpp_mov_f64(double x) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(x);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xF, 0xF, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xF, 0xF, false);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double halving_sum8_dpp(const double (&v)[8], int lane) {
    double a4[4], a2[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) a4[j] = dpp_swap_add_f64(v[j], v[j + 4], true);
#pragma unroll
    for (int j = 0; j < 2; ++j) a2[j] = dpp_swap_add_f64(a4[j], a4[j + 2], false);
    const int b = (lane >> 3) & 1;
    const double keep = b ? a2[1] : a2[0], send = b ? a2[0] : a2[1];
    double r = keep + dpp_mov_f64<0x128>(send);
    r += dpp_mov_f64<0xB1>(r);
    r += dpp_mov_f64<0x4E>(r);
    r += dpp_mov_f64<0x141>(r);
    return r;
}

__device__ __forceinline__ double wide_halving_dpp(const double (&v)[8], int lane) { return halving_sum8_dpp(v, lane); }

__device__ __forceinline__ void wr_store_f32(float* p, float v, bool wt) {
    if (wt) __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
__device__ __forceinline__ float wr_load_f32(const float* p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// Second cut (measured at BASELINE configs[2], 32 workgroups x 64 rows, everything above in place): 10.9 us per iteration -- the
// all-to-all of the partial dots is G^2 nS floats per iteration (4 MB at nS = 1000: every workgroup reads every workgroup's
// partial of every column: ~1 us of L2 requests, twice that beyond 512 columns) and every workgroup does the x-update, the dots
// and A x for ALL columns.  Third cut, this one: a 2-D split.  The G <= 32 workgroups form R row groups x C column groups
// (R = ceil(n / 256), C = 32 / R: 8 x 4 at n = 2000); workgroup (r, c) keeps the 256-row slices of the listed columns
// j = c mod C.  Per iteration
//   (1) x-update of ITS columns from the R partial dots of its column group (x of a column group is replicated R times, not 32),
//   (2) its partial of A x over its columns for its rows                      -> hand-over A inside the row group (C x 256 floats),
//   (3) A x, z, y, norms, the speculative next t of its rows (replicated C times), the partial dots of its columns over its rows
//                                                                             -> hand-over B inside the column group (R x nS / C
//       floats) + the norm shares of the row groups (published by column group 0),
//   (4) the decision, by everybody.
// Two hand-overs per iteration like the round-3 stretch, but their payloads are 1 KB and ~1 KB instead of 8 KB x 32 and the
// column slices never leave the registers.
constexpr int kRRS = 256;                 // rows per row group
constexpr int kRKRES = 24;                // register-resident columns per wave (float4 each: 96 registers)

// CS (column-sharded solver, PEER exchange): the same stretch on this rank's column block.  Two things cross the ranks, both through
// the AUX region of the exchange buffers (peer_device.h), numbered on the device:
//   * once per launch, the AGREEMENT: every rank's count of active columns.  A stretch runs only if every rank can carry its list
//     (and somebody has one); the hint that skips hopeless launches is formed from the largest count, so it is replicated too;
//   * once per iteration, A x = sum over the ranks of the row group's 256 local sums: workgroup (r, 0) writes them into every rank's
//     slot and raises the row group's flag there; every workgroup (r, c) of every rank waits for the nranks flags of ITS row group
//     and adds the nranks shares in rank order -- identical sums on every rank and in every column group, so z, y, the norm shares
//     and the decisions stay replicated bit for bit.  (The re-formed t after a rho change needs no exchange: A x, z, y are replicated.)
template <bool CS>
__global__ void __launch_bounds__(kRNW * 64)
wide_rows_persist_kernel(WideParams q, int cpar, WideRows ps, PeerAux ax_) {
    if ((blockIdx.x & 7) != 0) return;
    const int g = blockIdx.x >> 3, G = ps.G, R = ps.R, C = ps.C;
    if (g >= G) return;
    const int r = g / C, c = g % C;                         // row group, column group
    constexpr int NW = kRNW, T = NW * 64, RS = kRRS, KRES = kRKRES;
    constexpr int KMAX = kRCMAX / NW;                       // list positions per wave (256)
    __shared__ float xa[kRCMAX];                            // x of the listed columns (only this column group's entries are kept up to date)
    __shared__ int cidx[kRCMAX];                            // their column numbers
    __shared__ __attribute__((aligned(16))) float td[RS];   // t / gamma of this workgroup's rows
    __shared__ __attribute__((aligned(16))) float comb[NW][RS];     // the waves' partials of A x (this workgroup's rows, its columns)
    __shared__ int cnt[(262144 / 4 / T + 1) * NW + 64];     // compaction: non-zeros per (pass, wave), then their exclusive prefix
    __shared__ double nsh[4][8];
    __shared__ int loff[kRG + 2];
    __shared__ int s_ok, s_same, s_ns;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    WideCtl in = wide_ctl_uniform(q.ctl[cpar]);
    const double* hin = ps.hint + (size_t)(ps.seq & 1) * 2;
    double* hout = ps.hint + (size_t)((ps.seq + 1) & 1) * 2;
    const double h_nnz = hin[0], h_wait = hin[1];
    const bool leader = g == 0 && tid == 0;
    if (in.done || in.first) { if (leader) { hout[0] = h_nnz; hout[1] = h_wait; } return; }
    if (h_nnz > (double)kRCMAX && h_wait > 0.0) { if (leader) { hout[0] = h_nnz; hout[1] = h_wait - 1.0; } return; }
    double sums[5];
    {
        const WideNormRaw nraw = wide_norms_request(q, lane);
        wide_norms_finish(q, lane, nraw, sums);
    }
    WideDecision dec = wide_decide<true>(q, in, sums, lane);
    WideCtl out = wide_ctl_uniform(dec.out);
    if (out.done || out.type != W_ACT || __builtin_amdgcn_readfirstlane(dec.lam_finished) >= 0) { if (leader) { hout[0] = h_nnz; hout[1] = h_wait; } return; }
    const long long tick0 = wall_clock64();
    // CS: the exchanges of this launch are numbered ebase + 1 (agreement), ebase + 2 (iteration 0), ...; the word is rewritten by the
    // leader after every workgroup of the launch has read it (they all pass the hand-overs below first)
    const unsigned long long ebase = CS ? __hip_atomic_load(ax_.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;

    // ---- the active columns.  Workgroup g lists the non-zeros of ITS slice of x (two-pass ballot compaction: counts per (pass,
    // wave), exclusive prefix, ordered write), publishes count + list, and after the first hand-over of the launch every
    // workgroup concatenates the G lists in workgroup order: the same list everywhere, 1 / G of the scan each.  (First cut: every
    // workgroup scanned all of x -- 60 us per stretch at p = 2 * 10^5, 1.5 us per iteration of a 40-iteration stretch.)
    unsigned long long hs = 0;                              // hand-overs made in this launch
    auto publish = [&](unsigned long long h) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(ps.flag + (size_t)g * 8, (ps.seq << 32) | h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // every wave polls the words it needs itself: lane l < G watches workgroup l if `need(l)`
    auto wait_for = [&](unsigned long long h, int kind) -> bool {      // kind 0: everybody, 1: my row group, 2: my column group + the share publishers (column group 0)
        bool ok = true;
        const bool need = lane < G && (kind == 0 || (kind == 1 && lane / C == r) || (kind == 2 && lane % C == c));
        if (need) {
            const unsigned long long* fw = ps.flag + (size_t)lane * 8;
            const unsigned long long val = (ps.seq << 32) | h;
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(fw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < val) {
                __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > 200000000ll) { ok = false; break; }        // 2 s
            }
        }
        ok = __all(ok) != 0;
        if (!ok && lane == 0) __hip_atomic_store(ps.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return ok;
    };
    const int nq = (q.p + 3) / 4;                           // float4 groups of x (padded with zeros to a multiple of 32)
    const int per = (nq + G - 1) / G;                       // ... per workgroup
    const int q0 = g * per, q1 = min(nq, q0 + per);
    const int npass = (per + T - 1) / T;
    for (int it0 = 0; it0 < npass; it0 += 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e4 = q0 + (it0 + u) * T + tid;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (it0 + u < npass && e4 < q1) v[u] = *reinterpret_cast<const float4*>(q.x + (size_t)e4 * 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int cc = __popcll(__ballot(v[u].x != 0.f)) + __popcll(__ballot(v[u].y != 0.f)) + __popcll(__ballot(v[u].z != 0.f)) + __popcll(__ballot(v[u].w != 0.f));
            if (lane == 0 && it0 + u < npass) cnt[(it0 + u) * NW + wid] = cc;
        }
    }
    __syncthreads();
    if (wid == 0) {                                         // exclusive prefix over the npass * NW counts (one wave, a run per lane)
        const int m = npass * NW, pl = (m + 63) / 64;
        int loc = 0;
        for (int k = 0; k < pl; ++k) { const int i = lane * pl + k; if (i < m) loc += cnt[i]; }
        int inc = loc;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
        int run = inc - loc;
        for (int k = 0; k < pl; ++k) { const int i = lane * pl + k; if (i < m) { const int cc = cnt[i]; cnt[i] = run; run += cc; } }
        if (lane == 63) s_ns = inc;
    }
    __syncthreads();
    const int nloc = s_ns;
    {
        int* li = ps.lst_idx + (size_t)g * kRCMAX;
        float* lx = ps.lst_x + (size_t)g * kRCMAX;
        if (nloc <= kRCMAX) {
            for (int it = 0; it < npass; ++it) {
                int base = cnt[it * NW + wid];
                const int next = it * NW + wid + 1 < npass * NW ? cnt[it * NW + wid + 1] : nloc;
                if (next == base) continue;                 // nothing in this wave's 256 entries (wave uniform): no load
                const int e4 = q0 + it * T + tid;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e4 < q1) v = *reinterpret_cast<const float4*>(q.x + (size_t)e4 * 4);
                const unsigned long long lt = (1ull << lane) - 1ull;
                const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    const unsigned long long m = __ballot(vv[cc] != 0.f);
                    if (vv[cc] != 0.f) {
                        const int pos = base + __popcll(m & lt);
                        __hip_atomic_store(li + pos, e4 * 4 + cc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        wr_store_f32(lx + pos, vv[cc], true);
                    }
                    base += __popcll(m);
                }
            }
        }
        if (tid == 0) __hip_atomic_store(ps.lcount + g, nloc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0) __hip_atomic_store(ps.flagX + (size_t)g * 8, (ps.seq << 32) | (unsigned long long)(wp_xcc_id() + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    hs++;
    publish(hs);
    if (!wait_for(hs, 0)) return;
    if (wid == 0) {                                         // offsets of the G lists
        const int cc = lane < G ? __hip_atomic_load(ps.lcount + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        int inc = cc;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
        if (lane <= G) loff[lane] = lane < G ? inc - cc : 0;
        if (lane == G - 1) { loff[G] = inc; s_ns = inc; }
        const int anybig = __any(cc > kRCMAX);
        if (lane == 0 && anybig) s_ns = kRCMAX + 1;
    }
    __syncthreads();
    const int nS = s_ns;
    int nS_all = nS;                                        // what the go / no-go and the hint are formed from (CS: the largest count of any rank)
    if (CS) {
        const unsigned long long e = ebase + 1;
        if (g == 0 && wid == 0) {
            for (int dst = lane; dst < ax_.nranks; dst += 64)
                peer_store_u64(reinterpret_cast<unsigned long long*>(aux_slot(ax_.remote[dst], ax_, e, ax_.rank) + kAuxRows), (unsigned long long)(unsigned)nS);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (int dst = lane; dst < ax_.nranks; dst += 64)
                peer_store_u64(aux_flag(ax_.remote[dst], ax_, e, ax_.rank, kAuxEntry), e);
        }
        if (!aux_wait(ax_, e, kAuxEntry)) { if (lane == 0) __hip_atomic_store(ps.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
        int mx = 0, any = 0;
        for (int r0 = 0; r0 < ax_.nranks; r0 += 64) {
            const int rr = r0 + lane;
            const int v = rr < ax_.nranks ? (int)peer_load_u64(reinterpret_cast<const unsigned long long*>(aux_slot(ax_.local, ax_, e, rr) + kAuxRows)) : 0;
            mx = max(mx, v); any |= v;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { mx = max(mx, __shfl_xor(mx, d, 64)); any |= __shfl_xor(any, d, 64); }
        nS_all = any == 0 ? 0 : mx;
    }
    if (nS_all > kRCMAX || nS_all == 0) {
        if (leader) { hout[0] = (double)nS_all; hout[1] = 64.0; if (CS) __hip_atomic_store(ax_.seq, ebase + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        return;
    }
    for (int j = tid; j < nS; j += T) {
        int lo = 0, hi = G;                                 // the list j falls into: loff[lo] <= j < loff[lo + 1]
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (loff[mid] <= j) lo = mid; else hi = mid; }
        const int k = j - loff[lo];
        cidx[j] = __hip_atomic_load(ps.lst_idx + (size_t)lo * kRCMAX + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        xa[j] = wr_load_f32(ps.lst_x + (size_t)lo * kRCMAX + k);
    }
    // this column group's part of the list: positions j = c + C jj, jj < nC; wave w takes jj = w + NW k
    const int nC = nS > c ? (nS - c + C - 1) / C : 0;
    auto jpos = [&](int w, int k) -> int { return c + C * (w + NW * k); };      // list position of the wave's k-th column
    // ---- this workgroup's rows in registers (thread tid < RS owns row RS r + tid; replicated in the C workgroups of the row group)
    const int row = r * RS + tid;
    const bool own = tid < RS && row < q.n;
    float ax_r = own ? q.Ax[row] : 0.f, z_r = own ? q.z[row] : 0.f, y_r = own ? q.y[row] : 0.f;
    const float yd_r = own ? q.Y[row] : 0.f;
    const int rl = r * RS + lane * 4;                       // the 4 rows a lane holds of a column slice
    const bool rlok = rl < q.ldx;                           // padding rows up to ldx are zero; beyond that the next column begins
    if (tid < RS) {                                         // first t from the vectors the tail launch left
        const float rho_f = (float)out.rho;
        const float t = own ? (ax_r + z_r) + y_r / rho_f : 0.f;
        td[tid] = t / q.gamma;
    }
    __syncthreads();                                        // list, xa, td complete
    // the wave's first KRES columns stay in registers for the whole stretch
    float4 rc[KRES];
#pragma unroll
    for (int k0 = 0; k0 < KRES; k0 += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) rc[k0 + u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (wid + NW * k0 < nC && rlok) {                   // nothing beyond the list (wave uniform)
            int cc[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) cc[u] = cidx[min(jpos(wid, k0 + u), nS - 1)];        // clamped: every load below is unconditional
#pragma unroll
            for (int u = 0; u < 8; ++u) rc[k0 + u] = *reinterpret_cast<const float4*>(q.X + (size_t)cc[u] * q.ldx + rl);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (wid + NW * (k0 + u) >= nC) rc[k0 + u] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if (!wp_wait_xcc(ps.flagX, ps.seq, ps.err, &s_ok, &s_same, G)) return;
    const bool wt = s_same == 0;
    unsigned long long k_done = 0, redo = 0;
    long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp = wall_clock64();          // diagnostics: ticks per phase, as seen by workgroup 0
#define WR_PHASE(i) if (ps.diag) { const long long tn = wall_clock64(); ph[i] += tn - tp; tp = tn; }

    // x of the wave's resident columns, lane-indexed (lane k holds column k): one LDS round trip for all of them; column k's value
    // is then a readlane with a compile-time lane -- no LDS latency and no branch per column
    auto wave_x = [&]() -> float {
        const int j = jpos(wid, lane);
        return (lane < KRES && wid + NW * lane < nC) ? xa[j] : 0.f;
    };
    auto slice = [&](int j) -> float4 {
        return rlok ? *reinterpret_cast<const float4*>(q.X + (size_t)cidx[j] * q.ldx + rl) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    // Exchange buffers are indexed by the parity of the ITERATION they belong to, not of the hand-over number (A and B alternate: every
    // B would get the same parity -- one buffer -- and a workgroup that ran ahead overwrote dots a slower member of its column group
    // was still reading: found by the 16-process soak, where waves are context-switched, as an x-update 17 yardsticks off in one of
    // 1432 wide cases).  Two buffers suffice although a workgroup only waits for its row / column group: the reader of pd(k) -- a
    // column-group mate -- publishes B(k + 1) after it has read, and the writer of pd(k + 2) has waited for that B(k + 1); the reader
    // of pa(k) -- a row-group mate -- publishes A(k + 1) after it has read, and the writer of pa(k + 2) has waited for that A(k + 1);
    // the norm shares of iteration k are read by everybody before they publish A(k + 1), which (r'', 0) waits for before it
    // publishes B(k + 1), which the share publisher (r, 0) waits for before iteration k + 2.
    // partial dots of this workgroup's rows for its live columns, to pd[it & 1][g][wid * KMAX + k].  Eight columns at a time: the
    // lanes' products, then ONE halving butterfly for the eight sums (lane 8 u ends up with the total of column u).
    auto dots = [&](unsigned long long it) {            // `it`: the iteration whose x-update consumes these dots (buffer = its parity)
        float* dst = ps.pd + ((size_t)(it & 1) * G + g) * kRCMAX + (size_t)wid * KMAX;
        const float4 tv = *reinterpret_cast<const float4*>(td + lane * 4);
        auto dot1 = [&](const float4& cv) -> float {
            float d = cv.x * tv.x;
            d = fmaf(cv.y, tv.y, d); d = fmaf(cv.z, tv.z, d); d = fmaf(cv.w, tv.w, d);
            return d;
        };
        const float xr = wave_x();
#pragma unroll
        for (int k0 = 0; k0 < KRES; k0 += 8) {              // register-resident columns
            if (wid + NW * k0 < nC) {
                float v8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float xv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xr), k0 + u));
                    v8[u] = xv != 0.f ? dot1(rc[k0 + u]) : 0.f;  // (a select: a dead column's slice stays in its registers)
                }
                const float tot = halving_sum8_dpp(v8, lane);
                if ((lane & 7) == 0) wr_store_f32(dst + k0 + (lane >> 3), tot, wt);
            }
        }
        for (int k0 = KRES; wid + NW * k0 < nC; k0 += 8) {  // the others: eight slices in flight (from the L2)
            float4 cv[8];
            float xv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool in_list = wid + NW * (k0 + u) < nC;
                const int j = jpos(wid, k0 + u);
                xv[u] = in_list ? xa[j] : 0.f;
                cv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (xv[u] != 0.f) cv[u] = slice(j);
            }
            float v8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v8[u] = xv[u] != 0.f ? dot1(cv[u]) : 0.f;
            const float tot = halving_sum8_dpp(v8, lane);
            if ((lane & 7) == 0) wr_store_f32(dst + k0 + (lane >> 3), tot, wt);
        }
    };
    // the R partial dots of listed position j (of this column group) in row-group order: all requests in flight together
    auto pd_get = [&](int j, unsigned long long it) -> float {     // eight requests in flight (R = 8 at n = 2000: one round)
        const int jj = j / C;
        const float* src = ps.pd + ((size_t)(it & 1) * G + c) * kRCMAX + (jj % NW) * KMAX + jj / NW;      // workgroup (0, c), then (1, c), ...
        float d = 0.f;
        for (int u0 = 0; u0 < R; u0 += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = wr_load_f32(src + (size_t)min(u0 + u, R - 1) * C * kRCMAX);
#pragma unroll
            for (int u = 0; u < 8; ++u) d = (u0 + u == 0) ? v[0] : (u0 + u < R ? d + v[u] : d);
        }
        return d;
    };
    dots(0);                                                // iteration 0 of the stretch, under `out`: rho known, nothing speculative
    hs++;
    publish(hs);
    bool failed = false;
    const long long tick1 = wall_clock64();
    tp = tick1;
    if (!wait_for(hs, 2)) return;
    // the thread's first column: its partial dots are requested together with the norm shares, BEFORE the decision (one memory
    // round trip for both), and summed right away
    const int j0 = c + C * tid;                             // list position of the thread's first column
    float d0 = 0.f;
    bool have0 = tid < nC && xa[j0] != 0.f;
    if (have0) d0 = pd_get(j0, 0);
    for (;;) {
        WR_PHASE(0)
        // ---- (record the decision being acted on: what the x-update launch writes)
        if (leader && q.trace != nullptr && in.total < q.trace_cap) {
            double* t = q.trace + (size_t)in.total * ADMM_TRACE_FIELDS;
            t[0] = in.lam_idx; t[1] = in.iter; t[2] = in.eps_primal; t[3] = in.eps_dual; t[4] = dec.rp; t[5] = dec.rd;
            t[6] = out.rho; t[7] = out.type; t[8] = dec.code; t[9] = in.rho; t[10] = out.rho; t[11] = in.lam;
        }
        // ---- x-update of this column group's columns from the R partial dots (ADMMLassoWide.h:86-118 / ADMMEnet.h:85-122)
        {
            const double pen_d = (double)out.lam / (out.rho * (double)q.gamma);
            const float penalty = (float)pen_d;
            const float thresh_a = q.enet ? q.alpha * penalty : penalty;
            const float denom_a = q.enet ? (float)(1.0 + (double)penalty * (1.0 - (double)q.alpha)) : 1.f;
            if (have0) xa[j0] = prox_f(xa[j0] - d0, thresh_a, denom_a, q.enet != 0);        // requested before the decision
            for (int jj = tid + T; jj < nC; jj += T) {
                const int j = c + C * jj;
                const float xv = xa[j];
                if (xv != 0.f) xa[j] = prox_f(xv - pd_get(j, k_done), thresh_a, denom_a, q.enet != 0);
            }
        }
        __syncthreads();
        WR_PHASE(1)
        // ---- partial of A x over this workgroup's columns, its rows: a wave adds up x_j X_j over its columns, the waves in order
        {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            const float xr = wave_x();
#pragma unroll
            for (int k0 = 0; k0 < KRES; k0 += 8) {          // x_j = 0 (dead, or beyond the list: slice 0) adds an exact zero: no branch per column
                if (wid + NW * k0 < nC) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const float xv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xr), k0 + u));
                        acc.x = fmaf(xv, rc[k0 + u].x, acc.x); acc.y = fmaf(xv, rc[k0 + u].y, acc.y);
                        acc.z = fmaf(xv, rc[k0 + u].z, acc.z); acc.w = fmaf(xv, rc[k0 + u].w, acc.w);
                    }
                }
            }
            for (int k0 = KRES; wid + NW * k0 < nC; k0 += 8) {
                float4 cv[8];
                float xv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const bool in_list = wid + NW * (k0 + u) < nC;
                    const int j = jpos(wid, k0 + u);
                    xv[u] = in_list ? xa[j] : 0.f;
                    cv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (xv[u] != 0.f) cv[u] = slice(j);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (xv[u] != 0.f) {
                        acc.x = fmaf(xv[u], cv[u].x, acc.x); acc.y = fmaf(xv[u], cv[u].y, acc.y);
                        acc.z = fmaf(xv[u], cv[u].z, acc.z); acc.w = fmaf(xv[u], cv[u].w, acc.w);
                    }
                }
            }
            *reinterpret_cast<float4*>(&comb[wid][lane * 4]) = acc;
        }
        __syncthreads();
        if (tid < RS) {                                     // the workgroup's partial: its waves in order            [hand-over A]
            float a = comb[0][tid];
#pragma unroll
            for (int w = 1; w < NW; ++w) a += comb[w][tid];
            wr_store_f32(ps.pa + ((size_t)(k_done & 1) * G + g) * RS + tid, a, wt);
        }
        hs++;
        publish(hs);
        WR_PHASE(2)
        if (!wait_for(hs, 1)) { failed = true; break; }
        WR_PHASE(3)
        double nacc[5] = {0, 0, 0, 0, 0};
        float ax = 0.f;
        if (tid < RS) {
            const float* src = ps.pa + ((size_t)(k_done & 1) * G + (size_t)r * C) * RS + tid;      // workgroups (r, 0), (r, 1), ...
            for (int u0 = 0; u0 < C; u0 += 8) {             // the C partials in column-group order, eight requests in flight
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = wr_load_f32(src + (size_t)min(u0 + u, C - 1) * RS);
#pragma unroll
                for (int u = 0; u < 8; ++u) ax = (u0 + u == 0) ? v[0] : (u0 + u < C ? ax + v[u] : ax);
            }
        }
        if (CS) {                                           // A x of these rows summed over the ranks                 [exchange over the links]
            const unsigned long long e = ebase + 2 + k_done;
            if (c == 0) {
                if (tid < RS) {
                    const float mine = own ? ax : 0.f;
                    for (int dst = 0; dst < ax_.nranks; ++dst) peer_store_f32(aux_slot(ax_.remote[dst], ax_, e, ax_.rank) + r * RS + tid, mine);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (wid == 0)
                    for (int dst = lane; dst < ax_.nranks; dst += 64) peer_store_u64(aux_flag(ax_.remote[dst], ax_, e, ax_.rank, r), e);
            }
            if (!aux_wait(ax_, e, r)) { if (lane == 0) __hip_atomic_store(ps.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); failed = true; break; }
            if (tid < RS) {
                float a = 0.f;
                for (int u0 = 0; u0 < ax_.nranks; u0 += 8) {   // the ranks' shares in rank order, eight requests in flight
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = peer_load_f32(aux_slot(ax_.local, ax_, e, min(u0 + u, ax_.nranks - 1)) + r * RS + tid);
#pragma unroll
                    for (int u = 0; u < 8; ++u) a = (u0 + u == 0) ? v[0] : (u0 + u < ax_.nranks ? a + v[u] : a);
                }
                ax = a;
            }
        }
        if (tid < RS) {
            if (own) {
#pragma clang fp contract(off)
                const float rho_f = (float)out.rho;
                const float den = (float)(-1.0 - out.rho);
                const float zn = (yd_r + y_r + rho_f * ax) / den;                      // next_z (:156-165)
                const float dz = zn - z_r;
                const float rr = ax + zn;                                              // next_residual (:166-170)
                const float yn = y_r + rho_f * rr;                                     // ADMMBase.h:183
                ax_r = ax; z_r = zn; y_r = yn;
                nacc[0] = (double)rr * rr; nacc[1] = (double)dz * dz; nacc[2] = (double)ax * ax; nacc[3] = (double)zn * zn; nacc[4] = (double)yn * yn;
                // t of the NEXT active-set step, speculating that the decision keeps rho
                const float t = (ax_r + z_r) + y_r / rho_f;
                td[tid] = t / q.gamma;
            } else {
                td[tid] = 0.f;
            }
        }
        if (q.state != nullptr && out.total < q.state_cap) {      // iterate dump: x by row group 0 (zeros elsewhere: the dump is cleared), rows by column group 0
            float* srec = q.state + (size_t)out.total * ((size_t)q.p + 3 * (size_t)q.n);
            if (r == 0) for (int jj = tid; jj < nC; jj += T) srec[cidx[c + C * jj]] = xa[c + C * jj];
            if (own && c == 0) { srec[(size_t)q.p + row] = ax_r; srec[(size_t)q.p + q.n + row] = z_r; srec[(size_t)q.p + 2 * (size_t)q.n + row] = y_r; }
        }
        if (wid < 4) {                                      // norm share of the 256 rows: lanes (one halving butterfly for the five sums), then the four waves in order
            const double v8[8] = {nacc[0], nacc[1], nacc[2], nacc[3], nacc[4], 0.0, 0.0, 0.0};
            const double tot = halving_sum8_dpp(v8, lane);
            if ((lane & 7) == 0 && lane < 40) nsh[wid][lane >> 3] = tot;
        }
        __syncthreads();                                    // td, nsh complete
        k_done++;
        // the norm shares leave AT ONCE, with a word of their own (the iteration number): the other workgroups take the decision
        // from them while this one is still forming its partial dots and while the dots travel -- the decision's ~2 us of shares
        // round trip, square roots and double divisions are off the critical path of hand-over B
        if (c == 0 && wid == 0) {
            if (lane < 5) {
                const double v = ((nsh[0][lane] + nsh[1][lane]) + nsh[2][lane]) + nsh[3][lane];
                wp_store_f64(ps.np + ((size_t)(k_done & 1) * kRG + r) * 8 + lane, v, wt);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(ps.flagS + (size_t)r * 8, (ps.seq << 32) | k_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        dots(k_done);                                       // (k_done already counts this iteration: the NEXT one consumes them)  [hand-over B]
        hs++;
        publish(hs);
        WR_PHASE(4)
        // ---- the decision on the iteration just finished, by everybody from the same numbers
        {
            bool ok = true;
            if (lane < R) {
                const unsigned long long* fw = ps.flagS + (size_t)lane * 8;
                const unsigned long long val = (ps.seq << 32) | k_done;
                const long long t0 = wall_clock64();
                while (__hip_atomic_load(fw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < val) {
                    __builtin_amdgcn_s_sleep(1);
                    if (wall_clock64() - t0 > 200000000ll) { ok = false; break; }
                }
            }
            ok = __all(ok) != 0;
            if (!ok) { if (lane == 0) __hip_atomic_store(ps.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); failed = true; break; }
        }
        WR_PHASE(5)
        in = out;
        {
            const double* sh = ps.np + (size_t)(k_done & 1) * kRG * 8;
#pragma unroll
            for (int m = 0; m < 5; ++m)
                sums[m] = lane < R ? __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(sh + (size_t)lane * 8 + m),
                                                                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0.0;
        }
        dec = wide_decide<true>(q, in, sums, lane);
        out = wide_ctl_uniform(dec.out);
        const bool go_on = !out.done && out.type == W_ACT && __builtin_amdgcn_readfirstlane(dec.lam_finished) < 0;
        WR_PHASE(6)
        if (!go_on) break;                                  // (the speculative dots in flight are dropped)
        if (out.rho != in.rho) {                            // the speculation was wrong: t with the new rho, the partial dots once more
            __syncthreads();                                // (everybody has read td for the speculative dots)
            if (tid < RS) {
                const float rho_f = (float)out.rho;
                const float t = own ? (ax_r + z_r) + y_r / rho_f : 0.f;
                td[tid] = t / q.gamma;
            }
            __syncthreads();
            dots(k_done);
            hs++;
            publish(hs);
            redo++;
        }
        if (!wait_for(hs, 2)) { failed = true; break; }     // the partial dots of the next iteration (speculative ones, or those formed with the new rho)
        have0 = tid < nC && xa[j0] != 0.f;
        if (have0) d0 = pd_get(j0, k_done);
        WR_PHASE(7)
    }
    if (failed) return;
    if (leader) {
        if (CS) __hip_atomic_store(ax_.seq, ebase + 1 + k_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // agreement + one exchange per iteration made
        hout[0] = (double)nS_all; hout[1] = 64.0;
        ps.stat[0] += k_done; ps.stat[1] += 1; ps.stat[2] += (unsigned long long)(wall_clock64() - tick0); ps.stat[3] += wt ? 1 : 0; ps.stat[4] += redo;
        ps.stat[5] += (unsigned long long)(tick1 - tick0);
        for (int i = 0; i < 8; ++i) ps.stat[6 + i] += (unsigned long long)ph[i];
    }
    // ---- leave everything as a tail launch would have: x, A x, z, y, the norm partials, the control block
    if (r == 0) for (int jj = tid; jj < nC; jj += T) q.x[cidx[c + C * jj]] = xa[c + C * jj];
    if (own && c == 0) { q.Ax[row] = ax_r; q.z[row] = z_r; q.y[row] = y_r; }
    if (g == 0) {
        // norm partials: the share of row group r in row r (exactly the lanes' inputs of the decision above, so that the next launch,
        // which repeats it from P, adds the same numbers in the same order), zero in every other row the next decision adds up
        const double* sh = ps.np + (size_t)(k_done & 1) * kRG * 8;
        const int rows = q.nwg_tail > kWideNormRows ? q.nwg_tail : kWideNormRows;
        for (int idx = tid; idx < rows * 8; idx += T) {
            const int rr = idx >> 3, m = idx & 7;
            double v = 0.0;
            if (rr < R && m < 5)
                v = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(sh + (size_t)rr * 8 + m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            q.P[idx] = v;
        }
        if (tid == 0) q.ctl[cpar] = in;                     // the state the breaking decision was taken FROM: the next launch repeats it
    }
}

// Result read-back (round 4): the coefficient snapshots are (p x nlambda) floats of mostly zeros -- 80 MB at BASELINE configs[2], whose
// dense copy, dense host buffers and dense recovery loop (a division and a dependent float add per entry) cost 60 ms of a 0.32 s fit.
// The non-zeros of every column are listed in ascending order on the device (count, then ordered ballot compaction), only the lists
// cross PCIe, and the host scatters the recovered values into the caller's cleared buffer.
__global__ void __launch_bounds__(256) wide_beta_count_kernel(const float* __restrict__ beta, int p, int* __restrict__ cnt) {
    __shared__ int sh[4];
    const float* col = beta + (size_t)blockIdx.x * p;
    int c = 0;
    for (int j = threadIdx.x; j < p; j += 256) c += col[j] != 0.f;
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) cnt[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ void __launch_bounds__(256) wide_beta_compact_kernel(const float* __restrict__ beta, int p, const long long* __restrict__ off,
                                                                int* __restrict__ idx, float* __restrict__ val) {
    __shared__ int sh[4];
    const float* col = beta + (size_t)blockIdx.x * p;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    long long base = off[blockIdx.x];
    for (int j0 = 0; j0 < p; j0 += 256) {
        const int j = j0 + threadIdx.x;
        const float v = j < p ? col[j] : 0.f;
        const bool nz = v != 0.f;
        const unsigned long long m = __ballot(nz);
        __syncthreads();                                            // (the previous round's readers of sh are done)
        if (lane == 0) sh[wid] = __popcll(m);
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wid; ++w) woff += sh[w];
        if (nz) { const long long pos = base + woff + __popcll(m & ((1ull << lane) - 1ull)); idx[pos] = j; val[pos] = v; }
        base += sh[0] + sh[1] + sh[2] + sh[3];
    }
}

__global__ void wide_init_kernel(WideParams q, double rho, float lam0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < q.p) q.x[i] = 0.f;
    if (i < q.n) { q.Ax[i] = 0.f; q.z[i] = 0.f; q.y[i] = 0.f; }
    if (i < q.nwg_tail * 8) q.P[i] = 0.0;
    if (i == 0) {
        WideCtl c;
        c.rho = rho; c.eps_primal = 0; c.eps_dual = 0; c.lam = lam0; c.type = W_REG;
        c.iter = 0; c.counter = 0; c.lam_idx = 0; c.done = 0; c.first = 1; c.total = 0; c.pad1 = c.pad2 = 0;
        q.ctl[0] = c; q.ctl[1] = c;
        *q.done = 0;
    }
}

// Setup of the regular steps' screen (wide_x_kernel): column j of X rounded to fp16 (one wave per column) and its bound
//   s_j >= ||X_j - Xh_j||_2 + 2 gamma_n max(||X_j||_2, ||Xh_j||_2),   gamma_n = n u / (1 - n u) <= 1.001 n 2^-24  (n <= 8192),
// sums in double, the result rounded up to float.  A rounding that is not finite (|X_ij| > 65504) is stored as zero -- the whole
// entry then counts as rounding error -- and a column holding a NaN / Inf gets s_j = +Inf: it always takes the exact path.
__global__ void __launch_bounds__(256)
wide_screen_prep_kernel(const float* __restrict__ X, long long ldx, int n, int p, unsigned short* __restrict__ Xh, long long ldh,
                        float* __restrict__ s, int NW, int S, double loosen) {
    const int lane = threadIdx.x & 63;
    const long long j = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= p) return;
    const float* col = X + (size_t)j * ldx;
    unsigned short* hc = Xh + (size_t)j * ldh;
    double e2 = 0.0, x2 = 0.0, h2 = 0.0;
    for (long long r = (long long)lane * 8; r < ldh; r += 512) {       // ldh <= ldx, both multiples of 8: whole 32-byte pieces
        const float4 a0 = *reinterpret_cast<const float4*>(col + r), a1 = *reinterpret_cast<const float4*>(col + r + 4);
        const float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
        half8_t hv;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = r + e < n ? v[e] : 0.f;                    // rows [n, ldx) may hold anything
            _Float16 hh = (_Float16)x;
            float back = (float)hh;
            if (!(fabsf(back) <= 65504.f)) { hh = (_Float16)0.f; back = 0.f; }
            hv[e] = hh;
            const double dx = (double)x, db = (double)back;
            e2 = fma(dx - db, dx - db, e2); x2 = fma(dx, dx, x2); h2 = fma(db, db, h2);
        }
        *reinterpret_cast<uint4*>(hc + r) = __builtin_bit_cast(uint4, hv);
    }
    e2 = wave_sum(e2); x2 = wave_sum(x2); h2 = wave_sum(h2);
    if (lane == 0) {
        const double gn = 1.001 * (double)n * 5.9604644775390625e-8;   // gamma_n
        const double sd = (sqrt(e2) + 2.0 * gn * sqrt(fmax(x2, h2))) * (1.0 + 1e-6) * loosen;
        float sv = __double2float_ru(sd);
        if (!(sv < __builtin_huge_valf())) sv = __builtin_huge_valf();
        s[(size_t)(j % NW) * S + (size_t)(j / NW)] = sv;
    }
}

// The same for a linear 8-bit code: q_ij = round(X_ij / scale_j) in [-127, 127], scale_j = max_i |X_ij| / 127; the copy stands for
// scale_j q_j in the bound (||X_j - scale_j q_j||_2 measured here, in double).  The screen's product is fl32(q_j't) times scale_j in double.
__global__ void __launch_bounds__(256)
wide_screen_prep8_kernel(const float* __restrict__ X, long long ldx, int n, int p, signed char* __restrict__ Xq, long long ldq,
                         float* __restrict__ s, float* __restrict__ scale, int NW, int S, double loosen) {
    const int lane = threadIdx.x & 63;
    const long long j = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= p) return;
    const float* col = X + (size_t)j * ldx;
    signed char* qc = Xq + (size_t)j * ldq;
    float mx = 0.f;
    bool bad = false;
    for (long long r = (long long)lane * 4; r < n; r += 256) {
        const float4 a = *reinterpret_cast<const float4*>(col + r);
        const float v[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) if (r + e < n) { const float av = fabsf(v[e]); if (!(av <= 3.0e38f)) bad = true; mx = fmaxf(mx, av); }
    }
    mx = wave_max(mx);
    bad = __any(bad) != 0;
    const float sc = bad ? 0.f : mx / 127.f;
    double e2 = 0.0, x2 = 0.0, h2 = 0.0;
    for (long long r = (long long)lane * 16; r < ldq; r += 1024) {                       // the column again (from the cache)
        unsigned wq[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float x = r + e < n ? col[r + e] : 0.f;
            int qi = 0;
            if (sc > 0.f) { const float t = rintf(x / sc); qi = (int)fminf(fmaxf(t, -127.f), 127.f); }
            wq[e >> 2] |= ((unsigned)qi & 0xffu) << (8 * (e & 3));
            const double dx = (double)x, db = (double)sc * (double)qi;
            e2 = fma(dx - db, dx - db, e2); x2 = fma(dx, dx, x2); h2 = fma(db, db, h2);
        }
        *reinterpret_cast<uint4*>(qc + r) = make_uint4(wq[0], wq[1], wq[2], wq[3]);
    }
    e2 = wave_sum(e2); x2 = wave_sum(x2); h2 = wave_sum(h2);
    if (lane == 0) {
        const double gn = 1.001 * (double)n * 5.9604644775390625e-8;
        const double sd = (sqrt(e2) + 2.0 * gn * sqrt(fmax(x2, h2))) * (1.0 + 1e-6) * loosen;
        float sv = __double2float_ru(sd);
        if (bad || !(sv < __builtin_huge_valf())) sv = __builtin_huge_valf();
        const size_t pos = (size_t)(j % NW) * S + (size_t)(j / NW);
        s[pos] = sv; scale[pos] = sc;
    }
}

// Which copy, if any: the bound of column j costs exact steps in proportion to its size in units of the spread of X_j't over random t,
// r_j = s_j sqrt(n) / ||X_j||_2 (Gaussian-looking columns: 0.02 for fp16, 0.0082 sqrt(n) for the 8-bit code; a column with outliers
// wastes the 8-bit code's range).  Per workgroup (four columns): the sum of min(r_j, 4) for the 8-bit code; the host adds them up in order.
__global__ void __launch_bounds__(256)
wide_screen_rate8_kernel(const float* __restrict__ X, long long ldx, int n, int p, double* __restrict__ part) {
    __shared__ double sh[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const long long j = (long long)blockIdx.x * 4 + wid;
    double rj = 0.0;
    if (j < p) {
        const float* col = X + (size_t)j * ldx;
        float mx = 0.f;
        for (long long r = lane; r < n; r += 64) mx = fmaxf(mx, fabsf(col[r]));
        mx = wave_max(mx);
        const float sc = mx / 127.f;
        double e2 = 0.0, x2 = 0.0;
        for (long long r = lane; r < n; r += 64) {
            const float x = col[r];
            const double db = sc > 0.f ? (double)sc * (double)fminf(fmaxf(rintf(x / sc), -127.f), 127.f) : 0.0;
            e2 = fma((double)x - db, (double)x - db, e2); x2 = fma((double)x, (double)x, x2);
        }
        e2 = wave_sum(e2); x2 = wave_sum(x2);
        const double gn = 1.001 * (double)n * 5.9604644775390625e-8;
        rj = x2 > 0.0 ? (sqrt(e2) + 4.0 * gn * sqrt(x2)) * sqrt((double)n) / sqrt(x2) : 0.0;
        if (!(rj <= 4.0)) rj = 4.0;
    }
    if (lane == 0) sh[wid] = rj;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

struct WidePlan final : LassoPlan {
    DeviceData<float> d;
    LassoProblem pb;
    hipStream_t st;
    admm_stats setup_stats{};
    int n = 0, p = 0, nlam = 0, nwg_tail = 0, nwg_x = 0, fuse_rt = 0;
    bool t_global = false;               // n too large for the LDS: t through global memory (wide_t_kernel)
    bool cshard = false;                 // columns spread over the ranks: per iteration one all-reduce of Ax (n floats)
    bool peer_fused = false;             // ... done by the solver's own kernels over the PEER exchange (no launches of the exchange layer)
    bool peer_one = false;               // ... producer and consumer in one launch (wide_tail_kernel<2>)
    CommInfo ci;
    long long p_total = 0, col_offset = 0;
    DevBuf<float> axl;                   // [ldn] this rank's share of Ax, all-reduced in place
    size_t lds_x = 0;
    long long ldn = 0;
    float sprad = 0.f, lambda0 = 0.f;
    double rho0 = 0;
    std::vector<double> lam_user;
    std::vector<float> lam_int;
    DevBuf<float> x, Ax, z, y, axpart, tbuf, beta, dlam;
    DevBuf<unsigned char> Xh;            // the regular steps' screen: X rounded to fp16 or coded in 8 bits ...
    DevBuf<float> scr_s, scr_scale;      // ... the per-column bounds, the 8-bit code's scales (wide_screen_prep_kernel / wide_screen_prep8_kernel)
    int screen_fmt = 0;
    DevBuf<unsigned long long> scr_stat;
    bool screened = false;
    DevBuf<int> niter, done;
    DevBuf<double> P;
    DevBuf<WideCtl> ctl;
    PinnedFlag hflag;
#ifdef ADMM_HIP_PROBE
    DevBuf<long long> probe;
#endif
    WideParams q{};
    DevBuf<double> trace;
    long long trace_cap = 0, trace_n = 0;

    void enable_trace(long long cap) override {
        trace.alloc((size_t)cap * ADMM_TRACE_FIELDS);
        trace_cap = cap; trace_n = 0;
        q.trace = trace.get(); q.trace_cap = cap;
    }
    long long read_trace(double* out, long long cap) override {
        const long long nrec = std::min(std::min(trace_n, trace_cap), cap);
        if (nrec > 0) read_back(out, trace.get(), (size_t)nrec * ADMM_TRACE_FIELDS * sizeof(double), st);
        return nrec;
    }

    // iterate dump (test facility): record s = x | A x | z | y (p + 3 n floats; column-sharded: x of this rank's p columns) the decision of trace record s judged
    DevBuf<float> state;
    long long state_cap = 0;
    void enable_state(long long cap) override {
        const size_t rec = (size_t)p + 3 * (size_t)n;
        state.alloc((size_t)cap * rec);
        ADMM_HIP_CHECK(hipMemsetAsync(state.get(), 0, (size_t)cap * rec * sizeof(float), st));
        state_cap = cap;
        q.state = state.get(); q.state_cap = cap;
    }
    long long read_state(float* out, long long cap, long long* rec_floats) override {
        const size_t rec = (size_t)p + 3 * (size_t)n;
        if (rec_floats) *rec_floats = (long long)rec;
        if (!out) return std::min(trace_n, state_cap);                             // size query
        const long long nrec = std::min(std::min(trace_n, state_cap), cap);
        if (nrec > 0) read_back(out, state.get(), (size_t)nrec * rec * sizeof(float), st);
        return nrec;
    }
    // the standardised data as this solver holds them (test hook admm_hip_lasso_plan_data_read)
    void read_data(float* x_out, long long ld, float* y_out) override {
        comm_stream_sync(st);
        if (x_out) ADMM_HIP_CHECK(hipMemcpy2D(x_out, (size_t)ld * sizeof(float), d.X.get(), (size_t)d.ldx * sizeof(float), (size_t)n * sizeof(float), (size_t)p, hipMemcpyDeviceToHost));
        if (y_out) ADMM_HIP_CHECK(hipMemcpy(y_out, d.Y.get(), (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    }

    WidePlan(DeviceData<float>&& data, const LassoProblem& prob, hipStream_t stream) : d(std::move(data)), pb(prob), st(stream) {
        n = d.n; p = d.p;
        // Column-sharded mode (the north_star's "column-block ... all-reduce of X_i beta_i"; not in the compiled reference,
        // whose only column-block code is the dead TODO/PADMMBP.h:137-156): the SAME linearised ADMM as ADMMLassoWide, with
        // rank i holding the columns X_i and x_i.  Everything of length p is local (X_i't, the prox, the active set, the
        // column moments of DataStd); everything of length n is replicated (z, y, t, the decisions); the only exchange
        // per iteration is the sum over ranks of A x = sum_i X_i x_i (n floats).  Setup: lambda_0 = max over ranks,
        // X X' = sum_i X_i X_i' (one all-reduce), the Lanczos value replicated.
        cshard = pb.p_total > 0;
        ci = cshard ? comm_info() : CommInfo();
        peer_fused = cshard && ci.backend == COMM_PEER;
        if (const char* e = option("PEER_FUSED")) { if (std::string(e) == "0") peer_fused = false; }
        p_total = cshard ? pb.p_total : p; col_offset = cshard ? pb.col_offset : 0;
        admm_stats& S = setup_stats;
        S.branch = 1; S.t_h2d = d.t_h2d; S.t_standardize = d.t_std;
        const long long ldp = round_up(p, 32);
        ldn = round_up(n, 256);

        // lambda_0 (ADMMLassoWide.h:197; ADMMEnet.h:152)
        {
            DevBuf<float> XY(ldp); XY.zero(st);
            gemv_t_simple<float>(d.X.get(), d.ldx, n, p, d.Y.get(), XY.get(), st);
            lambda0 = device_absmax<float>(XY.get(), p, st);
            if (cshard) {                                            // max over ranks through the sum all-reduce of a one-hot vector
                std::vector<float> h(ci.nranks, 0.f);
                h[ci.rank] = lambda0;
                DevBuf<float> dm(round_up(ci.nranks, 4)); dm.zero(st);
                ADMM_HIP_CHECK(hipMemcpyAsync(dm.get(), h.data(), ci.nranks * sizeof(float), hipMemcpyHostToDevice, st));
                allreduce_sum_f32(dm.get(), (size_t)ci.nranks, st);
                ADMM_HIP_CHECK(hipMemcpyAsync(h.data(), dm.get(), ci.nranks * sizeof(float), hipMemcpyDeviceToHost, st));
                comm_stream_sync(st);
                comm_check();
                for (float v : h) lambda0 = std::max(lambda0, v);
            }
        }
        // spectral radius estimate from XX' (ADMMLassoWide.h:200-207).  Single process: Gram-FREE -- the Lanczos products are
        // formed as X (X' v) on the stored X (GramFreeWideOp, prep.h), the n x n Gram is never built (nothing else needs it).
        // ADMM_HIP_WIDE_SPRAD=gram takes the Gram-based product (the column-sharded solver always does: its ranks sum one
        // n x n matrix once instead of exchanging a p-vector per product).
        double t0 = now_s();
        bool gram_free = !cshard;
        if (const char* e = option("WIDE_SPRAD")) gram_free = !cshard && std::string(e) != "gram";
        if (gram_free) {
            GramFreeWideOp op(d.X.get(), d.ldx, n, p, st);
            comm_stream_sync(st);
            S.t_gram = 0.0;
            int nmatop = 0;
            sprad = lanczos_largest_f32([&](const float* v, float* w) { op(v, w); }, n, &nmatop);
            S.eig_est = sprad; S.t_eigs = now_s() - t0;
        } else {
            const long long ldg = round_up(n, 32);
            DevBuf<float> G((size_t)ldg * n); G.zero(st);
            gram_full<float>(d.X.get(), d.ldx, n, p, false, G.get(), ldg, st);
            if (cshard) allreduce_sum_f32(G.get(), (size_t)ldg * n, st);       // X X' = sum over the ranks' column blocks
            comm_stream_sync(st);
            comm_check();
            S.t_gram = now_s() - t0; t0 = now_s();
            SymMatVec<float> op(G.get(), ldg, n, st);
            int nmatop = 0;
            sprad = lanczos_largest_f32([&](const float* v, float* w) { op(v, w); }, n, &nmatop);
            S.eig_est = sprad; S.t_eigs = now_s() - t0;
        }
        if (pb.enet) lambda0 = (float)((double)lambda0 / ((double)(float)pb.alpha + 0.0001));

        lam_user = make_lambda_grid(pb, lambda0, n, (double)d.scaleY);
        nlam = (int)lam_user.size();
        lam_int.resize(nlam);
        for (int i = 0; i < nlam; ++i) lam_int[i] = (float)(lam_user[i] * n / (double)d.scaleY);
        rho0 = pb.opts.rho;
        if (rho0 <= 0) rho0 = std::pow((double)lam_int[0] / (double)sprad, 1.0 / 3);       // :227-228
        S.rho = rho0;

        nwg_tail = (n + kWtElems - 1) / kWtElems;                    // 32 elements per workgroup (8 lanes each)
        if (peer_fused) {
            int occ = 0;
            ADMM_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(wide_tail_kernel<2>), kWideThreads, 0));
            peer_one = (long long)nwg_tail * 2 <= resident_workgroups(occ);
            if (const char* e = option("PEER_FUSED")) { if (std::string(e) == "2") peer_one = false; }
        }
        x.alloc(ldp); x.zero(st);
        for (DevBuf<float>* b : {&Ax, &z, &y}) { b->alloc(std::max<long long>(ldn, 8192)); b->zero(st); }   // the fused x-update reads up to 32 x 256 entries unconditionally
        // ADMM_HIP_WIDE_FUSE=0: always three launches per iteration
        fuse_rt = n <= 1024 ? 4 : (n <= 2048 ? 8 : (n <= 4096 ? 16 : (n <= 6144 ? 24 : (n <= 8192 ? 32 : 0))));
        if (const char* e = option("WIDE_FUSE")) if (std::string(e) == "0") fuse_rt = 0;
        lds_x = std::max((size_t)((n + 255) / 256 * 256) * 2 * sizeof(float), (size_t)std::min(std::max(fuse_rt, 4), 8) * kWideThreads * sizeof(float4));
        // The x-update stages t and t / gamma (2 n floats) in dynamic LDS: up to 64 KB by default, up to the device's
        // opt-in limit (160 KB on gfx950) after raising the kernel's attribute; beyond that (n > ~20 000) t goes through
        // global memory instead (one more small launch per iteration).  ADMM_HIP_WIDE_TGLOBAL=1 forces that mode.
        if (lds_x > device_info().lds_per_block) {
            if (lds_x <= device_info().lds_optin) {
                ADMM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(wide_x_kernel<0, false>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_x));
            } else {
                t_global = true;
            }
        }
        if (const char* e = option("WIDE_TGLOBAL")) if (std::string(e) == "1") t_global = true;
        if (t_global) { fuse_rt = 0; lds_x = 0; tbuf.alloc(ldn); tbuf.zero(st); }
        // grid of the x-update launch: exactly ONE resident round of workgroups (a regular step streams all of X; a
        // partial second round runs at a fraction of the occupancy: 342 us instead of 280 us at C3 with 4 per CU
        // when 3 fit).  ADMM_HIP_WIDE_WGX overrides the workgroups per CU.
        int wgx = 0;
        {
            const void* fn = fuse_rt == 4 ? reinterpret_cast<const void*>(wide_x_kernel<4>)
                           : fuse_rt == 8 ? reinterpret_cast<const void*>(wide_x_kernel<8>)
                           : fuse_rt == 16 ? reinterpret_cast<const void*>(wide_x_kernel<16>)
                           : fuse_rt == 24 ? reinterpret_cast<const void*>(wide_x_kernel<24>)
                           : fuse_rt == 32 ? reinterpret_cast<const void*>(wide_x_kernel<32>)
                           : t_global ? reinterpret_cast<const void*>(wide_x_kernel<0, true>)
                                      : reinterpret_cast<const void*>(wide_x_kernel<0, false>);
            ADMM_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&wgx, fn, kWideThreads, lds_x));
            wgx = std::max(1, std::min(wgx, 4));
        }
        if (const char* e = option("WIDE_WGX")) wgx = std::max(1, std::atoi(e));
        nwg_x = std::max(wgx * device_info().num_cu, kActWG);
        axpart.alloc((size_t)(fuse_rt ? nwg_x : kAxWG) * ldn); axpart.zero(st);
        beta.alloc((size_t)nlam * p); niter.alloc(nlam); done.alloc(1); dlam.alloc(nlam);
        P.alloc((size_t)std::max(nwg_tail, kWideNormRows) * 8); P.zero(st); ctl.alloc(2);       // rows beyond nwg_tail stay zero (wide_load_norms)
        ADMM_HIP_CHECK(hipMemcpyAsync(dlam.get(), lam_int.data(), nlam * sizeof(float), hipMemcpyHostToDevice, st));

        q.n = n; q.p = p; q.maxit = pb.opts.maxit; q.nlam = nlam; q.enet = pb.enet ? 1 : 0; q.nwg_tail = nwg_tail;
        q.ldx = d.ldx; q.X = d.X.get(); q.Y = d.Y.get();
        q.gamma = sprad; q.lambda0 = lambda0; q.alpha = (float)pb.alpha;
        q.eps_abs = pb.opts.eps_abs; q.eps_rel = pb.opts.eps_rel;
        q.sqrt_n = std::sqrt((double)n); q.sqrt_p = std::sqrt((double)p_total); q.sqrt_gamma = (double)std::sqrt(sprad);
        if (cshard) { axl.alloc(ldn); axl.zero(st); }
        q.ax_given = cshard ? axl.get() : nullptr;
        q.lambdas = dlam.get(); q.x = x.get(); q.Ax = Ax.get(); q.z = z.get(); q.y = y.get();
        q.axpart = axpart.get(); q.tbuf = tbuf.get(); q.ldn = ldn; q.fused = fuse_rt ? 1 : 0; q.nwg_x = nwg_x;
        q.x_nt = gemv_stream_nt((size_t)d.ldx * (size_t)p * sizeof(float)) ? 1 : 0;
        q.ctl = ctl.get(); q.P = P.get(); q.beta = beta.get(); q.niter = niter.get(); q.done = done.get();
        q.done_host = hflag.p;
#ifdef ADMM_HIP_PROBE
        probe.alloc((size_t)4096 * 4 * 8); probe.zero(st);
        q.probe = probe.get();
#endif
        setup_screen();
        setup_persist_rows();
        comm_stream_sync(st);
    }

    // Safe screening of the regular steps (wide_x_kernel): worth its copy when X is a stream from HBM at all (beyond the Infinity Cache
    // the regular step is bandwidth, below it launch latency).  WIDE_SCREEN = 0 never, 1 / 16 the fp16 copy, 8 the 8-bit code, auto = the default's choice at any size (tests);
    // default: the 8-bit code when its bounds are tight enough on these columns (wide_screen_rate8_kernel: mean r_j <= 0.6), else
    // fp16.  When the copy does not fit the device the solver simply runs unscreened.
    void setup_screen() {
        const size_t xbytes = (size_t)d.ldx * (size_t)p * sizeof(float);
        if (fuse_rt == 0) return;
        int fmt = xbytes >= ((size_t)256 << 20) ? -1 : 0;                // -1: choose
        if (const char* e = option("WIDE_SCREEN")) { const int v = std::atoi(e); fmt = std::string(e) == "auto" ? -1 : (v == 0 ? 0 : (v == 8 ? 8 : 16)); }
        if (fmt == 0 || d.ldx % 16 != 0) return;
        if (fmt < 0) {
            const int nb = (p + 3) / 4;
            DevBuf<double> part(nb);
            hipLaunchKernelGGL(wide_screen_rate8_kernel, dim3((unsigned)nb), dim3(256), 0, st, d.X.get(), d.ldx, n, p, part.get());
            std::vector<double> hp(nb);
            read_back(hp.data(), part.get(), (size_t)nb * sizeof(double), st);
            double sum = 0.0;
            for (double v : hp) sum += v;
            fmt = sum / (double)p <= 0.6 ? 8 : 16;
        }
        const int epl = fmt == 8 ? 16 : 8;
        const long long ldh = round_up(n, epl);
        if (ldh > d.ldx) return;
        const int NW = nwg_x * (kWideThreads / 64);
        const int S = (int)round_up((p + NW - 1) / NW, 64);
        try {
            Xh.alloc((size_t)p * (size_t)ldh * (size_t)(fmt / 8));
            scr_s.alloc((size_t)NW * (size_t)S);
            if (fmt == 8) scr_scale.alloc((size_t)NW * (size_t)S);
        } catch (const Error&) {
            Xh.release(); scr_s.release(); scr_scale.release();
            return;
        }
        scr_s.zero(st);
        double loosen = 1.0;                                          // WIDE_SCREEN_SLACK >= 1: bounds that much looser (what a coarser copy would cost in exact steps; diagnosis)
        if (const char* e = option("WIDE_SCREEN_SLACK")) loosen = std::max(1.0, std::atof(e));
        if (fmt == 8) {
            scr_scale.zero(st);
            hipLaunchKernelGGL(wide_screen_prep8_kernel, dim3((unsigned)((p + 3) / 4)), dim3(256), 0, st, d.X.get(), d.ldx, n, p,
                               reinterpret_cast<signed char*>(Xh.get()), ldh, scr_s.get(), scr_scale.get(), NW, S, loosen);
        } else {
            hipLaunchKernelGGL(wide_screen_prep_kernel, dim3((unsigned)((p + 3) / 4)), dim3(256), 0, st, d.X.get(), d.ldx, n, p,
                               reinterpret_cast<unsigned short*>(Xh.get()), ldh, scr_s.get(), NW, S, loosen);
        }
        q.Xh = Xh.get(); q.ldh = ldh; q.scr_fmt = fmt; q.scr_s = scr_s.get(); q.scr_S = S; q.scr_scale = scr_scale.get();
        if (option("WIDE_SCREEN_STATS")) { scr_stat.alloc(2); scr_stat.zero(st); q.scr_stat = scr_stat.get(); }
        screened = true; screen_fmt = fmt;
    }

    // persistent active-set stretch (wide_rows_persist_kernel)
    bool persist_rows = false;
    int rows_G = 0, rows_R = 1, rows_C = 1;
    DevBuf<unsigned long long> rflags, rstat;
    DevBuf<float> rpd, rlx, rpa;
    DevBuf<double> rnp, rhint;
    DevBuf<int> rerr, rli, rlc;
    unsigned long long rseq = 0;
    void setup_persist_rows() {
        // ADMM_HIP_WIDE_PERSIST=0: two launches per iteration only
        // column-sharded: only where the ranks exchange through peer-mapped memory (the stretch exchanges INSIDE its launch, which RCCL
        // and the host shared-memory back-end cannot do); ADMM_HIP_WIDE_PERSIST_COLS=0 switches it off there alone
        persist_rows = (!cshard || (peer_fused && ci.nranks <= 64)) && n <= kRRS * kRG && (long long)p <= 262144;
        static_assert(kRRS * kRG <= kAuxRows && kRG <= kAuxGroups, "the AUX region carries one float per row, one flag per row group");
        if (cshard) if (const char* e = option("WIDE_PERSIST_COLS")) if (std::string(e) == "0") persist_rows = false;
        if (const char* e = option("WIDE_PERSIST")) if (std::string(e) == "0") persist_rows = false;
        if (!persist_rows) return;
        rows_R = (n + kRRS - 1) / kRRS;                                     // row groups of 256 rows
        rows_C = std::max(1, kRG / rows_R);                                 // column groups: R C <= 32 workgroups (one XCD)
        if (const char* e = option("WIDE_ROWS_C")) { const int c = std::atoi(e); if (c >= 1 && c * rows_R <= kRG) rows_C = c; }
        rows_G = rows_R * rows_C;
        rflags.alloc((size_t)3 * kRG * 8); rflags.zero(st);
        rpd.alloc((size_t)2 * rows_G * kRCMAX); rpd.zero(st);
        rpa.alloc((size_t)2 * rows_G * kRRS); rpa.zero(st);
        rnp.alloc((size_t)2 * kRG * 8); rnp.zero(st);
        rhint.alloc(4); rhint.zero(st);
        rerr.alloc(1); rerr.zero(st);
        rstat.alloc(16); rstat.zero(st);
        rli.alloc((size_t)rows_G * kRCMAX); rlx.alloc((size_t)rows_G * kRCMAX); rlc.alloc(kRG); rlc.zero(st);
    }
    void launch_persist_rows(int cpar) {
        WideRows ps;
        ps.flag = rflags.get(); ps.flagX = rflags.get() + (size_t)kRG * 8; ps.flagS = rflags.get() + (size_t)2 * kRG * 8;
        ps.pd = rpd.get(); ps.np = rnp.get(); ps.err = rerr.get(); ps.seq = ++rseq; ps.stat = rstat.get(); ps.hint = rhint.get();
        ps.G = rows_G; ps.R = rows_R; ps.C = rows_C; ps.pa = rpa.get(); ps.diag = option("WIDE_PERSIST_STATS") ? 1 : 0;
        ps.lst_idx = rli.get(); ps.lst_x = rlx.get(); ps.lcount = rlc.get();
        if (cshard) hipLaunchKernelGGL(wide_rows_persist_kernel<true>, dim3(8 * rows_G), dim3(kRNW * 64), 0, st, q, cpar, ps, comm_peer_aux());
        else hipLaunchKernelGGL(wide_rows_persist_kernel<false>, dim3(8 * rows_G), dim3(kRNW * 64), 0, st, q, cpar, ps, PeerAux{});
    }

    void run(LassoResult& res) override {
        admm_stats S = setup_stats;
        res.lambda = lam_user;
        beta.zero(st); niter.zero(st);
        *hflag.p = 0;
        if (persist_rows) { rhint.zero(st); rerr.zero(st); }
        // records written inside a persistent stretch store x for the listed columns only ("zeros elsewhere"): a second run() on the
        // same plan must not see the previous run's entries
        if (q.state != nullptr) ADMM_HIP_CHECK(hipMemsetAsync(state.get(), 0, state.n * sizeof(float), st));
        const int init_n = std::max(std::max(n, p), nwg_tail * 8);
        hipLaunchKernelGGL(wide_init_kernel, dim3((init_n + 255) / 256), dim3(256), 0, st, q, rho0, lam_int[0]);
        const int batch = pb.batch_iters > 0 ? (pb.batch_iters + 1) / 2 * 2 : 16;
        LoopTimes lt = run_until_done(st, done.get(), batch, (long long)nlam * ((long long)pb.opts.maxit + 2) + 4, [&](long long g) {
            const int par = (int)(g & 1);
            switch (fuse_rt) {
                case 4: hipLaunchKernelGGL(wide_x_kernel<4>, dim3(nwg_x), dim3(kWideThreads), lds_x, st, q, par); break;
                case 8: hipLaunchKernelGGL(wide_x_kernel<8>, dim3(nwg_x), dim3(kWideThreads), lds_x, st, q, par); break;
                case 16: hipLaunchKernelGGL(wide_x_kernel<16>, dim3(nwg_x), dim3(kWideThreads), lds_x, st, q, par); break;
                case 24: hipLaunchKernelGGL(wide_x_kernel<24>, dim3(nwg_x), dim3(kWideThreads), lds_x, st, q, par); break;
                case 32: hipLaunchKernelGGL(wide_x_kernel<32>, dim3(nwg_x), dim3(kWideThreads), lds_x, st, q, par); break;
                default:
                    if (t_global) {
                        hipLaunchKernelGGL(wide_t_kernel, dim3((unsigned)((ldn + kWideThreads - 1) / kWideThreads)), dim3(kWideThreads), 0, st, q, par);
                        hipLaunchKernelGGL((wide_x_kernel<0, true>), dim3(nwg_x), dim3(kWideThreads), 0, st, q, par);
                    } else {
                        hipLaunchKernelGGL((wide_x_kernel<0, false>), dim3(nwg_x), dim3(kWideThreads), lds_x, st, q, par);
                    }
                    hipLaunchKernelGGL(wide_ax_kernel, dim3(kAxWG), dim3(kWideThreads), 0, st, q);
            }
            if (cshard && peer_fused) {                                // the only exchange, produced and consumed by the solver's own kernels
                const PeerExchange ex = comm_peer_begin((size_t)ldn * sizeof(float));
                if (peer_one) {
                    hipLaunchKernelGGL(wide_tail_kernel<2>, dim3(nwg_tail), dim3(kWideThreads), 0, st, q, par, ex);
                } else {
                    hipLaunchKernelGGL(wide_ax_push_kernel, dim3((unsigned)((ldn + kWtElems - 1) / kWtElems)), dim3(kWideThreads), 0, st, q, par, ex);
                    hipLaunchKernelGGL(wide_tail_kernel<1>, dim3(nwg_tail), dim3(kWideThreads), 0, st, q, par, ex);
                }
                if (q.state != nullptr) hipLaunchKernelGGL(wide_state_kernel, dim3(std::min(1024, (std::max(n, p) + kWideThreads - 1) / kWideThreads)), dim3(kWideThreads), 0, st, q, par);
                if (persist_rows) launch_persist_rows(par ^ 1);          // the stretch on this rank's column block, its A x summed over the ranks inside the launch
                return;
            }
            if (cshard) {                                              // the only exchange: A x summed over the ranks' column blocks
                hipLaunchKernelGGL(wide_ax_local_kernel, dim3((unsigned)((ldn + kWtElems - 1) / kWtElems)), dim3(kWideThreads), 0, st, q, par, axl.get());
                allreduce_sum_f32(axl.get(), (size_t)n, st);
            }
            hipLaunchKernelGGL(wide_tail_kernel<0>, dim3(nwg_tail), dim3(kWideThreads), 0, st, q, par, PeerExchange{});
            if (q.state != nullptr) hipLaunchKernelGGL(wide_state_kernel, dim3(std::min(1024, (std::max(n, p) + kWideThreads - 1) / kWideThreads)), dim3(kWideThreads), 0, st, q, par);
            if (persist_rows) launch_persist_rows(par ^ 1);             // takes over from the state the next x-update launch would start from
        }, cshard ? nullptr : hflag.p);       // column-sharded: every rank must enqueue the same number of exchanges -> stream-ordered sampling of `done`
        S.t_loop = lt.wall_s; S.loop_ms_events = lt.events_ms; S.xupdate_launches = lt.launched;
        S.exchange_variant = !cshard ? 0 : (!peer_fused ? 1 : (peer_one ? 3 : 2));
        S.xupdate_variant = !screened ? 0 : (screen_fmt == 8 ? 2 : 1);
        if (q.scr_stat != nullptr) {
            unsigned long long hs[2] = {0, 0};
            ADMM_HIP_CHECK(hipMemcpy(hs, scr_stat.get(), sizeof(hs), hipMemcpyDeviceToHost));
            std::fprintf(stderr, "[wide screen] %d-bit copy: %llu columns screened on regular steps, %llu took the exact path (%.3f %%)\n", screen_fmt, hs[0], hs[1], hs[0] ? 100.0 * (double)hs[1] / (double)hs[0] : 0.0);
            scr_stat.zero(st);
        }
        if (persist_rows) {
            int herr = 0;
            ADMM_HIP_CHECK(hipMemcpy(&herr, rerr.get(), sizeof(int), hipMemcpyDeviceToHost));
            // A hand-over that timed out (the stretch's workgroups were not co-resident: a busy or shared device).  The waits are per
            // workgroup, so one workgroup may have left without its write-back while the others completed theirs: the state behind
            // such a launch cannot be trusted.  The whole path is discarded and run again with the stretch switched off
            // for the rest of this plan's life; persist_iter = -1 in the stats of that run says so.
            if (herr && cshard)                                         // (the ranks cannot agree on a re-run after the fact)
                throw Error(ADMM_ERR_COMM, "column-sharded wide solver: a hand-over of the persistent stretch timed out (a rank missing, or the stretch's workgroups not co-resident); ADMM_HIP_WIDE_PERSIST_COLS=0 runs without it");
            if (herr) {
                persist_rows = false;
                rstat.zero(st);
                comm_stream_sync(st);
                run(res);
                res.stats.persist_iter = -1;
                return;
            }
            unsigned long long hs[16] = {0};
            ADMM_HIP_CHECK(hipMemcpy(hs, rstat.get(), sizeof(hs), hipMemcpyDeviceToHost));
            S.persist_iter = (long long)hs[0];
            rstat.zero(st);
            if (option("WIDE_PERSIST_STATS"))
            {
                const double it = hs[0] ? (double)hs[0] : 1.0;
                std::fprintf(stderr, "[wide rows persist] %llu iterations in %llu stretches (%llu not on one XCD, %llu extra hand-overs after a rho change), %.2f us per iteration inside; %d row groups x %d column groups\n"
                             "[wide rows persist] start-up %.2f us per stretch; per iteration, workgroup 0: loop %.2f | x-update %.2f | A x partial + publish %.2f | wait A %.2f | rows + dots + publish %.2f | wait shares %.2f | decide %.2f | wait B + dots of my column %.2f us\n",
                             hs[0], hs[1], hs[3], hs[4], 0.01 * (double)hs[2] / it, rows_R, rows_C, hs[1] ? 0.01 * (double)hs[5] / (double)hs[1] : 0.0,
                             0.01 * hs[6] / it, 0.01 * hs[7] / it, 0.01 * hs[8] / it, 0.01 * hs[9] / it, 0.01 * hs[10] / it, 0.01 * hs[11] / it, 0.01 * hs[12] / it, 0.01 * hs[13] / it);
            }
        }
#ifdef ADMM_HIP_PROBE
        if (const char* f = option("PROBE_OUT")) {
            std::vector<long long> hp((size_t)4096 * 4 * 8);
            ADMM_HIP_CHECK(hipMemcpy(hp.data(), probe.get(), hp.size() * sizeof(long long), hipMemcpyDeviceToHost));
            if (FILE* fp = std::fopen(f, "wb")) { std::fwrite(hp.data(), sizeof(long long), hp.size(), fp); std::fclose(fp); }
        }
#endif

        res.niter.assign(nlam, 0);
        ADMM_HIP_CHECK(hipMemcpy(res.niter.data(), niter.get(), nlam * sizeof(int), hipMemcpyDeviceToHost));
        const size_t pt1 = (size_t)p_total + 1;
        std::vector<double> icpt(nlam, 0.0);                          // sum_j beta_j meanX_j over this rank's columns
        long long tot = 0;
        if (!cshard && res.beta_dst != nullptr) {
            // sparse read-back straight into the caller's buffer (kernels above)
            DevBuf<int> dcnt(nlam);
            hipLaunchKernelGGL(wide_beta_count_kernel, dim3(nlam), dim3(256), 0, st, beta.get(), p, dcnt.get());
            std::vector<int> hcnt(nlam);
            ADMM_HIP_CHECK(hipMemcpyAsync(hcnt.data(), dcnt.get(), (size_t)nlam * sizeof(int), hipMemcpyDeviceToHost, st));
            comm_stream_sync(st);
            std::vector<long long> hoff(nlam + 1, 0);
            for (int l = 0; l < nlam; ++l) hoff[l + 1] = hoff[l] + hcnt[l];
            const size_t tot_nz = (size_t)hoff[nlam];
            DevBuf<long long> doff(nlam);
            DevBuf<int> didx(std::max<size_t>(tot_nz, 1));
            DevBuf<float> dval(std::max<size_t>(tot_nz, 1));
            std::vector<int> hidx(tot_nz);
            std::vector<float> hval(tot_nz);
            ADMM_HIP_CHECK(hipMemcpyAsync(doff.get(), hoff.data(), (size_t)nlam * sizeof(long long), hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(wide_beta_compact_kernel, dim3(nlam), dim3(256), 0, st, beta.get(), p, doff.get(), didx.get(), dval.get());
            if (tot_nz) {
                read_back(hidx.data(), didx.get(), tot_nz * sizeof(int), st);
                read_back(hval.data(), dval.get(), tot_nz * sizeof(float), st);
            }
            std::memset(res.beta_dst, 0, pt1 * nlam * sizeof(float));
            comm_stream_sync(st);
            for (int l = 0; l < nlam; ++l) {
                float b0 = 0.f;
                recover_coef_sparse<float>(d, hidx.data() + hoff[l], hval.data() + hoff[l], hcnt[l], &b0, res.beta_dst + (size_t)l * pt1 + 1);
                res.beta_dst[(size_t)l * pt1] = b0;
                tot += res.niter[l];
            }
            res.beta_written = true;
            res.beta.clear();
        } else {
        std::vector<float> hb((size_t)nlam * p);
        read_back(hb.data(), beta.get(), hb.size() * sizeof(float), st);
        res.beta.assign(pt1 * nlam, 0.f);
        for (int l = 0; l < nlam; ++l) {
            float b0 = 0.f;
            recover_coef<float>(d, hb.data() + (size_t)l * p, &b0, res.beta.data() + (size_t)l * pt1 + 1 + col_offset);
            res.beta[(size_t)l * pt1] = b0;
            icpt[l] = (double)d.meanY - (double)b0;
            tot += res.niter[l];
        }
        }
        if (cshard) {
            // every rank returns the full coefficient matrix: blocks summed into a zero-padded copy, the intercept
            // beta_0 = meanY - sum over ALL columns (DataStd::recover) from the ranks' partial sums
            const bool has_icpt = (d.flag & 2) != 0;
            DevBuf<float> db(res.beta.size()); DevBuf<double> di(nlam);
            for (int l = 0; l < nlam; ++l) res.beta[(size_t)l * pt1] = 0.f;
            ADMM_HIP_CHECK(hipMemcpyAsync(db.get(), res.beta.data(), res.beta.size() * sizeof(float), hipMemcpyHostToDevice, st));
            ADMM_HIP_CHECK(hipMemcpyAsync(di.get(), icpt.data(), nlam * sizeof(double), hipMemcpyHostToDevice, st));
            allreduce_sum_f32(db.get(), res.beta.size(), st);
            allreduce_sum_f64(di.get(), (size_t)nlam, st);
            read_back(res.beta.data(), db.get(), res.beta.size() * sizeof(float), st);
            ADMM_HIP_CHECK(hipMemcpyAsync(icpt.data(), di.get(), nlam * sizeof(double), hipMemcpyDeviceToHost, st));
            comm_stream_sync(st);
            comm_check();
            for (int l = 0; l < nlam; ++l) res.beta[(size_t)l * pt1] = has_icpt ? (float)((double)d.meanY - icpt[l]) : 0.f;
        }
        S.total_iter = tot;
        {   // decisions taken = the cold-start one + one per ADMM iteration
            WideCtl hc[2];
            ADMM_HIP_CHECK(hipMemcpy(hc, ctl.get(), sizeof(hc), hipMemcpyDeviceToHost));
            trace_n = std::max(hc[0].total, hc[1].total);
        }
        res.stats = S;
    }
};

std::unique_ptr<LassoPlan> make_wide_plan(DeviceData<float>&& d, const LassoProblem& pb, hipStream_t st) {
    return std::unique_ptr<LassoPlan>(new WidePlan(std::move(d), pb, st));
}

}  // namespace admm
