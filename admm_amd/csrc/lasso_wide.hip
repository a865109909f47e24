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
__device__ __forceinline__ WideDecision wide_decide(const WideParams& q, const WideCtl& in, const double (&sums)[5], int lane) {
    int lam_finished = -1, niter_val = 0;
    // Every wave reduces the norm partials itself (fixed order, no LDS, no barrier): the five sums through one halving
    // butterfly (bit-identical to five wave_sum calls), then ONE square-root sequence with lane 8 k working on sum k,
    // and the five results read back as wave-uniform scalars.
    const double v8[8] = {sums[0], sums[1], sums[2], sums[3], sums[4], 0.0, 0.0, 0.0};
    const double root = sqrt(halving_sum8(v8, lane));
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
                // regular step, fused mode: two columns requested before the first is consumed (16 KB in flight per wave;
                // same order of columns: bit-identical partial sums)
                while (reg && mask) {
                    const int l0 = __ffsll((long long)mask) - 1;
                    mask &= mask - 1;
                    const long long j0 = (long long)(s0 + l0) * NW + w;
                    float4 c0[NRT], c1[NRT];
                    col_request(j0, c0, q.x_nt != 0);
                    int l1 = -1;
                    long long j1 = 0;
                    if (mask) {
                        l1 = __ffsll((long long)mask) - 1;
                        mask &= mask - 1;
                        j1 = (long long)(s0 + l1) * NW + w;
                        col_request(j1, c1, q.x_nt != 0);
                    }
                    const float xn0 = col_finish(__shfl(xj, l0, 64), c0);
                    if (lane == l0) q.x[j0] = xn0;
                    if (l1 >= 0) {
                        const float xn1 = col_finish(__shfl(xj, l1, 64), c1);
                        if (lane == l1) q.x[j1] = xn1;
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
// Round 3.  Almost every iteration of a wide path is an ACTIVE-SET step (ADMMLassoWide.h:86-118: only the current
// non-zeros are updated; 17 191 of 17 613 iterations at BASELINE configs[2]) on a few dozen to a few hundred columns -- two
// latency-bound launches, 16.6 us per iteration however little there is to do (profiles/r02_probe_timelines.md: prologue round
// trip, column round trip, two kernel boundaries, the z / y launch's own round trip).  This kernel runs a whole STRETCH of
// them -- from wherever the two-launch path stands until the next regular step, a finished lambda, or an active set too
// large for it -- inside ONE launch: kPG workgroups (blocks b with b % 8 == 0 of a 256-block grid: observed to share one XCD
// and its L2, a speed-up only, never relied upon) keep iterating and hand their results to one another through global
// memory with write-through stores and cache-bypassing loads, two hand-overs per iteration:
//   (B) every wave keeps the x values of ITS column slots in registers (column j <-> slot (j / NW) of wave j mod NW, the
//       same deal as the two-launch kernel), updates the non-zero ones against t / gamma staged in LDS and accumulates
//       x_j X_j; the workgroup publishes its partial of A x                                      [hand-over A: partials]
//   (D) workgroup g owns n / kPG rows: it sums the kPG partials of its rows in workgroup order, forms z_new, y_new, r and its
//       share of the five norms, and publishes A x + z, y and the norm shares                   [hand-over B: everything else]
//   (E) every workgroup reduces the norm shares, takes the decision of ADMMBase::solve (wide_decide: stop test, rho adaptation,
//       schedule), and -- if another active-set step follows -- forms t = (A x + z) + y / rho itself.
// The arithmetic of a step is the two-launch path's (same functions, same roundings); only the ORDER in which the partials
// of A x and of the norms are added differs (kPG workgroups instead of 256, per-workgroup norm shares), so the two paths
// agree to summation rounding, not bit for bit, and both are held to the oracle by the same trace rule.  When the stretch
// ends the kernel leaves x, A x, z, y, the norm partials and the control block exactly as a tail launch would, and the next
// x-update launch simply continues.  ADMM_HIP_WIDE_PERSIST=0 disables it.
constexpr int kPG = 32;                   // workgroups of the persistent stretch
constexpr int kPNU = 32;                  // x slots per lane a wave can own: p <= kPNU * 64 * (4 kPG) = 262144
struct WidePersist {
    unsigned long long* flagA; unsigned long long* flagB;      // [kPG][8] monotonic hand-over words, one 64-byte line each
    unsigned long long* flagX;                                 // [kPG][8] (launch << 32) | (XCD + 1) of every workgroup
    float* sz; float* yv;                  // [npad] A x + z and y of the iteration just finished (write-through)
    double* np;                            // [kPG][8] norm shares: |r|^2, |dz|^2, |Ax|^2, |z|^2, |y|^2, non-zeros
    int* err;                              // device word: non-zero after a timed-out wait
    unsigned long long seq;                // launch number (host): hand-over words only ever grow
    int max_cols;                          // leave the stretch when the active set exceeds this many columns
    unsigned long long* stat;              // [4] iterations done in stretches, stretches, 100 MHz ticks inside them (diagnostics)
    double* hint;                          // [2][2] per launch parity: {non-zeros when the kernel last ran, launches to sit out}: while the
                                           // active set is too large for this kernel it only looks again every 64th launch
};

// 8-byte payload store.  wt = true: write-through (sc1), visible to a cache-bypassing load anywhere on the device.  wt = false
// (all workgroups of the stretch were FOUND on one XCD at the start of this launch, wide_act_persist_kernel): a plain store, which
// stays in the XCD's shared L2 where the readers' cache-bypassing loads find it at L2 latency instead of the fabric's.
__device__ __forceinline__ void wp_store2(float* p, float a, float b, bool wt) {
    if (wt) __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b) << 32),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *reinterpret_cast<float2*>(p) = make_float2(a, b);
}
__device__ __forceinline__ void wp_store_f64(double* p, double v, bool wt) {
    if (wt) __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
__device__ __forceinline__ unsigned wp_xcc_id() {                                    // which XCD this wave runs on (0..7)
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}
__device__ __forceinline__ float2 wp_load2(const float* p) {                        // 8-byte load that bypasses this CU's L1
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32)));
}
// every storing wave drains its stores, then ONE lane raises the workgroup's word
__device__ __forceinline__ void wp_publish(unsigned long long* flag, unsigned long long val) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// lanes < kPG of wave 0 poll one word each (bounded); everybody leaves together.  Returns false after a time-out.
__device__ __forceinline__ bool wp_wait(const unsigned long long* flags, unsigned long long val, int* err, int* s_ok) {
    if (threadIdx.x < 64) {
        bool ok = true;
        if (threadIdx.x < kPG) {
            const unsigned long long* f = flags + (size_t)threadIdx.x * 8;
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < val) {
                __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > 200000000ll) { ok = false; break; }        // 2 s
            }
        }
        ok = __all(ok) != 0;
        if (threadIdx.x == 0) { *s_ok = ok ? 1 : 0; if (!ok) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
    __syncthreads();
    return *s_ok != 0;
}
// The same wait on words that carry (launch << 32) | (XCD of the workgroup + 1); *s_same = 1 if all kPG workgroups sit on one XCD.
__device__ __forceinline__ bool wp_wait_xcc(const unsigned long long* flags, unsigned long long seq, int* err, int* s_ok, int* s_same, int npart = kPG) {
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

template <int RT>      // a column is RT float4 per lane (n <= RT * 256)
__global__ void __launch_bounds__(kWideThreads)
wide_act_persist_kernel(WideParams q, int cpar, WidePersist ps) {
    if ((blockIdx.x & 7) != 0) return;
    const int g = blockIdx.x >> 3;
    if (g >= kPG) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int npad = (q.n + 255) / 256 * 256;
    float* tdl = reinterpret_cast<float*>(smem_raw);                       // t / gamma [npad]
    float4* red = reinterpret_cast<float4*>(smem_raw + (size_t)npad * sizeof(float));      // [RT][256] wave partials of A x
    __shared__ int s_ok, s_same;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    constexpr int NWp = kPG * (kWideThreads / 64);
    const int w = g * (kWideThreads / 64) + wid;
    // ---- the decision the next x-update launch would take: go on only if it is an active-set step
    WideCtl in = q.ctl[cpar];
    in = wide_ctl_uniform(in);
    // (this launch READS hint[seq & 1] and WRITES hint[(seq + 1) & 1]: every workgroup sees the same words)
    const double* hin = ps.hint + (size_t)(ps.seq & 1) * 2;
    double* hout = ps.hint + (size_t)((ps.seq + 1) & 1) * 2;
    const double h_nnz = hin[0], h_wait = hin[1];
    const bool leader = g == 0 && threadIdx.x == 0;
    if (in.done || in.first) { if (leader) { hout[0] = h_nnz; hout[1] = h_wait; } return; }
    if (h_nnz > (double)ps.max_cols && h_wait > 0.0) { if (leader) { hout[0] = h_nnz; hout[1] = h_wait - 1.0; } return; }
    double sums[5];
    {
        const WideNormRaw nraw = wide_norms_request(q, lane);
        wide_norms_finish(q, lane, nraw, sums);
    }
    WideDecision dec = wide_decide(q, in, sums, lane);
    WideCtl out = wide_ctl_uniform(dec.out);
    if (out.done || out.type != W_ACT || __builtin_amdgcn_readfirstlane(dec.lam_finished) >= 0) { if (leader) { hout[0] = h_nnz; hout[1] = h_wait; } return; }
    // ---- where do we run?  Every workgroup announces its XCD (write-through, valid under any placement); the answer is
    // collected below, after the loads of the prologue have been issued
    if (threadIdx.x == 0) __hip_atomic_store(ps.flagX + (size_t)g * 8, (ps.seq << 32) | (unsigned long long)(wp_xcc_id() + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ---- this wave's x slots, its share of the rows, in registers for the whole stretch
    float xs[kPNU];
#pragma unroll
    for (int u = 0; u < kPNU; ++u) {
        const long long jl = (long long)(u * 64 + lane) * NWp + w;
        xs[u] = jl < q.p ? q.x[jl] : 0.f;
    }
    const int R = npad / kPG;                                   // rows per workgroup, a multiple of 8
    // reducer mapping: 8 lanes share a PAIR of rows (4 of the kPG partials each); pairs beyond R / 2 idle
    const int sub = threadIdx.x & 7, pair = threadIdx.x >> 3;
    const int npairs_pass = kWideThreads / 8;                    // 32 pairs = 64 rows per pass
    constexpr int MAXP = 4;                                      // R <= 256 rows per workgroup (n <= 8192)
    float ax_r[MAXP][2], z_r[MAXP][2], y_r[MAXP][2], yd_r[MAXP][2];
#pragma unroll
    for (int ps_ = 0; ps_ < MAXP; ++ps_) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int i = g * R + (ps_ * npairs_pass + pair) * 2 + e;
            const bool own = (ps_ * npairs_pass + pair) * 2 < R && i < q.n;
            ax_r[ps_][e] = own ? q.Ax[i] : 0.f; z_r[ps_][e] = own ? q.z[i] : 0.f; y_r[ps_][e] = own ? q.y[i] : 0.f; yd_r[ps_][e] = own ? q.Y[i] : 0.f;
        }
    }
    // first t from the vectors the tail launch left (plain loads: written by an earlier launch)
    {
        const float rho_f = (float)out.rho;
        for (int i = threadIdx.x; i < npad; i += kWideThreads) {
            const float t = i < q.n ? (q.Ax[i] + q.z[i]) + q.y[i] / rho_f : 0.f;
            tdl[i] = t / q.gamma;
        }
    }
    const int nv = (q.n + 3) / 4 * 4;
    // the wave's first non-zero column stays in registers for the whole stretch (the active set only shrinks inside it:
    // a zero never comes back before the next regular step): no column round trip in most iterations
    int cu0 = -1, cl0 = -1;
#pragma unroll
    for (int u = 0; u < kPNU; ++u) {
        const unsigned long long m = __ballot(xs[u] != 0.f);
        if (cu0 < 0 && m != 0) { cu0 = u; cl0 = __ffsll((long long)m) - 1; }
    }
    float4 cv0[RT];
#pragma unroll
    for (int kk = 0; kk < RT; ++kk) cv0[kk] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cu0 >= 0) {
        const float* col = q.X + (size_t)((long long)(cu0 * 64 + cl0) * NWp + w) * q.ldx;
#pragma unroll
        for (int kk = 0; kk < RT; ++kk) {
            const int r = kk * 256 + lane * 4;
            if (r < nv) cv0[kk] = *reinterpret_cast<const float4*>(col + r);
        }
    }
    if (!wp_wait_xcc(ps.flagX, ps.seq, ps.err, &s_ok, &s_same)) return;       // (also the barrier after staging t)
    const bool wt = s_same == 0;                                 // not all on one XCD: payloads must be written through
    unsigned long long k = 0;                                    // iterations completed in this launch
    bool failed = false;
    double nz_last = 0.0;
    const long long tick0 = wall_clock64();
    long long ph[6] = {0, 0, 0, 0, 0, 0}, tp = tick0;              // diagnostics: ticks per phase, as seen by workgroup 0
#define WP_PHASE(i) { const long long tn = wall_clock64(); ph[i] += tn - tp; tp = tn; }
    for (;;) {
        // ---- (record the decision being acted on: what the x-update launch writes)
        if (g == 0 && threadIdx.x == 0 && q.trace != nullptr && in.total < q.trace_cap) {
            double* t = q.trace + (size_t)in.total * ADMM_TRACE_FIELDS;
            t[0] = in.lam_idx; t[1] = in.iter; t[2] = in.eps_primal; t[3] = in.eps_dual; t[4] = dec.rp; t[5] = dec.rd;
            t[6] = out.rho; t[7] = out.type; t[8] = dec.code; t[9] = in.rho; t[10] = out.rho; t[11] = in.lam;
        }
        // ---- (B) active-set update of this wave's non-zero columns (ADMMLassoWide.h:86-118 / ADMMEnet.h:85-122)
        const double pen_d = (double)out.lam / (out.rho * (double)q.gamma);
        const float penalty = (float)pen_d;
        const float thresh_a = q.enet ? q.alpha * penalty : penalty;
        const float denom_a = q.enet ? (float)(1.0 + (double)penalty * (1.0 - (double)q.alpha)) : 1.f;
        float4 acc[RT];
#pragma unroll
        for (int kk = 0; kk < RT; ++kk) acc[kk] = make_float4(0.f, 0.f, 0.f, 0.f);
        int nnz_w = 0;
#pragma unroll
        for (int u = 0; u < kPNU; ++u) {
            unsigned long long mask = __ballot(xs[u] != 0.f);
            while (mask) {
                const int l = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const long long jj = (long long)(u * 64 + l) * NWp + w;
                const float xv = __shfl(xs[u], l, 64);
                const float* col = q.X + (size_t)jj * q.ldx;
                float4 cv[RT];
                if (u == cu0 && l == cl0) {
#pragma unroll
                    for (int kk = 0; kk < RT; ++kk) cv[kk] = cv0[kk];
                } else {
#pragma unroll
                    for (int kk = 0; kk < RT; ++kk) {
                        const int r = kk * 256 + lane * 4;
                        cv[kk] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (r < nv) cv[kk] = *reinterpret_cast<const float4*>(col + r);
                    }
                }
                float d0 = 0.f, d1 = 0.f;
#pragma unroll
                for (int kk = 0; kk < RT; ++kk) {
                    const int r = kk * 256 + lane * 4;
                    if (r < nv) {
                        const float4 b = *reinterpret_cast<const float4*>(tdl + r);
                        float& dd = (kk & 1) ? d1 : d0;
                        dd = fmaf(cv[kk].x, b.x, dd); dd = fmaf(cv[kk].y, b.y, dd); dd = fmaf(cv[kk].z, b.z, dd); dd = fmaf(cv[kk].w, b.w, dd);
                    }
                }
                const float xn = prox_f(xv - wave_sum(d0 + d1), thresh_a, denom_a, q.enet != 0);
                if (xn != 0.f) {
                    nnz_w++;
#pragma unroll
                    for (int kk = 0; kk < RT; ++kk) {
                        acc[kk].x = fmaf(xn, cv[kk].x, acc[kk].x); acc[kk].y = fmaf(xn, cv[kk].y, acc[kk].y);
                        acc[kk].z = fmaf(xn, cv[kk].z, acc[kk].z); acc[kk].w = fmaf(xn, cv[kk].w, acc[kk].w);
                    }
                }
                if (lane == l) xs[u] = xn;
            }
        }
        WP_PHASE(0)
        // ---- (C) the workgroup's partial of A x: the 4 waves in order, written through
        __syncthreads();                                         // everybody is done with tdl's neighbour `red` of the last round
#pragma unroll
        for (int kk = 0; kk < RT; ++kk) red[kk * kWideThreads + threadIdx.x] = acc[kk];
        __syncthreads();
        for (int e = threadIdx.x; e < npad / 2; e += kWideThreads) {          // element pair e: rows 2 e, 2 e + 1
            const int r0 = 2 * e, kk = r0 / 256, ln = (r0 % 256) / 4, c = r0 & 3;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int ww = 0; ww < kWideThreads / 64; ++ww) {
                const float4 v = red[kk * kWideThreads + ww * 64 + ln];
                const float a = c == 0 ? v.x : v.z, b = c == 0 ? v.y : v.w;
                s0 = ww == 0 ? a : s0 + a; s1 = ww == 0 ? b : s1 + b;
            }
            wp_store2(q.axpart + (size_t)g * q.ldn + r0, s0, s1, wt);
        }
        const unsigned long long tag = (ps.seq << 32) | (k + 1);
        wp_publish(ps.flagA + (size_t)g * 8, tag);
        WP_PHASE(1)
        // ---- (D) rows of this workgroup: A x from the kPG partials in workgroup order, z / y update, norm shares
        if (!wp_wait(ps.flagA, tag, ps.err, &s_ok)) { failed = true; break; }
        WP_PHASE(2)
        double nacc[5] = {0, 0, 0, 0, 0};
        {
            const float rho_f = (float)out.rho;
            const float den = (float)(-1.0 - out.rho);
#pragma unroll
            for (int ps_ = 0; ps_ < MAXP; ++ps_) {
                const int pr = ps_ * npairs_pass + pair;
                if (pr * 2 < R) {
                    const int i0 = g * R + pr * 2;
                    float2 v[kPG / 8];
#pragma unroll
                    for (int m = 0; m < kPG / 8; ++m) v[m] = wp_load2(q.axpart + (size_t)(m * 8 + sub) * q.ldn + i0);
                    float a0 = v[0].x, a1 = v[0].y;
#pragma unroll
                    for (int m = 1; m < kPG / 8; ++m) { a0 += v[m].x; a1 += v[m].y; }
#pragma unroll
                    for (int m = 1; m < 8; m <<= 1) { a0 += __shfl_xor(a0, m, 64); a1 += __shfl_xor(a1, m, 64); }
                    if (sub == 0) {
#pragma clang fp contract(off)
                        const float axn[2] = {a0, a1};
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            if (i0 + e < q.n) {
                                const float ax = axn[e];
                                const float zn = (yd_r[ps_][e] + y_r[ps_][e] + rho_f * ax) / den;      // next_z (:156-165)
                                const float dz = zn - z_r[ps_][e];
                                const float r = ax + zn;                                               // next_residual (:166-170)
                                const float yn = y_r[ps_][e] + rho_f * r;                              // ADMMBase.h:183
                                ax_r[ps_][e] = ax; z_r[ps_][e] = zn; y_r[ps_][e] = yn;
                                nacc[0] += (double)r * r; nacc[1] += (double)dz * dz; nacc[2] += (double)ax * ax;
                                nacc[3] += (double)zn * zn; nacc[4] += (double)yn * yn;
                            }
                        }
                        wp_store2(ps.sz + i0, ax_r[ps_][0] + z_r[ps_][0], ax_r[ps_][1] + z_r[ps_][1], wt);
                        wp_store2(ps.yv + i0, y_r[ps_][0], y_r[ps_][1], wt);
                    }
                }
            }
        }
        {   // the workgroup's norm shares and non-zero count: fixed order (lanes, then waves)
            double* scr = reinterpret_cast<double*>(red);                     // `red` has been consumed (barrier inside wp_wait)
            double v6[6] = {nacc[0], nacc[1], nacc[2], nacc[3], nacc[4], (double)nnz_w};
#pragma unroll
            for (int m = 0; m < 6; ++m) v6[m] = wave_sum(v6[m]);
            // nnz_w is wave uniform: wave_sum multiplied it by 64
            if (lane == 0) {
#pragma unroll
                for (int m = 0; m < 6; ++m) scr[wid * 8 + m] = v6[m];
            }
            __syncthreads();
            if (threadIdx.x < 6) {
                double t = 0;
                for (int ww = 0; ww < kWideThreads / 64; ++ww) t += scr[ww * 8 + threadIdx.x];
                if (threadIdx.x == 5) t *= 1.0 / 64.0;
                wp_store_f64(ps.np + (size_t)g * 8 + threadIdx.x, t, wt);
            }
        }
        if (q.state != nullptr && out.total < q.state_cap) {         // iterate dump: this iteration's x | A x | z | y (record = the trace record that judges it)
            float* srec = q.state + (size_t)out.total * ((size_t)q.p + 3 * (size_t)q.n);
#pragma unroll
            for (int u = 0; u < kPNU; ++u) {
                const long long jl = (long long)(u * 64 + lane) * NWp + w;
                if (jl < q.p) srec[jl] = xs[u];
            }
            if (sub == 0) {
#pragma unroll
                for (int ps_ = 0; ps_ < MAXP; ++ps_) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int pr = ps_ * npairs_pass + pair;
                        const int i = g * R + pr * 2 + e;
                        if (pr * 2 < R && i < q.n) {
                            srec[(size_t)q.p + i] = ax_r[ps_][e]; srec[(size_t)q.p + q.n + i] = z_r[ps_][e]; srec[(size_t)q.p + 2 * (size_t)q.n + i] = y_r[ps_][e];
                        }
                    }
                }
            }
        }
        wp_publish(ps.flagB + (size_t)g * 8, tag);
        WP_PHASE(3)
        k++;
        // ---- (E) the decision on this iteration, by everybody from the same numbers
        if (!wp_wait(ps.flagB, tag, ps.err, &s_ok)) { failed = true; break; }
        WP_PHASE(4)
        in = out;
        // everything the rest of the iteration needs is requested at once: the norm shares AND the vectors of the next t
        constexpr int NE = (RT * 256 / 2 + kWideThreads - 1) / kWideThreads;          // element pairs per thread
        float2 sza[NE], yva[NE];
#pragma unroll
        for (int ee = 0; ee < NE; ++ee) {
            const int e = ee * kWideThreads + threadIdx.x;
            sza[ee] = make_float2(0.f, 0.f); yva[ee] = make_float2(0.f, 0.f);
            if (e < npad / 2) { sza[ee] = wp_load2(ps.sz + 2 * e); yva[ee] = wp_load2(ps.yv + 2 * e); }
        }
        double nz_total;
        {
            double sh[6];
#pragma unroll
            for (int m = 0; m < 6; ++m)
                sh[m] = lane < kPG ? __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(ps.np + (size_t)lane * 8 + m),
                                                                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0.0;
#pragma unroll
            for (int m = 0; m < 5; ++m) sums[m] = sh[m];
            nz_total = wave_sum(sh[5]);
        }
        nz_last = nz_total;
        dec = wide_decide(q, in, sums, lane);
        out = wide_ctl_uniform(dec.out);
        const bool go_on = !out.done && out.type == W_ACT && __builtin_amdgcn_readfirstlane(dec.lam_finished) < 0 && nz_total <= (double)ps.max_cols;
        if (!go_on) { WP_PHASE(5) break; }
        {   // t of the next active-set step from what the row owners published
            const float rho_f = (float)out.rho;
#pragma unroll
            for (int ee = 0; ee < NE; ++ee) {
                const int e = ee * kWideThreads + threadIdx.x;
                if (e < npad / 2) {
                    const float2 a = sza[ee], b = yva[ee];
                    const float t0 = 2 * e < q.n ? a.x + b.x / rho_f : 0.f, t1 = 2 * e + 1 < q.n ? a.y + b.y / rho_f : 0.f;
                    tdl[2 * e] = t0 / q.gamma; tdl[2 * e + 1] = t1 / q.gamma;
                }
            }
        }
        __syncthreads();
        WP_PHASE(5)
    }
    if (failed) return;
    if (leader) {
        hout[0] = nz_last; hout[1] = 64.0;
        ps.stat[0] += k; ps.stat[1] += 1; ps.stat[2] += (unsigned long long)(wall_clock64() - tick0); ps.stat[3] += wt ? 1 : 0;
        for (int i = 0; i < 6; ++i) ps.stat[4 + i] += (unsigned long long)ph[i];
    }
    // ---- leave everything as a tail launch would have: x, A x, z, y, the norm partials, the control block
#pragma unroll
    for (int u = 0; u < kPNU; ++u) {
        const long long jl = (long long)(u * 64 + lane) * NWp + w;
        if (jl < q.p) q.x[jl] = xs[u];
    }
    if (sub == 0) {
#pragma unroll
        for (int ps_ = 0; ps_ < MAXP; ++ps_) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int pr = ps_ * npairs_pass + pair;
                const int i = g * R + pr * 2 + e;
                if (pr * 2 < R && i < q.n) { q.Ax[i] = ax_r[ps_][e]; q.z[i] = z_r[ps_][e]; q.y[i] = y_r[ps_][e]; }
            }
        }
    }
    if (g == 0) {
        // the norm partials as a tail launch leaves them: rows [0, nwg_tail) hold the sums (here: row 0 holds all of it, added
        // in workgroup order), every other row is ZERO -- the decision of the next launch adds up max(nwg_tail, 128) rows, and a
        // later tail launch only overwrites the first nwg_tail of them (a stale share left in row nwg_tail .. kPG - 1 was added
        // to every later decision: found by the random sweep, case 41 of seed 7, n = 19)
        for (int idx = threadIdx.x; idx < q.nwg_tail * 8 || idx < kPG * 8; idx += kWideThreads) {
            const int row = idx >> 3, m = idx & 7;
            double v = 0.0;
            if (row == 0 && m < 5) {
                for (int r = 0; r < kPG; ++r)
                    v += __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(ps.np + (size_t)r * 8 + m),
                                                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            }
            q.P[idx] = v;                                       // P holds max(nwg_tail, kWideNormRows) >= kPG rows
        }
        if (threadIdx.x == 0) q.ctl[cpar] = in;                 // the state the breaking decision was taken FROM: the next launch repeats it
    }
}

// Measured and rejected (round 3): the same stretch with ONE hand-over per iteration -- 16 workgroups of 512 threads, every
// workgroup adds up all 16 partials of A x itself (128 KB from the XCD's L2 per iteration) and keeps the whole z / y / A x in
// registers, so that norms, decision and the next t need no second exchange.  Correct (same tests), but 12.1 us per iteration
// inside against 10.5 us for the two-hand-over form above on C3: the 8-wave combine and the 16 x 8 KB of partial loads per
// workgroup cost more than the second hop they remove.

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
        if (nrec > 0) ADMM_HIP_CHECK(hipMemcpy(out, trace.get(), (size_t)nrec * ADMM_TRACE_FIELDS * sizeof(double), hipMemcpyDeviceToHost));
        return nrec;
    }

    // iterate dump (test facility): record s = x | A x | z | y (p + 3 n floats) the decision of trace record s judged
    DevBuf<float> state;
    long long state_cap = 0;
    void enable_state(long long cap) override {
        if (cshard) throw Error(ADMM_ERR_INVALID_ARG, "the column-sharded wide solver records no iterate dump");
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
        if (nrec > 0) ADMM_HIP_CHECK(hipMemcpy(out, state.get(), (size_t)nrec * rec * sizeof(float), hipMemcpyDeviceToHost));
        return nrec;
    }
    // the standardised data as this solver holds them (test hook admm_hip_lasso_plan_data_read)
    void read_data(float* x_out, long long ld, float* y_out) override {
        ADMM_HIP_CHECK(hipStreamSynchronize(st));
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
        if (const char* e = std::getenv("ADMM_HIP_PEER_FUSED")) { if (std::string(e) == "0") peer_fused = false; }
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
                ADMM_HIP_CHECK(hipStreamSynchronize(st));
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
        if (const char* e = std::getenv("ADMM_HIP_WIDE_SPRAD")) gram_free = !cshard && std::string(e) != "gram";
        if (gram_free) {
            GramFreeWideOp op(d.X.get(), d.ldx, n, p, st);
            ADMM_HIP_CHECK(hipStreamSynchronize(st));
            S.t_gram = 0.0;
            int nmatop = 0;
            sprad = lanczos_largest_f32([&](const float* v, float* w) { op(v, w); }, n, &nmatop);
            S.eig_est = sprad; S.t_eigs = now_s() - t0;
        } else {
            const long long ldg = round_up(n, 32);
            DevBuf<float> G((size_t)ldg * n); G.zero(st);
            gram_full<float>(d.X.get(), d.ldx, n, p, false, G.get(), ldg, st);
            if (cshard) allreduce_sum_f32(G.get(), (size_t)ldg * n, st);       // X X' = sum over the ranks' column blocks
            ADMM_HIP_CHECK(hipStreamSynchronize(st));
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
            if (const char* e = std::getenv("ADMM_HIP_PEER_FUSED")) { if (std::string(e) == "2") peer_one = false; }
        }
        x.alloc(ldp); x.zero(st);
        for (DevBuf<float>* b : {&Ax, &z, &y}) { b->alloc(std::max<long long>(ldn, 8192)); b->zero(st); }   // the fused x-update reads up to 32 x 256 entries unconditionally
        // ADMM_HIP_WIDE_FUSE=0: always three launches per iteration
        fuse_rt = n <= 1024 ? 4 : (n <= 2048 ? 8 : (n <= 4096 ? 16 : (n <= 6144 ? 24 : (n <= 8192 ? 32 : 0))));
        if (const char* e = std::getenv("ADMM_HIP_WIDE_FUSE")) if (std::string(e) == "0") fuse_rt = 0;
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
        if (const char* e = std::getenv("ADMM_HIP_WIDE_TGLOBAL")) if (std::string(e) == "1") t_global = true;
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
        if (const char* e = std::getenv("ADMM_HIP_WIDE_WGX")) wgx = std::max(1, std::atoi(e));
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
        setup_persist();
        ADMM_HIP_CHECK(hipStreamSynchronize(st));
    }

    // persistent active-set stretch (wide_act_persist_kernel)
    bool persist = false;
    DevBuf<unsigned long long> pflags;
    DevBuf<float> psz, pyv;
    DevBuf<double> pnp, phint;
    DevBuf<int> perr;
    DevBuf<unsigned long long> pstat;
    unsigned long long pseq = 0;
    int pmax_cols = 512;
    size_t lds_persist = 0;


    void setup_persist() {
        persist = (fuse_rt == 4 || fuse_rt == 8) && !cshard && (long long)p <= (long long)kPNU * 64 * kPG * (kWideThreads / 64);
        if (const char* e = std::getenv("ADMM_HIP_WIDE_PERSIST")) if (std::string(e) == "0") persist = false;
        if (const char* e = std::getenv("ADMM_HIP_WIDE_PERSIST_COLS")) pmax_cols = std::max(1, std::atoi(e));
        if (!persist) return;
        const size_t npad = (size_t)(n + 255) / 256 * 256;
        lds_persist = npad * sizeof(float) + (size_t)fuse_rt * kWideThreads * sizeof(float4);
        pflags.alloc((size_t)3 * kPG * 8); pflags.zero(st);
        psz.alloc(npad); pyv.alloc(npad); psz.zero(st); pyv.zero(st);
        pnp.alloc((size_t)kPG * 8); pnp.zero(st);
        phint.alloc(4); phint.zero(st);
        perr.alloc(1); perr.zero(st);
        pstat.alloc(16); pstat.zero(st);
    }
    void launch_persist(int cpar) {
        WidePersist ps;
        ps.flagA = pflags.get(); ps.flagB = pflags.get() + (size_t)kPG * 8; ps.flagX = pflags.get() + (size_t)2 * kPG * 8;
        ps.sz = psz.get(); ps.yv = pyv.get(); ps.np = pnp.get(); ps.err = perr.get(); ps.seq = ++pseq; ps.max_cols = pmax_cols; ps.hint = phint.get(); ps.stat = pstat.get();
        if (fuse_rt == 4) hipLaunchKernelGGL(wide_act_persist_kernel<4>, dim3(8 * kPG), dim3(kWideThreads), lds_persist, st, q, cpar, ps);
        else hipLaunchKernelGGL(wide_act_persist_kernel<8>, dim3(8 * kPG), dim3(kWideThreads), lds_persist, st, q, cpar, ps);
    }

    void run(LassoResult& res) override {
        admm_stats S = setup_stats;
        res.lambda = lam_user;
        beta.zero(st); niter.zero(st);
        *hflag.p = 0;
        if (persist) { phint.zero(st); perr.zero(st); }
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
                    return;
                }
                hipLaunchKernelGGL(wide_ax_push_kernel, dim3((unsigned)((ldn + kWtElems - 1) / kWtElems)), dim3(kWideThreads), 0, st, q, par, ex);
                hipLaunchKernelGGL(wide_tail_kernel<1>, dim3(nwg_tail), dim3(kWideThreads), 0, st, q, par, ex);
                return;
            }
            if (cshard) {                                              // the only exchange: A x summed over the ranks' column blocks
                hipLaunchKernelGGL(wide_ax_local_kernel, dim3((unsigned)((ldn + kWtElems - 1) / kWtElems)), dim3(kWideThreads), 0, st, q, par, axl.get());
                allreduce_sum_f32(axl.get(), (size_t)n, st);
            }
            hipLaunchKernelGGL(wide_tail_kernel<0>, dim3(nwg_tail), dim3(kWideThreads), 0, st, q, par, PeerExchange{});
            if (q.state != nullptr) hipLaunchKernelGGL(wide_state_kernel, dim3(std::min(1024, (std::max(n, p) + kWideThreads - 1) / kWideThreads)), dim3(kWideThreads), 0, st, q, par);
            if (persist) launch_persist(par ^ 1);                      // takes over from the state the next x-update launch would start from
        }, cshard ? nullptr : hflag.p);       // column-sharded: every rank must enqueue the same number of exchanges -> stream-ordered sampling of `done`
        S.t_loop = lt.wall_s; S.loop_ms_events = lt.events_ms; S.xupdate_launches = lt.launched;
        S.exchange_variant = !cshard ? 0 : (!peer_fused ? 1 : (peer_one ? 3 : 2));
        if (persist) {
            int herr = 0;
            ADMM_HIP_CHECK(hipMemcpy(&herr, perr.get(), sizeof(int), hipMemcpyDeviceToHost));
            if (herr) throw Error(ADMM_ERR_INTERNAL, "wide solver: a hand-over inside the persistent active-set launch timed out");
            unsigned long long hs[16] = {0};
            ADMM_HIP_CHECK(hipMemcpy(hs, pstat.get(), sizeof(hs), hipMemcpyDeviceToHost));
            S.persist_iter = (long long)hs[0];
            pstat.zero(st);
            if (std::getenv("ADMM_HIP_WIDE_PERSIST_STATS")) {
                const double it = hs[0] ? (double)hs[0] : 1.0;
                std::fprintf(stderr, "[wide persist] %llu iterations in %llu stretches (%llu of them not on one XCD), %.2f us per iteration inside; %lld host iterations enqueued\n"
                             "[wide persist] per iteration, workgroup 0: columns %.2f | combine + publish partial %.2f | wait partials %.2f | rows + publish %.2f | wait rows %.2f | decide + t %.2f us\n",
                             hs[0], hs[1], hs[3], hs[0] ? 0.01 * (double)hs[2] / it : 0.0, (long long)lt.launched,
                             0.01 * hs[4] / it, 0.01 * hs[5] / it, 0.01 * hs[6] / it, 0.01 * hs[7] / it, 0.01 * hs[8] / it, 0.01 * hs[9] / it);
            }
        }
#ifdef ADMM_HIP_PROBE
        if (const char* f = std::getenv("ADMM_HIP_PROBE_OUT")) {
            std::vector<long long> hp((size_t)4096 * 4 * 8);
            ADMM_HIP_CHECK(hipMemcpy(hp.data(), probe.get(), hp.size() * sizeof(long long), hipMemcpyDeviceToHost));
            if (FILE* fp = std::fopen(f, "wb")) { std::fwrite(hp.data(), sizeof(long long), hp.size(), fp); std::fclose(fp); }
        }
#endif

        res.niter.assign(nlam, 0);
        ADMM_HIP_CHECK(hipMemcpy(res.niter.data(), niter.get(), nlam * sizeof(int), hipMemcpyDeviceToHost));
        std::vector<float> hb((size_t)nlam * p);
        ADMM_HIP_CHECK(hipMemcpy(hb.data(), beta.get(), hb.size() * sizeof(float), hipMemcpyDeviceToHost));
        const size_t pt1 = (size_t)p_total + 1;
        res.beta.assign(pt1 * nlam, 0.f);
        std::vector<double> icpt(nlam, 0.0);                          // sum_j beta_j meanX_j over this rank's columns
        long long tot = 0;
        for (int l = 0; l < nlam; ++l) {
            float b0 = 0.f;
            recover_coef<float>(d, hb.data() + (size_t)l * p, &b0, res.beta.data() + (size_t)l * pt1 + 1 + col_offset);
            res.beta[(size_t)l * pt1] = b0;
            icpt[l] = (double)d.meanY - (double)b0;
            tot += res.niter[l];
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
            ADMM_HIP_CHECK(hipMemcpyAsync(res.beta.data(), db.get(), res.beta.size() * sizeof(float), hipMemcpyDeviceToHost, st));
            ADMM_HIP_CHECK(hipMemcpyAsync(icpt.data(), di.get(), nlam * sizeof(double), hipMemcpyDeviceToHost, st));
            ADMM_HIP_CHECK(hipStreamSynchronize(st));
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
