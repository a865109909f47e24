// Tall (n > p) Lasso / Elastic-net lambda path, device resident.
//
// Replaces, for n > p:  ADMMLassoTall / ADMMEnetTall driven by FADMMBase::solve
//   /root/reference/src/FADMMBase.h:185-265, ADMMLassoTall.h:55-231, ADMMEnet.h:19-58,
//   and the lambda loop of Lasso.cpp:97-124.
//
// Design (MI355X-first, not a translation):
//  * rho is fixed along the whole path (ADMMLassoTall.h:97, init only at i == 0), so the
//    Cholesky solve of (X'X + rho I) is replaced by ONE cached symmetric inverse Minv (p x p
//    fp32, 4p^2 bytes) and the x-update becomes a bandwidth-bound dense mat-vec.
//  * Goldstein acceleration/restart makes the right-hand side depend on a decision taken from
//    global norms (FADMMBase.h:243-256), but only between TWO vectors that are both known one
//    iteration earlier: the momentum ratio (a-1)/a' follows from the previous a alone, so
//        u = fl(fl(X'y - adj_y) + rho adj_z)  with adj = (1+ratio) new - ratio old     ("accelerate")
//        w = fl(fl(X'y - y_old) + rho z_old)                                          ("restart": adj = old)
//    are formed by the tail of the previous iteration exactly as ADMMLassoTall.h:70-80 rounds them.
//    The x-update kernel streams Minv ONCE against the pair (u, w), a = Minv u, b = Minv w, and the
//    tail picks x = a or x = b.  The scalar decision itself (norm reduction, convergence,
//    acceleration/restart, lambda schedule) runs as ONE EXTRA WORKGROUP of the x-update launch,
//    concurrently with the streaming workgroups.  No host round trip, no grid barrier, no atomics:
//    two launches per ADMM iteration.
//    (An earlier version used x = Minv u' + tau Minv w' with u' = X'y - y + rho z: equal in exact
//    arithmetic, but it bypasses the float rounding of the right-hand side, and with it the dead
//    band that lets the reference's stopping rule fire when rho * ulp(z) exceeds eps_dual -- tiny
//    lambda, unstandardised data.  tests/tools/fuzz_parity.py found paths running to maxit.)
//  * Minv is symmetric: for p >= 2048 the x-update reads only its lower triangle (symv_kernels.h,
//    2p^2 bytes); small problems use the full-matrix gemv_t (fewer, larger workgroups).
//  * Convergence test, acceleration scalars, the lambda schedule (init_warm), niter[] and the
//    beta snapshot all live on the device; the host only enqueues iteration batches and polls
//    a `done` word asynchronously.
//  * After a converged lambda the reference re-solves with an unchanged right-hand side
//    (adj_z/adj_y are not updated on the exit iteration, FADMMBase.h:237-238), so the first
//    x of the next lambda equals the last x: `mode == 0` reuses it instead of a mat-vec.
#include "prep.h"
#include "gemv_kernels.h"
#include "symv_kernels.h"
#include "solvers.h"
#include "comm.h"
#include "loop_driver.h"
#include "peer_device.h"
#include "probe.h"

namespace admm {

struct TallCtl {
    double rho, lam, eps_primal, eps_dual, adj_a, adj_c, tau;
    double a_next, tau_next;   // a' and ratio (a - 1) / a' the NEXT decision uses if it accelerates (computed once, here)
    int mode;       // 1: x = a (accelerate) or b (restart), adj from (cur, old);  0: keep stored x / adj (first iteration after convergence)
    int restart;    // with mode 1: adj = old exactly
    int iter;       // index i of the iteration this block describes
    int lam_idx;
    int done;
    int first;
    int total;
    int fin_idx;    // >= 0: the lambda that finished at this decision (snapshot z, record niter)
    int fin_niter;
    int pad;
};

struct TallParams {
    int p, nwg, nseg, maxit, nlam, enet;
    long long part_stride;
    double eps_abs, eps_rel, alpha, sqrt_p;
    const double* lambdas;      // device, nlam values already rounded to float
    const float* XY;
    const float* a_part; const float* b_part;            // gemv_t partials [nseg][part_stride]   (full-matrix x-update)
    const float* dot0; const float* dot1; const float* axp0; const float* axp1;   // symv partials (lower-triangle x-update)
    long long ldo; int nrb, p32; SymvSched sched;
    float* x; float* z0; float* z1; float* y0; float* y1; float* adj_z; float* adj_y; float* u; float* w;
    TallCtl* ctl;               // [2]
    double* P;                  // [2][nwg][8]
    float* beta;                // [nlam][p] snapshots of z (standardised scale)
    int* niter;                 // [nlam]
    int* done_host;             // pinned host word set when the path has finished (loop_driver.h: PinnedFlag)
#ifdef ADMM_HIP_PROBE
    long long* probe;           // dev build only (probe.h)
#endif
    double* trace;              // optional [trace_cap][kTraceFields] decision records (admm_hip_lasso_plan_trace_*), or NULL
    long long trace_cap;
    const float* refine_ab;     // refined x-update (ADMM_HIP_REFINE=1): the first solve's a | b (leading dimension refine_ld), to which the tail adds
    long long refine_ld;        // the correction it sums from the partials; NULL otherwise
    float* state;               // optional [state_cap][5][p] iterates x, z, y, adj_z, adj_y of every iteration (admm_hip_lasso_plan_state_*), or NULL
    long long state_cap;
};

constexpr int kTailThreads = 256;
constexpr int kTailLanes = kSySumLanes;   // lanes cooperating on one element's partial sums
constexpr int kTailElems = kTailThreads / kTailLanes;

// Scalar control of one iteration, run by a single workgroup: reduce the previous iteration's norm
// partials, evaluate convergence / acceleration / restart / lambda schedule, publish ctl[par ^ 1].
// IN_LAUNCH: the norm partials were written (write-through) by other workgroups of the SAME launch (single-launch
// iteration below): they are read with agent-scope loads that bypass this XCD's L2.
__device__ __forceinline__ double tall_load_partial(const double* p, bool in_launch) {
    if (!in_launch) return *p;
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void tall_store_wt(float* p, float v) {       // write-through store (visible to other XCDs without a fence)
    __hip_atomic_store(reinterpret_cast<unsigned int*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void tall_store_wt(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool IN_LAUNCH = false>
__device__ void tall_decide(const TallParams& q, int par) {
    __shared__ double dscratch[6 * (kTailThreads / 64)];
    WIDE_PROBE_DECL
    WIDE_PROBE(0);
    const TallCtl in = q.ctl[par];
    TallCtl* outp = &q.ctl[par ^ 1];
    if (in.done) {
        if (threadIdx.x == 0) { TallCtl o = in; o.fin_idx = -1; *outp = o; }   // keep `done` sticky in both slots
        return;
    }
    WIDE_PROBE(1);
    double acc[6] = {0, 0, 0, 0, 0, 0};
    const double* Pin = q.P + (size_t)par * q.nwg * 8;
    for (int w = threadIdx.x; w < q.nwg; w += kTailThreads) {
#pragma unroll
        for (int k = 0; k < 6; ++k) acc[k] += tall_load_partial(Pin + (size_t)w * 8 + k, IN_LAUNCH);
    }
    block_sum<double, 6>(acc, dscratch);
    if (threadIdx.x != 0) return;
    WIDE_PROBE(2);
    const double r2 = acc[0], dz2 = acc[1], daz2 = acc[2], x2 = acc[3], z2 = acc[4], y2 = acc[5];
    double tr_rp = 0, tr_rd = 0, tr_c = 0, tr_code = ADMM_TRACE_COLD;
    TallCtl out = in;
    out.first = 0;
    out.fin_idx = -1; out.fin_niter = 0;
    if (!in.first) {
        const double rp = sqrt(r2);                    // resid_primal            FADMMBase.h:208
        const double rd = in.rho * sqrt(dz2);          // resid_dual              ADMMLassoTall.h:150-153
        tr_rp = rp; tr_rd = rd;
        if (rp < in.eps_primal && rd < in.eps_dual) {  // converged()             FADMMBase.h:213-217
            out.fin_idx = in.lam_idx; out.fin_niter = in.iter + 1;
            out.mode = 0;
            tr_code = ADMM_TRACE_CONVERGED;
        } else {
            const double old_c = in.adj_c;
            const double c = in.rho * rp * rp + in.rho * daz2;     // compute_resid_combined  ADMMLassoTall.h:154-161
            tr_c = c;
            if (c < 0.999 * old_c) {                   // FADMMBase.h:243-249
                out.adj_a = in.a_next; out.adj_c = c; out.tau = in.tau_next; out.restart = 0;
                tr_code = ADMM_TRACE_ACCELERATE;
            } else {                                   // restart                 FADMMBase.h:250-256
                out.adj_a = 1.0; out.adj_c = old_c / 0.999; out.tau = -1.0; out.restart = 1;
                tr_code = ADMM_TRACE_RESTART;
            }
            out.mode = 1;
            out.iter = in.iter + 1;
            if (in.iter + 1 >= q.maxit) {              // loop ran out: `return i + 1` with i == maxit
                out.fin_idx = in.lam_idx; out.fin_niter = q.maxit + 1;
            }
        }
        if (out.fin_idx >= 0) {                        // next lambda: init_warm keeps x,z,y,adj,a,c,rho (ADMMLassoTall.h:219-230)
            out.lam_idx = in.lam_idx + 1;
            out.iter = 0;
            if (out.lam_idx >= q.nlam) out.done = 1;
            else out.lam = q.lambdas[out.lam_idx];
            q.niter[out.fin_idx] = out.fin_niter;
        }
    } else {
        out.mode = 1; out.tau = 0.0; out.restart = 0;  // cold start: adj = 0, x = Minv X'y
    }
    // eps for the iteration about to run, from the CURRENT iterate (FADMMBase.h:187-188, ADMMLassoTall.h:141-149)
    out.eps_primal = fmax(sqrt(x2), sqrt(z2)) * q.eps_rel + q.sqrt_p * q.eps_abs;
    out.eps_dual = sqrt(y2) * q.eps_rel + q.sqrt_p * q.eps_abs;
    out.a_next = 0.5 + 0.5 * sqrt(1.0 + 4.0 * out.adj_a * out.adj_a);
    out.tau_next = (out.adj_a - 1.0) / out.a_next;
    out.total = in.total + 1;
    *outp = out;
    if (out.done) *q.done_host = 1;
    WIDE_PROBE(3);
    WIDE_PROBE_FLUSH(2, in.total);
    if (q.trace != nullptr && in.total < q.trace_cap) {      // what FADMMBase.h:135-170 (print_row, commented out there) would print
        double* t = q.trace + (size_t)in.total * ADMM_TRACE_FIELDS;
        t[0] = in.lam_idx; t[1] = in.iter; t[2] = in.eps_primal; t[3] = in.eps_dual; t[4] = tr_rp; t[5] = tr_rd;
        t[6] = tr_c; t[7] = in.adj_c; t[8] = tr_code; t[9] = in.rho; t[10] = in.rho; t[11] = in.lam;
    }
}

struct TallDecideExtra {
    static constexpr bool kHas = true;
    TallParams q; int par;
    __device__ void operator()() const { tall_decide<false>(q, par); }
};

// (1 + ratio) * cur - ratio * old without contraction, like the reference build (FADMMBase.h:247-248): the
// same value is formed twice, as the candidate right-hand side and as adj of the next iteration.
__device__ __forceinline__ float tall_extrapolate(float t1, float t, float cur, float old) {
    // plain operators under contract(off): HIP's __fmul_rn / __fsub_rn are inline `x * y` / `x - y` written in a header that is
    // compiled with contraction allowed, and the two were fused into v_pk_fma_f32 all the same
#pragma clang fp contract(off)
    const float a = t1 * cur;
    const float b = t * old;
    return a - b;
}

// State of one coordinate and its update: everything after the x-update in one iteration
// (next_z, residual, dual update, norms, right-hand sides of the next x-update).
struct TallElem { float zc, yc, zo, yo, adjz, adjy, x, xy; };

__device__ __forceinline__ TallElem tall_load_elem(const TallParams& q, int par, int i) {
    const float* zc_ = par ? q.z1 : q.z0; const float* yc_ = par ? q.y1 : q.y0;
    const float* zo_ = par ? q.z0 : q.z1; const float* yo_ = par ? q.y0 : q.y1;
    TallElem e;
    e.zc = zc_[i]; e.yc = yc_[i]; e.zo = zo_[i]; e.yo = yo_[i]; e.adjz = q.adj_z[i]; e.adjy = q.adj_y[i]; e.x = q.x[i]; e.xy = q.XY[i];
    return e;
}

template <bool WT = false>      // WT: u, w are consumed by other workgroups of the same launch -> write-through stores
__device__ __forceinline__ void tall_update_elem(const TallParams& q, const TallCtl& c, int par, int i, const TallElem& e, float a, float b, double (&acc)[6]) {
    // The reference is built without fused multiply-adds (R's default flags on x86-64: no -march, /root/reference/src/Makevars):
    // every product below rounds before it is added, as there.  (__fmul_rn / __fadd_rn are plain operators in HIP and were
    // contracted all the same: found by the stepwise check of oracle/stepcheck.py, round 3.)
#pragma clang fp contract(off)
    float* zo_ = par ? q.z0 : q.z1; float* yo_ = par ? q.y0 : q.y1;
    const float zc = e.zc, yc = e.yc, zo = e.zo, yo = e.yo;
    if (c.fin_idx >= 0) q.beta[(size_t)c.fin_idx * q.p + i] = zc;     // get_z() snapshot (Lasso.cpp:108)
    if (c.done) return;
    float adjz, adjy, x;
    if (c.mode) {
        if (c.restart) { adjz = zo; adjy = yo; x = b; }
        else {
            const float t = (float)c.tau, t1 = (float)(1.0 + c.tau);
            adjz = tall_extrapolate(t1, t, zc, zo);       // (1 + ratio) * aux_z - ratio * old_z   FADMMBase.h:247-248
            adjy = tall_extrapolate(t1, t, yc, yo);
            x = a;
        }
    } else { adjz = e.adjz; adjy = e.adjy; x = e.x; }
    const float rho_f = (float)c.rho;
    const float vec = x + adjy / rho_f;        // next_z: main_x + adj_y / rho            ADMMLassoTall.h:83
    const double pen = c.lam / c.rho;
    float zn;
    if (!q.enet) {                             // soft_threshold, double compare          ADMMLassoTall.h:55-69
        const double v = (double)vec;
        zn = v > pen ? (float)(v - pen) : (v < -pen ? (float)(v + pen) : 0.f);
    } else {                                   // enet()                                  ADMMEnet.h:24-40
        const float thresh = (float)(q.alpha * pen);
        const float denom = (float)(1.0 + pen * (1.0 - q.alpha));
        zn = vec > thresh ? (vec - thresh) / denom : (vec < -thresh ? (vec + thresh) / denom : 0.f);
    }
    const float r = x - zn;                    // next_residual                            ADMMLassoTall.h:86-95
    const float yn = adjy + rho_f * r;         // dual_y = adj_y + rho * newr              FADMMBase.h:210
    const float dz = zn - zc, daz = zn - adjz;
    acc[0] = (double)r * r; acc[1] = (double)dz * dz; acc[2] = (double)daz * daz;
    acc[3] = (double)x * x; acc[4] = (double)zn * zn; acc[5] = (double)yn * yn;
    q.x[i] = x; zo_[i] = zn; yo_[i] = yn; q.adj_z[i] = adjz; q.adj_y[i] = adjy;
    if (q.state != nullptr && c.total < q.state_cap) {     // record c.total = the trace record that will judge this iteration
        float* s = q.state + (size_t)c.total * 5 * q.p;
        s[i] = x; s[q.p + i] = zn; s[2 * (size_t)q.p + i] = yn; s[3 * (size_t)q.p + i] = adjz; s[4 * (size_t)q.p + i] = adjy;
    }
    // both possible right-hand sides of the next x-update, rounded as ADMMLassoTall.h:70-80 does
    const float tn = (float)c.tau_next, tn1 = (float)(1.0 + c.tau_next);
    const float adjz_a = tall_extrapolate(tn1, tn, zn, zc), adjy_a = tall_extrapolate(tn1, tn, yn, yc);
    const float un = (float)((double)(e.xy - adjy_a) + c.rho * (double)adjz_a);
    const float wn = (float)((double)(e.xy - yc) + c.rho * (double)zc);
    if (WT) { tall_store_wt(q.u + i, un); tall_store_wt(q.w + i, wn); }
    else { q.u[i] = un; q.w[i] = wn; }
}

// Element-wise part of one iteration.  `c` = the control block published by this iteration's decision.
// MODE 0: x-update results as gemv_t partial rows; 1: as the symmetric mat-vec's partial arrays; 2: row-sharded over the
// PEER exchange -- wait for the K flags, then sum the K ranks' shares straight out of the exchange slots.
enum { TAIL_GEMV = 0, TAIL_SYMV = 1, TAIL_PEER = 2, TAIL_PEER1 = 3 };
// TAIL_PEER1: producer and consumer of the exchange in ONE launch -- every workgroup sums its elements of this rank's share,
// writes them into every rank's slot and counts itself in (the last one raises the flags), then waits for the K flags like
// TAIL_PEER.  A workgroup waits for flags that need ALL workgroups of this launch (on every rank) to have published, so
// the launch must be resident as a whole; the host only chooses it when the grid is at most half of what the device holds.
template <int MODE>
__global__ void __launch_bounds__(kTailThreads)
tall_tail_kernel(TallParams q, int par, PeerExchange ex) {
    __shared__ double scratch[6 * (kTailThreads / 64)];
    WIDE_PROBE_DECL
    WIDE_PROBE(0);
    const TallCtl c = q.ctl[par ^ 1];
    // Every load below is independent of `c` (the ping-pong parity equals the launch parity because
    // the decision advances `total` once per launch), so the whole kernel is one memory round trip.
    const int sub = threadIdx.x & (kTailLanes - 1);
    const int i = blockIdx.x * kTailElems + threadIdx.x / kTailLanes;
    const bool valid = i < q.p;
    const bool owner = valid && sub == 0;
    TallElem e = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (owner) e = tall_load_elem(q, par, i);
    // ---- x-update results a = Minv u, b = Minv w: kTailLanes lanes share one element and issue all
    // their partial loads at once, then combine with shuffles.
    float a = 0.f, b = 0.f;
    if (MODE == TAIL_SYMV) {
        symv_sum_partials<kTailLanes>(q.dot0, q.dot1, q.axp0, q.axp1, q.ldo, q.nrb, q.sched, q.p32, i, sub, valid, a, b);
        if (q.refine_ab != nullptr && valid) {            // x = x1 + Minv (rhs - M x1): the partials hold the correction
            a = q.refine_ab[i] + a;
            b = q.refine_ab[q.refine_ld + i] + b;
        }
    } else if (MODE == TAIL_PEER || MODE == TAIL_PEER1) {
        const bool live = !q.ctl[par].done;                          // finished in an earlier launch: nothing pushed, nothing to wait for (replicated flag: all ranks agree)
        if (MODE == TAIL_PEER1 && live) {
            float sa, sb;
            symv_sum_partials<kTailLanes>(q.dot0, q.dot1, q.axp0, q.axp1, q.ldo, q.nrb, q.sched, q.p32, i, sub, valid, sa, sb);
            if (valid) {
                for (int dst = sub; dst < ex.nranks; dst += kTailLanes)
                    peer_store_f32x2(reinterpret_cast<float*>(peer_dst_slot(ex, dst)) + 2 * (size_t)i, sa, sb);
            }
            peer_publish(ex, gridDim.x);
        }
        // the producing launch (or the block above) pushed unless the solve was already finished
        const bool ok = live ? peer_wait_relaxed(ex) : false;
        if (ok && valid) {
            for (int r = sub; r < ex.nranks; r += kTailLanes) {          // rank order fixed by the lane pattern: identical on every rank
                const float2 v = peer_load_f32x2(reinterpret_cast<const float*>(peer_src_slot(ex, r)) + 2 * (size_t)i);    // (a_i, b_i) interleaved
                a += v.x; b += v.y;
            }
        }
#pragma unroll
        for (int m = 1; m < kTailLanes; m <<= 1) { a += __shfl_xor(a, m, 64); b += __shfl_xor(b, m, 64); }
    } else {
        if (valid) {
            for (int k = sub; k < q.nseg; k += kTailLanes) {
                const size_t o = (size_t)k * q.part_stride + i;
                a += q.a_part[o]; b += q.b_part[o];
            }
        }
#pragma unroll
        for (int m = 1; m < kTailLanes; m <<= 1) { a += __shfl_xor(a, m, 64); b += __shfl_xor(b, m, 64); }
    }
    if (c.done && c.fin_idx < 0) return;
    WIDE_PROBE(1);

    double acc[6] = {0, 0, 0, 0, 0, 0};
    if (owner) tall_update_elem(q, c, par, i, e, a, b, acc);
    if (c.done) return;
    WIDE_PROBE(2);
    // Block sum of the six norms.  Only the owner lanes (sub == 0) hold values, so wave_sum's xor-4 / 2 / 1 steps would add
    // exact zeros: the top half of the halving butterfly (xor 32 / 16 / 8) leaves the wave total of value k in lane 8 k,
    // bit-identical to block_sum<double, 6> at 7 exchanges instead of 36.
    static_assert(kTailLanes == 8, "owner lanes are the multiples of 8");
    {
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
        const double v8[8] = {acc[0], acc[1], acc[2], acc[3], acc[4], acc[5], 0.0, 0.0};
        const double tot = halving_sum8_top(v8, lane);
        if ((lane & 7) == 0 && lane < 48) scratch[(lane >> 3) * (kTailThreads / 64) + wid] = tot;
        __syncthreads();
        if (threadIdx.x < 6) {
            double sum = 0;
            for (int ww = 0; ww < kTailThreads / 64; ++ww) sum += scratch[threadIdx.x * (kTailThreads / 64) + ww];
            q.P[((size_t)(par ^ 1) * q.nwg + blockIdx.x) * 8 + threadIdx.x] = sum;
        }
    }
    WIDE_PROBE(3);
    WIDE_PROBE_FLUSH(blockIdx.x == 0 ? 0 : (blockIdx.x == gridDim.x - 1 ? 1 : -1), c.total - 1);
}

// Round 4 also built the whole path as ONE persistent launch with a third of the inverse's triangle resident in registers (two
// grid barriers per iteration, decision off the critical path; bit-identical): 66.7 us per iteration against 39.25 at C2 -- the
// barriers + an in-launch tail cost 16.6 us where the kernel boundaries cost 5.6, residency saves at most 10.8 us of the stream, and
// the stream itself falls to 2.5 TB/s with half the registers taken.  Measurements and the bound that rules out tuning it into a
// win: profiles/r04_tall_persist.md; the code: commit 6d31127.

// Row-sharded mode: this rank's share of the two products (the partial arrays of its tiles) summed into ab[2][ld], the
// vectors the ranks then all-reduce.  Same lane geometry and order as the single-GPU tail.
__global__ void __launch_bounds__(kTailThreads)
tall_shard_reduce_kernel(TallParams q, float* ab, long long ld, const int* skip) {
    if (*skip) return;
    const int sub = threadIdx.x & (kTailLanes - 1);
    const int i = blockIdx.x * kTailElems + threadIdx.x / kTailLanes;
    float a, b;
    symv_sum_partials<kTailLanes>(q.dot0, q.dot1, q.axp0, q.axp1, q.ldo, q.nrb, q.sched, q.p32, i, sub, i < q.p, a, b);
    if (i < q.p && sub == 0) { ab[i] = a; ab[ld + i] = b; }
}

// Refined x-update, residual step: r = rhs - M x1 for both candidates, M x1 from the double partial arrays of
// symv2_lower_f64acc_kernel, the subtraction in double, the result rounded to float once.
__global__ void __launch_bounds__(kTailThreads)
tall_refine_resid_kernel(TallParams q, const double* d0, const double* d1, const double* x0, const double* x1, float* ru, float* rw, const int* skip) {
    if (*skip) return;
    const int sub = threadIdx.x & (kTailLanes - 1);
    const int i = blockIdx.x * kTailElems + threadIdx.x / kTailLanes;
    double ma, mb;
    symv_sum_partials<kTailLanes, double>(d0, d1, x0, x1, q.ldo, q.nrb, q.sched, q.p32, i, sub, i < q.p, ma, mb);
    if (i < q.p && sub == 0) { ru[i] = (float)((double)q.u[i] - ma); rw[i] = (float)((double)q.w[i] - mb); }
}

// The same over the PEER exchange, without launches of the exchange layer: every workgroup writes its elements of
// (a, b) straight into this rank's slot of EVERY rank's buffer (lane `sub` of an element's group serves rank sub,
// sub + 8, ...), and the last workgroup to finish raises the flags (peer_device.h).
__global__ void __launch_bounds__(kTailThreads)
tall_shard_push_kernel(TallParams q, PeerExchange ex, long long ld, const int* skip) {
    if (*skip) return;
    const int sub = threadIdx.x & (kTailLanes - 1);
    const int i = blockIdx.x * kTailElems + threadIdx.x / kTailLanes;
    float a, b;
    symv_sum_partials<kTailLanes>(q.dot0, q.dot1, q.axp0, q.axp1, q.ldo, q.nrb, q.sched, q.p32, i, sub, i < q.p, a, b);
    if (i < q.p) {
        for (int dst = sub; dst < ex.nranks; dst += kTailLanes)
            peer_store_f32x2(reinterpret_cast<float*>(peer_dst_slot(ex, dst)) + 2 * (size_t)i, a, b);       // (a_i, b_i) interleaved: one 8-byte store
    }
    peer_publish(ex, gridDim.x);
}

__global__ void tall_init_kernel(TallParams q, double rho, double lam0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < q.p) {
        q.x[i] = 0.f; q.z0[i] = 0.f; q.z1[i] = 0.f; q.y0[i] = 0.f; q.y1[i] = 0.f;
        q.adj_z[i] = 0.f; q.adj_y[i] = 0.f; q.u[i] = q.XY[i]; q.w[i] = 0.f;
    }
    if (i < 2 * q.nwg * 8) q.P[i] = 0.0;
    if (i == 0) {
        TallCtl c;
        c.rho = rho; c.lam = lam0; c.eps_primal = 0.0; c.eps_dual = 0.0;
        c.adj_a = 1.0; c.adj_c = 9999.0; c.tau = 0.0;
        c.a_next = 0.5 + 0.5 * sqrt(5.0); c.tau_next = 0.0;
        c.mode = 1; c.restart = 0; c.iter = 0; c.lam_idx = 0; c.done = 0; c.first = 1; c.total = 0; c.fin_idx = -1; c.fin_niter = 0; c.pad = 0;
        q.ctl[0] = c; q.ctl[1] = c;
    }
}

// ----------------------------------------------------------------------------------------------
struct TallPlan final : LassoPlan {
    DeviceData<float> d;
    LassoProblem pb;
    hipStream_t st;
    admm_stats setup_stats{};
    int p = 0, nlam = 0, nwg = 0;
    long long ldp = 0;
    double rho = 0;
    std::vector<double> lam_user, lam_int;
    GemvTPlan pl;
    SymvPlan sy;
    bool use_sym = false;
    bool shard = false;                                 // x-update spread over the ranks of the attached communicator
    bool peer_fused = false;                            // ... with the exchange done by the solver's own kernels (PEER backend)
    bool peer_one = false;                              // ... producer and consumer in one launch (tall_tail_kernel<TAIL_PEER1>)
    CommInfo ci;
    DevBuf<float> ab;                                   // [2][ldp] this rank's share of (a, b), all-reduced in place
    double dist_flops = 0;                              // distributed factorisation: flops this rank performed (0: replicated)
    long long ldv = 0;
    DevBuf<float> XY, M, a_part, b_part, x, z0, z1, y0, y1, adj_z, adj_y, u, w, beta;
    DevBuf<int> niter;
    DevBuf<double> P, dlam;
    DevBuf<TallCtl> ctl;
    TallParams q{};
    DevBuf<double> trace;
    long long trace_cap = 0, trace_n = 0;
    DevBuf<float> state;
    long long state_cap = 0;
    // mixed-precision refinement of the x-update (ADMM_HIP_REFINE=1)
    bool refine = false;
    DevBuf<float> Mg, rab, ruw;                          // the float system X'X + rho I; first solve a | b; residuals r_u | r_w
    DevBuf<double> dD0, dD1, xD0, xD1;                   // double partial arrays of M x1
    TallCtl* hctl = nullptr;
    PinnedFlag hflag;
#ifdef ADMM_HIP_PROBE
    DevBuf<long long> probe;
#endif
    float* hbeta = nullptr;                             // pinned landing buffer of the beta snapshots (nlam x p)

    std::vector<hipEvent_t> ev_pool;                    // start/stop events of sampled x-update launches, reused by every run()

    ~TallPlan() override {
        if (hctl) (void)hipHostFree(hctl);
        if (hbeta) (void)hipHostFree(hbeta);
        for (auto e : ev_pool) (void)hipEventDestroy(e);
    }

    TallPlan(DeviceData<float>&& data, const LassoProblem& prob, hipStream_t stream) : d(std::move(data)), pb(prob), st(stream) {
        const int n = d.n;
        p = d.p;
        admm_stats& S = setup_stats;
        S.branch = 0;
        S.t_h2d = d.t_h2d; S.t_standardize = d.t_std;
        ldp = round_up(p, 128);                         // whole 128-row blocks for the matrix-core setup kernels

        // X'y and lambda_0 (ADMMLassoTall.h:172-173; ADMMEnet.h:56 divides by alpha + 1e-4)
        // Row-sharded mode (not in the reference): d holds this rank's ROWS of the globally standardised
        // data.  X'y and X'X are sums over the row blocks (split-K over the ranks + one all-reduce each); the Lanczos call
        // and the factorisation are replicated (identical inputs -> identical rho and inverse on every rank); each rank
        // then streams 1/nranks of the inverse's lower-triangle tiles per iteration.
        shard = pb.dist;
        ci = shard ? comm_info() : CommInfo();
        const long long nt = d.n_total > 0 ? d.n_total : n;
        if (d.xy.get()) {                                  // Gram-form data (a cross-validation fold formed as a down-date, cv.hip)
            ADMM_REQUIRE(!shard && d.gram.get() && d.ldgram == ldp, "Gram-form data needs X'X next to X'y");
            XY = std::move(d.xy);
        } else {
            XY.alloc(ldp); XY.zero(st);
            gemv_t_simple<float>(d.X.get(), d.ldx, n, p, d.Y.get(), XY.get(), st);
            if (shard) allreduce_sum_f32(XY.get(), (size_t)p, st);
        }
        float lambda0 = device_absmax<float>(XY.get(), p, st);
        if (pb.enet) lambda0 = (float)(lambda0 / ((double)(float)pb.alpha + 0.0001));

        // lambda grid (Lasso.cpp:78-89) and internal lambdas (Lasso.cpp:99), stored as float like `Scalar lambda`
        lam_user = make_lambda_grid(pb, lambda0, (int)nt, (double)d.scaleY);
        nlam = (int)lam_user.size();
        lam_int.resize(nlam);
        for (int i = 0; i < nlam; ++i) lam_int[i] = (double)(float)(lam_user[i] * (double)nt / (double)d.scaleY);

        // Memory wall: this solver keeps (X'X + rho I)^-1, ldp^2 floats, next to X (until the loop starts), the Gram and -- below
        // p = 4096 or with ADMM_HIP_INVERSE=f64 -- a double copy during the factorisation.  Refuse clearly instead of failing
        // inside some allocation (ADMM_HIP_TEST_FREE_BYTES: test hook that pretends the device has that much memory left).
        {
            size_t free_b = 0, total_b = 0;
            ADMM_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
            free_b += pool_cached_bytes();                        // blocks this library holds for re-use are free for it
            if (const char* e = option("TEST_FREE_BYTES")) free_b = (size_t)std::atoll(e);
            const bool have_gram = d.gram.get() && d.ldgram == ldp;
            const bool inv64_wanted = p < 4096 || (option("INVERSE") && std::string(option("INVERSE")) == "f64");
            const double mat = (double)ldp * (double)ldp * 4.0;
            const double need = (have_gram ? 0.0 : mat) + (inv64_wanted ? 2.0 * mat : 0.5 * mat) + (option("REFINE") ? mat : 0.0) +
                                (shard && ci.nranks > 1 ? mat + mat / ci.nranks : 0.0);      // packed send / receive buffers of the Gram's reduce-scatter
            if (need > 0.97 * (double)free_b) {
                char msg[320];
                std::snprintf(msg, sizeof msg, "tall solver: the cached %d x %d inverse and its workspace need %.1f GB, the device has %.1f GB free; "
                              "use $parallel() (row blocks, admm_hip_parlasso) or fewer columns", p, p, need / 1e9, (double)free_b / 1e9);
                throw Error(ADMM_ERR_MEMORY, msg);
            }
        }
        // Gram (cross_prod_lower, ADMMLassoTall.h:191-192) -- both triangles
        double t0 = now_s();
        const bool gram_given = d.gram.get() && d.ldgram == ldp;
        if (gram_given) {                                 // computed under the host-to-device transfer (upload_standardize_gram_f32)
            M = std::move(d.gram);
            S.t_gram = d.t_gram_tail;
        } else {
            M.alloc((size_t)ldp * ldp); M.zero(st);
            gram_full<float>(d.X.get(), d.ldx, n, p, true, M.get(), ldp, st);      // sharded: this rank's term of the split-K sum (reduced below)
            comm_stream_sync(st);
            S.t_gram = now_s() - t0;
        }
        // Row-sharded solver: decided here because it chooses how the split-K Gram is reduced (below)
        bool inv64 = p < 4096;
        if (const char* e = option("INVERSE")) inv64 = std::string(e) == "f64";
        bool dist_factor = shard && ci.nranks > 1 && !inv64 && (p + 127) / 128 >= 2 * ci.nranks && p >= 256;
        if (const char* e = option("DIST_FACTOR")) dist_factor = dist_factor && std::string(e) != "0";

        // rho (ADMMLassoTall.h:194-202)
        rho = pb.opts.rho;
        t0 = now_s();
        if (rho <= 0) {
            SymMatVec<float> op(M.get(), ldp, p, st);
            int nmatop = 0;
            // Row-sharded: M still holds this rank's term X_r'X_r of the Gram, and the product is (sum_r X_r'X_r) v = sum_r (X_r'X_r v):
            // the local product, then ONE all-reduce of p floats per Lanczos step (3-5 steps) -- so the whole matrix never has to exist
            // on any rank (round 3 all-reduced the p x p Gram first).  Identical sums in identical order on every rank: identical rho.
            DevBuf<float> wsum;
            if (shard && ci.nranks > 1) wsum.alloc(ldp);
            const float ev = lanczos_largest_f32([&](const float* v, float* w_) {
                op(v, w_);
                if (shard && ci.nranks > 1) {
                    ADMM_HIP_CHECK(hipMemcpyAsync(wsum.get(), w_, (size_t)p * sizeof(float), hipMemcpyHostToDevice, st));
                    allreduce_sum_f32(wsum.get(), (size_t)p, st);
                    ADMM_HIP_CHECK(hipMemcpyAsync(w_, wsum.get(), (size_t)p * sizeof(float), hipMemcpyDeviceToHost, st));
                    comm_stream_sync(st);
                    comm_check();
                }
            }, p, &nmatop);
            S.eig_est = ev;
            rho = std::pow((double)ev, 1.0 / 3) * std::pow(lam_int[0], 2.0 / 3);
        }
        S.rho = rho;
        S.t_eigs = now_s() - t0;

        // Row-sharded: the split-K sum of the Gram over the ranks.  With the distributed factorisation a rank
        // only ever reads the block columns it owns (k mod N == rank: chol_inverse.h), so the sum is a REDUCE-SCATTER: the block columns
        // are packed owner by owner (a block column is 128 x ldp contiguous floats), every rank receives the sum of its own -- 1 / N of the
        // matrix -- and unpacks it in place; the other block columns of M are dead from here on (the panels arrive by broadcast).
        // Replicated factorisation (ADMM_HIP_DIST_FACTOR=0, double-built inverses, few blocks): the all-reduce of round 3.
        if (shard && ci.nranks > 1 && !gram_given) {            // (a Gram that arrived with the data is a single-process set-up: never sharded)
            const double tr0 = now_s();
            if (dist_factor) {
                const int nb128 = (p + 127) / 128, N = ci.nranks;
                const int nown = (nb128 + N - 1) / N;                           // slots per rank (the last ones of some ranks stay zero)
                const size_t blk = (size_t)128 * ldp, cnt = (size_t)nown * blk;
                DevBuf<float> send((size_t)N * cnt), recv(cnt);
                send.zero(st);
                for (int k = 0; k < nb128; ++k) {
                    const size_t cols = (size_t)std::min(128, (int)ldp - k * 128);
                    ADMM_HIP_CHECK(hipMemcpyAsync(send.get() + ((size_t)(k % N) * nown + k / N) * blk, M.get() + (size_t)k * blk, cols * ldp * sizeof(float),
                                                  hipMemcpyDeviceToDevice, st));
                }
                reduce_scatter_sum_f32(send.get(), recv.get(), cnt, st);
                for (int k = ci.rank; k < nb128; k += N) {
                    const size_t cols = (size_t)std::min(128, (int)ldp - k * 128);
                    ADMM_HIP_CHECK(hipMemcpyAsync(M.get() + (size_t)k * blk, recv.get() + (size_t)(k / N) * blk, cols * ldp * sizeof(float), hipMemcpyDeviceToDevice, st));
                }
            } else {
                allreduce_sum_f32(M.get(), (size_t)ldp * ldp, st);
            }
            comm_stream_sync(st);
            comm_check();
            S.t_gram += now_s() - tr0;
        }

        // (X'X + rho I)^-1, cached for the whole path (rho never changes: ADMMLassoTall.h:97)
        // ADMM_HIP_INVERSE=f32: factorise and invert in float (the reference's LLT is a float factorisation, XX.diagonal() += rho in float).
        // ADMM_HIP_INVERSE=f64: the same float Gram factorised and inverted in double, rounded to float once -- each entry
        // of the cached inverse then carries half an ulp instead of cond * ulp (useful when n ~ p); it costs 150 ms more at
        // p = 10^4 and does not change how often the stopping rule flips against a float Cholesky solve (measured with
        // tests/tools/flip_floor.py: 45 vs 44 of 185 lambdas, the reference's own solve against the exact one: 39).
        // ADMM_HIP_REFINE=1 (opt-in): every x-update is refined once, x = x1 + Minv (rhs - M x1) with the residual in double
        // from the float system M = X'X + rho I (the reference's: XX.diagonal() += rho in float, ADMMLassoTall.h:204) -- the
        // x-update's error against the exact solve of that system drops from cond(M) ulps to about one ulp, i.e. onto what
        // oracle/variants.py calls the `exact` variant, at three passes over the triangle per iteration instead of one.
        if (const char* e = option("REFINE")) refine = std::string(e) == "1" && !shard;
        if (refine) {
            Mg.alloc((size_t)ldp * ldp);
            ADMM_HIP_CHECK(hipMemcpyAsync(Mg.get(), M.get(), (size_t)ldp * ldp * sizeof(float), hipMemcpyDeviceToDevice, st));
            add_diag<float>(Mg.get(), ldp, p, (float)rho, st);
        }
        t0 = now_s();
        // Policy: double below p = 4096 (a few ms there, and the float inverse is 30-100x less accurate: 1-3e-6 against
        // 3e-8 of the largest entry at cond ~ 30, tests/test_gpu_kernels.py), float above.
        // Row-sharded solver: the factorisation's block columns dealt out to the ranks, and of the inverse
        // only the tiles this rank's share of the x-update reads -- 1 / N of the 2 p^3 / 3 + p^3 / 3 flops per rank, bit-identical to the
        // replicated factorisation (chol_inverse.h).  ADMM_HIP_DIST_FACTOR=0: every rank factorises the whole matrix (round 3).
        if (inv64) {
            spd_inverse_f32_via_f64(M.get(), ldp, p, (double)(float)rho, st);
        } else if (dist_factor) {
            add_diag<float>(M.get(), ldp, p, (float)rho, st);
            sy.init(p, st, ci.rank, ci.nranks);               // (the tile list of this rank's share; initialised again below, identically)
            std::vector<int> need;
            {
                const int nb128 = (p + 127) / 128;
                std::vector<char> mark((size_t)nb128 * nb128, 0);
                for (const int4& t : sy.htiles)
                    for (int bi = 2 * t.x; bi <= 2 * t.x + 1 && bi < nb128; ++bi)
                        for (int bj = t.y / 128; bj <= (t.y + t.z - 1) / 128 && bj < nb128; ++bj)
                            if (bj <= bi && !mark[(size_t)bi * nb128 + bj]) { mark[(size_t)bi * nb128 + bj] = 1; need.push_back(bi << 16 | bj); }
            }
            double fl = 0;
            spd_inverse_mfma_f32_dist(M.get(), ldp, p, need, &fl, st);
            dist_flops = fl;
        } else {
            add_diag<float>(M.get(), ldp, p, (float)rho, st);
            spd_inverse_f32(M.get(), ldp, p, st);
        }
        comm_stream_sync(st);
        S.t_factor = now_s() - t0;
        // X itself is no longer needed by the loop (only X'y and Minv are): release 4np bytes.
        d.X.release();

        // ---- loop state
        // x-update variant: lower-triangle symmetric mat-vec (2p^2 bytes) for large p, full-matrix
        // gemv_t (4p^2 bytes, fewer and larger workgroups) for small p.  ADMM_HIP_XUPDATE=full|sym overrides.
        use_sym = p >= 2048;
        if (const char* e = option("XUPDATE")) use_sym = std::string(e) == "sym";
        if (refine) use_sym = true;                      // the refinement is built on the symmetric kernel's partial layout
        if (shard) use_sym = true;                       // the sharded x-update is the tile list of the symmetric kernel dealt out to the ranks
        pl = plan_gemv_t<float>(p, p, 2, 4);
        nwg = (p + kTailElems - 1) / kTailElems;
        ldv = round_up(p, 256);                         // symv reads the right-hand vectors in 256-row blocks
        if (use_sym) sy.init(p, st, shard ? ci.rank : 0, shard ? ci.nranks : 1);
        if (refine) {
            rab.alloc((size_t)2 * ldv); rab.zero(st);
            ruw.alloc((size_t)2 * ldv); ruw.zero(st);
            dD0.alloc((size_t)sy.nrb * sy.ldo); dD1.alloc((size_t)sy.nrb * sy.ldo);
            xD0.alloc((size_t)sy.nax_rows * sy.ldo); xD1.alloc((size_t)sy.nax_rows * sy.ldo);
            dD0.zero(st); dD1.zero(st); xD0.zero(st); xD1.zero(st);
        }
        if (shard) { ab.alloc((size_t)2 * ldp); ab.zero(st); }
        else if (!use_sym) { a_part.alloc((size_t)pl.nseg * ldp); b_part.alloc((size_t)pl.nseg * ldp); a_part.zero(st); b_part.zero(st); }
        // ADMM_HIP_PEER_FUSED=0: go through the generic all-reduce of the exchange layer also on the PEER backend
        peer_fused = shard && ci.backend == COMM_PEER;
        if (const char* e = option("PEER_FUSED")) { if (std::string(e) == "0") peer_fused = false; }
        if (peer_fused) {
            // one launch only when the whole grid is resident with room to spare (its workgroups wait for one another)
            int occ = 0;
            ADMM_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(tall_tail_kernel<TAIL_PEER1>), kTailThreads, 0));
            const int nwg_tail = (p + kTailElems - 1) / kTailElems;
            peer_one = (long long)nwg_tail * 2 <= resident_workgroups(occ);
            if (const char* e = option("PEER_FUSED")) { if (std::string(e) == "2") peer_one = false; }
        }
        // (A single-launch iteration -- tail, decision and tiles in one launch -- and a hipGraph replay of the batch were built, measured
        // bit-identical and SLOWER on C2 in rounds 4 / 5 (44.5 and 46.9 us per iteration against 42.1 / 46.6): removed in round 6,
        // profiles/HISTORY.md.)
        x.alloc(ldv); z0.alloc(ldv); z1.alloc(ldv); y0.alloc(ldv); y1.alloc(ldv);
        adj_z.alloc(ldv); adj_y.alloc(ldv); u.alloc(ldv); w.alloc(ldv);
        beta.alloc((size_t)nlam * p); niter.alloc(nlam);
        P.alloc((size_t)2 * nwg * 8); dlam.alloc(nlam); ctl.alloc(2);
        u.zero(st); w.zero(st);
        ADMM_HIP_CHECK(hipMemcpyAsync(dlam.get(), lam_int.data(), nlam * sizeof(double), hipMemcpyHostToDevice, st));

        q.p = p; q.nwg = nwg; q.nseg = pl.nseg; q.maxit = pb.opts.maxit; q.nlam = nlam; q.enet = pb.enet ? 1 : 0;
        q.part_stride = ldp;
        q.eps_abs = pb.opts.eps_abs; q.eps_rel = pb.opts.eps_rel; q.alpha = (double)(float)pb.alpha; q.sqrt_p = std::sqrt((double)p);
        q.lambdas = dlam.get(); q.XY = XY.get(); q.a_part = a_part.get(); q.b_part = b_part.get();
        if (shard) { q.a_part = ab.get(); q.b_part = ab.get() + ldp; q.nseg = 1; }      // the tail reads the all-reduced pair (generic exchange)
        q.dot0 = sy.dot0.get(); q.dot1 = sy.dot1.get(); q.axp0 = sy.axp0.get(); q.axp1 = sy.axp1.get();
        q.ldo = sy.ldo; q.nrb = sy.nrb; q.p32 = sy.p32; q.sched = sy.sched;
        q.refine_ab = refine ? rab.get() : nullptr; q.refine_ld = ldv;
        q.x = x.get(); q.z0 = z0.get(); q.z1 = z1.get(); q.y0 = y0.get(); q.y1 = y1.get();
        q.adj_z = adj_z.get(); q.adj_y = adj_y.get(); q.u = u.get(); q.w = w.get();
        q.ctl = ctl.get(); q.P = P.get(); q.beta = beta.get(); q.niter = niter.get();
        q.done_host = hflag.p;
#ifdef ADMM_HIP_PROBE
        probe.alloc((size_t)4096 * 4 * 8); probe.zero(st);
        q.probe = probe.get();
        sy.probe = probe.get();
#endif

        // Pinned mirror of the control block for asynchronous polling.
        ADMM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&hctl), 2 * sizeof(TallCtl), hipHostMallocDefault));
        ADMM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&hbeta), (size_t)nlam * p * sizeof(float), hipHostMallocDefault));
        comm_stream_sync(st);
    }

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

    void enable_state(long long cap) override {
        state.alloc((size_t)cap * 5 * p);
        // on the solver's own (non-blocking) stream: a null-stream memset is not ordered against it and, on a busy device, landed
        // AFTER run() had copied record 0 into the dump (suspected cause of the one unreadable record 0 of the 40-process soak, case 546:23)
        ADMM_HIP_CHECK(hipMemsetAsync(state.get(), 0, (size_t)cap * 5 * p * sizeof(float), st));
        state_cap = cap;
        q.state = state.get(); q.state_cap = cap;
    }
    long long read_state(float* out, long long cap, long long* rec_floats) override {
        if (rec_floats) *rec_floats = 5ll * p;
        if (!out) return std::min(trace_n, state_cap);                             // size query
        const long long nrec = std::min(std::min(trace_n, state_cap), cap);       // one record per decision, same numbering as the trace
        if (nrec > 0 && out) read_back(out, state.get(), (size_t)nrec * 5 * p * sizeof(float), st);
        return nrec;
    }

    void read_system(float* out, long long ld) override {
        if (!refine) throw Error(ADMM_ERR_INVALID_ARG, "the system matrix is only kept with ADMM_HIP_REFINE=1");
        ADMM_REQUIRE(out != nullptr && ld >= p, "bad output for the system matrix");
        ADMM_HIP_CHECK(hipMemcpy2D(out, (size_t)ld * sizeof(float), Mg.get(), (size_t)ldp * sizeof(float), (size_t)p * sizeof(float), (size_t)p, hipMemcpyDeviceToHost));
    }

    void debug_dump(const char* tag, const float* dptr, size_t n) {
        std::vector<float> h(n);
        ADMM_HIP_CHECK(hipMemcpy(h.data(), dptr, n * sizeof(float), hipMemcpyDeviceToHost));
        size_t bad = 0; double mx = 0;
        for (size_t k = 0; k < n; ++k) { const float v = h[k]; if (!(v == v) || std::fabs(v) > 1e30f) { if (bad < 4) fprintf(stderr, "[dbg]   %s[%zu] = %g\n", tag, k, (double)v); ++bad; } else mx = std::max(mx, (double)std::fabs(v)); }
        fprintf(stderr, "[dbg] %-8s n=%zu nonfinite=%zu max|.|=%g\n", tag, n, bad, mx);
    }

    // One warm-started lambda path from a cold start (init at the first lambda, init_warm after).
    void run(LassoResult& res) override {
        if (option("DEBUG_DUMP")) {
            if (M.get()) debug_dump("M", M.get(), (size_t)ldp * ldp);
            debug_dump("XY", XY.get(), ldp);
        }
        admm_stats S = setup_stats;
        S.xupdate_variant = shard ? 2 : (use_sym ? 1 : 0);
        S.exchange_variant = !shard ? 0 : (!peer_fused ? 1 : (peer_one ? 3 : 2));
        S.refine = refine ? 1 : 0;
        S.factor_flops = dist_flops;
        res.lambda = lam_user;
        beta.zero(st); niter.zero(st);
        const int init_n = std::max(p, 2 * nwg * 8);
        hipLaunchKernelGGL(tall_init_kernel, dim3((init_n + 255) / 256), dim3(256), 0, st, q, rho, lam_int[0]);
        if (q.state != nullptr)      // record 0 of the iterate dump (the cold start has no iterates): X'y as this solver holds it, in the x slot
            ADMM_HIP_CHECK(hipMemcpyAsync(q.state, XY.get(), (size_t)p * sizeof(float), hipMemcpyDeviceToDevice, st));
        hctl[0].done = hctl[1].done = 0;
        *hflag.p = 0;
#ifdef ADMM_HIP_PROBE
        sy.probe_idx = 0;
#endif

        const int batch = pb.batch_iters > 0 ? (pb.batch_iters + 1) / 2 * 2 : 32;    // even
        const int stride = pb.profile_stride;                          // sample every stride-th x-update with events
        size_t nev = 0;                                                // events of ev_pool used by this run
        Event ev_loop0, ev_loop1, ev_poll[2];

        comm_stream_sync(st);
        const CommLockstep lockstep;                                   // per-iteration exchanges: the short wait bound (comm.h)
        const TraceRange trace_range("admm:loop");
        const double tl0 = now_s();
        ADMM_HIP_CHECK(hipEventRecord(ev_loop0.e, st));
        long long g = 0, launches = 0;
        const long long max_total = (long long)nlam * ((long long)pb.opts.maxit + 2) + 4 + 2 * batch;
        auto enqueue_batch = [&](int slot) {
            for (int k = 0; k < batch; ++k, ++g) {
                const int par = (int)(g & 1);
                const bool sample = stride > 0 && (g % stride) == stride / 2 && nev < 8192;      // mid-batch: not the launch right behind the poll event
                hipEvent_t e0 = nullptr, e1 = nullptr;
                if (sample) {
                    while (ev_pool.size() < nev + 2) {
                        hipEvent_t e; ADMM_HIP_CHECK(hipEventCreate(&e));
                        ev_pool.push_back(e);
                    }
                    e0 = ev_pool[nev]; e1 = ev_pool[nev + 1];
                    nev += 2;
                }
                // sampled launches carry start/stop events that time exactly the x-update kernel on this stream
                // the decision of this iteration rides along as one extra workgroup of the x-update launch
                const TallDecideExtra dec{q, par};
                if (shard && peer_fused) {
                    // this rank's tiles -> its share of (a, b) written into every rank's exchange slot by the reduction
                    // launch itself -> the (replicated) tail waits for the K flags and sums the K slots: three launches,
                    // none of them the exchange layer's
                    sy.launch(M.get(), ldp, u.get(), w.get(), &ctl.get()[par].done, st, dec, e0, e1);
                    const PeerExchange ex = comm_peer_begin((size_t)2 * ldp * sizeof(float));
                    if (peer_one) {
                        hipLaunchKernelGGL(tall_tail_kernel<TAIL_PEER1>, dim3(nwg), dim3(kTailThreads), 0, st, q, par, ex);
                    } else {
                        hipLaunchKernelGGL(tall_shard_push_kernel, dim3(nwg), dim3(kTailThreads), 0, st, q, ex, ldp, &ctl.get()[par].done);
                        hipLaunchKernelGGL(tall_tail_kernel<TAIL_PEER>, dim3(nwg), dim3(kTailThreads), 0, st, q, par, ex);
                    }
                } else if (shard) {
                    // this rank's tiles -> its share of (a, b) -> ONE all-reduce of 2 ldp floats -> the (replicated) tail
                    sy.launch(M.get(), ldp, u.get(), w.get(), &ctl.get()[par].done, st, dec, e0, e1);
                    hipLaunchKernelGGL(tall_shard_reduce_kernel, dim3(nwg), dim3(kTailThreads), 0, st, q, ab.get(), ldp, &ctl.get()[par].done);
                    allreduce_sum_f32(ab.get(), (size_t)2 * ldp, st);
                    hipLaunchKernelGGL(tall_tail_kernel<TAIL_GEMV>, dim3(nwg), dim3(kTailThreads), 0, st, q, par, PeerExchange{});
                } else if (use_sym && refine) {
                    // x1 = Minv rhs (both candidates) -> vectors; r = rhs - M x1 in double; correction Minv r -> partials; the tail adds
                    const int* skip = &ctl.get()[par].done;
                    sy.launch(M.get(), ldp, u.get(), w.get(), skip, st, dec, e0, e1);
                    hipLaunchKernelGGL(tall_shard_reduce_kernel, dim3(nwg), dim3(kTailThreads), 0, st, q, rab.get(), ldv, skip);
                    SymvArgsD ad;
                    ad.A = Mg.get(); ad.lda = ldp; ad.p = p; ad.v0 = rab.get(); ad.v1 = rab.get() + ldv;
                    ad.dot0 = dD0.get(); ad.dot1 = dD1.get(); ad.axp0 = xD0.get(); ad.axp1 = xD1.get(); ad.ldo = sy.ldo; ad.tiles = sy.tiles.get(); ad.skip = skip;
                    hipLaunchKernelGGL(symv2_lower_f64acc_kernel, dim3(sy.ntiles), dim3(kSyThreads), 0, st, ad);
                    hipLaunchKernelGGL(tall_refine_resid_kernel, dim3(nwg), dim3(kTailThreads), 0, st, q, dD0.get(), dD1.get(), xD0.get(), xD1.get(),
                                       ruw.get(), ruw.get() + ldv, skip);
                    sy.launch(M.get(), ldp, ruw.get(), ruw.get() + ldv, skip, st, SymvNoExtra());
                    hipLaunchKernelGGL(tall_tail_kernel<TAIL_SYMV>, dim3(nwg), dim3(kTailThreads), 0, st, q, par, PeerExchange{});
                } else if (use_sym) {
                    sy.launch(M.get(), ldp, u.get(), w.get(), &ctl.get()[par].done, st, dec, e0, e1);
                    hipLaunchKernelGGL(tall_tail_kernel<TAIL_SYMV>, dim3(nwg), dim3(kTailThreads), 0, st, q, par, PeerExchange{});
                } else {
                    launch_gemv_t<float, 2, 4, TallDecideExtra>(pl, M.get(), ldp, p, p, u.get(), w.get(), a_part.get(), b_part.get(), ldp,
                                                                &ctl.get()[par].done, st, dec, e0, e1);
                    hipLaunchKernelGGL(tall_tail_kernel<TAIL_GEMV>, dim3(nwg), dim3(kTailThreads), 0, st, q, par, PeerExchange{});
                }
                ++launches;
            }
            // Single rank: no copy per poll, the deciding workgroup sets the pinned flag itself when the path has finished.
            // Sharded over ranks: every rank must enqueue the SAME number of exchanges, so the stop decision has to be a
            // deterministic function of the stream position -- the control block sampled in stream order after each
            // batch -- not of when a flag store happens to become visible to this host.
            if (shard) ADMM_HIP_CHECK(hipMemcpyAsync(&hctl[slot], &ctl.get()[(int)(g & 1)], sizeof(TallCtl), hipMemcpyDeviceToHost, st));
            ADMM_HIP_CHECK(hipEventRecord(ev_poll[slot].e, st));
        };
        int slot = 0;
        enqueue_batch(slot);
        ADMM_HIP_CHECK(hipGetLastError());                 // launch failures surface here
        bool done = false;
        while (!done) {
            enqueue_batch(slot ^ 1);      // keep one batch in flight while polling the previous one
            comm_event_sync(ev_poll[slot].e);
            comm_check();
            done = shard ? hctl[slot].done != 0 : *static_cast<volatile int*>(hflag.p) != 0;
            slot ^= 1;
            if (!done && g > max_total) throw Error(ADMM_ERR_INTERNAL, "tall path: iteration bound exceeded without completion");
        }
        ADMM_HIP_CHECK(hipEventRecord(ev_loop1.e, st));
        comm_stream_sync(st);
        S.t_loop = now_s() - tl0;
        ADMM_HIP_CHECK(hipMemcpy(hctl, ctl.get(), 2 * sizeof(TallCtl), hipMemcpyDeviceToHost));      // both slots: decisions taken
#ifdef ADMM_HIP_PROBE
        if (const char* f = option("PROBE_OUT")) {
            std::vector<long long> hp((size_t)4096 * 4 * 8);
            ADMM_HIP_CHECK(hipMemcpy(hp.data(), probe.get(), hp.size() * sizeof(long long), hipMemcpyDeviceToHost));
            if (FILE* fp = std::fopen(f, "wb")) { std::fwrite(hp.data(), sizeof(long long), hp.size(), fp); std::fclose(fp); }
        }
#endif
        float ms = 0.f;
        ADMM_HIP_CHECK(hipEventElapsedTime(&ms, ev_loop0.e, ev_loop1.e));
        S.loop_ms_events = ms;
        S.xupdate_launches = launches;
        if (nev > 0) {
            double tot = 0;
            for (size_t k = 0; k + 1 < nev; k += 2) {
                float m1 = 0.f;
                ADMM_HIP_CHECK(hipEventElapsedTime(&m1, ev_pool[k], ev_pool[k + 1]));
                tot += m1;
            }
            S.xupdate_samples = (long long)(nev / 2);
            S.xupdate_ms_avg = tot / (double)(nev / 2);
        }

        if (option("DEBUG_DUMP")) {
            debug_dump("x", x.get(), ldv); debug_dump("u", u.get(), ldv); debug_dump("w", w.get(), ldv);
            debug_dump("z0", z0.get(), ldv); debug_dump("y0", y0.get(), ldv);
            if (!use_sym) { debug_dump("a_part", a_part.get(), (size_t)pl.nseg * ldp); debug_dump("b_part", b_part.get(), (size_t)pl.nseg * ldp); }
            std::vector<double> hp((size_t)2 * nwg * 8);
            ADMM_HIP_CHECK(hipMemcpy(hp.data(), P.get(), hp.size() * sizeof(double), hipMemcpyDeviceToHost));
            for (size_t k = 0; k < hp.size() && k < 16; ++k) fprintf(stderr, "[dbg] P[%zu]=%g\n", k, hp[k]);
            fprintf(stderr, "[dbg] nseg=%d nwg=%d ldp=%lld ldv=%lld\n", pl.nseg, nwg, ldp, ldv);
        }

        // ---- results: niter, beta on the original scale (DataStd::recover, Lasso.cpp:108-111)
        res.niter.assign(nlam, 0);
        ADMM_HIP_CHECK(hipMemcpy(res.niter.data(), niter.get(), nlam * sizeof(int), hipMemcpyDeviceToHost));
        ADMM_HIP_CHECK(hipMemcpyAsync(hbeta, beta.get(), (size_t)nlam * p * sizeof(float), hipMemcpyDeviceToHost, st));
        comm_stream_sync(st);
        res.beta.assign((size_t)(p + 1) * nlam, 0.f);
        long long tot_it = 0;
        for (int l = 0; l < nlam; ++l) {
            float b0 = 0.f;
            recover_coef<float>(d, hbeta + (size_t)l * p, &b0, res.beta.data() + (size_t)l * (p + 1) + 1);
            res.beta[(size_t)l * (p + 1)] = b0;
            tot_it += res.niter[l];
        }
        S.total_iter = tot_it;
        trace_n = hctl[0].done ? std::max(hctl[0].total, hctl[1].total) : 0;      // decisions taken (the sticky no-op decisions after `done` do not write)
        res.stats = S;
    }
};

std::unique_ptr<LassoPlan> make_tall_plan(DeviceData<float>&& d, const LassoProblem& pb, hipStream_t st) {
    return std::unique_ptr<LassoPlan>(new TallPlan(std::move(d), pb, st));
}

}  // namespace admm
