// Row-block consensus Lasso (`$parallel()`), device resident.
//
// Replaces PADMMLasso_Master / PADMMLasso_Worker driven by PADMMBase_Master::solve:
//   /root/reference/src/PADMMBase.h:57-78 (worker update_y), :117-145 (eps / resid), :174-237 (solve)
//   /root/reference/src/PADMMLasso.h:17-31 (worker next_x: Cholesky or Woodbury), :48-68 (init, add_xu_to),
//   :99-108 (master next_z), :149-152, :163-179 (row partition), :193-223 (init / init_warm)
//   and the lambda loop of ParLasso.cpp:89-102.
//
// Workers are the reference's contiguous row blocks (last block takes the remainder).  Each
// worker's solve uses a cached inverse: (A'A + rho I)^-1 for a tall block, or the Woodbury form
// x = (rhs - A'(AA' + rho I)^-1 A rhs) / rho for a wide block, with A and A' both stored so that
// every product streams contiguous columns (gemv_t).  rho never changes (PADMMBase.h:147-159).
// Per iteration: head (convergence decision of the previous iteration + lambda schedule + rhs_k),
// the workers' mat-vecs, pack (x_k and the consensus sum  sum_k x_k + y_k/rho), z (soft-threshold,
// dual updates, norms).  The consensus sum is the only cross-worker exchange; it is a separate
// buffer so that a multi-process build can all-reduce it between `pack` and `z`.
#include "prep.h"
#include "gemv_kernels.h"
#include "gather_kernels.h"
#include "solvers.h"
#include "loop_driver.h"
#include "comm.h"
#include "peer_device.h"
#include "probe.h"

namespace admm {

struct ParCtl {
    double lam, eps_primal, eps_dual;
    int iter, lam_idx, done, first, total, pad0, pad1, pad2, pad3, pad4;      // 64 bytes: whole 16-byte words (load_ctl_vector)
};
static_assert(sizeof(ParCtl) == 64, "ParCtl layout");

constexpr int kParMaxWorkers = 64;
constexpr int kParThreads = 256;

// One-pass form of a Woodbury worker (round 5).  The reference forms t_k = A_k rhs_k and then A_k's_k with s_k = (A_k A_k' + rho I)^-1 t_k
// (PADMMLasso.h:23-29): two passes over the block.  With rhs_k = A_k'b_k - y_k + rho z:
//     t_k = c_k - q_k + rho A_k z,     c_k = A_k (A_k'b_k) (once),   q_k = A_k y_k,   z = the prox output (sparse: gather_kernels.h)
// and from y_k <- y_k + rho (x_k - z_new), x_k = (rhs_k - A_k's_k) / rho:  A_k x_k = (t_k - A_k A_k's_k) / rho = s_k, hence
//     q_k <- q_k + rho (s_k - A_k z_new)
// -- everything rows_k-sized and in double except the ONE stream A_k's_k.  The recurrence is dead-beat: an error e in the held q_k
// puts -e into t_k, (A_k A_k' + rho I)^-1 carries it into s_k, and the true A_k x_k differs from s_k by exactly -e / rho (plus the
// residual of the cached inverse), so the true A_k y_k,new and the held q_k,new agree again up to this iteration's own roundings.
struct ParWb {
    const float* spart;            // partial rows of s_k = Minv t_k (the worker's gM product), summed in row order like the next product's staging does
    long long sstride;
    int snseg, rows;               // rows = 0: a tall block (Cholesky branch), nothing to do
    const float* G; long long ldg; // the float Gram A_k A_k' the cached inverse was formed from (par_wb_resid_kernel)
};

struct ParParams {
    int p, K, Kl, maxit, nlam, nwg;     // K: row blocks of the whole problem; Kl: blocks owned by this process
    long long ldv;                 // stride between the per-worker p-vectors
    double rho, eps_abs, eps_rel;
    const double* lambdas;
    const float* Ab;               // [Kl][ldv]
    float* rhs;                    // [Kl][ldv]
    float* x;                      // [Kl][ldv]
    float* y;                      // [Kl][ldv]
    float* z;                      // [ldv]
    float* wsum;                   // [ldv]  consensus sum (the all-reduce payload, together with nsum[0..2])
    double* nsum;                  // [8]: sum_k|x_k|^2, sum_k|y_k|^2, sum_k|x_k - z|^2 (summed over ranks), |z|^2, |z_new - z_old|^2
    // per local worker: result of the last mat-vec of the x-update
    const float* gout[kParMaxWorkers]; int gnseg[kParMaxWorkers]; long long gstride[kParMaxWorkers];
    int wide[kParMaxWorkers];      // 1: Woodbury branch, x = (rhs - gout) / rho ; 0: x = gout
    ParCtl* ctl;                   // [2]
    double* P;                     // [nwg][8] per-workgroup partials of the five sums
    double* trace; long long trace_cap;      // optional decision records (admm_hip_lasso_plan_trace_*), or NULL
    float* state; long long state_cap;       // optional [state_cap][(1 + 2 Kl) p]: z, x_0 .. x_{Kl-1}, y_0 .. y_{Kl-1} of every iteration, or NULL
    float* beta; int* niter; int* done;
    // one-pass Woodbury workers (wb = NULL: the two-pass form): flat [Kl][wb_ld] vectors, gather partial rows [Kl][az_ng][wb_ld]
    const ParWb* wb; int wb_ld, az_ng;
    const double* cv; double* qv; float* tvec; const double* azpart;
    double* azv; double* tv64; int* wbflag; const double* dpart; double wb_tau2;      // cancellation fall-back (par_wb_flag_kernel)
    unsigned long long* wbcount;                                                         // fall-back passes taken (diagnostic)
    int wb_nb;                                                                           // workgroups per worker of the head's one-pass part (ceil(wb_ld / 256))
    double* wbnorm;                                                                      // [Kl][wb_nb][2] their shares of |t_k|^2, |q_k|^2
    unsigned int* wbarrive;                                                              // [2 Kl] arrival counters: head's workgroups per worker, fall-back pass's workgroups per worker
    double* dlt;                                                                         // [Kl][wb_ld] residual of the small solve (par_wb_resid_kernel)
#ifdef ADMM_HIP_PROBE
    long long* probe;
#endif
};

// head: rhs_k = A_k'b_k - y_k + rho z (PADMMLasso.h:19-21) for the local workers.
// Grid (round 6: three launches fewer per iteration): the first Kl * wb_nb workgroups form the one-pass workers' t_k / q_k -- 256 rows of one
// worker each -- and the cancellation guard's flag with them (the last of a worker's workgroups to arrive adds up their shares of
// |t_k|^2, |q_k|^2 in workgroup order: the same decision whoever arrives last); the next workgroup folds the previous iteration's norm
// partials into nsum (was: workgroup 0 of `pack`); the others form rhs.
__global__ void __launch_bounds__(kParThreads)
par_head_kernel(ParParams q) {
    // no fused multiply-adds in the elementwise arithmetic: the reference is built without them (see lasso_tall.hip, tall_update_elem)
#pragma clang fp contract(off)
    __shared__ double scratch[5 * (kParThreads / 64)];
    if (load_flag_vector(q.done)) return;
    const int nwb = q.wb != nullptr ? q.Kl * q.wb_nb : 0;
    if ((int)blockIdx.x < nwb) {
        // one-pass Woodbury workers: A_k z from the gather launch's partial rows (group order), q_k = A_k y_k advanced by the dual
        // update the z kernel just made (its rho is the float one, PADMMBase.h:70-78), t_k = A_k rhs_k for the product with the inverse
        const double rho_f = (double)(float)q.rho;
        const int k = blockIdx.x / q.wb_nb, jb = blockIdx.x - k * q.wb_nb;
        const int i = jb * kParThreads + threadIdx.x;
        const ParWb wb = q.wb[k];
        double nacc[2] = {0.0, 0.0};
        if (i < wb.rows) {
            const int gi = k * q.wb_ld + i;
            const double* ap = q.azpart + (size_t)k * q.az_ng * q.wb_ld + i;
            double az = 0.0;
            for (int c0 = 0; c0 < q.az_ng; c0 += 8) {
                double tv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) tv[u] = ap[(size_t)min(c0 + u, q.az_ng - 1) * q.wb_ld];
#pragma unroll
                for (int u = 0; u < 8; ++u) az += c0 + u < q.az_ng ? tv[u] : 0.0;
            }
            float sv = 0.f;
            for (int r = 0; r < wb.snseg; ++r) sv += wb.spart[(size_t)r * wb.sstride + i];
            const double qn = q.qv[gi] + rho_f * ((double)sv - az) - q.dlt[gi];       // A_k x_k = s_k - delta_k / rho (par_wb_resid_kernel)
            const double tn = q.cv[gi] - qn + q.rho * az;
            q.qv[gi] = qn; q.azv[gi] = az; q.tv64[gi] = tn;
            q.tvec[gi] = (float)tn;
            nacc[0] = tn * tn; nacc[1] = qn * qn;
        }
        // the cancellation guard's flag (see below): this workgroup's share, then -- last arrival of the worker -- the flag
        block_sum<double, 2>(nacc, scratch);
        if (threadIdx.x == 0) {
            double* sh = q.wbnorm + ((size_t)k * q.wb_nb + jb) * 2;
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(sh), (unsigned long long)__double_as_longlong(nacc[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(sh + 1), (unsigned long long)__double_as_longlong(nacc[1]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned prev = __hip_atomic_fetch_add(q.wbarrive + k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == (unsigned)q.wb_nb - 1) {
                __hip_atomic_store(q.wbarrive + k, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                double t2 = 0.0, q2 = 0.0;
                for (int j = 0; j < q.wb_nb; ++j) {
                    const unsigned long long* o = reinterpret_cast<const unsigned long long*>(q.wbnorm + ((size_t)k * q.wb_nb + j) * 2);
                    t2 += __longlong_as_double((long long)__hip_atomic_load(o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    q2 += __longlong_as_double((long long)__hip_atomic_load(o + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                }
                const int f = (wb.rows > 0 && t2 < q.wb_tau2 * ((double)wb.rows / (double)q.p) * q2) ? 1 : 0;      // tau_k = tau0 sqrt(rows_k / p)
                q.wbflag[k] = f;
                if (f) atomicAdd(q.wbcount, 1ull);
            }
        }
        return;
    }
    const int b = (int)blockIdx.x - nwb;
    if (b == 0) {                                       // the norm partials of the previous iteration's z launch -> nsum[0..4]
        double acc[5] = {0, 0, 0, 0, 0};
        for (int w = threadIdx.x; w < q.nwg; w += kParThreads) {
#pragma unroll
            for (int k = 0; k < 5; ++k) acc[k] += q.P[(size_t)w * 8 + k];
        }
        block_sum<double, 5>(acc, scratch);
        if (threadIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < 5; ++k) q.nsum[k] = acc[k];
        }
    }
    const int nb = (int)gridDim.x - nwb;
    // workers in groups of 8: the 16 loads of a group are requested together (one worker at a time was a chain of dependent
    // round trips: kernel arguments -> addresses -> values, per worker)
    for (int i = b * kParThreads + threadIdx.x; i < q.p; i += nb * kParThreads) {
        const double rz = q.rho * (double)q.z[i];
        for (int k0 = 0; k0 < q.Kl; k0 += 8) {
            float ab[8], yv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const size_t o = (size_t)min(k0 + u, q.Kl - 1) * q.ldv + i;
                ab[u] = q.Ab[o]; yv[u] = q.y[o];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (k0 + u < q.Kl) {
                    const float r0 = ab[u] - yv[u];
                    q.rhs[(size_t)(k0 + u) * q.ldv + i] = (float)((double)r0 + rz);   // rhs[idx] += rho * value (double)
                }
            }
        }
    }
}

// One-pass Woodbury workers, the guard against CANCELLATION.  t_k = c_k - q_k + rho A_k z is a difference; where the iteration
// approaches a null model (z = 0, x_k -> 0: the whole first lambda of an automatic grid) A_k y_k -> A_k A_k'b_k and t_k -> 0.  The
// reference forms rhs_k = A_k'b_k - y_k in float, EXACTLY when the two are close (Sterbenz), so its t_k = A_k rhs_k keeps its
// relative accuracy down to zero and sees every rounding the stored y_k carries; q_k follows the un-rounded dual, so t_k misses
// A_k (rounding of y_k) -- an absolute error of u |A_k||y_k| that the stepwise instrument (oracle/stepcheck.py) flags once
// |t_k| << |q_k| (measured: 160 x the float-solve yardstick at |t| / |q| = 1 / 4000, fresh samples 811:42 / 812:42; below the
// reference's own x-update error from |t| / |q| >= 1 / 64 on).  The missing term is the part of the dual's rounding noise that lies
// in range(A_k'), sqrt(rows_k / p) of it, so its size against the yardstick goes like sqrt(rows_k / p) |q| / |t|.  So: per worker and
// iteration, |t_k|_2 < tau_k |q_k|_2 with tau_k = sqrt(rows_k / p) / 16 (1/16 for a nearly square block: the soak's ill-conditioned
// 947:142 is at the two-pass form's own 12 x there, 18 x at 1/128; 1/143 for C4's 1250 x 10^5 blocks) -> t_k is formed the
// reference's way for this iteration, one dense pass A_k rhs_k through the same gather kernel (double accumulation), and q_k is
// re-anchored on it.  Elsewhere the one-pass form is MORE accurate than the float product it replaces.
// (the flag itself: par_head_kernel, last arrival of the worker's workgroups)
// The residual of the small solve, delta_k = (A_k A_k' + rho I) s_k - t_k, in double from the float Gram, the float s_k and the float t_k
// the solve was given: with it A_k x_k = (t_k - A_k A_k's_k) / rho = s_k - delta_k / rho exactly, and the recurrence of q_k carries no
// term of the cached inverse's own error (u cond(A_k A_k' + rho I) |t_k|; without it the x-update of ill-conditioned blocks was 1.2 .. 1.4
// x the two-pass form's error in the rms over a run, emulation and soak 940:147 / 947:142; with it the two forms are level).
// One wave per row of the symmetric Gram (its column, contiguous), all local workers in one launch.
__device__ __forceinline__ void par_wb_resid_row(const ParParams& q, int wave) {
    const int lane = threadIdx.x & 63;
    const int k = wave / q.wb_ld, i = wave - k * q.wb_ld;
    if (k >= q.Kl) return;
    const ParWb wb = q.wb[k];
    if (i >= wb.rows) return;
    const float* g = wb.G + (size_t)i * wb.ldg;
    double acc = 0.0;
    // 16 bytes per lane and request (the Gram's columns start on 512-byte boundaries, the partial rows of s on 128-byte ones); rows
    // beyond wb.rows of the Gram column are zero (the Gram buffer is zero padded), so whole float4 pieces are safe
    const int r4 = (wb.rows + 3) / 4;
    for (int j4 = lane; j4 < r4; j4 += 64) {
        const float4 gv = reinterpret_cast<const float4*>(g)[j4];
        float4 sv = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int r = 0; r < wb.snseg; ++r) {
            const float4 pv = reinterpret_cast<const float4*>(wb.spart + (size_t)r * wb.sstride)[j4];
            sv.x += pv.x; sv.y += pv.y; sv.z += pv.z; sv.w += pv.w;
        }
        const int j = 4 * j4;
        acc = fma((double)gv.x, (double)sv.x, acc);
        if (j + 1 < wb.rows) acc = fma((double)gv.y, (double)sv.y, acc);
        if (j + 2 < wb.rows) acc = fma((double)gv.z, (double)sv.z, acc);
        if (j + 3 < wb.rows) acc = fma((double)gv.w, (double)sv.w, acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) {
        float si = 0.f;
        for (int r = 0; r < wb.snseg; ++r) si += wb.spart[(size_t)r * wb.sstride + i];
        q.dlt[(size_t)k * q.wb_ld + i] = acc + (double)(float)q.rho * (double)si - (double)q.tvec[(size_t)k * q.wb_ld + i];
    }
}
__global__ void __launch_bounds__(256)
par_wb_resid_kernel(ParParams q) {
    if (load_flag_vector(q.done)) return;
    par_wb_resid_row(q, blockIdx.x * 4 + (threadIdx.x >> 6));
}
// The workers' A_k's_k products (gemv_t_batch_kernel's body) with the residual rows as one more slice of the grid (blockIdx.y = 0,
// dispatched first): delta_k only needs s_k, like the products, and is done in the first microseconds of their 0.6 ms instead of in a
// launch of its own behind them (15 us per iteration at C4).
template <bool NT>
__global__ void __launch_bounds__(kGemvThreads)
par_A_batch_resid_kernel(const GemvTArgs<float>* __restrict__ batch, ParParams q) {
    if (blockIdx.y == 0) {
        if (load_flag_vector(q.done)) return;
        const int nw = q.Kl * q.wb_ld;
        for (int w = blockIdx.x * 4 + (threadIdx.x >> 6); w < nw; w += gridDim.x * 4) par_wb_resid_row(q, w);
        return;
    }
    const GemvTArgs<float> a = batch[blockIdx.y - 1];
    if (a.skip != nullptr && *a.skip != 0) return;
    const int ngroups = (a.k + 4 - 1) / 4;
    const int grid = a.nseg * ((ngroups + a.groups_per_wg - 1) / a.groups_per_wg);
    if ((int)blockIdx.x >= grid) return;
    gemv_t_body<float, 1, 4, NT>(a, (int)blockIdx.x);
}
// The fall-back pass of the cancellation guard and what follows it, in ONE launch (round 6; was gather_batch_kernel + par_wb_fix_kernel):
// the workgroups of a flagged worker run the dense gather A_k rhs_k (gather_kernels.h), and the last of them to finish forms
// t_k = A_k rhs_k from the partial rows (group order) and re-anchors q_k = c_k + rho A_k z - t_k (= A_k y_k as stored).  A worker whose
// flag is down costs its workgroups one flag load.
__global__ void __launch_bounds__(kGatherThreads)
par_gather_fix_kernel(const GatherArgs<float>* __restrict__ batch, ParParams q) {
    __shared__ int s_last;
    const GatherArgs<float> a = batch[blockIdx.z];
    if (a.skip != nullptr && load_flag_vector(a.skip) != 0) return;
    if (a.only_if != nullptr && load_flag_vector(a.only_if) == 0) return;
    constexpr int TILE = kGatherThreads * Vec16<float>::N;
    const int k = blockIdx.z;
    const int tiles = (a.rows + TILE - 1) / TILE;
    if ((int)blockIdx.y >= a.ngroups || (int)blockIdx.x >= tiles) return;
    gather_body<float>(a, (int)blockIdx.x, (int)blockIdx.y);
    __threadfence();                                     // the partial rows are visible device-wide before this workgroup counts itself in
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned total = (unsigned)(tiles * a.ngroups);
        const unsigned prev = __hip_atomic_fetch_add(q.wbarrive + q.Kl + k, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_last = prev == total - 1 ? 1 : 0;
        if (s_last) __hip_atomic_store(q.wbarrive + q.Kl + k, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    const int rows = q.wb[k].rows;
    for (int i = threadIdx.x; i < rows; i += kGatherThreads) {
        const int gi = k * q.wb_ld + i;
        double t = 0.0;
        for (int g = 0; g < q.az_ng; ++g)
            t += __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(q.dpart + ((size_t)k * q.az_ng + g) * q.wb_ld + i),
                                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        q.tvec[gi] = (float)t;
        q.qv[gi] = q.cv[gi] + q.rho * q.azv[gi] - t;
    }
}

// c_k = A_k (A_k'b_k) from the setup gather's partial rows (dense right-hand side), in double
__global__ void par_wb_c_kernel(ParParams q, double* cv) {
    const int gi = blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= q.Kl * q.wb_ld) return;
    const int k = gi / q.wb_ld, i = gi - k * q.wb_ld;
    double c = 0.0;
    if (i < q.wb[k].rows)
        for (int g = 0; g < q.az_ng; ++g) c += q.azpart[((size_t)k * q.az_ng + g) * q.wb_ld + i];
    cv[gi] = c;
}

// pack: x_k from the mat-vec results, consensus sum w = sum_k (x_k + y_k / rho)   (PADMMLasso.h:65-68,101-105);
// workgroup 0 also folds the previous iteration's per-workgroup norm partials into nsum[0..4].
// PEER = 1 (multi-process run over the PEER exchange, round 3): this launch is the PRODUCER of the iteration's one exchange --
// every workgroup writes its elements of the consensus sum straight into this rank's slot of EVERY rank's exchange buffer
// (pairs of elements as one 8-byte write-through store), workgroup 0 adds the three worker-summed norms behind them, and the
// last workgroup to finish raises the flags (peer_device.h); par_z_kernel<1> is the consumer.  No launches of the exchange
// layer, like the sharded tall and wide solvers (replaces the shared-memory reads of PADMMLasso.h:99-108, PADMMBase.h:200-214).
constexpr size_t par_peer_norm_offset(int p) { return ((size_t)p * sizeof(float) + 15) / 16 * 16; }
// x_k of element i from the mat-vec results, returns this process's share of the consensus sum  sum_k (x_k + y_k / rho)  (PADMMLasso.h:23-30,65-68)
__device__ __forceinline__ float par_pack_elem(const ParParams& q, int i, float rho_f) {
#pragma clang fp contract(off)
    float w = 0.f;
    for (int k0 = 0; k0 < q.Kl; k0 += 8) {                                      // 8 workers' operands requested together, consumed in order
        float g0[8], rh[8], yv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = min(k0 + u, q.Kl - 1);
            const size_t o = (size_t)k * q.ldv + i;
            g0[u] = q.gout[k][i]; rh[u] = q.rhs[o]; yv[u] = q.y[o];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + u;
            if (k < q.Kl) {
                float g = 0.f;
                g += g0[u];
                for (int s = 1; s < q.gnseg[k]; ++s) g += q.gout[k][(size_t)s * q.gstride[k] + i];
                const float x = q.wide[k] ? (rh[u] - g) / rho_f : g;            // PADMMLasso.h:23-30
                q.x[(size_t)k * q.ldv + i] = x;
                w += x + yv[u] / rho_f;
            }
        }
    }
    return w;
}

template <int PEER>
__global__ void __launch_bounds__(kParThreads)
par_pack_kernel(ParParams q, PeerExchange ex) {
    // no fused multiply-adds in the elementwise arithmetic: the reference is built without them (see lasso_tall.hip, tall_update_elem)
#pragma clang fp contract(off)
    if (load_flag_vector(q.done)) return;
    if (PEER && blockIdx.x == 0 && threadIdx.x == 0) {               // the three worker-summed norms behind the payload (nsum: folded by the head launch)
        for (int dst = 0; dst < ex.nranks; ++dst) {
            unsigned long long* nd = reinterpret_cast<unsigned long long*>(peer_dst_slot(ex, dst) + par_peer_norm_offset(q.p));
#pragma unroll
            for (int k = 0; k < 3; ++k) peer_store_u64(nd + k, (unsigned long long)__double_as_longlong(q.nsum[k]));
        }
    }
    const float rho_f = (float)q.rho;
    const int pe = PEER ? (q.p + 1) / 2 * 2 : q.p;        // PEER: whole pairs (the partner lane of the last odd element stores a zero)
    for (int i = blockIdx.x * kParThreads + threadIdx.x; i < pe; i += gridDim.x * kParThreads) {
        const float w = i < q.p ? par_pack_elem(q, i, rho_f) : 0.f;
        if (PEER) {
            const float wn = __shfl_down(w, 1, 64);        // i is even in even lanes (grid stride and block size are even)
            if ((i & 1) == 0) {
                for (int dst = 0; dst < ex.nranks; ++dst)
                    peer_store_f32x2(reinterpret_cast<float*>(peer_dst_slot(ex, dst)) + i, w, wn);
            }
        } else {
            q.wsum[i] = w;
        }
    }
    if (PEER) peer_publish(ex, gridDim.x);
}

// The decision for the iteration that has just been summed up (PADMMBase.h:216-221,230-231; eps :117-139) and the lambda schedule.
struct ParDecision { ParCtl out; int lam_finished, niter_val; double rp, rd, code; };
__device__ __forceinline__ ParDecision par_decide(const ParParams& q, const ParCtl& in, double x2, double y2, double r2, double z2, double dz2) {
    ParDecision d;
    d.out = in;
    d.out.first = 0;
    d.lam_finished = -1; d.niter_val = 0; d.rp = 0; d.rd = 0; d.code = ADMM_TRACE_COLD;
    if (!in.first) {
        const double rp = sqrt(r2);                                  // sqrt(sum_k |x_k - z|^2)        PADMMBase.h:213
        const double rd = q.rho * sqrt((double)q.K * dz2);           // rho sqrt(K |z_new - z|^2)      PADMMLasso.h:149-152
        d.rp = rp; d.rd = rd; d.code = (rp < in.eps_primal && rd < in.eps_dual) ? ADMM_TRACE_CONVERGED : ADMM_TRACE_CONTINUE;
        if (rp < in.eps_primal && rd < in.eps_dual) { d.lam_finished = in.lam_idx; d.niter_val = in.iter + 1; }
        else {
            d.out.iter = in.iter + 1;
            if (in.iter + 1 >= q.maxit) { d.lam_finished = in.lam_idx; d.niter_val = q.maxit + 1; }
        }
        if (d.lam_finished >= 0) {
            d.out.lam_idx = in.lam_idx + 1; d.out.iter = 0;
            if (d.out.lam_idx >= q.nlam) d.out.done = 1;
            else d.out.lam = q.lambdas[d.out.lam_idx];
        }
    }
    const double sK = sqrt((double)q.K), spK = sqrt((double)q.p * (double)q.K);
    d.out.eps_primal = fmax(sqrt(x2), sqrt(z2) * sK) * q.eps_rel + spK * q.eps_abs;   // PADMMBase.h:117-128
    d.out.eps_dual = sqrt(y2) * q.eps_rel + spK * q.eps_abs;                           // PADMMBase.h:129-139
    d.out.total = in.total + 1;
    return d;
}
__device__ __forceinline__ void par_record(const ParParams& q, const ParCtl& in, const ParDecision& d, ParCtl* outp) {
    if (d.lam_finished >= 0) q.niter[d.lam_finished] = d.niter_val;
    *outp = d.out;
    if (d.out.done) *q.done = 1;
    if (q.trace != nullptr && in.total < q.trace_cap) {
        double* t = q.trace + (size_t)in.total * ADMM_TRACE_FIELDS;
        t[0] = in.lam_idx; t[1] = in.iter; t[2] = in.eps_primal; t[3] = in.eps_dual; t[4] = d.rp; t[5] = d.rd;
        t[6] = q.rho; t[7] = 0.0; t[8] = d.code; t[9] = q.rho; t[10] = q.rho; t[11] = in.lam;
    }
}
// z_new = soft(w / K, lambda / (rho K)); y_k += rho (x_k - z_new); norm shares   (PADMMLasso.h:99-108, PADMMBase.h:70-78) of element i
__device__ __forceinline__ void par_z_elem(const ParParams& q, int i, float wtot, const ParDecision& d, double pen, float rho_f, double (&acc)[5]) {
#pragma clang fp contract(off)
    const float zo = q.z[i];
    if (d.lam_finished >= 0) q.beta[(size_t)d.lam_finished * q.p + i] = zo;           // get_z()  ParLasso.cpp:98
    if (d.out.done) return;
    const float v = wtot / (float)q.K;
    const double vd = (double)v;
    const float zn = vd > pen ? (float)(vd - pen) : (vd < -pen ? (float)(vd + pen) : 0.f);
    for (int k0 = 0; k0 < q.Kl; k0 += 8) {                                      // 8 workers' operands requested together
        float xv[8], yv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const size_t o = (size_t)min(k0 + u, q.Kl - 1) * q.ldv + i;
            xv[u] = q.x[o]; yv[u] = q.y[o];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (k0 + u < q.Kl) {
                const float x = xv[u];
                const float r = x - zn;
                const float yn = yv[u] + rho_f * r;
                q.y[(size_t)(k0 + u) * q.ldv + i] = yn;
                if (q.state != nullptr && d.out.total < q.state_cap) {     // record out.total = the trace record that will judge this iteration
                    float* s = q.state + (size_t)d.out.total * (1 + 2 * (size_t)q.Kl) * q.p;
                    s[(size_t)(1 + k0 + u) * q.p + i] = x; s[(size_t)(1 + q.Kl + k0 + u) * q.p + i] = yn;
                }
                acc[0] += (double)x * x; acc[1] += (double)yn * yn; acc[2] += (double)r * r;
            }
        }
    }
    const float dz = zn - zo;
    acc[3] += (double)zn * zn; acc[4] += (double)dz * dz;
    q.z[i] = zn;
    if (q.state != nullptr && d.out.total < q.state_cap) q.state[(size_t)d.out.total * (1 + 2 * (size_t)q.Kl) * q.p + i] = zn;
}

// z(g): decision for iteration g-1 from nsum (after the all-reduce every rank holds identical numbers and
// takes identical decisions: PADMMBase.h:216-221,230-231; eps :117-139), lambda schedule, then
// z_new = soft(w / K, lambda / (rho K)); y_k += rho (x_k - z_new); norms   (PADMMLasso.h:99-108, PADMMBase.h:70-78)
// PEER 0: the consensus sum is in q.wsum (single process, or all-reduced by the exchange layer between `pack` and this launch).
// PEER 1: the consensus sum and the worker-summed norms arrive in the K exchange slots (par_pack_kernel<1> of every rank).
// FUSED (round 6): `pack` and `z` in ONE launch -- every thread forms x_k and the consensus share of ITS elements first.  PEER 0
// (single process): the share IS the sum, nothing travels.  PEER 1: the launch is producer and consumer of the exchange (the host
// chooses it only when the grid is resident with room to spare: its workgroups wait for one another, and for the other ranks).
template <int PEER, bool FUSED>
__global__ void __launch_bounds__(kParThreads)
par_z_kernel(ParParams q, int par, PeerExchange ex) {
    // no fused multiply-adds in the elementwise arithmetic: the reference is built without them (see lasso_tall.hip, tall_update_elem)
#pragma clang fp contract(off)
    __shared__ double scratch[5 * (kParThreads / 64)];
    WIDE_PROBE_DECL
    WIDE_PROBE(0);
    const ParCtl in = load_ctl_vector(q.ctl + par);
    ParCtl* outp = &q.ctl[par ^ 1];
    if (in.done) {
        if (blockIdx.x == 0 && threadIdx.x == 0) *outp = in;
        return;
    }
    WIDE_PROBE(4);
    const float rho_f = (float)q.rho;
    double x2 = q.nsum[0], y2 = q.nsum[1], r2 = q.nsum[2];
    const double z2 = q.nsum[3], dz2 = q.nsum[4];
    if (FUSED && PEER) {                                                  // producer half: this workgroup's elements into every rank's slot
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            for (int dst = 0; dst < ex.nranks; ++dst) {
                unsigned long long* nd = reinterpret_cast<unsigned long long*>(peer_dst_slot(ex, dst) + par_peer_norm_offset(q.p));
                peer_store_u64(nd + 0, (unsigned long long)__double_as_longlong(x2));
                peer_store_u64(nd + 1, (unsigned long long)__double_as_longlong(y2));
                peer_store_u64(nd + 2, (unsigned long long)__double_as_longlong(r2));
            }
        }
        const int pe = (q.p + 1) / 2 * 2;
        for (int i = blockIdx.x * kParThreads + threadIdx.x; i < pe; i += gridDim.x * kParThreads) {
            const float w = i < q.p ? par_pack_elem(q, i, rho_f) : 0.f;
            const float wn = __shfl_down(w, 1, 64);
            if ((i & 1) == 0) {
                for (int dst = 0; dst < ex.nranks; ++dst)
                    peer_store_f32x2(reinterpret_cast<float*>(peer_dst_slot(ex, dst)) + i, w, wn);
            }
        }
        peer_publish(ex, gridDim.x);
    }
    if (PEER) {
        if (!peer_wait_relaxed(ex)) return;                               // a rank did not arrive in time: ADMM_ERR_COMM at the next poll
        x2 = 0.0; y2 = 0.0; r2 = 0.0;
        for (int r = 0; r < ex.nranks; ++r) {                             // rank order: identical sums on every rank
            const unsigned long long* nd = reinterpret_cast<const unsigned long long*>(peer_src_slot(ex, r) + par_peer_norm_offset(q.p));
            x2 += __longlong_as_double((long long)__hip_atomic_load(nd + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
            y2 += __longlong_as_double((long long)__hip_atomic_load(nd + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
            r2 += __longlong_as_double((long long)__hip_atomic_load(nd + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
        }
    }
#ifdef ADMM_HIP_PROBE
    asm volatile("" :: "v"(x2), "v"(y2), "v"(r2), "v"(z2), "v"(dz2));
#endif
    WIDE_PROBE(5);
    const ParDecision d = par_decide(q, in, x2, y2, r2, z2, dz2);
#ifdef ADMM_HIP_PROBE
    asm volatile("" :: "v"(d.out.eps_primal), "v"(d.out.eps_dual));
#endif
    WIDE_PROBE(6);
    if (blockIdx.x == 0 && threadIdx.x == 0) par_record(q, in, d, outp);
    WIDE_PROBE(1);
    const double pen = d.out.lam / (q.rho * (double)q.K);
    double acc[5] = {0, 0, 0, 0, 0};
    for (int i = blockIdx.x * kParThreads + threadIdx.x; i < q.p; i += gridDim.x * kParThreads) {
        float wtot;
        if (PEER) {
            wtot = 0.f;
            for (int r = 0; r < ex.nranks; ++r) {
                const float2 pr = peer_load_f32x2(reinterpret_cast<const float*>(peer_src_slot(ex, r)) + (i & ~1));
                wtot += (i & 1) ? pr.y : pr.x;
            }
        } else if (FUSED) {
            wtot = par_pack_elem(q, i, rho_f);
        } else {
            wtot = q.wsum[i];
        }
        par_z_elem(q, i, wtot, d, pen, rho_f, acc);
    }
    if (d.out.done) return;
    WIDE_PROBE(2);
    block_sum<double, 5>(acc, scratch);
    if (threadIdx.x == 0) {
        double* Pout = q.P + (size_t)blockIdx.x * 8;
#pragma unroll
        for (int k = 0; k < 5; ++k) Pout[k] = acc[k];
    }
    WIDE_PROBE(3);
    WIDE_PROBE_FLUSH(blockIdx.x == 0 ? 0 : (blockIdx.x == gridDim.x - 1 ? 1 : (blockIdx.x == gridDim.x / 2 ? 2 : -1)), in.total);
}

__global__ void par_init_kernel(ParParams q, double lam0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < q.p) {
        q.z[i] = 0.f; q.wsum[i] = 0.f;
        for (int k = 0; k < q.Kl; ++k) { const size_t o = (size_t)k * q.ldv + i; q.x[o] = 0.f; q.y[o] = 0.f; q.rhs[o] = 0.f; }
    }
    if (i < q.nwg * 8) q.P[i] = 0.0;
    if (i < 8) q.nsum[i] = 0.0;
    if (i == 0) {
        ParCtl c;
        c.lam = lam0; c.eps_primal = 0; c.eps_dual = 0; c.iter = 0; c.lam_idx = 0; c.done = 0; c.first = 1; c.total = 0;
        c.pad0 = c.pad1 = c.pad2 = 0;
        q.ctl[0] = c; q.ctl[1] = c;
        *q.done = 0;
    }
}

__global__ void copy_rows_kernel(const float* X, long long ldx, int row0, int nrows, int p, float* A, long long lda) {
    const int j = blockIdx.x;
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < nrows; i += gridDim.y * blockDim.x)
        A[(size_t)j * lda + i] = X[(size_t)j * ldx + row0 + i];
}

// (G + rho I)^-1 of a worker's small system, cached for the whole path (rho never changes: PADMMBase.h:147-159).  Same policy
// as the tall solver (lasso_tall.hip): below order 4096 the float Gram is factorised and inverted in DOUBLE and rounded to
// float once.  The reference solves with a float Cholesky factor (PADMMLasso.h:23-29); a float-built explicit inverse loses
// cond(G + rho I) ulps per entry, and the Woodbury form of a wide block then amplifies that by (sigma^2 + rho) / rho when it
// subtracts A'(...)A rhs from rhs: the stepwise check (oracle/stepcheck.py, round 3) measured x_k errors of up to 15 x the
// reference's own on ill-conditioned blocks with the float-built inverse.
static void par_inverse(float* M, long long ldm, int order, double rho, hipStream_t st) {
    bool inv64 = order < 4096;
    if (const char* e = option("INVERSE")) inv64 = std::string(e) == "f64";
    if (inv64) {
        spd_inverse_f32_via_f64(M, ldm, order, (double)(float)rho, st);
    } else {
        add_diag<float>(M, ldm, order, (float)rho, st);
        spd_inverse_f32(M, ldm, order, st);
    }
}

struct ParWorker {
    int rows = 0;
    bool wide = false;
    long long lda = 0, ldat = 0, ldm = 0;
    DevBuf<float> A, At, Minv, tvec, svec, G;
    GemvT<float> gA, gAt, gM;       // gA: A' v (p outputs); gAt: A v via the stored transpose (rows outputs); gM: cached inverse
};

struct ParPlan final : LassoPlan {
    DeviceData<float> d;
    LassoProblem pb;
    hipStream_t st;
    admm_stats setup_stats{};
    int p = 0, K = 0, Kl = 0, nlam = 0, nwg = 0;
    long long ldv = 0;
    CommInfo ci;
    DevBuf<double> nsum;
    double rho = 0;
    std::vector<double> lam_user, lam_int;
    std::vector<ParWorker> W;
    DevBuf<float> Ab, rhs, x, y, z, wsum, beta;
    DevBuf<int> niter, done;
    DevBuf<double> P, dlam;
    DevBuf<ParCtl> ctl;
#ifdef ADMM_HIP_PROBE
    DevBuf<long long> probe;
#endif
    ParParams q{};
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

    // several row blocks in this process, all of one branch: the workers' products of one kind go out as ONE launch each
    bool batched = false, bt_wide = false, bt_nt = false;
    DevBuf<GemvTArgs<float>> bAt, bM, bA;
    int gridAt = 0, gridM = 0, gridA = 0;
    size_t ldsAt = 0, ldsM = 0, ldsA = 0;
    bool peer_fused = false;          // multi-process over the PEER backend: the exchange is done by pack / z themselves
    // one-pass Woodbury workers (ParWb): default; ADMM_HIP_PAR_ONEPASS=0 keeps the reference's two products and the stored transpose
    bool onepass = false;
    GatherPlan gp;
    int wb_ld = 0, gather_tiles = 0;
    DevBuf<ParWb> wbd;
    DevBuf<double> cv, qv, azpart, azv, tv64, dpart, dlt;
    DevBuf<float> tvec;
    DevBuf<int> wbflag;
    DevBuf<unsigned long long> wbcount;
    DevBuf<double> wbnorm;
    DevBuf<unsigned int> wbarrive;
    int wb_nb = 0;
    bool fuse_pz = false;             // pack + z in one launch (single process; PEER: when the grid is resident with room to spare)
    DevBuf<GatherArgs<float>> bG, bGd;
    long long fallback_passes = 0;
    DevBuf<float> state;
    long long state_cap = 0;
    void enable_state(long long cap) override {
        const size_t rec = (size_t)(1 + 2 * Kl) * p;
        state.alloc((size_t)cap * rec);
        // on the solver's own (non-blocking) stream: a null-stream memset is not ordered against it and, on a busy device, landed
        // AFTER run() had copied record 0 into the dump (suspected cause of the one unreadable record 0 of the 40-process soak, case 546:23)
        ADMM_HIP_CHECK(hipMemsetAsync(state.get(), 0, (size_t)cap * rec * sizeof(float), st));
        state_cap = cap;
        q.state = state.get(); q.state_cap = cap;
    }
    long long read_state(float* out, long long cap, long long* rec_floats) override {
        const size_t rec = (size_t)(1 + 2 * Kl) * p;
        if (rec_floats) *rec_floats = (long long)rec;
        if (!out) return std::min(trace_n, state_cap);                             // size query
        const long long nrec = std::min(std::min(trace_n, state_cap), cap);
        if (nrec > 0 && out) read_back(out, state.get(), (size_t)nrec * rec * sizeof(float), st);
        return nrec;
    }

    ParPlan(DeviceData<float>&& data, const LassoProblem& prob, hipStream_t stream) : d(std::move(data)), pb(prob), st(stream) {
        const int n = d.n;
        const long long nt = d.n_total;
        p = d.p; K = pb.nworkers;
        ci = pb.dist ? comm_info() : CommInfo();
        ADMM_REQUIRE(K >= 1, "number of row blocks must be >= 1");
        ADMM_REQUIRE(K % ci.nranks == 0, "the number of row blocks must be a multiple of the number of ranks");
        Kl = K / ci.nranks;
        ADMM_REQUIRE(Kl <= kParMaxWorkers, "at most 64 row blocks per process");
        // the reference's partition of the GLOBAL rows (PADMMLasso.h:163-179): chunk = n / K, last block takes the remainder
        const long long chunk = nt / K;
        ADMM_REQUIRE(chunk >= 1, "more row blocks than rows");
        const long long expect = (long long)Kl * chunk + (ci.rank == ci.nranks - 1 ? nt - chunk * K : 0);
        ADMM_REQUIRE(expect == n, "this rank's row count does not match the reference row partition");
        admm_stats& S = setup_stats;
        S.branch = 2; S.t_h2d = d.t_h2d; S.t_standardize = d.t_std;
        ldv = round_up(p, 32);

        // lambda_0 from the full data (PADMMLasso.h:161)
        DevBuf<float> XY(ldv); XY.zero(st);
        gemv_t_simple<float>(d.X.get(), d.ldx, n, p, d.Y.get(), XY.get(), st);
        if (pb.dist) { allreduce_sum_f32(XY.get(), p, st); comm_stream_sync(st); }
        const float lambda0 = device_absmax<float>(XY.get(), p, st);
        lam_user = make_lambda_grid(pb, lambda0, (int)nt, (double)d.scaleY);
        nlam = (int)lam_user.size();
        lam_int.resize(nlam);
        for (int i = 0; i < nlam; ++i) lam_int[i] = lam_user[i] * (double)nt / (double)d.scaleY;   // `double lambda` in the master
        rho = pb.opts.rho;
        if (rho <= 0) rho = lam_int[0] / K;                                                // PADMMLasso.h:199-200
        S.rho = rho;

        // row partition (PADMMLasso.h:163-179) and per-worker factorisations (:48-63)
        Ab.alloc((size_t)Kl * ldv); Ab.zero(st);
        W.resize(Kl);
        onepass = true;
        if (const char* e = option("PAR_ONEPASS")) onepass = std::string(e) != "0";
        double t_gram = 0, t_fac = 0;
        for (int k = 0; k < Kl; ++k) {
            ParWorker& w = W[k];
            const int lo = (int)(k * chunk);
            w.rows = (k < Kl - 1) ? (int)chunk : n - lo;
            w.wide = w.rows < p;                                   // subA.rows() >= subA.cols() -> Cholesky branch
            w.lda = round_up(w.rows, 32);
            w.A.alloc((size_t)w.lda * p); w.A.zero(st);
            hipLaunchKernelGGL(copy_rows_kernel, dim3(p, std::min(64, (w.rows + 255) / 256)), dim3(256), 0, st, d.X.get(), d.ldx, lo, w.rows, p, w.A.get(), w.lda);
            DevBuf<float> bk(w.lda); bk.zero(st);
            ADMM_HIP_CHECK(hipMemcpyAsync(bk.get(), d.Y.get() + lo, (size_t)w.rows * sizeof(float), hipMemcpyDeviceToDevice, st));
            gemv_t_simple<float>(w.A.get(), w.lda, w.rows, p, bk.get(), Ab.get() + (size_t)k * ldv, st);   // A_k' b_k  (:42)
            double t0 = now_s();
            if (!w.wide) {
                w.ldm = round_up(p, 128);                      // whole 128-blocks for the matrix-core inverse
                w.Minv.alloc((size_t)w.ldm * w.ldm); w.Minv.zero(st);
                gram_full<float>(w.A.get(), w.lda, w.rows, p, true, w.Minv.get(), w.ldm, st);
                comm_stream_sync(st);
                t_gram += now_s() - t0; t0 = now_s();
                par_inverse(w.Minv.get(), w.ldm, p, rho, st);
                w.gM.init(w.Minv.get(), w.ldm, p, p);
                A_release_if_tall(w);
            } else {
                w.ldm = round_up(w.rows, 128);
                w.Minv.alloc((size_t)w.ldm * w.ldm); w.Minv.zero(st);
                gram_full<float>(w.A.get(), w.lda, w.rows, p, false, w.Minv.get(), w.ldm, st);
                comm_stream_sync(st);
                t_gram += now_s() - t0; t0 = now_s();
                if (onepass) {                               // the float Gram itself, for the residual of the small solve (par_wb_resid_kernel)
                    w.G.alloc((size_t)w.ldm * w.ldm);
                    ADMM_HIP_CHECK(hipMemcpyAsync(w.G.get(), w.Minv.get(), (size_t)w.ldm * w.ldm * sizeof(float), hipMemcpyDeviceToDevice, st));
                }
                par_inverse(w.Minv.get(), w.ldm, w.rows, rho, st);
                if (!onepass) {
                    w.ldat = round_up(p, 32);
                    w.At.alloc((size_t)w.ldat * w.rows); w.At.zero(st);
                    transpose<float>(w.A.get(), w.lda, w.rows, p, w.At.get(), w.ldat, st);
                    w.gAt.init(w.At.get(), w.ldat, p, w.rows);
                }
                w.gM.init(w.Minv.get(), w.ldm, w.rows, w.rows);
                w.gA.init(w.A.get(), w.lda, w.rows, p);
                w.tvec.alloc(w.ldm); w.svec.alloc(w.ldm); w.tvec.zero(st); w.svec.zero(st);
            }
            comm_stream_sync(st);
            t_fac += now_s() - t0;
        }
        S.t_gram = t_gram; S.t_factor = t_fac;
        d.X.release();
        {   // streaming policy of the products from the working set of ONE iteration on this GPU: all local workers' matrices
            size_t ws = 0;
            for (int k = 0; k < Kl; ++k) ws += W[k].wide ? (onepass ? 0 : W[k].gAt.bytes()) + W[k].gM.bytes() + W[k].gA.bytes() : W[k].gM.bytes();
            const bool nt = gemv_stream_nt(ws);
            for (int k = 0; k < Kl; ++k) { W[k].gM.set_nt(nt); if (W[k].wide) { if (!onepass) W[k].gAt.set_nt(nt); W[k].gA.set_nt(nt); } }
        }
        {   // one-pass form only where a Woodbury worker exists
            bool any_wide = false;
            for (int k = 0; k < Kl; ++k) any_wide = any_wide || W[k].wide;
            onepass = onepass && any_wide;
        }

        peer_fused = pb.dist && ci.active && ci.backend == COMM_PEER;
        if (const char* e = option("PEER_FUSED")) { if (std::string(e) == "0") peer_fused = false; }
        nwg = std::max(1, std::min(1024, (p + kParThreads - 1) / kParThreads));     // one element per thread up to p = 262144 (was <= 64 workgroups: 31 us for p = 10^5)
        // ADMM_HIP_PAR_FUSE_PZ=0: `pack` and `z` as two launches everywhere
        fuse_pz = !pb.dist;
        if (peer_fused) {
            int occ = 0;
            ADMM_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(par_z_kernel<1, true>), kParThreads, 0));
            fuse_pz = (long long)nwg * 2 <= resident_workgroups(occ);       // its workgroups wait for one another: resident with room to spare
        }
        if (const char* e = option("PAR_FUSE_PZ")) { if (std::string(e) == "0") fuse_pz = false; }
        rhs.alloc((size_t)Kl * ldv); x.alloc((size_t)Kl * ldv); y.alloc((size_t)Kl * ldv); nsum.alloc(8);
        z.alloc(ldv); wsum.alloc(ldv);
        rhs.zero(st); x.zero(st); y.zero(st);
        beta.alloc((size_t)nlam * p); niter.alloc(nlam); done.alloc(1);
        P.alloc((size_t)nwg * 8); dlam.alloc(nlam); ctl.alloc(2);
        ADMM_HIP_CHECK(hipMemcpyAsync(dlam.get(), lam_int.data(), nlam * sizeof(double), hipMemcpyHostToDevice, st));
        q.p = p; q.K = K; q.Kl = Kl; q.nsum = nsum.get(); q.maxit = pb.opts.maxit; q.nlam = nlam; q.nwg = nwg; q.ldv = ldv;
        q.rho = rho; q.eps_abs = pb.opts.eps_abs; q.eps_rel = pb.opts.eps_rel;
        q.lambdas = dlam.get(); q.Ab = Ab.get(); q.rhs = rhs.get(); q.x = x.get(); q.y = y.get(); q.z = z.get(); q.wsum = wsum.get();
        for (int k = 0; k < Kl; ++k) {
            ParWorker& w = W[k];
            GemvT<float>& last = w.wide ? w.gA : w.gM;
            q.gout[k] = last.part.get(); q.gnseg[k] = last.pl.nseg; q.gstride[k] = last.stride; q.wide[k] = w.wide ? 1 : 0;
        }
        q.ctl = ctl.get(); q.P = P.get(); q.beta = beta.get(); q.niter = niter.get(); q.done = done.get();
#ifdef ADMM_HIP_PROBE
        probe.alloc((size_t)4096 * 4 * 8); probe.zero(st);
        q.probe = probe.get();
#endif
        if (onepass) {
            int max_rows = 0, nwide = 0;
            for (int k = 0; k < Kl; ++k) if (W[k].wide) { max_rows = std::max(max_rows, W[k].rows); ++nwide; }
            wb_ld = round_up(max_rows, 32);
            gp = plan_gather<float>(max_rows, p, nwide);
            gather_tiles = gp.tiles;
            std::vector<ParWb> hwb(Kl);
            std::vector<GatherArgs<float>> hG(Kl), hG0(Kl), hGd(Kl);
            cv.alloc((size_t)Kl * wb_ld); qv.alloc((size_t)Kl * wb_ld); tvec.alloc((size_t)Kl * wb_ld);
            azv.alloc((size_t)Kl * wb_ld); tv64.alloc((size_t)Kl * wb_ld); wbflag.alloc(Kl); dlt.alloc((size_t)Kl * wb_ld); dlt.zero(st);
            azpart.alloc((size_t)Kl * gp.ngroups * wb_ld); dpart.alloc((size_t)Kl * gp.ngroups * wb_ld);
            cv.zero(st); qv.zero(st); tvec.zero(st); azpart.zero(st); azv.zero(st); tv64.zero(st); wbflag.zero(st); dpart.zero(st);
            GatherPlan gk = gp; gk.pstride = wb_ld;
            for (int k = 0; k < Kl; ++k) {
                ParWorker& w = W[k];
                hwb[k].spart = w.wide ? w.gM.part.get() : nullptr; hwb[k].sstride = w.gM.stride; hwb[k].snseg = w.gM.pl.nseg;
                hwb[k].rows = w.wide ? w.rows : 0;
                hwb[k].G = w.wide ? w.G.get() : nullptr; hwb[k].ldg = w.ldm;
                double* part = azpart.get() + (size_t)k * gp.ngroups * wb_ld;
                hG[k] = gather_args<float>(gk, w.A.get(), w.lda, hwb[k].rows, p, z.get(), part, done.get());
                hG0[k] = gather_args<float>(gk, w.A.get(), w.lda, hwb[k].rows, p, Ab.get() + (size_t)k * ldv, part, nullptr);
                // the fall-back pass of an iteration: A_k rhs_k (dense right-hand side), only where this worker's flag is up
                hGd[k] = gather_args<float>(gk, w.A.get(), w.lda, hwb[k].rows, p, rhs.get() + (size_t)k * ldv, dpart.get() + (size_t)k * gp.ngroups * wb_ld, done.get());
                hGd[k].only_if = wbflag.get() + k;
            }
            wbd.alloc(Kl); bG.alloc(Kl); bGd.alloc(Kl);
            ADMM_HIP_CHECK(hipMemcpyAsync(bGd.get(), hGd.data(), Kl * sizeof(GatherArgs<float>), hipMemcpyHostToDevice, st));
            ADMM_HIP_CHECK(hipMemcpyAsync(wbd.get(), hwb.data(), Kl * sizeof(ParWb), hipMemcpyHostToDevice, st));
            q.wb = wbd.get(); q.wb_ld = wb_ld; q.az_ng = gp.ngroups;
            q.cv = cv.get(); q.qv = qv.get(); q.tvec = tvec.get(); q.azpart = azpart.get();
            wbcount.alloc(1); wbcount.zero(st);
            wb_nb = (wb_ld + kParThreads - 1) / kParThreads;
            wbnorm.alloc((size_t)Kl * wb_nb * 2); wbnorm.zero(st);
            wbarrive.alloc((size_t)2 * Kl); wbarrive.zero(st);
            q.wb_nb = wb_nb; q.wbnorm = wbnorm.get(); q.wbarrive = wbarrive.get();
            q.azv = azv.get(); q.tv64 = tv64.get(); q.wbflag = wbflag.get(); q.dpart = dpart.get(); q.wbcount = wbcount.get(); q.dlt = dlt.get();
            {
                double tau = 1.0 / 16.0;                                                       // tau0: see par_wb_flag_kernel
                if (const char* e = option("PAR_ONEPASS_TAU")) tau = std::atof(e);       // 0: never fall back (the measurement of the failure); 1e30: always
                q.wb_tau2 = tau * tau;
            }
            // c_k = A_k (A_k'b_k): the same gather with the dense A_k'b_k as right-hand side, once
            ADMM_HIP_CHECK(hipMemcpyAsync(bG.get(), hG0.data(), Kl * sizeof(GatherArgs<float>), hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL((gather_batch_kernel<float>), dim3(gather_tiles, gp.ngroups, Kl), dim3(kGatherThreads), 0, st, bG.get());
            hipLaunchKernelGGL(par_wb_c_kernel, dim3((Kl * wb_ld + 255) / 256), dim3(256), 0, st, q, cv.get());
            comm_stream_sync(st);              // hG0 is a host temporary; the per-iteration blocks replace it
            ADMM_HIP_CHECK(hipMemcpyAsync(bG.get(), hG.data(), Kl * sizeof(GatherArgs<float>), hipMemcpyHostToDevice, st));
            comm_stream_sync(st);
        }
        // ---- batched launches of the workers' products (ADMM_HIP_PAR_BATCH=0: one launch per worker and product, as before)
        {
            const char* e = option("PAR_BATCH");
            // (one-pass workers: also for ONE block per process -- what K = N GPUs runs -- so that the residual of the small solve rides in the
            // first slice of the streaming launch there too instead of in a launch of its own)
            bool same = (Kl > 1 || onepass) && !(e && std::string(e) == "0");
            for (int k = 1; k < Kl && same; ++k) same = W[k].wide == W[0].wide && W[k].gM.pl.nt == W[0].gM.pl.nt;
            if (same) {
                const int* skip = done.get();
                std::vector<GemvTArgs<float>> hAt, hM, hA;
                bt_wide = W[0].wide; bt_nt = W[0].gM.pl.nt;
                for (int k = 0; k < Kl; ++k) {
                    ParWorker& w = W[k];
                    const float* rk = rhs.get() + (size_t)k * ldv;
                    if (!bt_wide) {
                        hM.push_back(w.gM.args_partials(rk, skip));
                    } else {
                        if (onepass) {
                            hM.push_back(w.gM.args_partials(tvec.get() + (size_t)k * wb_ld, skip));
                        } else {
                            hAt.push_back(w.gAt.args_partials(rk, skip));
                            hM.push_back(w.gM.args_partials_from(w.gAt, skip));
                            gridAt = std::max(gridAt, w.gAt.pl.grid); ldsAt = std::max(ldsAt, w.gAt.pl.lds_bytes);
                        }
                        hA.push_back(w.gA.args_partials_from(w.gM, skip));
                        gridA = std::max(gridA, w.gA.pl.grid); ldsA = std::max(ldsA, w.gA.pl.lds_bytes);
                    }
                    gridM = std::max(gridM, w.gM.pl.grid); ldsM = std::max(ldsM, w.gM.pl.lds_bytes);
                }
                auto up = [&](DevBuf<GemvTArgs<float>>& d, const std::vector<GemvTArgs<float>>& h) {
                    if (h.empty()) return;
                    d.alloc(h.size());
                    ADMM_HIP_CHECK(hipMemcpyAsync(d.get(), h.data(), h.size() * sizeof(GemvTArgs<float>), hipMemcpyHostToDevice, st));
                };
                up(bAt, hAt); up(bM, hM); up(bA, hA);
                batched = true;
            }
        }
        comm_stream_sync(st);
    }

    static void A_release_if_tall(ParWorker& w) { w.A.release(); }   // a tall block only needs A'b and the inverse

    void run(LassoResult& res) override {
        admm_stats S = setup_stats;
        res.lambda = lam_user;
        beta.zero(st); niter.zero(st);
        const int init_n = std::max(p, nwg * 8);
        hipLaunchKernelGGL(par_init_kernel, dim3((init_n + 255) / 256), dim3(256), 0, st, q, lam_int[0]);
        if (onepass) {               // y_k = 0, z = 0: q_k = A_k z = 0, and no s_k yet
            qv.zero(st); azpart.zero(st); dlt.zero(st);
            for (int k = 0; k < Kl; ++k) if (W[k].wide) W[k].gM.part.zero(st);
        }
        if (q.state != nullptr)      // record 0 of the iterate dump: A_k'b_k as the workers hold them, in the x_k slots
            for (int k = 0; k < Kl; ++k)
                ADMM_HIP_CHECK(hipMemcpyAsync(q.state + (size_t)(1 + k) * p, Ab.get() + (size_t)k * ldv, (size_t)p * sizeof(float), hipMemcpyDeviceToDevice, st));
        const int* skip = done.get();
        const int nwg_e = std::max(1, std::min(4 * device_info().num_cu, (p + kParThreads - 1) / kParThreads));
        const int batch = pb.batch_iters > 0 ? (pb.batch_iters + 1) / 2 * 2 : 16;
        LoopTimes lt = run_until_done(st, skip, batch, (long long)nlam * ((long long)pb.opts.maxit + 2) + 4, [&](long long g) {
            const int par = (int)(g & 1);
            hipLaunchKernelGGL(par_head_kernel, dim3(nwg_e + (onepass ? Kl * wb_nb : 0)), dim3(kParThreads), 0, st, q);      // rhs_k; one-pass: t_k, q_k and the guard's flag
            if (onepass)        // cancellation guard: the fall-back pass where a worker's flag is up (its workgroups leave at once otherwise), t_k / q_k from it
                hipLaunchKernelGGL(par_gather_fix_kernel, dim3(gather_tiles, gp.ngroups, Kl), dim3(kGatherThreads), 0, st, bGd.get(), q);
            if (batched) {
                // all workers' products of one kind in ONE launch (bit-identical to the per-worker launches below)
                if (bt_wide) {
                    if (!onepass) launch_gemv_t_batch<float>(bAt.get(), Kl, gridAt, ldsAt, bt_nt, st);      // t_k = A_k rhs_k   (one-pass form: formed by the head)
                    launch_gemv_t_batch<float>(bM.get(), Kl, gridM, ldsM, bt_nt, st);         // s_k = (A_k A_k' + rho I)^-1 t_k
                    if (onepass) {                                                            // A_k' s_k, and delta_k in the first slice of the same grid
                        if (bt_nt) hipLaunchKernelGGL((par_A_batch_resid_kernel<true>), dim3(gridA, Kl + 1), dim3(kGemvThreads), (std::uint32_t)ldsA, st, bA.get(), q);
                        else hipLaunchKernelGGL((par_A_batch_resid_kernel<false>), dim3(gridA, Kl + 1), dim3(kGemvThreads), (std::uint32_t)ldsA, st, bA.get(), q);
                    } else
                    launch_gemv_t_batch<float>(bA.get(), Kl, gridA, ldsA, bt_nt, st);         // A_k' s_k
                } else {
                    launch_gemv_t_batch<float>(bM.get(), Kl, gridM, ldsM, bt_nt, st);         // x_k = (A_k'A_k + rho I)^-1 rhs_k
                }
            } else
            for (int k = 0; k < Kl; ++k) {
                ParWorker& w = W[k];
                const float* rk = rhs.get() + (size_t)k * ldv;
                if (!w.wide) {
                    w.gM.run_partials(rk, skip, st);                       // x = (A'A + rho I)^-1 rhs
                } else {
                    // chained without reduction launches: each product sums the previous one's partial rows while staging
                    if (onepass) {
                        w.gM.run_partials(tvec.get() + (size_t)k * wb_ld, skip, st);      // s = (AA' + rho I)^-1 t, t from the head
                    } else {
                        w.gAt.run_partials(rk, skip, st);                  // t = A rhs
                        w.gM.run_partials_from(w.gAt, skip, st);           // s = (AA' + rho I)^-1 t
                    }
                    w.gA.run_partials_from(w.gM, skip, st);                // A' s
                }
            }
            if (onepass && !batched) hipLaunchKernelGGL(par_wb_resid_kernel, dim3((Kl * wb_ld + 3) / 4), dim3(256), 0, st, q);        // delta_k of every Woodbury worker
            if (peer_fused) {
                // the only cross-worker exchange, produced by `pack` and consumed by `z` themselves (PEER slots)
                const PeerExchange ex = comm_peer_begin(par_peer_norm_offset(p) + 3 * sizeof(double));
                if (fuse_pz) {
                    hipLaunchKernelGGL((par_z_kernel<1, true>), dim3(nwg), dim3(kParThreads), 0, st, q, par, ex);
                } else {
                    hipLaunchKernelGGL(par_pack_kernel<1>, dim3(nwg_e), dim3(kParThreads), 0, st, q, ex);
                    hipLaunchKernelGGL((par_z_kernel<1, false>), dim3(nwg), dim3(kParThreads), 0, st, q, par, ex);
                }
                if (onepass) hipLaunchKernelGGL((gather_batch_kernel<float>), dim3(gather_tiles, gp.ngroups, Kl), dim3(kGatherThreads), 0, st, bG.get());
                return;
            }
            if (fuse_pz) {       // single process: nothing travels between `pack` and `z`
                hipLaunchKernelGGL((par_z_kernel<0, true>), dim3(nwg), dim3(kParThreads), 0, st, q, par, PeerExchange{});
            } else {
                hipLaunchKernelGGL(par_pack_kernel<0>, dim3(nwg_e), dim3(kParThreads), 0, st, q, PeerExchange{});
                // the only cross-worker exchange: consensus sum (p floats) + the three worker-summed norms, one grouped
                // RCCL all-reduce over xGMI (no-op in a single process)
                if (pb.dist) allreduce_sum_f32_f64(wsum.get(), (size_t)p, nsum.get(), 3, st);
                hipLaunchKernelGGL((par_z_kernel<0, false>), dim3(nwg), dim3(kParThreads), 0, st, q, par, PeerExchange{});
            }
            if (onepass) hipLaunchKernelGGL((gather_batch_kernel<float>), dim3(gather_tiles, gp.ngroups, Kl), dim3(kGatherThreads), 0, st, bG.get());      // A_k z_new over the non-zeros of z
        });
        S.t_loop = lt.wall_s; S.loop_ms_events = lt.events_ms; S.xupdate_launches = lt.launched;
        S.exchange_variant = !pb.dist ? 0 : (peer_fused ? (fuse_pz ? 3 : 2) : 1);      // 3: `pack` and `z` as ONE launch, producer and consumer of the exchange
        if (onepass) {
            unsigned long long hc = 0;
            ADMM_HIP_CHECK(hipMemcpy(&hc, wbcount.get(), sizeof(hc), hipMemcpyDeviceToHost));
            fallback_passes = (long long)hc;
            wbcount.zero(st);
            if (option("PAR_ONEPASS_STATS"))
                std::fprintf(stderr, "[consensus one-pass] %lld worker-iterations took the dense fall-back pass (cancellation guard) of %lld x %d\n", fallback_passes, (long long)lt.launched, Kl);
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
        std::vector<float> hb((size_t)nlam * p);
        read_back(hb.data(), beta.get(), hb.size() * sizeof(float), st);
        res.beta.assign((size_t)(p + 1) * nlam, 0.f);
        long long tot = 0;
        for (int l = 0; l < nlam; ++l) {
            float b0 = 0.f;
            recover_coef<float>(d, hb.data() + (size_t)l * p, &b0, res.beta.data() + (size_t)l * (p + 1) + 1);
            res.beta[(size_t)l * (p + 1)] = b0;
            tot += res.niter[l];
        }
        S.total_iter = tot;
        {   // decisions taken = the cold-start one + one per ADMM iteration
            ParCtl hc[2];
            ADMM_HIP_CHECK(hipMemcpy(hc, ctl.get(), sizeof(hc), hipMemcpyDeviceToHost));
            trace_n = std::max(hc[0].total, hc[1].total);
        }
        res.stats = S;
    }
};

std::unique_ptr<LassoPlan> make_par_plan(DeviceData<float>&& d, const LassoProblem& pb, hipStream_t st) {
    return std::unique_ptr<LassoPlan>(new ParPlan(std::move(d), pb, st));
}

}  // namespace admm
