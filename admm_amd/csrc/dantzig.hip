// Dantzig selector (fp64), device resident; admm_hip_dantzig.
//
//   min ||beta||_1  s.t.  ||X'(X beta - y)||_inf <= lambda
//
// What R's admm_dantzig(x, y)$fit() asks for, .Call("admm_dantzig", ...) (/root/reference/R/50_admm_dantzig.R:30-46): a
// symbol the reference never builds.  Its source, /root/reference/src/TODO/ADMMDantzig.h (:132-144 x-update, :170-191 z-update /
// residual / dual, :196-204 thresholds, :222-259 setup) and TODO/Dantzig.cpp:32-99 (DataStd<double>, lambda grid, warm-started
// loop, recover), is written against an older ADMMBase; it is restated on the CURRENT driver ADMMBase::solve
// (/root/reference/src/ADMMBase.h:158-216, update_rho :85-109) in oracle/solvers.py (class Dantzig), which this file follows.
// SURVEY.md section 8f row n3.
//
// Linearised ADMM on x = beta with A = X'X, c = X'y, gamma = (loose Lanczos value of A)^2, rho = 1 / sqrt(gamma) unless given:
//   x-update   rhs = (A x + z + y / rho - c) / (-gamma);   x <- soft(A rhs + x, 1 / (rho gamma))
//   z-update   zz = A x + y / rho - c;   z_i = -min(zz_i, lambda) if zz_i > 0 else min(-zz_i, lambda)
//   dual       r = A x + z - c;   y <- y + rho r
// A is formed explicitly when n > p and p <= 1000 (ADMMDantzig.h:222), otherwise every product is X'(X v) on the two stored
// layouts of X.  Per iteration: `head` (the decision of the previous iteration -- convergence, next lambda with a warm start
// and the coefficient snapshot, rho adaptation from its fifth iteration -- evaluated identically by every workgroup from the
// norm partials; then rhs), one or two streaming mat-vecs, `mid` (soft threshold), the same mat-vecs on x, `tail` (z, r, y and
// the five squared norms).  The host enqueues iterations in batches and polls a sticky flag.
//
// KNOWN BEHAVIOUR, pinned in tests/test_oracle_dantzig.py: the iteration converges on comfortably tall problems (n >= 5 p) and
// does not for p > n within the R default maxit at any tolerance -- the step 1 / gamma uses the square of the LOOSE Lanczos
// value, 6-16 % below ||X'X||^2, so the linearisation does not majorise.  That is the algorithm the reference holds; it is
// reproduced, not repaired.
#include "prep.h"
#include "gemv_kernels.h"
#include "solvers.h"
#include "loop_driver.h"

namespace admm {

struct DzCtl {                                    // 64 bytes: whole 16-byte words (load_ctl_vector)
    double rho, eps_primal, eps_dual, lam;
    int iter, done, lam_idx, total, first, niter_last, pad0, pad1;
};

struct DzParams {
    int p, maxit, nlam, nwg_tail;
    double eps_abs, eps_rel, sqrt_p, gamma, sqrt_gamma, xy_norm, lambda0;
    const double* lam;                            // [nlam] internal lambdas
    const double* XY;
    double *x, *z, *y, *Ax, *rhs;
    const double* vec;                            // A rhs
    double* beta;                                 // [nlam][p] snapshots of x
    int* niter;                                   // [nlam]
    DzCtl* ctl;                                   // [2]
    double* P;                                    // [nwg_tail][8]
    int* done; int* hflag;
    double* trace; long long trace_cap;
};

constexpr int kDzThreads = 256;

__global__ void __launch_bounds__(kDzThreads)
dz_head_kernel(DzParams q, int par) {
    __shared__ double red[8 * 4];
    const DzCtl in = load_ctl_vector(q.ctl + par);
    DzCtl* outp = &q.ctl[par ^ 1];
    if (in.done) {
        if (blockIdx.x == 0 && threadIdx.x == 0) *outp = in;
        return;
    }
    double s[5] = {0, 0, 0, 0, 0};
    for (int w = threadIdx.x; w < q.nwg_tail; w += kDzThreads) {
#pragma unroll
        for (int k = 0; k < 5; ++k) s[k] += q.P[w * 8 + k];
    }
    block_sum<double, 5>(s, red);
    const double r2 = s[0], dz2 = s[1], ax2 = s[2], z2 = s[3], y2 = s[4];
    DzCtl out = in;
    out.first = 0;
    bool snapshot = false;
    double tr_rp = 0, tr_rd = 0, tr_code = ADMM_TRACE_COLD;
    if (!in.first) {                                           // the iteration just finished: index in.iter - 1 of lambda in.lam_idx
        const double rp = sqrt(r2), rd = in.rho * q.sqrt_gamma * sqrt(dz2);
        tr_rp = rp; tr_rd = rd; tr_code = ADMM_TRACE_CONTINUE;
        const bool conv = rp < in.eps_primal && rd < in.eps_dual;
        if (conv || in.iter >= q.maxit) {                      // ADMMBase::solve returns i + 1 (converged) or maxit + 1 (:206-215)
            out.niter_last = conv ? in.iter : q.maxit + 1;
            snapshot = true;
            if (conv) tr_code = ADMM_TRACE_CONVERGED;
            out.iter = 0;
            out.lam_idx = in.lam_idx + 1;
            if (out.lam_idx >= q.nlam) out.done = 1;
            else out.lam = q.lam[out.lam_idx];                 // init_warm: lambda and the iteration counter only (rho stays)
        }
        if (!conv && in.iter - 1 > 3) {                        // update_rho() after iterations i > 3 (ADMMBase.h:211-212; :85-109)
            double rho = in.rho;
            if (rp / in.eps_primal > 10 * rd / in.eps_dual) rho *= 2;
            else if (rd / in.eps_dual > 10 * rp / in.eps_primal) rho /= 2;
            if (rp < in.eps_primal) rho /= 1.2;
            if (rd < in.eps_dual) rho *= 1.2;
            out.rho = rho;
        }
    }
    out.eps_primal = fmax(fmax(sqrt(ax2), sqrt(z2)), q.xy_norm) * q.eps_rel + q.sqrt_p * q.eps_abs;
    out.eps_dual = q.sqrt_gamma * sqrt(y2) * q.eps_rel + q.sqrt_p * q.eps_abs;
    if (!out.done) out.iter = out.iter + 1;                    // the iteration this launch starts has index out.iter - 1
    out.total = in.total + 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *outp = out;
        if (snapshot) q.niter[in.lam_idx] = out.niter_last;
        if (out.done) { *q.done = 1; if (q.hflag) *q.hflag = 1; }
        if (q.trace != nullptr && in.total < q.trace_cap) {
            double* t = q.trace + (size_t)in.total * ADMM_TRACE_FIELDS;
            t[0] = in.lam_idx; t[1] = in.iter - 1; t[2] = in.eps_primal; t[3] = in.eps_dual; t[4] = tr_rp; t[5] = tr_rd;
            t[6] = 0.0; t[7] = 0.0; t[8] = tr_code; t[9] = in.rho; t[10] = out.rho; t[11] = in.lam;
        }
    }
    const double rho = out.rho;
    const bool null_model = out.lam > q.lambda0 - 1e-5;       // ADMMDantzig.h:134: x stays zero
    for (int i = blockIdx.x * kDzThreads + threadIdx.x; i < q.p; i += gridDim.x * kDzThreads) {
#pragma clang fp contract(off)
        if (snapshot) q.beta[(size_t)in.lam_idx * q.p + i] = q.x[i];
        if (!out.done && !null_model) q.rhs[i] = (((q.Ax[i] + q.z[i]) + q.y[i] / rho) - q.XY[i]) / (-q.gamma);
    }
}

__global__ void __launch_bounds__(kDzThreads)
dz_mid_kernel(DzParams q, int par) {
    const DzCtl c = load_ctl_vector(q.ctl + par);             // written by this iteration's head
    if (c.done) return;
    const bool null_model = c.lam > q.lambda0 - 1e-5;
    const double pen = 1.0 / (c.rho * q.gamma);
    for (int i = blockIdx.x * kDzThreads + threadIdx.x; i < q.p; i += gridDim.x * kDzThreads) {
#pragma clang fp contract(off)
        double xn = 0.0;
        if (!null_model) {
            const double v = q.vec[i] + q.x[i];
            xn = v > pen ? v - pen : (v < -pen ? v + pen : 0.0);
        }
        q.x[i] = xn;
    }
}

__global__ void __launch_bounds__(kDzThreads)
dz_tail_kernel(DzParams q, int par) {
    __shared__ double red[8 * 4];
    const DzCtl c = load_ctl_vector(q.ctl + par);
    if (c.done) return;
    const double rho = c.rho, lam = c.lam;
    double s[5] = {0, 0, 0, 0, 0};
    for (int i = blockIdx.x * kDzThreads + threadIdx.x; i < q.p; i += gridDim.x * kDzThreads) {
#pragma clang fp contract(off)
        const double ax = q.Ax[i], yo = q.y[i], xy = q.XY[i];
        const double zz = (ax + yo / rho) - xy;
        const double zn = zz > 0 ? -fmin(zz, lam) : fmin(-zz, lam);
        const double dz = zn - q.z[i];
        q.z[i] = zn;
        const double r = (ax + zn) - xy;
        const double yn = yo + rho * r;
        q.y[i] = yn;
        s[0] += r * r; s[1] += dz * dz; s[2] += ax * ax; s[3] += zn * zn; s[4] += yn * yn;
    }
    block_sum<double, 5>(s, red);
    if (threadIdx.x == 0) {
        double* P = q.P + (size_t)blockIdx.x * 8;
#pragma unroll
        for (int k = 0; k < 5; ++k) P[k] = s[k];
    }
}

static int dz_batch() {
    const char* e = option("BATCH_ITERS");
    const int v = e ? std::atoi(e) : 0;
    return v > 0 ? (v + 1) / 2 * 2 : 16;
}

// d: the standardised data (DataStd<double>, Dantzig.cpp:52-55).  pb: lambda grid / options.  res.beta: (p + 1) x nlambda doubles.
void solve_dantzig(DeviceData<double>& d, const LassoProblem& pb, DantzigResult& res, hipStream_t st) {
    const int n = d.n, p = d.p;
    admm_stats& S = res.stats;
    S.branch = 7;
    const long long ldv = round_up(p, 32);
    // ---- X'y, lambda_0, the lambda grid (Dantzig.cpp:57-79)
    DevBuf<double> XY(ldv);
    XY.zero(st);
    gemv_t_simple<double>(d.X.get(), d.ldx, n, p, d.Y.get(), XY.get(), st);
    std::vector<double> hxy(p);
    ADMM_HIP_CHECK(hipMemcpy(hxy.data(), XY.get(), (size_t)p * sizeof(double), hipMemcpyDeviceToHost));
    double lambda0 = 0, xyn = 0;
    for (int j = 0; j < p; ++j) { lambda0 = std::max(lambda0, std::fabs(hxy[j])); xyn += hxy[j] * hxy[j]; }
    xyn = std::sqrt(xyn);
    res.lambda = make_lambda_grid(pb, lambda0, n, (double)d.scaleY);
    const int nlam = (int)res.lambda.size();
    std::vector<double> lam_int(nlam);
    for (int i = 0; i < nlam; ++i) lam_int[i] = res.lambda[i] * (double)n / (double)d.scaleY;

    // ---- the operator A = X'X: explicit for small tall problems, two streaming products otherwise (ADMMDantzig.h:222-224)
    const bool use_xx = n > p && p <= 1000;
    double t0 = now_s();
    DevBuf<double> XX, Xt, tn(round_up(n, 32));
    GemvT<double> gA, gX, gXt;
    if (use_xx) {
        XX.alloc((size_t)ldv * p); XX.zero(st);
        gram_full<double>(d.X.get(), d.ldx, n, p, true, XX.get(), ldv, st);
        gA.init(XX.get(), ldv, p, p);
    } else {
        const long long ldt = round_up(p, 32);
        Xt.alloc((size_t)ldt * n); Xt.zero(st);
        transpose<double>(d.X.get(), d.ldx, n, p, Xt.get(), ldt, st);
        gXt.init(Xt.get(), ldt, p, n);                          // t = X v  (the columns of X' are the rows of X)
        gX.init(d.X.get(), d.ldx, n, p);                        // w = X' t
        const bool nt = gemv_stream_nt(gXt.bytes() + gX.bytes());
        gXt.set_nt(nt); gX.set_nt(nt);
    }
    ADMM_HIP_CHECK(hipStreamSynchronize(st));
    S.t_gram = now_s() - t0;
    auto amult = [&](const double* v, double* w, const int* skip) {     // w = A v, device vectors
        if (use_xx) { gA.run(v, w, skip, st); return; }
        gXt.run_partials(v, skip, st);
        gX.run_from(gXt, w, skip, st);
    };

    // ---- gamma = (loose Lanczos value)^2, rho (ADMMDantzig.h:226-233, 256-259)
    t0 = now_s();
    DevBuf<double> dv(ldv), dw(ldv);
    dv.zero(st); dw.zero(st);
    int nmatop = 0;
    const double ev = lanczos_largest_f64([&](const double* vh, double* wh) {
        ADMM_HIP_CHECK(hipMemcpyAsync(dv.get(), vh, (size_t)p * sizeof(double), hipMemcpyHostToDevice, st));
        amult(dv.get(), dw.get(), nullptr);
        ADMM_HIP_CHECK(hipMemcpyAsync(wh, dw.get(), (size_t)p * sizeof(double), hipMemcpyDeviceToHost, st));
        ADMM_HIP_CHECK(hipStreamSynchronize(st));
    }, p, &nmatop);
    const double gamma = ev * ev;
    const double rho0 = pb.opts.rho > 0 ? pb.opts.rho : 1.0 / std::sqrt(gamma);
    S.eig_est = ev; S.rho = rho0; S.t_eigs = now_s() - t0;

    // ---- loop state
    int ncu = 256;
    { hipDeviceProp_t prop; int dev = 0; ADMM_HIP_CHECK(hipGetDevice(&dev)); ADMM_HIP_CHECK(hipGetDeviceProperties(&prop, dev)); ncu = prop.multiProcessorCount; }
    const int nwg = std::max(1, std::min(ncu, (p + kDzThreads - 1) / kDzThreads));
    DevBuf<double> x(ldv), z(ldv), y(ldv), Ax(ldv), rhs(ldv), vec(ldv), beta((size_t)nlam * p), P((size_t)nwg * 8), dlam(nlam), trace;
    DevBuf<int> dniter(nlam), ddone(1);
    DevBuf<DzCtl> ctl(2);
    x.zero(st); z.zero(st); y.zero(st); Ax.zero(st); rhs.zero(st); vec.zero(st); beta.zero(st); P.zero(st); dniter.zero(st); ddone.zero(st);
    ADMM_HIP_CHECK(hipMemcpyAsync(dlam.get(), lam_int.data(), (size_t)nlam * sizeof(double), hipMemcpyHostToDevice, st));
    DzCtl c0{};
    c0.rho = rho0; c0.lam = lam_int[0]; c0.first = 1;
    ADMM_HIP_CHECK(hipMemcpyAsync(ctl.get(), &c0, sizeof(DzCtl), hipMemcpyHostToDevice, st));
    ADMM_HIP_CHECK(hipMemcpyAsync(ctl.get() + 1, &c0, sizeof(DzCtl), hipMemcpyHostToDevice, st));
    if (res.trace_cap > 0) { trace.alloc((size_t)res.trace_cap * ADMM_TRACE_FIELDS); trace.zero(st); }
    PinnedFlag hflag;
    DzParams q{};
    q.p = p; q.maxit = pb.opts.maxit; q.nlam = nlam; q.nwg_tail = nwg;
    q.eps_abs = pb.opts.eps_abs; q.eps_rel = pb.opts.eps_rel; q.sqrt_p = std::sqrt((double)p);
    q.gamma = gamma; q.sqrt_gamma = std::sqrt(gamma); q.xy_norm = xyn; q.lambda0 = lambda0;
    q.lam = dlam.get(); q.XY = XY.get();
    q.x = x.get(); q.z = z.get(); q.y = y.get(); q.Ax = Ax.get(); q.rhs = rhs.get(); q.vec = vec.get();
    q.beta = beta.get(); q.niter = dniter.get(); q.ctl = ctl.get(); q.P = P.get(); q.done = ddone.get(); q.hflag = hflag.p;
    q.trace = res.trace_cap > 0 ? trace.get() : nullptr; q.trace_cap = res.trace_cap;

    ADMM_HIP_CHECK(hipStreamSynchronize(st));
    const int* skip = ddone.get();
    LoopTimes lt = run_until_done(st, ddone.get(), dz_batch(), (long long)pb.opts.maxit * nlam + 2 * nlam + 2, [&](long long g) {
        const int par = (int)(g & 1);
        hipLaunchKernelGGL(dz_head_kernel, dim3(nwg), dim3(kDzThreads), 0, st, q, par);
        amult(rhs.get(), vec.get(), skip);
        hipLaunchKernelGGL(dz_mid_kernel, dim3(nwg), dim3(kDzThreads), 0, st, q, par ^ 1);
        amult(x.get(), Ax.get(), skip);
        hipLaunchKernelGGL(dz_tail_kernel, dim3(nwg), dim3(kDzThreads), 0, st, q, par ^ 1);
    }, hflag.p);

    // ---- results: recover every column (Dantzig.cpp:88-93)
    std::vector<double> hb((size_t)nlam * p);
    std::vector<int> hn(nlam);
    read_back(hb.data(), beta.get(), hb.size() * sizeof(double), st);
    ADMM_HIP_CHECK(hipMemcpy(hn.data(), dniter.get(), (size_t)nlam * sizeof(int), hipMemcpyDeviceToHost));
    res.beta.assign((size_t)(p + 1) * nlam, 0.0);
    res.niter = hn;
    long long tot = 0;
    for (int i = 0; i < nlam; ++i) {
        double b0 = 0;
        recover_coef<double>(d, hb.data() + (size_t)i * p, &b0, res.beta.data() + (size_t)i * (p + 1) + 1);
        res.beta[(size_t)i * (p + 1)] = b0;
        tot += std::min(hn[i], pb.opts.maxit);
    }
    if (res.trace_cap > 0) {
        DzCtl hc[2];
        ADMM_HIP_CHECK(hipMemcpy(hc, ctl.get(), sizeof(hc), hipMemcpyDeviceToHost));
        const long long nrec = std::min<long long>(std::max(hc[0].total, hc[1].total), res.trace_cap);
        res.trace.assign((size_t)nrec * ADMM_TRACE_FIELDS, 0.0);
        if (nrec > 0) read_back(res.trace.data(), trace.get(), res.trace.size() * sizeof(double), st);
    }
    S.total_iter = tot;
    S.t_loop = lt.wall_s;
    S.loop_ms_events = lt.events_ms;
    S.xupdate_variant = use_xx ? 0 : 1;
    S.xupdate_samples = nmatop;
}

}  // namespace admm
