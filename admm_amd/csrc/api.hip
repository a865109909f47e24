// extern "C" boundary of libadmm_hip.so (see include/admm_hip.h).
#include <cstring>
#include <unordered_map>
#include "solvers.h"
#include "comm.h"
#include <mutex>

namespace admm {
const std::string& last_error_ref();
void test_symv(const float* A, int p, const float* v0, const float* v1, float* y0, float* y1);
template <typename T> void test_gram(const T* A, int rows, int cols, bool atA, T* G);
template <typename T> void test_spd_inverse(const T* A, int n, T* Ainv, bool via64);
template <typename T> void test_gemv_t(const T* A, int rows, int cols, const T* v, T* y);
template <typename T> void test_gather(const T* A, int rows, int cols, const T* v, double* y);
int comm_unique_id(void* out);
void comm_init(int nranks, int rank, const void* idbytes);
void comm_finalize();
void comm_peer_prepare(int nranks, void* handle_out);
void comm_init_peer(int nranks, int rank, const void* handles);
void comm_init_shm(int nranks, int rank, const char* name, unsigned long long token);
void cv_gather(const double* x, long long ldx, const double* y, const int* d_idx, int m, int p, double* xo, double* yo, hipStream_t st);
std::vector<double> cv_score(const double* xt, const double* yt, int m, int p, const float* beta_host, int nlam, hipStream_t st);

// ---- variant selectors / tuning values (admm_internal.h: option)
}  // namespace admm
extern char** environ;
namespace admm {
namespace {
using OptMap = std::unordered_map<std::string, std::string>;
// the ADMM_HIP_* variables of the environment the library was first used in -- read once, never again
const OptMap& option_overlay() {
    static const OptMap* m = []() {
        OptMap* o = new OptMap();
        for (char** e = environ; e && *e; ++e) {
            if (std::strncmp(*e, "ADMM_HIP_", 9) != 0) continue;
            const char* eq = std::strchr(*e, '=');
            if (eq && eq > *e + 9) (*o)[std::string(*e + 9, (size_t)(eq - (*e + 9)))] = std::string(eq + 1);
        }
        return o;
    }();
    return *m;
}
OptMap& option_thread() { static thread_local OptMap m; return m; }
}  // namespace
const char* option(const char* name) {
    const OptMap& t = option_thread();
    if (!t.empty()) { auto it = t.find(name); if (it != t.end()) return it->second.c_str(); }
    const OptMap& o = option_overlay();
    auto it = o.find(name);
    return it != o.end() ? it->second.c_str() : nullptr;
}
int option_int(const char* name, int dflt) {
    const char* v = option(name);
    return v ? std::atoi(v) : dflt;
}
void option_set_thread(const char* name, const char* value) {
    if (value) option_thread()[name] = value; else option_thread().erase(name);
}
void options_reset_thread() { option_thread().clear(); }

// ---- cache of large device blocks (admm_internal.h, DevBuf)
namespace {
struct PoolBlock { void* p; size_t bytes; int dev; };
struct Pool {
    std::mutex mu;
    std::vector<PoolBlock> blocks;
    size_t cached = 0;
};
Pool& pool() { static Pool* p = new Pool(); return *p; }             // never destroyed: the runtime may be gone at process exit
constexpr size_t kPoolMinBytes = size_t(32) << 20;
// Cap of the cache: ADMM_HIP_POOL_MB, default the smaller of 16 GB and an eighth of the device's memory (24 GB held back from
// every other allocator of the process -- library workspaces, RCCL, torch, an R session -- was too much to keep silently; what a warm
// C2 setup re-uses is ~10 GB).  Allocators of this library that bypass DevBuf (the PEER exchange buffer) trim the cache and retry on
// out-of-memory themselves; admm_hip_trim_memory() hands everything back on request.
size_t pool_cap_bytes() {
    if (const char* e = option("POOL_MB")) { const long long mb = std::atoll(e); return mb > 0 ? (size_t)mb << 20 : size_t(0); }
    static const size_t dflt = []() {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); total_b = size_t(128) << 30; }
        return std::min(size_t(16) << 30, total_b / 8);
    }();
    return dflt;
}
void pool_drop_locked(Pool& P, int dev) {                               // dev < 0: every device
    size_t w = 0;
    for (size_t i = 0; i < P.blocks.size(); ++i) {
        if (dev < 0 || P.blocks[i].dev == dev) { (void)hipFree(P.blocks[i].p); P.cached -= P.blocks[i].bytes; }
        else P.blocks[w++] = P.blocks[i];
    }
    P.blocks.resize(w);
}
}  // namespace

void* pool_alloc(size_t bytes, size_t* granted) {
    *granted = bytes;
    int dev = 0;
    if (bytes >= kPoolMinBytes && pool_cap_bytes() > 0) {
        ADMM_HIP_CHECK(hipGetDevice(&dev));
        Pool& P = pool();
        std::lock_guard<std::mutex> lk(P.mu);
        int best = -1;
        for (size_t i = 0; i < P.blocks.size(); ++i) {
            const PoolBlock& b = P.blocks[i];
            if (b.dev == dev && b.bytes >= bytes && b.bytes <= bytes + bytes / 4 && (best < 0 || b.bytes < P.blocks[best].bytes)) best = (int)i;
        }
        if (best >= 0) {
            void* p = P.blocks[best].p;
            *granted = P.blocks[best].bytes;
            P.cached -= P.blocks[best].bytes;
            P.blocks.erase(P.blocks.begin() + best);
            return p;
        }
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation) {      // give the cached blocks back and try once more
        (void)hipGetLastError();
        { Pool& P = pool(); std::lock_guard<std::mutex> lk(P.mu); pool_drop_locked(P, -1); }
        e = hipMalloc(&p, bytes);
    }
    if (e != hipSuccess)
        throw Error(ADMM_ERR_HIP, std::string("hipMalloc(") + std::to_string(bytes) + " bytes): " + hipGetErrorString(e));
    return p;
}

void pool_free(void* p, size_t granted) {
    if (!p) return;
    if (granted >= kPoolMinBytes && pool_cap_bytes() > 0) {
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess) {
            hipPointerAttribute_t at;
            if (hipPointerGetAttributes(&at, p) == hipSuccess) dev = at.device;      // the device the block lives on (the caller may have switched)
            Pool& P = pool();
            bool keep = false;
            {   // reserve the room first, synchronise OUTSIDE the lock (other threads' allocations must not queue behind this device's streams)
                std::lock_guard<std::mutex> lk(P.mu);
                if (P.cached + granted <= pool_cap_bytes()) { P.cached += granted; keep = true; }
            }
            if (keep) {
                // a block may be handed to another stream's work next: everything enqueued on ITS device must have finished (hipFree would
                // have waited too) -- under a device guard: the caller may have switched devices since the block was allocated
                int cur = dev;
                (void)hipGetDevice(&cur);
                if (cur != dev) (void)hipSetDevice(dev);
                (void)hipDeviceSynchronize();
                if (cur != dev) (void)hipSetDevice(cur);
                std::lock_guard<std::mutex> lk(P.mu);
                P.blocks.push_back({p, granted, dev});          // (its bytes are already counted)
                return;
            }
        }
    }
    (void)hipFree(p);
}

size_t pool_cached_bytes() {
    Pool& P = pool();
    std::lock_guard<std::mutex> lk(P.mu);
    int dev = 0;
    (void)hipGetDevice(&dev);
    size_t s = 0;
    for (const PoolBlock& b : P.blocks) if (b.dev == dev) s += b.bytes;
    return s;
}

void pool_trim() {
    Pool& P = pool();
    std::lock_guard<std::mutex> lk(P.mu);
    pool_drop_locked(P, -1);
}

std::vector<double> make_lambda_grid(const LassoProblem& pb, double lambda0, int n, double scaleY) {
    if (!pb.lambda_in.empty()) return pb.lambda_in;
    // Lasso.cpp:78-89: lmax = lambda0 / n * scaleY; log-spaced down to lmin_ratio * lmax
    const int nl = pb.nlambda_auto;
    const double lmax = lambda0 / n * scaleY;
    const double lmin = pb.lmin_ratio * lmax;
    std::vector<double> lam(nl);
    const double lo = std::log(lmax), hi = std::log(lmin);
    for (int i = 0; i < nl; ++i) {
        const double t = nl > 1 ? lo + (hi - lo) * ((double)i / (double)(nl - 1)) : lo;
        lam[i] = std::exp(i == nl - 1 && nl > 1 ? hi : t);
    }
    return lam;
}


template <typename F>
static int guarded(F&& f) {
    try {
        f();
        return ADMM_OK;
    } catch (const Error& e) {
        set_last_error(e.what());
        return e.code;
    } catch (const std::bad_alloc&) {
        set_last_error("host allocation failed");
        return ADMM_ERR_INTERNAL;
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return ADMM_ERR_INTERNAL;
    }
}

static void check_common(const double* x, const double* y, int n, int p, int mem, const admm_opts* opts) {
    ADMM_REQUIRE(x != nullptr && y != nullptr, "x and y must not be NULL");
    ADMM_REQUIRE(n > 0 && p > 0, "n and p must be positive");
    ADMM_REQUIRE(mem == ADMM_MEM_HOST || mem == ADMM_MEM_DEVICE, "mem must be ADMM_MEM_HOST or ADMM_MEM_DEVICE");
    ADMM_REQUIRE(opts != nullptr, "opts must not be NULL");
    ADMM_REQUIRE(opts->maxit > 0, "maxit should be positive");                                  // R/30_admm_lasso.R:119-120
    ADMM_REQUIRE(opts->eps_abs >= 0 && opts->eps_rel >= 0, "eps_abs and eps_rel should be nonnegative");
}

struct PlanHandle {
    Stream st;
    std::unique_ptr<LassoPlan> plan;
    int p = 0, nlam = 0;
    double t_create = 0;
};

static LassoProblem make_problem(const double* lambda_in, int nlambda_in, int nlambda_auto, double lmin_ratio, bool enet, double alpha,
                                 int nworkers, bool dist, const admm_opts* opts) {
    LassoProblem pb;
    pb.opts = *opts;
    pb.lambda_in.assign(lambda_in, lambda_in + nlambda_in);
    pb.nlambda_auto = nlambda_auto;
    pb.lmin_ratio = lmin_ratio;
    pb.enet = enet;
    pb.alpha = alpha;
    pb.nworkers = nworkers;
    pb.dist = dist;
    pb.batch_iters = option_int("BATCH_ITERS", 0);
    pb.profile_stride = option_int("PROFILE_STRIDE", 0);
    return pb;
}

static PlanHandle* create_plan(const double* x, const double* y, int n, int p, int mem,
                               const double* lambda_in, int nlambda_in, int nlambda_auto, double lmin_ratio,
                               int standardize, int intercept, bool enet, double alpha, int nworkers,
                               const admm_opts* opts, long long n_total = 0) {
    check_common(x, y, n, p, mem, opts);
    ADMM_REQUIRE(nlambda_in >= 0, "nlambda_in must be >= 0");
    ADMM_REQUIRE(nlambda_in > 0 ? lambda_in != nullptr : nlambda_auto > 0, "need a lambda grid or nlambda_auto > 0");
    if (nlambda_in == 0) ADMM_REQUIRE(lmin_ratio > 0 && lmin_ratio < 1, "lambda_min_ratio must be within (0, 1)");
    for (int i = 0; i < nlambda_in; ++i) ADMM_REQUIRE(lambda_in[i] > 0, "lambda must be positive");
    const bool dist = n_total > 0;
    if (nworkers > 0 && !dist) ADMM_REQUIRE(nworkers <= n, "more row blocks than rows");
    if (dist) {
        ADMM_REQUIRE(nworkers >= 0, "nthread must be >= 0");
        ADMM_REQUIRE(comm_info().active, "no communicator: call admm_hip_comm_init first");
        if (nworkers == 0) ADMM_REQUIRE(n_total > p, "the row-sharded serial solver is the tall one: it needs n_total > p");
    }
    require_device();
    const double t0 = now_s();
    std::unique_ptr<PlanHandle> h(new PlanHandle());
    const LassoProblem pb = make_problem(lambda_in, nlambda_in, nlambda_auto, lmin_ratio, enet, alpha, nworkers, dist, opts);
    DeviceData<float> d;
    // Host input of a large tall problem: standardisation and X'X run under the PCIe transfer (bit-identical result).
    const char* eg = option("GRAM");
    const bool pipelined = mem == ADMM_MEM_HOST && !dist && nworkers <= 0 && n > p && p >= 4096 &&
                           !(eg && (std::string(eg) == "rocblas" || std::string(eg) == "oneshot"));
    if (pipelined) upload_standardize_gram_f32(d, x, y, n, p, standardize != 0, intercept != 0, h->st.s);
    else upload_standardize<float>(d, x, y, n, p, mem, standardize != 0, intercept != 0, h->st.s, dist ? n_total : 0);
    if (nworkers > 0) h->plan = make_par_plan(std::move(d), pb, h->st.s);
    else if ((dist ? n_total : (long long)n) > p) h->plan = make_tall_plan(std::move(d), pb, h->st.s);      // Lasso.cpp:73
    else h->plan = make_wide_plan(std::move(d), pb, h->st.s);
    h->p = p;
    h->nlam = nlambda_in > 0 ? nlambda_in : nlambda_auto;
    h->t_create = now_s() - t0;
    return h.release();
}

// Column-sharded wide solver: this rank holds columns [col_offset, col_offset + p_local) of the n x p_total problem.
static PlanHandle* create_plan_cols(const double* x_cols, const double* y, int n, int p_local, long long p_total, long long col_offset, int mem,
                                    const double* lambda_in, int nlambda_in, int nlambda_auto, double lmin_ratio,
                                    int standardize, int intercept, bool enet, double alpha, const admm_opts* opts) {
    check_common(x_cols, y, n, p_local, mem, opts);
    ADMM_REQUIRE(nlambda_in >= 0, "nlambda_in must be >= 0");
    ADMM_REQUIRE(nlambda_in > 0 ? lambda_in != nullptr : nlambda_auto > 0, "need a lambda grid or nlambda_auto > 0");
    if (nlambda_in == 0) ADMM_REQUIRE(lmin_ratio > 0 && lmin_ratio < 1, "lambda_min_ratio must be within (0, 1)");
    for (int i = 0; i < nlambda_in; ++i) ADMM_REQUIRE(lambda_in[i] > 0, "lambda must be positive");
    ADMM_REQUIRE(p_total >= p_local && col_offset >= 0 && col_offset + p_local <= p_total, "column block outside [0, p_total)");
    ADMM_REQUIRE(p_total < (1ll << 31) - 1, "p_total too large");
    ADMM_REQUIRE((long long)n <= p_total, "the column-sharded solver is the wide one: it needs n <= p_total (Lasso.cpp:73)");
    ADMM_REQUIRE(comm_info().active, "no communicator: call admm_hip_comm_init first");
    require_device();
    const double t0 = now_s();
    std::unique_ptr<PlanHandle> h(new PlanHandle());
    LassoProblem pb;
    pb.opts = *opts;
    pb.lambda_in.assign(lambda_in, lambda_in + nlambda_in);
    pb.nlambda_auto = nlambda_auto;
    pb.lmin_ratio = lmin_ratio;
    pb.enet = enet;
    pb.alpha = alpha;
    pb.p_total = p_total;
    pb.col_offset = col_offset;
    pb.batch_iters = option_int("BATCH_ITERS", 0);
    DeviceData<float> d;
    upload_standardize<float>(d, x_cols, y, n, p_local, mem, standardize != 0, intercept != 0, h->st.s, 0);   // column moments are local, y is replicated
    h->plan = make_wide_plan(std::move(d), pb, h->st.s);
    h->p = (int)p_total;
    h->nlam = nlambda_in > 0 ? nlambda_in : nlambda_auto;
    h->t_create = now_s() - t0;
    return h.release();
}

static void run_plan(PlanHandle* h, double* lambda_out, float* beta_out, int* niter_out, admm_stats* stats, double t_extra) {
    ADMM_REQUIRE(h != nullptr && h->plan, "plan is NULL");
    ADMM_REQUIRE(lambda_out && beta_out && niter_out, "output pointers must not be NULL");
    const double t0 = now_s();
    LassoResult res;
    res.beta_dst = beta_out;
    h->plan->run(res);
    const int nl = (int)res.lambda.size();
    for (int i = 0; i < nl; ++i) { lambda_out[i] = res.lambda[i]; niter_out[i] = res.niter[i]; }
    if (!res.beta_written) std::memcpy(beta_out, res.beta.data(), sizeof(float) * (size_t)(h->p + 1) * nl);
    res.stats.t_total = now_s() - t0 + t_extra;
    if (stats) *stats = res.stats;
}

static int lasso_family(const double* x, const double* y, int n, int p, int mem,
                        const double* lambda_in, int nlambda_in, int nlambda_auto, double lmin_ratio,
                        int standardize, int intercept, bool enet, double alpha, int nworkers,
                        const admm_opts* opts, double* lambda_out, float* beta_out, int* niter_out, admm_stats* stats) {
    return guarded([&] {
        ADMM_REQUIRE(lambda_out && beta_out && niter_out, "output pointers must not be NULL");
        std::unique_ptr<PlanHandle> h(create_plan(x, y, n, p, mem, lambda_in, nlambda_in, nlambda_auto, lmin_ratio,
                                                  standardize, intercept, enet, alpha, nworkers, opts));
        run_plan(h.get(), lambda_out, beta_out, niter_out, stats, h->t_create);
    });
}

// K-fold cross-validation (cv.hip).  The full-data fit fixes the lambda grid; fold f is the ordinary plan on the rows with
// fold_id != f, scored on the rows with fold_id == f.  With a communicator the folds are dealt out to the ranks (fold f on
// rank f mod nranks: independent replicas, nothing exchanged on the data path) and the score / iteration tables are
// summed over the ranks at the end.
static void lasso_cv(const double* x, const double* y, int n, int p, int mem, const int* fold_id, int nfolds,
                     const double* lambda_in, int nlambda_in, int nlambda_auto, double lmin_ratio,
                     int standardize, int intercept, double alpha, const admm_opts* opts,
                     double* lambda_out, float* beta_out, int* niter_out,
                     double* cv_mean, double* cv_se, double* fold_mse, int* fold_niter, float* fold_beta,
                     int* idx_min, int* idx_1se, admm_stats* stats) {
    check_common(x, y, n, p, mem, opts);
    ADMM_REQUIRE(nfolds >= 2 && nfolds <= n, "nfolds must be within [2, n]");
    ADMM_REQUIRE(lambda_out && cv_mean && cv_se, "lambda_out, cv_mean and cv_se must not be NULL");
    const bool enet = alpha >= 0.0;
    if (enet) ADMM_REQUIRE(alpha <= 1.0, "alpha must be within [0, 1]");
    require_device();
    const double t0 = now_s();
    std::vector<int> fid(n);
    for (int i = 0; i < n; ++i) {
        fid[i] = fold_id ? fold_id[i] : i % nfolds;
        ADMM_REQUIRE(fid[i] >= 0 && fid[i] < nfolds, "fold_id entries must be within [0, nfolds)");
    }
    std::vector<int> cnt(nfolds, 0);
    for (int i = 0; i < n; ++i) ++cnt[fid[i]];
    for (int f = 0; f < nfolds; ++f) ADMM_REQUIRE(cnt[f] > 0 && cnt[f] < n, "every fold needs at least one held-out row and one training row");

    Stream st;
    // one resident copy of the data (doubles, as handed over); folds are gathered from it on the device
    DevBuf<double> xd_own, yd_own;
    const double* xd = x; const double* yd = y;
    if (mem == ADMM_MEM_HOST) {
        xd_own.alloc((size_t)n * p); yd_own.alloc(n);
        write_device(xd_own.get(), x, (size_t)n * p * sizeof(double));
        write_device(yd_own.get(), y, (size_t)n * sizeof(double));
        comm_stream_sync(st.s);
        xd = xd_own.get(); yd = yd_own.get();
    }
    // Folds as down-dates of the full-data Gram (cv.hip): when every fit of the call is the tall solver's and the Gram is
    // what setup costs (p >= 1024; ADMM_HIP_CV_DOWNDATE=1 / 0 forces it on for any tall call / off).
    int min_tr = n;
    for (int f = 0; f < nfolds; ++f) min_tr = std::min(min_tr, n - cnt[f]);
    bool downdate = min_tr > p && p >= 1024;
    if (const char* e = option("CV_DOWNDATE")) downdate = min_tr > p && std::string(e) == "1";
    CvBase base;
    if (downdate) {
        ADMM_REQUIRE(nlambda_in >= 0, "nlambda_in must be >= 0");
        ADMM_REQUIRE(nlambda_in > 0 ? lambda_in != nullptr : nlambda_auto > 0, "need a lambda grid or nlambda_auto > 0");
        if (nlambda_in == 0) ADMM_REQUIRE(lmin_ratio > 0 && lmin_ratio < 1, "lambda_min_ratio must be within (0, 1)");
        for (int i = 0; i < nlambda_in; ++i) ADMM_REQUIRE(lambda_in[i] > 0, "lambda must be positive");
        cv_downdate_prepare(base, xd, yd, n, p, standardize != 0, intercept != 0, st.s);
    }
    // ---- full-data fit: the lambda grid (and, if asked for, the coefficients)
    int nlam = 0;
    std::vector<double> lam;
    {
        std::unique_ptr<PlanHandle> h;
        if (downdate) {
            h.reset(new PlanHandle());
            DeviceData<float> d;
            cv_downdate_full(d, base, h->st.s);
            h->plan = make_tall_plan(std::move(d), make_problem(lambda_in, nlambda_in, nlambda_auto, lmin_ratio, enet, enet ? alpha : 1.0, 0, false, opts), h->st.s);
        } else {
            h.reset(create_plan(xd, yd, n, p, ADMM_MEM_DEVICE, lambda_in, nlambda_in, nlambda_auto, lmin_ratio,
                                standardize, intercept, enet, enet ? alpha : 1.0, 0, opts));
        }
        LassoResult res;
        h->plan->run(res);
        nlam = (int)res.lambda.size();
        lam = res.lambda;
        for (int l = 0; l < nlam; ++l) lambda_out[l] = lam[l];
        if (beta_out) std::memcpy(beta_out, res.beta.data(), sizeof(float) * (size_t)(p + 1) * nlam);
        if (niter_out) for (int l = 0; l < nlam; ++l) niter_out[l] = res.niter[l];
        if (stats) *stats = res.stats;
    }
    // ---- folds
    const CommInfo ci = comm_info();
    const int nranks = ci.active ? ci.nranks : 1, rank = ci.active ? ci.rank : 0;
    std::vector<double> mse((size_t)nfolds * nlam, 0.0);
    std::vector<double> nit((size_t)nfolds * nlam, 0.0);         // as doubles: summed over ranks with the scores
    if (fold_beta) std::memset(fold_beta, 0, sizeof(float) * (size_t)(p + 1) * nlam * nfolds);
    DevBuf<int> didx(n);
    for (int f = 0; f < nfolds; ++f) {
        if (f % nranks != rank) continue;
        std::vector<int> tr, te;
        for (int i = 0; i < n; ++i) (fid[i] == f ? te : tr).push_back(i);
        const int ntr = (int)tr.size(), nte = (int)te.size();
        std::vector<int> both(tr);
        both.insert(both.end(), te.begin(), te.end());
        ADMM_HIP_CHECK(hipMemcpyAsync(didx.get(), both.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, st.s));
        DevBuf<double> xtr, ytr, xte((size_t)nte * p), yte(nte);
        cv_gather(xd, n, yd, didx.get() + ntr, nte, p, xte.get(), yte.get(), st.s);
        LassoResult res;
        if (downdate) {
            std::unique_ptr<PlanHandle> h(new PlanHandle());
            DeviceData<float> d;
            cv_downdate_fold(d, base, yd, didx.get(), ntr, didx.get() + ntr, nte, st.s);
            h->plan = make_tall_plan(std::move(d), make_problem(lam.data(), nlam, 0, lmin_ratio, enet, enet ? alpha : 1.0, 0, false, opts), h->st.s);
            h->plan->run(res);
        } else {
            xtr.alloc((size_t)ntr * p); ytr.alloc(ntr);
            cv_gather(xd, n, yd, didx.get(), ntr, p, xtr.get(), ytr.get(), st.s);
            comm_stream_sync(st.s);
            std::unique_ptr<PlanHandle> h(create_plan(xtr.get(), ytr.get(), ntr, p, ADMM_MEM_DEVICE, lam.data(), nlam, 0, lmin_ratio,
                                                      standardize, intercept, enet, enet ? alpha : 1.0, 0, opts));
            h->plan->run(res);
        }
        const std::vector<double> sse = cv_score(xte.get(), yte.get(), nte, p, res.beta.data(), nlam, st.s);
        for (int l = 0; l < nlam; ++l) { mse[(size_t)f * nlam + l] = sse[l] / nte; nit[(size_t)f * nlam + l] = res.niter[l]; }
        if (fold_beta) std::memcpy(fold_beta + (size_t)f * (p + 1) * nlam, res.beta.data(), sizeof(float) * (size_t)(p + 1) * nlam);
    }
    if (nranks > 1) {                                             // folds of the other ranks: one sum all-reduce of the tables
        DevBuf<double> t((size_t)2 * nfolds * nlam);
        ADMM_HIP_CHECK(hipMemcpyAsync(t.get(), mse.data(), mse.size() * sizeof(double), hipMemcpyHostToDevice, st.s));
        ADMM_HIP_CHECK(hipMemcpyAsync(t.get() + mse.size(), nit.data(), nit.size() * sizeof(double), hipMemcpyHostToDevice, st.s));
        allreduce_sum_f64(t.get(), (size_t)2 * nfolds * nlam, st.s);
        ADMM_HIP_CHECK(hipMemcpyAsync(mse.data(), t.get(), mse.size() * sizeof(double), hipMemcpyDeviceToHost, st.s));
        ADMM_HIP_CHECK(hipMemcpyAsync(nit.data(), t.get() + mse.size(), nit.size() * sizeof(double), hipMemcpyDeviceToHost, st.s));
        comm_stream_sync(st.s);
        comm_check();
        if (fold_beta) {
            const size_t nfb = (size_t)(p + 1) * nlam * nfolds;
            DevBuf<float> fb(nfb);
            ADMM_HIP_CHECK(hipMemcpyAsync(fb.get(), fold_beta, nfb * sizeof(float), hipMemcpyHostToDevice, st.s));
            allreduce_sum_f32(fb.get(), nfb, st.s);
            read_back(fold_beta, fb.get(), nfb * sizeof(float), st.s);
            comm_stream_sync(st.s);
            comm_check();
        }
    }
    // ---- summary: mean over folds, standard error sd / sqrt(K) (sample sd over the folds), minimum and one-standard-error rule
    int imin = 0;
    for (int l = 0; l < nlam; ++l) {
        double m = 0;
        for (int f = 0; f < nfolds; ++f) m += mse[(size_t)f * nlam + l];
        m /= nfolds;
        double v = 0;
        for (int f = 0; f < nfolds; ++f) { const double d = mse[(size_t)f * nlam + l] - m; v += d * d; }
        cv_mean[l] = m;
        cv_se[l] = std::sqrt(v / (nfolds - 1) / nfolds);
        if (cv_mean[l] < cv_mean[imin]) imin = l;
    }
    int i1se = imin;
    for (int l = 0; l < nlam; ++l) if (lam[l] > lam[i1se] && cv_mean[l] <= cv_mean[imin] + cv_se[imin]) i1se = l;   // largest lambda within one standard error
    if (idx_min) *idx_min = imin;
    if (idx_1se) *idx_1se = i1se;
    if (fold_mse) std::memcpy(fold_mse, mse.data(), mse.size() * sizeof(double));
    if (fold_niter) for (size_t k = 0; k < nit.size(); ++k) fold_niter[k] = (int)std::llround(nit[k]);
    if (stats) stats->t_total = now_s() - t0;
}

// Several responses of one design matrix ("batched / multi-response"): response j is the ordinary
// fit of (x, Y[:, j]) -- bit-identical to admm_hip_lasso / admm_hip_enet on that pair -- but x is uploaded, converted and
// standardised once, and for the tall solver X'X is formed once (it does not depend on y; the cached inverse does, through
// rho, and is rebuilt per response).  With a communicator the responses are dealt out to the ranks (response j on rank
// j mod nranks, independent replicas) and the outputs are summed over the ranks at the end.
static void lasso_multi(const double* x, const double* Y, int n, int p, int m, int mem,
                        const double* lambda_in, int nlambda_in, int nlambda_auto, double lmin_ratio,
                        int standardize, int intercept, double alpha, const admm_opts* opts,
                        double* lambda_out, float* beta_out, int* niter_out, admm_stats* stats) {
    check_common(x, Y, n, p, mem, opts);
    ADMM_REQUIRE(m >= 1, "the number of responses must be >= 1");
    ADMM_REQUIRE(lambda_out && beta_out && niter_out, "output pointers must not be NULL");
    ADMM_REQUIRE(nlambda_in >= 0, "nlambda_in must be >= 0");
    ADMM_REQUIRE(nlambda_in > 0 ? lambda_in != nullptr : nlambda_auto > 0, "need a lambda grid or nlambda_auto > 0");
    const bool enet = alpha >= 0.0;
    if (enet) ADMM_REQUIRE(alpha <= 1.0, "alpha must be within [0, 1]");
    require_device();
    const int nlam = nlambda_in > 0 ? nlambda_in : nlambda_auto;
    const size_t bsz = (size_t)(p + 1) * nlam;
    Stream st;
    DevBuf<double> xd_own, yd_own;
    const double* xd = x; const double* yd = Y;
    if (mem == ADMM_MEM_HOST) {
        xd_own.alloc((size_t)n * p); yd_own.alloc((size_t)n * m);
        write_device(xd_own.get(), x, (size_t)n * p * sizeof(double));
        write_device(yd_own.get(), Y, (size_t)n * m * sizeof(double));
        comm_stream_sync(st.s);
        xd = xd_own.get(); yd = yd_own.get();
    }
    const CommInfo ci = comm_info();
    const int nranks = ci.active ? ci.nranks : 1, rank = ci.active ? ci.rank : 0;
    std::vector<double> lam((size_t)m * nlam, 0.0), nit((size_t)m * nlam, 0.0);
    std::memset(beta_out, 0, sizeof(float) * bsz * m);
    if (stats) std::memset(stats, 0, sizeof(admm_stats) * (size_t)m);

    LassoProblem pb;
    pb.opts = *opts;
    pb.lambda_in.assign(lambda_in, lambda_in + nlambda_in);
    pb.nlambda_auto = nlambda_auto;
    pb.lmin_ratio = lmin_ratio;
    pb.enet = enet;
    pb.alpha = enet ? alpha : 1.0;
    pb.batch_iters = option_int("BATCH_ITERS", 0);
    pb.profile_stride = 0;
    if (nlambda_in == 0) ADMM_REQUIRE(lmin_ratio > 0 && lmin_ratio < 1, "lambda_min_ratio must be within (0, 1)");
    for (int i = 0; i < nlambda_in; ++i) ADMM_REQUIRE(lambda_in[i] > 0, "lambda must be positive");

    const bool tall = n > p;                                           // Lasso.cpp:73
    DeviceData<float> base;
    DevBuf<float> G;
    long long ldg = 0;
    double t_shared = 0;
    int first = -1;
    for (int j = 0; j < m; ++j) if (j % nranks == rank) { first = j; break; }
    if (first >= 0) {
        const double t0 = now_s();
        upload_standardize<float>(base, xd, yd + (size_t)first * n, n, p, ADMM_MEM_DEVICE, standardize != 0, intercept != 0, st.s, 0);
        if (tall) {
            ldg = round_up(p, 128);
            G.alloc((size_t)ldg * ldg); G.zero(st.s);
            gram_full<float>(base.X.get(), base.ldx, n, p, true, G.get(), ldg, st.s);
            comm_stream_sync(st.s);
        }
        t_shared = now_s() - t0;
    }
    for (int j = 0; j < m; ++j) {
        if (j % nranks != rank) continue;
        const double t0 = now_s();
        DeviceData<float> d;
        clone_with_response_f32(d, base, tall ? G.get() : nullptr, ldg, yd + (size_t)j * n, st.s);
        std::unique_ptr<LassoPlan> plan = tall ? make_tall_plan(std::move(d), pb, st.s) : make_wide_plan(std::move(d), pb, st.s);
        LassoResult res;
        plan->run(res);
        ADMM_REQUIRE((int)res.lambda.size() == nlam, "internal: unexpected path length");
        for (int l = 0; l < nlam; ++l) { lam[(size_t)j * nlam + l] = res.lambda[l]; nit[(size_t)j * nlam + l] = res.niter[l]; }
        std::memcpy(beta_out + (size_t)j * bsz, res.beta.data(), sizeof(float) * bsz);
        if (stats) { stats[j] = res.stats; stats[j].t_total = now_s() - t0 + (j == first ? t_shared : 0.0); }
    }
    if (nranks > 1) {                                                  // the other ranks' responses: sum all-reduces of the outputs
        DevBuf<double> t((size_t)2 * m * nlam);
        ADMM_HIP_CHECK(hipMemcpyAsync(t.get(), lam.data(), lam.size() * sizeof(double), hipMemcpyHostToDevice, st.s));
        ADMM_HIP_CHECK(hipMemcpyAsync(t.get() + lam.size(), nit.data(), nit.size() * sizeof(double), hipMemcpyHostToDevice, st.s));
        allreduce_sum_f64(t.get(), (size_t)2 * m * nlam, st.s);
        ADMM_HIP_CHECK(hipMemcpyAsync(lam.data(), t.get(), lam.size() * sizeof(double), hipMemcpyDeviceToHost, st.s));
        ADMM_HIP_CHECK(hipMemcpyAsync(nit.data(), t.get() + lam.size(), nit.size() * sizeof(double), hipMemcpyDeviceToHost, st.s));
        DevBuf<float> fb(bsz * m);
        ADMM_HIP_CHECK(hipMemcpyAsync(fb.get(), beta_out, bsz * m * sizeof(float), hipMemcpyHostToDevice, st.s));
        allreduce_sum_f32(fb.get(), bsz * m, st.s);
        read_back(beta_out, fb.get(), bsz * m * sizeof(float), st.s);
        comm_stream_sync(st.s);
        comm_check();
    }
    for (size_t k = 0; k < lam.size(); ++k) { lambda_out[k] = lam[k]; niter_out[k] = (int)std::llround(nit[k]); }
}

}  // namespace admm

using namespace admm;

extern "C" {

int admm_hip_lasso(const double* x, const double* y, int n, int p, int mem,
                   const double* lambda_in, int nlambda_in, int nlambda_auto, double lmin_ratio,
                   int standardize, int intercept, const admm_opts* opts,
                   double* lambda_out, float* beta_out, int* niter_out, admm_stats* stats) {
    return lasso_family(x, y, n, p, mem, lambda_in, nlambda_in, nlambda_auto, lmin_ratio, standardize, intercept,
                        false, 1.0, 0, opts, lambda_out, beta_out, niter_out, stats);
}

int admm_hip_enet(const double* x, const double* y, int n, int p, int mem,
                  const double* lambda_in, int nlambda_in, int nlambda_auto, double lmin_ratio,
                  int standardize, int intercept, double alpha, const admm_opts* opts,
                  double* lambda_out, float* beta_out, int* niter_out, admm_stats* stats) {
    if (!(alpha >= 0.0 && alpha <= 1.0)) {                       // R/40_admm_enet.R:38-39
        set_last_error("alpha must be within [0, 1]");
        return ADMM_ERR_INVALID_ARG;
    }
    return lasso_family(x, y, n, p, mem, lambda_in, nlambda_in, nlambda_auto, lmin_ratio, standardize, intercept,
                        true, alpha, 0, opts, lambda_out, beta_out, niter_out, stats);
}

int admm_hip_lasso_cv(const double* x, const double* y, int n, int p, int mem, const int* fold_id, int nfolds,
                      const double* lambda_in, int nlambda_in, int nlambda_auto, double lmin_ratio,
                      int standardize, int intercept, double alpha, const admm_opts* opts,
                      double* lambda_out, float* beta_out, int* niter_out,
                      double* cv_mean, double* cv_se, double* fold_mse, int* fold_niter, float* fold_beta,
                      int* idx_min, int* idx_1se, admm_stats* stats) {
    return guarded([&] {
        lasso_cv(x, y, n, p, mem, fold_id, nfolds, lambda_in, nlambda_in, nlambda_auto, lmin_ratio, standardize, intercept, alpha, opts,
                 lambda_out, beta_out, niter_out, cv_mean, cv_se, fold_mse, fold_niter, fold_beta, idx_min, idx_1se, stats);
    });
}

int admm_hip_lasso_multi(const double* x, const double* Y, int n, int p, int m, int mem,
                         const double* lambda_in, int nlambda_in, int nlambda_auto, double lmin_ratio,
                         int standardize, int intercept, double alpha, const admm_opts* opts,
                         double* lambda_out, float* beta_out, int* niter_out, admm_stats* stats) {
    return guarded([&] {
        lasso_multi(x, Y, n, p, m, mem, lambda_in, nlambda_in, nlambda_auto, lmin_ratio, standardize, intercept, alpha, opts,
                    lambda_out, beta_out, niter_out, stats);
    });
}

int admm_hip_parlasso(const double* x, const double* y, int n, int p, int mem,
                      const double* lambda_in, int nlambda_in, int nlambda_auto, double lmin_ratio,
                      int standardize, int intercept, int nthread, const admm_opts* opts,
                      double* lambda_out, float* beta_out, int* niter_out, admm_stats* stats) {
    if (nthread < 1) {
        set_last_error("nthread must be >= 1");
        return ADMM_ERR_INVALID_ARG;
    }
    return lasso_family(x, y, n, p, mem, lambda_in, nlambda_in, nlambda_auto, lmin_ratio, standardize, intercept,
                        false, 1.0, nthread, opts, lambda_out, beta_out, niter_out, stats);
}

int admm_hip_lad(const double* x, const double* y, int n, int p, int mem, int intercept,
                 const admm_opts* opts, double* beta_out, int* niter_out, admm_stats* stats) {
    return admm_hip_lad_traced(x, y, n, p, mem, intercept, opts, beta_out, niter_out, stats, nullptr, 0, nullptr);
}

int admm_hip_lad_traced(const double* x, const double* y, int n, int p, int mem, int intercept, const admm_opts* opts,
                        double* beta_out, int* niter_out, admm_stats* stats, double* trace_out, long long trace_cap, long long* ntrace_out) {
    return admm_hip_lad_state(x, y, n, p, mem, intercept, opts, beta_out, niter_out, stats, trace_out, trace_cap, ntrace_out, nullptr, 0, nullptr);
}

// copies the iterate dump of a LAD / BP run into the caller's buffer (admm_hip_lad_state / admm_hip_bp_state)
static void dense_state_out(const DenseResult& res, double* state_out, long long state_cap, long long* nstate_out) {
    if (state_cap <= 0) return;
    std::memcpy(state_out, res.state.data(), res.state.size() * sizeof(double));
    *nstate_out = res.state_dim > 0 ? (long long)(res.state.size() / (5 * (size_t)res.state_dim)) : 0;
}

int admm_hip_lad_state(const double* x, const double* y, int n, int p, int mem, int intercept, const admm_opts* opts,
                       double* beta_out, int* niter_out, admm_stats* stats, double* trace_out, long long trace_cap, long long* ntrace_out,
                       double* state_out, long long state_cap, long long* nstate_out) {
    return guarded([&] {
        ADMM_REQUIRE(trace_cap == 0 || (trace_out != nullptr && ntrace_out != nullptr && trace_cap > 0), "bad trace arguments");
        ADMM_REQUIRE(state_cap == 0 || (state_out != nullptr && nstate_out != nullptr && state_cap > 0 && trace_cap > 0), "bad state arguments (the iterate dump needs the trace)");
        check_common(x, y, n, p, mem, opts);
        ADMM_REQUIRE(beta_out && niter_out, "output pointers must not be NULL");
        ADMM_REQUIRE(n > p, "nrow(x) must be greater than ncol(x)");            // R/20_admm_lad.R:21-22
        ADMM_REQUIRE(opts->rho > 0, "rho should be positive");
        require_device();
        const double t0 = now_s();
        Stream st;
        DeviceData<double> d;
        upload_standardize<double>(d, x, y, n, p, mem, true, intercept != 0, st.s);    // LAD.cpp:34: standardize always TRUE
        DenseResult res;
        res.trace_cap = trace_cap;
        res.state_cap = state_cap;
        res.stats.t_h2d = d.t_h2d;
        res.stats.t_standardize = d.t_std;
        solve_lad(d, *opts, res, st.s);
        for (int i = 0; i <= p; ++i) beta_out[i] = res.beta[i];
        niter_out[0] = res.niter;
        if (trace_cap > 0) { std::memcpy(trace_out, res.trace.data(), res.trace.size() * sizeof(double)); *ntrace_out = (long long)(res.trace.size() / ADMM_TRACE_FIELDS); }
        dense_state_out(res, state_out, state_cap, nstate_out);
        res.stats.t_total = now_s() - t0;
        if (stats) *stats = res.stats;
    });
}

int admm_hip_bp(const double* x, const double* y, int n, int p, int mem,
                const admm_opts* opts, double* beta_out, int* niter_out, admm_stats* stats) {
    return admm_hip_bp_traced(x, y, n, p, mem, opts, beta_out, niter_out, stats, nullptr, 0, nullptr);
}

int admm_hip_bp_traced(const double* x, const double* y, int n, int p, int mem, const admm_opts* opts,
                       double* beta_out, int* niter_out, admm_stats* stats, double* trace_out, long long trace_cap, long long* ntrace_out) {
    return admm_hip_bp_state(x, y, n, p, mem, opts, beta_out, niter_out, stats, trace_out, trace_cap, ntrace_out, nullptr, 0, nullptr);
}

int admm_hip_bp_state(const double* x, const double* y, int n, int p, int mem, const admm_opts* opts,
                      double* beta_out, int* niter_out, admm_stats* stats, double* trace_out, long long trace_cap, long long* ntrace_out,
                      double* state_out, long long state_cap, long long* nstate_out) {
    return guarded([&] {
        ADMM_REQUIRE(trace_cap == 0 || (trace_out != nullptr && ntrace_out != nullptr && trace_cap > 0), "bad trace arguments");
        ADMM_REQUIRE(state_cap == 0 || (state_out != nullptr && nstate_out != nullptr && state_cap > 0 && trace_cap > 0), "bad state arguments (the iterate dump needs the trace)");
        check_common(x, y, n, p, mem, opts);
        ADMM_REQUIRE(beta_out && niter_out, "output pointers must not be NULL");
        ADMM_REQUIRE(p > n, "ncol(x) must be greater than nrow(x)");            // R/10_admm_bp.R:30-31
        ADMM_REQUIRE(opts->rho > 0, "rho should be positive");
        require_device();
        const double t0 = now_s();
        Stream st;
        DeviceData<double> d;
        upload_standardize<double>(d, x, y, n, p, mem, false, false, st.s);     // BP.cpp:24-27: no standardisation
        DenseResult res;
        res.trace_cap = trace_cap;
        res.state_cap = state_cap;
        res.stats.t_h2d = d.t_h2d;
        res.stats.t_standardize = d.t_std;
        solve_bp(d, *opts, res, st.s);
        for (int i = 0; i < p; ++i) beta_out[i] = res.beta[i];
        niter_out[0] = res.niter;
        if (trace_cap > 0) { std::memcpy(trace_out, res.trace.data(), res.trace.size() * sizeof(double)); *ntrace_out = (long long)(res.trace.size() / ADMM_TRACE_FIELDS); }
        dense_state_out(res, state_out, state_cap, nstate_out);
        res.stats.t_total = now_s() - t0;
        if (stats) *stats = res.stats;
    });
}

// admm_dantzig (R/50_admm_dantzig.R:30-46; TODO/Dantzig.cpp:32-99)
int admm_hip_dantzig_traced(const double* x, const double* y, int n, int p, int mem,
                            const double* lambda_in, int nlambda_in, int nlambda_auto, double lmin_ratio,
                            int standardize, int intercept, const admm_opts* opts,
                            double* lambda_out, double* beta_out, int* niter_out, admm_stats* stats,
                            double* trace_out, long long trace_cap, long long* ntrace_out) {
    return guarded([&] {
        ADMM_REQUIRE(trace_cap == 0 || (trace_out != nullptr && ntrace_out != nullptr && trace_cap > 0), "bad trace arguments");
        check_common(x, y, n, p, mem, opts);
        ADMM_REQUIRE(lambda_out && beta_out && niter_out, "output pointers must not be NULL");
        ADMM_REQUIRE(nlambda_in >= 0, "nlambda_in must be >= 0");
        ADMM_REQUIRE(nlambda_in > 0 ? lambda_in != nullptr : nlambda_auto > 0, "need a lambda grid or nlambda_auto > 0");
        if (nlambda_in == 0) ADMM_REQUIRE(lmin_ratio > 0 && lmin_ratio < 1, "lambda_min_ratio must be within (0, 1)");
        for (int i = 0; i < nlambda_in; ++i) ADMM_REQUIRE(lambda_in[i] > 0, "lambda must be positive");
        ADMM_REQUIRE(p >= 3, "the spectral-radius estimate needs at least 3 columns");
        require_device();
        const double t0 = now_s();
        Stream st;
        DeviceData<double> d;
        upload_standardize<double>(d, x, y, n, p, mem, standardize != 0, intercept != 0, st.s);     // Dantzig.cpp:52-55
        const LassoProblem pb = make_problem(lambda_in, nlambda_in, nlambda_auto, lmin_ratio, false, 1.0, 0, false, opts);
        DantzigResult res;
        res.trace_cap = trace_cap;
        res.stats.t_h2d = d.t_h2d;
        res.stats.t_standardize = d.t_std;
        solve_dantzig(d, pb, res, st.s);
        const int nl = (int)res.lambda.size();
        for (int i = 0; i < nl; ++i) { lambda_out[i] = res.lambda[i]; niter_out[i] = res.niter[i]; }
        std::memcpy(beta_out, res.beta.data(), sizeof(double) * (size_t)(p + 1) * nl);
        if (trace_cap > 0) { std::memcpy(trace_out, res.trace.data(), res.trace.size() * sizeof(double)); *ntrace_out = (long long)(res.trace.size() / ADMM_TRACE_FIELDS); }
        res.stats.t_total = now_s() - t0;
        if (stats) *stats = res.stats;
    });
}

int admm_hip_dantzig(const double* x, const double* y, int n, int p, int mem,
                     const double* lambda_in, int nlambda_in, int nlambda_auto, double lmin_ratio,
                     int standardize, int intercept, const admm_opts* opts,
                     double* lambda_out, double* beta_out, int* niter_out, admm_stats* stats) {
    return admm_hip_dantzig_traced(x, y, n, p, mem, lambda_in, nlambda_in, nlambda_auto, lmin_ratio, standardize, intercept, opts,
                                   lambda_out, beta_out, niter_out, stats, nullptr, 0, nullptr);
}

// admm_parbp (R/10_admm_bp.R:111-116; TODO/ParBP.cppp:26-71): opts->rho carries rho_ratio.
static void parbp_common(const double* x_cols, const double* y, int n, int p_local, long long p_total, long long col_offset, int mem, int nthread,
                         const admm_opts* opts, double* beta_out, int* niter_out, admm_stats* stats,
                         double* trace_out, long long trace_cap, long long* ntrace_out) {
    ADMM_REQUIRE(trace_cap == 0 || (trace_out != nullptr && ntrace_out != nullptr && trace_cap > 0), "bad trace arguments");
    check_common(x_cols, y, n, p_local, mem, opts);
    ADMM_REQUIRE(beta_out && niter_out, "output pointers must not be NULL");
    ADMM_REQUIRE(p_total > n, "ncol(x) must be greater than nrow(x)");            // R/10_admm_bp.R:30-31
    ADMM_REQUIRE(opts->rho > 0, "rho should be positive");
    ADMM_REQUIRE(nthread >= 1 && nthread <= p_total, "nthread must be within [1, ncol(x)]");
    require_device();
    const double t0 = now_s();
    Stream st;
    DeviceData<double> d;
    upload_standardize<double>(d, x_cols, y, n, p_local, mem, false, false, st.s);     // ParBP.cppp:36-37: no standardisation
    DenseResult res;
    res.trace_cap = trace_cap;
    res.stats.t_h2d = d.t_h2d;
    res.stats.t_standardize = d.t_std;
    solve_parbp(d, *opts, nthread, p_total, col_offset, res, st.s);
    for (int i = 0; i < p_local; ++i) beta_out[i] = res.beta[i];
    niter_out[0] = res.niter;
    if (trace_cap > 0) { std::memcpy(trace_out, res.trace.data(), res.trace.size() * sizeof(double)); *ntrace_out = (long long)(res.trace.size() / ADMM_TRACE_FIELDS); }
    res.stats.t_total = now_s() - t0;
    if (stats) *stats = res.stats;
}

int admm_hip_parbp(const double* x, const double* y, int n, int p, int mem, int nthread, const admm_opts* opts,
                   double* beta_out, int* niter_out, admm_stats* stats) {
    return guarded([&] { parbp_common(x, y, n, p, p, 0, mem, nthread, opts, beta_out, niter_out, stats, nullptr, 0, nullptr); });
}

int admm_hip_parbp_traced(const double* x, const double* y, int n, int p, int mem, int nthread, const admm_opts* opts,
                          double* beta_out, int* niter_out, admm_stats* stats, double* trace_out, long long trace_cap, long long* ntrace_out) {
    return guarded([&] { parbp_common(x, y, n, p, p, 0, mem, nthread, opts, beta_out, niter_out, stats, trace_out, trace_cap, ntrace_out); });
}

int admm_hip_parbp_dist(const double* x_cols, const double* y, int n, int p_local, long long p_total, long long col_offset, int mem, int nthread,
                        const admm_opts* opts, double* beta_local_out, int* niter_out, admm_stats* stats) {
    return guarded([&] {
        ADMM_REQUIRE(comm_info().active, "no communicator: call admm_hip_comm_init first");
        ADMM_REQUIRE(p_total >= p_local && col_offset >= 0 && col_offset + p_local <= p_total, "column block outside [0, p_total)");
        parbp_common(x_cols, y, n, p_local, p_total, col_offset, mem, nthread, opts, beta_local_out, niter_out, stats, nullptr, 0, nullptr);
    });
}

int admm_hip_lasso_plan_create(const double* x, const double* y, int n, int p, int mem,
                               const double* lambda_in, int nlambda_in, int nlambda_auto, double lmin_ratio,
                               int standardize, int intercept, double alpha, int nthread, const admm_opts* opts,
                               admm_hip_plan** plan_out, int* nlambda_out) {
    return guarded([&] {
        ADMM_REQUIRE(plan_out != nullptr, "plan_out must not be NULL");
        const bool enet = alpha >= 0.0;
        if (enet) ADMM_REQUIRE(alpha <= 1.0, "alpha must be within [0, 1]");
        // the reference has no parallel elastic net (R/40_admm_enet.R:50-64 always calls admm_enet): refuse instead of silently running a consensus Lasso
        ADMM_REQUIRE(!(enet && nthread > 1), "the consensus solver has no elastic-net variant: alpha >= 0 cannot be combined with nthread > 1");
        PlanHandle* h = create_plan(x, y, n, p, mem, lambda_in, nlambda_in, nlambda_auto, lmin_ratio, standardize, intercept,
                                    enet, enet ? alpha : 1.0, nthread > 1 ? nthread : 0, opts);
        *plan_out = reinterpret_cast<admm_hip_plan*>(h);
        if (nlambda_out) *nlambda_out = h->nlam;
    });
}

int admm_hip_lasso_plan_create_dist(const double* x_local, const double* y_local, int n_local, long long n_total, int p, int mem,
                                    const double* lambda_in, int nlambda_in, int nlambda_auto, double lmin_ratio,
                                    int standardize, int intercept, int nthread, const admm_opts* opts,
                                    admm_hip_plan** plan_out, int* nlambda_out) {
    return guarded([&] {
        ADMM_REQUIRE(plan_out != nullptr, "plan_out must not be NULL");
        ADMM_REQUIRE(n_total >= n_local && n_local > 0, "n_total must be >= n_local > 0");
        PlanHandle* h = create_plan(x_local, y_local, n_local, p, mem, lambda_in, nlambda_in, nlambda_auto, lmin_ratio,
                                    standardize, intercept, false, 1.0, nthread, opts, n_total);
        *plan_out = reinterpret_cast<admm_hip_plan*>(h);
        if (nlambda_out) *nlambda_out = h->nlam;
    });
}

int admm_hip_lasso_dist(const double* x_local, const double* y_local, int n_local, long long n_total, int p, int mem,
                        const double* lambda_in, int nlambda_in, int nlambda_auto, double lmin_ratio,
                        int standardize, int intercept, double alpha, const admm_opts* opts,
                        double* lambda_out, float* beta_out, int* niter_out, admm_stats* stats) {
    return guarded([&] {
        ADMM_REQUIRE(lambda_out && beta_out && niter_out, "output pointers must not be NULL");
        ADMM_REQUIRE(n_total >= n_local && n_local > 0, "n_total must be >= n_local > 0");
        const bool enet = alpha >= 0.0;
        if (enet) ADMM_REQUIRE(alpha <= 1.0, "alpha must be within [0, 1]");
        std::unique_ptr<PlanHandle> h(create_plan(x_local, y_local, n_local, p, mem, lambda_in, nlambda_in, nlambda_auto, lmin_ratio,
                                                  standardize, intercept, enet, enet ? alpha : 1.0, 0, opts, n_total));
        run_plan(h.get(), lambda_out, beta_out, niter_out, stats, h->t_create);
    });
}

int admm_hip_parlasso_dist(const double* x_local, const double* y_local, int n_local, long long n_total, int p, int mem,
                           const double* lambda_in, int nlambda_in, int nlambda_auto, double lmin_ratio,
                           int standardize, int intercept, int nthread, const admm_opts* opts,
                           double* lambda_out, float* beta_out, int* niter_out, admm_stats* stats) {
    return guarded([&] {
        ADMM_REQUIRE(lambda_out && beta_out && niter_out, "output pointers must not be NULL");
        ADMM_REQUIRE(n_total >= n_local && n_local > 0, "n_total must be >= n_local > 0");
        std::unique_ptr<PlanHandle> h(create_plan(x_local, y_local, n_local, p, mem, lambda_in, nlambda_in, nlambda_auto, lmin_ratio,
                                                  standardize, intercept, false, 1.0, nthread, opts, n_total));
        run_plan(h.get(), lambda_out, beta_out, niter_out, stats, h->t_create);
    });
}

int admm_hip_lasso_dist_cols(const double* x_cols, const double* y, int n, int p_local, long long p_total, long long col_offset, int mem,
                             const double* lambda_in, int nlambda_in, int nlambda_auto, double lmin_ratio,
                             int standardize, int intercept, double alpha, const admm_opts* opts,
                             double* lambda_out, float* beta_out, int* niter_out, admm_stats* stats) {
    return guarded([&] {
        ADMM_REQUIRE(lambda_out && beta_out && niter_out, "output pointers must not be NULL");
        const bool enet = alpha >= 0.0;
        if (enet) ADMM_REQUIRE(alpha <= 1.0, "alpha must be within [0, 1]");
        std::unique_ptr<PlanHandle> h(create_plan_cols(x_cols, y, n, p_local, p_total, col_offset, mem, lambda_in, nlambda_in, nlambda_auto,
                                                       lmin_ratio, standardize, intercept, enet, enet ? alpha : 1.0, opts));
        run_plan(h.get(), lambda_out, beta_out, niter_out, stats, h->t_create);
    });
}

int admm_hip_lasso_plan_create_dist_cols(const double* x_cols, const double* y, int n, int p_local, long long p_total, long long col_offset, int mem,
                                         const double* lambda_in, int nlambda_in, int nlambda_auto, double lmin_ratio,
                                         int standardize, int intercept, double alpha, const admm_opts* opts,
                                         admm_hip_plan** plan_out, int* nlambda_out) {
    return guarded([&] {
        ADMM_REQUIRE(plan_out != nullptr, "plan_out must not be NULL");
        const bool enet = alpha >= 0.0;
        if (enet) ADMM_REQUIRE(alpha <= 1.0, "alpha must be within [0, 1]");
        PlanHandle* h = create_plan_cols(x_cols, y, n, p_local, p_total, col_offset, mem, lambda_in, nlambda_in, nlambda_auto,
                                         lmin_ratio, standardize, intercept, enet, enet ? alpha : 1.0, opts);
        *plan_out = reinterpret_cast<admm_hip_plan*>(h);
        if (nlambda_out) *nlambda_out = h->nlam;
    });
}

int admm_hip_options_default(admm_hip_options* o) {
    return guarded([&] {
        ADMM_REQUIRE(o != nullptr, "options must not be NULL");
        std::memset(o, 0, sizeof(*o));
        o->struct_size = (int)sizeof(*o);
    });
}
int admm_hip_options_reset(void) { return guarded([&] { options_reset_thread(); }); }
int admm_hip_option_set(const char* name, const char* value) {
    return guarded([&] {
        ADMM_REQUIRE(name != nullptr && name[0] != 0, "option name must not be empty");
        std::string n(name);
        if (n.rfind("ADMM_HIP_", 0) == 0) n = n.substr(9);
        for (char& c : n) c = (char)std::toupper((unsigned char)c);
        option_set_thread(n.c_str(), value);
    });
}
const char* admm_hip_option_get(const char* name) {
    if (!name) return nullptr;
    std::string n(name);
    if (n.rfind("ADMM_HIP_", 0) == 0) n = n.substr(9);
    for (char& c : n) c = (char)std::toupper((unsigned char)c);
    return option(n.c_str());
}
int admm_hip_options_set(const admm_hip_options* o) {
    return guarded([&] {
        options_reset_thread();
        if (o == nullptr) return;
        ADMM_REQUIRE(o->struct_size >= (int)(2 * sizeof(int)) && o->struct_size <= (int)sizeof(admm_hip_options), "options: bad struct_size");
        admm_hip_options v;
        std::memset(&v, 0, sizeof(v));
        std::memcpy(&v, o, (size_t)o->struct_size);
        auto set = [](const char* k, const char* val) { option_set_thread(k, val); };
        auto num = [](const char* k, int val) { option_set_thread(k, std::to_string(val).c_str()); };
        if (v.gram_backend == 1) set("GRAM", "rocblas");
        if (v.gram_split) { ADMM_REQUIRE(v.gram_split >= 1 && v.gram_split <= 3, "options: gram_split"); set("GRAM_SPLIT", v.gram_split == 1 ? "0" : (v.gram_split == 2 ? "f16x2" : "bf16x3")); }
        if (v.factor_backend == 1) set("FACTOR", "rocsolver");
        if (v.inverse_precision) { ADMM_REQUIRE(v.inverse_precision == 1 || v.inverse_precision == 2, "options: inverse_precision"); set("INVERSE", v.inverse_precision == 1 ? "f32" : "f64"); }
        if (v.tall_xupdate) { ADMM_REQUIRE(v.tall_xupdate == 1 || v.tall_xupdate == 2, "options: tall_xupdate"); set("XUPDATE", v.tall_xupdate == 1 ? "gemv" : "sym"); }
        if (v.tall_refine) set("REFINE", "1");
        if (v.consensus_two_pass) set("PAR_ONEPASS", "0");
        if (v.consensus_unfused >= 1) set("PAR_FUSE_PZ", "0");
        if (v.consensus_unfused >= 2) set("PAR_BATCH", "0");
        if (v.bp_two_pass) set("BP_ONEPASS", "0");
        if (v.lad_no_hat) set("LAD_HAT", "0");
        if (v.wide_no_persist == 1) set("WIDE_PERSIST", "0");
        if (v.wide_no_persist == 2) set("WIDE_PERSIST_COLS", "0");
        if (v.wide_unfused) set("WIDE_FUSE", "0");
        if (v.wide_gram_sprad) set("WIDE_SPRAD", "gram");
        if (v.sharing_bp_direct) set("SBP_GRAM", "0");
        if (v.cv_downdate) set("CV_DOWNDATE", v.cv_downdate == 1 ? "1" : "0");
        if (v.peer_exchange) set("PEER_FUSED", v.peer_exchange == 1 ? "2" : "0");
        if (v.batch_iters > 0) num("BATCH_ITERS", v.batch_iters);
        if (v.profile_stride > 0) num("PROFILE_STRIDE", v.profile_stride);
        if (v.pool_mb) num("POOL_MB", v.pool_mb < 0 ? 0 : v.pool_mb);
        if (v.lad_two_pass) set("LAD_ONEPASS", "0");
        if (v.screen) {
            ADMM_REQUIRE(v.screen >= 1 && v.screen <= 3, "options: screen");
            set("WIDE_SCREEN", v.screen == 1 ? "16" : (v.screen == 3 ? "8" : "0"));
            set("SBP_SCREEN", v.screen == 2 ? "0" : "1");
        }
    });
}
int admm_hip_comm_unique_id(void* id_out) {
    return guarded([&] { ADMM_REQUIRE(id_out != nullptr, "id_out must not be NULL"); comm_unique_id(id_out); });
}
int admm_hip_comm_init(int nranks, int rank, const void* id) {
    return guarded([&] { ADMM_REQUIRE(id != nullptr, "id must not be NULL"); require_device(); comm_init(nranks, rank, id); });
}
int admm_hip_comm_finalize(void) {
    return guarded([&] { comm_finalize(); });
}
int admm_hip_comm_info(int* nranks_out, int* rank_out, int* backend_out) {
    return guarded([&] {
        const CommInfo ci = comm_info_live();
        if (nranks_out) *nranks_out = ci.active ? ci.nranks : 1;
        if (rank_out) *rank_out = ci.active ? ci.rank : 0;
        if (backend_out) *backend_out = ci.active ? ci.backend : 0;
    });
}
int admm_hip_comm_peer_prepare(int nranks, void* handle_out) {
    return guarded([&] { ADMM_REQUIRE(handle_out != nullptr, "handle_out must not be NULL"); require_device(); comm_peer_prepare(nranks, handle_out); });
}
int admm_hip_comm_init_peer(int nranks, int rank, const void* handles) {
    return guarded([&] { ADMM_REQUIRE(handles != nullptr, "handles must not be NULL"); require_device(); comm_init_peer(nranks, rank, handles); });
}
int admm_hip_comm_init_shm(int nranks, int rank, const char* name, unsigned long long token) {
    return guarded([&] { require_device(); comm_init_shm(nranks, rank, name, token); });
}
int admm_hip_comm_test_allreduce(float* fbuf, long long nf, double* dbuf, long long nd, int mem) {
    return guarded([&] {
        ADMM_REQUIRE(nf >= 0 && nd >= 0 && (nf == 0 || fbuf) && (nd == 0 || dbuf), "bad arguments");
        ADMM_REQUIRE(comm_info().active, "no communicator");
        require_device();
        Stream st;
        DevBuf<float> df; DevBuf<double> dd;
        float* pf = fbuf; double* pd = dbuf;
        if (mem == ADMM_MEM_HOST) {
            df.alloc((size_t)nf); dd.alloc((size_t)nd);
            if (nf) ADMM_HIP_CHECK(hipMemcpyAsync(df.get(), fbuf, (size_t)nf * sizeof(float), hipMemcpyHostToDevice, st.s));
            if (nd) ADMM_HIP_CHECK(hipMemcpyAsync(dd.get(), dbuf, (size_t)nd * sizeof(double), hipMemcpyHostToDevice, st.s));
            pf = df.get(); pd = dd.get();
        }
        if (nf && nd) allreduce_sum_f32_f64(pf, (size_t)nf, pd, (size_t)nd, st.s);
        else if (nf) allreduce_sum_f32(pf, (size_t)nf, st.s);
        else if (nd) allreduce_sum_f64(pd, (size_t)nd, st.s);
        if (mem == ADMM_MEM_HOST) {
            if (nf) ADMM_HIP_CHECK(hipMemcpyAsync(fbuf, pf, (size_t)nf * sizeof(float), hipMemcpyDeviceToHost, st.s));
            if (nd) ADMM_HIP_CHECK(hipMemcpyAsync(dbuf, pd, (size_t)nd * sizeof(double), hipMemcpyDeviceToHost, st.s));
        }
        st.sync();
        comm_check();
    });
}

int admm_hip_comm_test_reduce_scatter(const float* send, long long count, float* recv) {
    return guarded([&] {
        ADMM_REQUIRE(count > 0 && count % 4 == 0 && send && recv, "bad arguments (count must be a positive multiple of 4)");
        require_device();
        const CommInfo ci = comm_info();
        const size_t nr = (size_t)(ci.active ? ci.nranks : 1);
        Stream st;
        DevBuf<float> ds(nr * (size_t)count), dr((size_t)count);
        ADMM_HIP_CHECK(hipMemcpyAsync(ds.get(), send, nr * (size_t)count * sizeof(float), hipMemcpyHostToDevice, st.s));
        reduce_scatter_sum_f32(ds.get(), dr.get(), (size_t)count, st.s);
        ADMM_HIP_CHECK(hipMemcpyAsync(recv, dr.get(), (size_t)count * sizeof(float), hipMemcpyDeviceToHost, st.s));
        st.sync();
        comm_check();
    });
}

int admm_hip_lasso_plan_run(admm_hip_plan* plan, double* lambda_out, float* beta_out, int* niter_out, admm_stats* stats) {
    return guarded([&] { run_plan(reinterpret_cast<PlanHandle*>(plan), lambda_out, beta_out, niter_out, stats, 0.0); });
}

int admm_hip_lasso_plan_destroy(admm_hip_plan* plan) {
    return guarded([&] { delete reinterpret_cast<PlanHandle*>(plan); });
}

int admm_hip_lasso_plan_trace_enable(admm_hip_plan* plan, long long capacity_records) {
    return guarded([&] {
        PlanHandle* h = reinterpret_cast<PlanHandle*>(plan);
        ADMM_REQUIRE(h != nullptr && h->plan, "plan is NULL");
        ADMM_REQUIRE(capacity_records > 0 && capacity_records <= (1ll << 26), "trace capacity must be within [1, 2^26] records");
        h->plan->enable_trace(capacity_records);
    });
}

int admm_hip_lasso_plan_trace_read(admm_hip_plan* plan, double* out, long long cap_records, long long* nrecords_out) {
    return guarded([&] {
        PlanHandle* h = reinterpret_cast<PlanHandle*>(plan);
        ADMM_REQUIRE(h != nullptr && h->plan, "plan is NULL");
        ADMM_REQUIRE(out != nullptr && nrecords_out != nullptr && cap_records >= 0, "bad trace output arguments");
        *nrecords_out = h->plan->read_trace(out, cap_records);
    });
}

int admm_hip_lasso_plan_state_enable(admm_hip_plan* plan, long long capacity_records) {
    return guarded([&] {
        PlanHandle* h = reinterpret_cast<PlanHandle*>(plan);
        ADMM_REQUIRE(h != nullptr && h->plan, "plan is NULL");
        ADMM_REQUIRE(capacity_records > 0 && capacity_records <= (1ll << 22), "state capacity must be within [1, 2^22] records");
        h->plan->enable_state(capacity_records);
    });
}

int admm_hip_lasso_plan_state_read(admm_hip_plan* plan, float* out, long long cap_records, long long* nrecords_out, long long* record_floats_out) {
    return guarded([&] {
        PlanHandle* h = reinterpret_cast<PlanHandle*>(plan);
        ADMM_REQUIRE(h != nullptr && h->plan, "plan is NULL");
        ADMM_REQUIRE(nrecords_out != nullptr && cap_records >= 0 && (out != nullptr || cap_records == 0), "bad state output arguments");
        *nrecords_out = h->plan->read_state(out, cap_records, record_floats_out);
    });
}

int admm_hip_lasso_plan_data_read(admm_hip_plan* plan, float* x_out, long long ld, float* y_out) {
    return guarded([&] {
        PlanHandle* h = reinterpret_cast<PlanHandle*>(plan);
        ADMM_REQUIRE(h != nullptr && h->plan, "plan is NULL");
        h->plan->read_data(x_out, ld, y_out);
    });
}

int admm_hip_lasso_plan_system_read(admm_hip_plan* plan, float* out, long long ld) {
    return guarded([&] {
        PlanHandle* h = reinterpret_cast<PlanHandle*>(plan);
        ADMM_REQUIRE(h != nullptr && h->plan, "plan is NULL");
        h->plan->read_system(out, ld);
    });
}

const char* admm_hip_last_error(void) { return last_error_ref().c_str(); }
const char* admm_hip_version(void) { return "admm_hip 0.3 (gfx950)"; }
int admm_hip_trim_memory(void) {
    admm::pool_trim();
    return ADMM_OK;
}

int admm_hip_device_count(void) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) return 0;
    return cnt;
}
int admm_hip_set_device(int device) {
    return guarded([&] { ADMM_HIP_CHECK(hipSetDevice(device)); });
}
int admm_hip_device_synchronize(void) {
    return guarded([&] { ADMM_HIP_CHECK(hipDeviceSynchronize()); });
}

// Host-only helper used by the CPU test-suite: runs the Lanczos host logic (lanczos.hip) against a
// dense symmetric matrix held in host memory.  Not part of any solver path.
int admm_hip_host_lanczos(const float* A, int n, float* eig_out, int* nmatop_out) {
    return guarded([&] {
        ADMM_REQUIRE(A && eig_out && n >= 3, "bad arguments");
        auto op = [&](const float* v, float* w) {
            for (int i = 0; i < n; ++i) w[i] = 0.f;
            for (int j = 0; j < n; ++j) {
                const float vj = v[j];
                const float* col = A + (size_t)j * n;
                for (int i = 0; i < n; ++i) w[i] += col[i] * vj;
            }
        };
        *eig_out = lanczos_largest_f32(op, n, nmatop_out);
    });
}

int admm_hip_test_symv(const float* A, int p, const float* v0, const float* v1, float* y0, float* y1) {
    return guarded([&] {
        ADMM_REQUIRE(A && v0 && v1 && y0 && y1 && p > 0, "bad arguments");
        test_symv(A, p, v0, v1, y0, y1);
    });
}

int admm_hip_test_gram(const void* A, int rows, int cols, int atA, int is_double, void* G) {
    return guarded([&] {
        ADMM_REQUIRE(A && G && rows > 0 && cols > 0, "bad arguments");
        if (is_double) test_gram<double>(static_cast<const double*>(A), rows, cols, atA != 0, static_cast<double*>(G));
        else test_gram<float>(static_cast<const float*>(A), rows, cols, atA != 0, static_cast<float*>(G));
    });
}

int admm_hip_test_gemv_t(const void* A, int rows, int cols, int is_double, const void* v, void* y) {
    return guarded([&] {
        ADMM_REQUIRE(A && v && y && rows > 0 && cols > 0, "bad arguments");
        if (is_double) test_gemv_t<double>(static_cast<const double*>(A), rows, cols, static_cast<const double*>(v), static_cast<double*>(y));
        else test_gemv_t<float>(static_cast<const float*>(A), rows, cols, static_cast<const float*>(v), static_cast<float*>(y));
    });
}

int admm_hip_test_gather(const void* A, int rows, int cols, int is_double, const void* v, double* y) {
    return guarded([&] {
        ADMM_REQUIRE(A && v && y && rows > 0 && cols > 0, "bad arguments");
        if (is_double) test_gather<double>(static_cast<const double*>(A), rows, cols, static_cast<const double*>(v), y);
        else test_gather<float>(static_cast<const float*>(A), rows, cols, static_cast<const float*>(v), y);
    });
}

int admm_hip_test_spd_inverse(const void* A, int n, int precision, void* Ainv) {
    return guarded([&] {
        ADMM_REQUIRE(A && Ainv && n > 0 && precision >= 0 && precision <= 2, "bad arguments");
        if (precision == 1) test_spd_inverse<double>(static_cast<const double*>(A), n, static_cast<double*>(Ainv), false);
        else test_spd_inverse<float>(static_cast<const float*>(A), n, static_cast<float*>(Ainv), precision == 2);
    });
}

int admm_hip_test_cv_fold_system(const double* x, const double* y, int n, int p, const int* fold_id, int nfolds, int fold,
                                 int standardize, int intercept, float* gram, float* xy, float* mean_x, float* scale_x, float* mean_scale_y) {
    return guarded([&] {
        ADMM_REQUIRE(x && y && gram && xy && mean_x && scale_x && mean_scale_y && n > 0 && p > 0, "bad arguments");
        ADMM_REQUIRE(nfolds >= 2 && fold >= 0 && fold < nfolds, "fold must be within [0, nfolds)");
        require_device();
        Stream st;
        DevBuf<double> xd((size_t)n * p), yd(n);
        write_device(xd.get(), x, (size_t)n * p * sizeof(double));
        write_device(yd.get(), y, (size_t)n * sizeof(double));
        std::vector<int> tr, te;
        for (int i = 0; i < n; ++i) ((fold_id ? fold_id[i] : i % nfolds) == fold ? te : tr).push_back(i);
        const int ntr = (int)tr.size(), nte = (int)te.size();
        ADMM_REQUIRE(ntr > p && nte > 0, "the fold needs held-out rows and more training rows than columns");
        std::vector<int> both(tr);
        both.insert(both.end(), te.begin(), te.end());
        DevBuf<int> didx(n);
        ADMM_HIP_CHECK(hipMemcpyAsync(didx.get(), both.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, st.s));
        CvBase base;
        cv_downdate_prepare(base, xd.get(), yd.get(), n, p, standardize != 0, intercept != 0, st.s);
        DeviceData<float> d;
        cv_downdate_fold(d, base, yd.get(), didx.get(), ntr, didx.get() + ntr, nte, st.s);
        ADMM_HIP_CHECK(hipMemcpy2D(gram, (size_t)p * sizeof(float), d.gram.get(), (size_t)d.ldgram * sizeof(float), (size_t)p * sizeof(float), p, hipMemcpyDeviceToHost));
        ADMM_HIP_CHECK(hipMemcpy(xy, d.xy.get(), (size_t)p * sizeof(float), hipMemcpyDeviceToHost));
        for (int j = 0; j < p; ++j) { mean_x[j] = d.meanX[j]; scale_x[j] = d.scaleX[j]; }
        mean_scale_y[0] = d.meanY; mean_scale_y[1] = d.scaleY;
    });
}

}  // extern "C"
