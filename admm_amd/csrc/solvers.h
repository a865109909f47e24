// Solver entry points behind the C ABI (one per reference problem class).
#pragma once
#include "prep.h"
#include <memory>

namespace admm {

struct LassoProblem {
    admm_opts opts{};
    std::vector<double> lambda_in;   // user grid (may be empty -> automatic)
    int nlambda_auto = 100;
    double lmin_ratio = 1e-4;
    bool enet = false;
    double alpha = 1.0;
    int nworkers = 0;                // > 0: row-block consensus (admm_parlasso)
    bool dist = false;               // consensus blocks / rows of the tall solver spread over the ranks of the attached communicator
    long long p_total = 0;           // > 0: COLUMN-sharded wide solver -- this rank holds columns [col_offset, col_offset + p) of p_total
    long long col_offset = 0;
    int batch_iters = 0;             // iterations enqueued per host poll (0 = default)
    int profile_stride = 0;          // > 0: time every stride-th x-update launch with HIP events
};

struct LassoResult {
    std::vector<double> lambda;
    std::vector<float> beta;         // (p+1) x nlambda column-major, row 0 = intercept
    std::vector<int> niter;
    admm_stats stats{};
    // optional: the caller's coefficient buffer ((p+1) x nlambda floats).  A solver that fills it directly sets beta_written and leaves
    // `beta` empty (the wide solver: its coefficient matrix is 80 MB of mostly zeros at BASELINE configs[2])
    float* beta_dst = nullptr;
    bool beta_written = false;
};

// Lasso.cpp:78-89
std::vector<double> make_lambda_grid(const LassoProblem& pb, double lambda0, int n, double scaleY);

// A prepared problem: construction does the one-time work (X'y, Gram, rho, factorisation ...),
// run() executes one cold-started warm-chained lambda path and may be called repeatedly.
struct LassoPlan {
    virtual ~LassoPlan() = default;
    virtual void run(LassoResult& res) = 0;
    // per-decision trace of the iteration control (admm_hip_lasso_plan_trace_*); solvers without one refuse
    virtual void enable_trace(long long) { throw Error(ADMM_ERR_INVALID_ARG, "this solver records no decision trace"); }
    virtual long long read_trace(double*, long long) { return 0; }
    // per-iteration iterate dump (admm_hip_lasso_plan_state_*): tall and consensus solvers
    virtual void enable_state(long long) { throw Error(ADMM_ERR_INVALID_ARG, "this solver records no iterate dump"); }
    virtual long long read_state(float*, long long, long long*) { return 0; }
    // the float system matrix X'X + rho I the x-update solves, p x p column-major (admm_hip_lasso_plan_system_read): only
    // the tall solver with ADMM_HIP_REFINE=1 keeps it
    // the standardised data (X n x p column-major with leading dimension ld, Y) as the solver holds them (admm_hip_lasso_plan_data_read): wide solver
    virtual void read_data(float*, long long, float*) { throw Error(ADMM_ERR_INVALID_ARG, "this plan does not keep its standardised data (wide solver only)"); }
    virtual void read_system(float*, long long) { throw Error(ADMM_ERR_INVALID_ARG, "this plan does not keep its system matrix (tall solver with ADMM_HIP_REFINE=1 only)"); }
};
std::unique_ptr<LassoPlan> make_tall_plan(DeviceData<float>&& d, const LassoProblem& pb, hipStream_t st);
std::unique_ptr<LassoPlan> make_wide_plan(DeviceData<float>&& d, const LassoProblem& pb, hipStream_t st);
std::unique_ptr<LassoPlan> make_par_plan(DeviceData<float>&& d, const LassoProblem& pb, hipStream_t st);

struct DenseResult {
    std::vector<double> beta;
    int niter = 0;
    admm_stats stats{};
    long long trace_cap = 0;         // > 0: record up to this many decisions (admm_hip_lad_traced / admm_hip_bp_traced)
    std::vector<double> trace;       // [nrec][ADMM_TRACE_FIELDS]
    long long state_cap = 0;         // > 0: also dump the iterates of up to this many decisions (admm_hip_lad_state / admm_hip_bp_state)
    std::vector<double> state;       // [nrec][5][state_dim]  x | z | y | adj_z | adj_y
    long long state_dim = 0;
};
void solve_lad(const DeviceData<double>& d, const admm_opts& opts, DenseResult& res, hipStream_t st);
void solve_bp(const DeviceData<double>& d, const admm_opts& opts, DenseResult& res, hipStream_t st);
// admm_dantzig (dantzig.hip): the Dantzig selector path in double.  res.beta: (p + 1) x nlambda column-major, row 0 = intercept.
struct DantzigResult {
    std::vector<double> lambda, beta;
    std::vector<int> niter;
    admm_stats stats{};
    long long trace_cap = 0;
    std::vector<double> trace;
};
void solve_dantzig(DeviceData<double>& d, const LassoProblem& pb, DantzigResult& res, hipStream_t st);
// admm_parbp: basis pursuit with the columns in `nblocks` blocks (sharing ADMM, sharing_bp.hip).  d holds this rank's columns
// [col_offset, col_offset + d.p) of p_total (whole blocks); opts.rho carries rho_ratio.  res.beta: this rank's coefficients.
void solve_parbp(const DeviceData<double>& d, const admm_opts& opts, int nblocks, long long p_total, long long col_offset,
                 DenseResult& res, hipStream_t st);

}  // namespace admm
