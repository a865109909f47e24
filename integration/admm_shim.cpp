// Rcpp shim that a maintainer of yixuan/ADMM adds to src/ in place of Lasso.cpp, Enet.cpp, ParLasso.cpp,
// LAD.cpp and BP.cpp.  It keeps the five `.Call` symbols the R code looks up by name (and supplies admm_parbp / admm_dantzig)
// (R/30_admm_lasso.R:140,149; R/40_admm_enet.R:53; R/20_admm_lad.R:60; R/10_admm_bp.R:104,111; R/50_admm_dantzig.R:38) and forwards the
// unpacked arguments to libadmm_hip.so (include/admm_hip.h).  R/ stays untouched.
//
// src/Makevars:   PKG_CPPFLAGS = -I/path/to/admm-mi355x/include
//                 PKG_LIBS     = -L/path/to/admm-mi355x/admm_amd/lib -ladmm_hip -Wl,-rpath,/path/to/admm-mi355x/admm_amd/lib
// (R is not available in the build image of this repository, so this file is compile-checked only where R exists.)
#include <Rcpp.h>
#include <vector>
#include "admm_hip.h"

using Rcpp::as;
using Rcpp::IntegerVector;
using Rcpp::List;
using Rcpp::Named;
using Rcpp::NumericMatrix;
using Rcpp::NumericVector;

static admm_opts unpack_opts(SEXP opts_) {
    List opts(opts_);
    admm_opts o;
    o.maxit = as<int>(opts["maxit"]);
    o.eps_abs = as<double>(opts["eps_abs"]);
    o.eps_rel = as<double>(opts["eps_rel"]);
    o.rho = as<double>(opts["rho"]);
    return o;
}

static void check(int rc) {
    if (rc != ADMM_OK) Rcpp::stop("libadmm_hip: %s", admm_hip_last_error());
}

// (p+1) x nlambda dense -> dgCMatrix with row 0 always stored (Lasso.cpp:22-30,131)
template <typename T>
static Rcpp::S4 to_dgCMatrix(const std::vector<T>& beta, int nrow, int ncol) {
    std::vector<int> ip(1, 0), ii;
    std::vector<double> xx;
    for (int j = 0; j < ncol; ++j) {
        for (int i = 0; i < nrow; ++i) {
            const T v = beta[(size_t)j * nrow + i];
            if (i == 0 || v != T(0)) { ii.push_back(i); xx.push_back((double)v); }
        }
        ip.push_back((int)ii.size());
    }
    Rcpp::S4 m("dgCMatrix");
    m.slot("i") = IntegerVector(ii.begin(), ii.end());
    m.slot("p") = IntegerVector(ip.begin(), ip.end());
    m.slot("x") = NumericVector(xx.begin(), xx.end());
    m.slot("Dim") = IntegerVector::create(nrow, ncol);
    return m;
}

static SEXP lasso_family(int which, SEXP x_, SEXP y_, SEXP lambda_, SEXP nlambda_, SEXP lmin_ratio_,
                         SEXP standardize_, SEXP intercept_, double alpha, int nthread, SEXP opts_) {
    NumericMatrix x(x_);
    NumericVector y(y_), lambda(lambda_);
    const int n = x.nrow(), p = x.ncol();
    const int nl_in = lambda.size();
    const int nl = nl_in > 0 ? nl_in : as<int>(nlambda_);
    admm_opts o = unpack_opts(opts_);
    NumericVector lambda_out(nl);
    IntegerVector niter(nl);
    std::vector<float> beta((size_t)(p + 1) * nl);
    const double* lam = nl_in > 0 ? lambda.begin() : nullptr;
    int rc;
    if (which == 0)
        rc = admm_hip_lasso(x.begin(), y.begin(), n, p, ADMM_MEM_HOST, lam, nl_in, as<int>(nlambda_), as<double>(lmin_ratio_),
                            as<bool>(standardize_), as<bool>(intercept_), &o, lambda_out.begin(), beta.data(), niter.begin(), nullptr);
    else if (which == 1)
        rc = admm_hip_enet(x.begin(), y.begin(), n, p, ADMM_MEM_HOST, lam, nl_in, as<int>(nlambda_), as<double>(lmin_ratio_),
                           as<bool>(standardize_), as<bool>(intercept_), alpha, &o, lambda_out.begin(), beta.data(), niter.begin(), nullptr);
    else
        rc = admm_hip_parlasso(x.begin(), y.begin(), n, p, ADMM_MEM_HOST, lam, nl_in, as<int>(nlambda_), as<double>(lmin_ratio_),
                               as<bool>(standardize_), as<bool>(intercept_), nthread, &o, lambda_out.begin(), beta.data(), niter.begin(), nullptr);
    check(rc);
    return List::create(Named("lambda") = lambda_out, Named("beta") = to_dgCMatrix(beta, p + 1, nl), Named("niter") = niter);
}

RcppExport SEXP admm_lasso(SEXP x_, SEXP y_, SEXP lambda_, SEXP nlambda_, SEXP lmin_ratio_,
                           SEXP standardize_, SEXP intercept_, SEXP opts_) {
BEGIN_RCPP
    return lasso_family(0, x_, y_, lambda_, nlambda_, lmin_ratio_, standardize_, intercept_, 1.0, 0, opts_);
END_RCPP
}

RcppExport SEXP admm_enet(SEXP x_, SEXP y_, SEXP lambda_, SEXP nlambda_, SEXP lmin_ratio_,
                          SEXP standardize_, SEXP intercept_, SEXP alpha_, SEXP opts_) {
BEGIN_RCPP
    return lasso_family(1, x_, y_, lambda_, nlambda_, lmin_ratio_, standardize_, intercept_, as<double>(alpha_), 0, opts_);
END_RCPP
}

RcppExport SEXP admm_parlasso(SEXP x_, SEXP y_, SEXP lambda_, SEXP nlambda_, SEXP lmin_ratio_,
                              SEXP standardize_, SEXP intercept_, SEXP nthread_, SEXP opts_) {
BEGIN_RCPP
    return lasso_family(2, x_, y_, lambda_, nlambda_, lmin_ratio_, standardize_, intercept_, 1.0, as<int>(nthread_), opts_);
END_RCPP
}

RcppExport SEXP admm_lad(SEXP x_, SEXP y_, SEXP intercept_, SEXP opts_) {
BEGIN_RCPP
    NumericMatrix x(x_);
    NumericVector y(y_);
    admm_opts o = unpack_opts(opts_);
    NumericVector beta(x.ncol() + 1);
    int niter = 0;
    check(admm_hip_lad(x.begin(), y.begin(), x.nrow(), x.ncol(), ADMM_MEM_HOST, as<bool>(intercept_), &o, beta.begin(), &niter, nullptr));
    return List::create(Named("beta") = beta, Named("niter") = niter);
END_RCPP
}

// The symbol R/50_admm_dantzig.R:38 asks for and the reference never builds (src/TODO/Dantzig.cpp:32-99): doubles throughout.
RcppExport SEXP admm_dantzig(SEXP x_, SEXP y_, SEXP lambda_, SEXP nlambda_, SEXP lmin_ratio_, SEXP standardize_, SEXP intercept_, SEXP opts_) {
BEGIN_RCPP
    NumericMatrix x(x_);
    NumericVector y(y_), lambda(lambda_);
    const int n = x.nrow(), p = x.ncol();
    const int nl_in = lambda.size();
    const int nl = nl_in > 0 ? nl_in : as<int>(nlambda_);
    admm_opts o = unpack_opts(opts_);
    NumericVector lambda_out(nl);
    IntegerVector niter(nl);
    // `beta` must be a dgCMatrix: R builds the result with do.call(ADMM_Dantzig_fit, res), and ADMM_Dantzig_fit contains
    // ADMM_Lasso_fit, whose field is beta = "dgCMatrix" (R/30_admm_lasso.R:18-21) -- a plain matrix would fail at object
    // construction.  TODO/Dantzig.cpp builds `SpMat beta(p + 1, nlambda)` through write_beta_matrix, row 0 always stored.
    std::vector<double> beta((size_t)(p + 1) * nl);
    check(admm_hip_dantzig(x.begin(), y.begin(), n, p, ADMM_MEM_HOST, nl_in > 0 ? lambda.begin() : nullptr, nl_in, as<int>(nlambda_),
                           as<double>(lmin_ratio_), as<bool>(standardize_), as<bool>(intercept_), &o, lambda_out.begin(), beta.data(),
                           niter.begin(), nullptr));
    return List::create(Named("lambda") = lambda_out, Named("beta") = to_dgCMatrix(beta, p + 1, nl), Named("niter") = niter);
END_RCPP
}

// Shared by admm_bp and admm_parbp: the p x 1 dgCMatrix of the non-zeros (BP.cpp:38-43, ParBP.cppp:58-61)
static Rcpp::S4 sparse_column(const std::vector<double>& b) {
    const int p = (int)b.size();
    std::vector<int> ii; std::vector<double> xx;
    for (int i = 0; i < p; ++i) if (b[i] != 0.0) { ii.push_back(i); xx.push_back(b[i]); }
    Rcpp::S4 m("dgCMatrix");
    m.slot("i") = IntegerVector(ii.begin(), ii.end());
    m.slot("p") = IntegerVector::create(0, (int)ii.size());
    m.slot("x") = NumericVector(xx.begin(), xx.end());
    m.slot("Dim") = IntegerVector::create(p, 1);
    return m;
}

// The symbol R/10_admm_bp.R:111 asks for and the reference never builds (src/TODO/ParBP.cppp:26-71): opts carries rho_ratio.
RcppExport SEXP admm_parbp(SEXP x_, SEXP y_, SEXP nthread_, SEXP opts_) {
BEGIN_RCPP
    NumericMatrix x(x_);
    NumericVector y(y_);
    List opts(opts_);
    admm_opts o;
    o.maxit = as<int>(opts["maxit"]); o.eps_abs = as<double>(opts["eps_abs"]); o.eps_rel = as<double>(opts["eps_rel"]);
    o.rho = as<double>(opts["rho_ratio"]);
    const int p = x.ncol();
    std::vector<double> b(p);
    int niter = 0;
    check(admm_hip_parbp(x.begin(), y.begin(), x.nrow(), p, ADMM_MEM_HOST, as<int>(nthread_), &o, b.data(), &niter, nullptr));
    return List::create(Named("beta") = sparse_column(b), Named("niter") = niter);
END_RCPP
}

RcppExport SEXP admm_bp(SEXP x_, SEXP y_, SEXP opts_) {
BEGIN_RCPP
    NumericMatrix x(x_);
    NumericVector y(y_);
    admm_opts o = unpack_opts(opts_);
    const int p = x.ncol();
    std::vector<double> b(p);
    int niter = 0;
    check(admm_hip_bp(x.begin(), y.begin(), x.nrow(), p, ADMM_MEM_HOST, &o, b.data(), &niter, nullptr));
    return List::create(Named("beta") = sparse_column(b), Named("niter") = niter);
END_RCPP
}

// Not in the reference (it keeps nothing between calls): libadmm_hip keeps EMPTY device blocks >= 32 MB for its next call
// (include/admm_hip.h, admm_hip_trim_memory).  The package would call this from .onUnload(libpath) -- and may offer it to users
// who share the GPU with other libraries:  .Call("admm_trim_memory", PACKAGE = "ADMM").
RcppExport SEXP admm_trim_memory() {
BEGIN_RCPP
    check(admm_hip_trim_memory());
    return R_NilValue;
END_RCPP
}
