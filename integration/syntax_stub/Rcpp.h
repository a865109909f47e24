// DECLARATIONS ONLY -- not Rcpp and not usable as Rcpp.  R is not available in the build image of this repository, so
// integration/admm_shim.cpp cannot be compiled here; this header declares just the names the shim uses, with the signatures
// Rcpp documents for them, so that `g++ -fsyntax-only` can type-check the shim's calls into include/admm_hip.h (argument
// counts, order and types of every admm_hip_* entry point) and its own control flow (tests/test_shim_syntax.py).  Nothing
// here has a definition; nothing links against it.  What it cannot check: Rcpp's own semantics, R's headers, the link.
#pragma once
#include <cstddef>
#include <string>
#include <vector>

struct SEXPREC;
typedef SEXPREC* SEXP;
extern SEXP R_NilValue;

#define RcppExport extern "C"
#define BEGIN_RCPP try {
#define END_RCPP } catch (...) { } return (SEXP)0;

namespace Rcpp {

template <typename T> T as(SEXP);

struct NamedValue;
struct Named {
    explicit Named(const char*);
    template <typename T> NamedValue operator=(const T&) const;
};
struct NamedValue { operator SEXP() const; };

template <typename T>
struct VectorOf {
    VectorOf();
    VectorOf(SEXP);
    explicit VectorOf(int);
    explicit VectorOf(std::size_t);
    template <typename It> VectorOf(It, It);
    T* begin();
    const T* begin() const;
    T* end();
    int size() const;
    T& operator[](int);
    operator SEXP() const;
    template <typename... A> static VectorOf create(const A&...);
};
typedef VectorOf<double> NumericVector;
typedef VectorOf<int> IntegerVector;

struct NumericMatrix {
    NumericMatrix(SEXP);
    NumericMatrix(int, int);
    int nrow() const;
    int ncol() const;
    double* begin();
    operator SEXP() const;
};

struct ListProxy {
    operator SEXP() const;
    template <typename T> ListProxy& operator=(const T&);
};
struct List {
    List();
    List(SEXP);
    ListProxy operator[](const char*);
    ListProxy operator[](const std::string&);
    operator SEXP() const;
    template <typename... A> static List create(const A&...);
};

struct SlotProxy { template <typename T> SlotProxy& operator=(const T&); };
struct S4 {
    explicit S4(const char*);
    SlotProxy slot(const char*);
    operator SEXP() const;
};

template <typename... A> [[noreturn]] void stop(const char*, A...);
template <typename T> SEXP wrap(const T&);

}  // namespace Rcpp
