// One-time device-side preparation shared by the solvers: upload/convert/standardise
// (the DataStd step of the reference), Gram matrices, Cholesky + cached inverse, Lanczos.
#pragma once
#include "admm_internal.h"
#include "gemv_plan.h"
#include <functional>

namespace admm {

// Device copy of the (standardised) problem data in solver precision T.
template <typename T>
struct DeviceData {
    int n = 0, p = 0;
    long long n_total = 0;      // rows of the global problem (== n unless the rows are spread over ranks)
    long long ldx = 0;          // leading dimension of X (>= n, multiple of 32 elements)
    DevBuf<T> X;                // n x p column-major, padding rows zero
    DevBuf<T> Y;                // n (allocated ldx, zero padded)
    int flag = 0;               // DataStd flag = standardize + 2*intercept (DataStd.h:21-29)
    std::vector<T> meanX, scaleX;   // host copies for recover()
    T meanY = 0, scaleY = 1;
    double t_h2d = 0, t_std = 0;
    // Filled by upload_standardize_gram_f32 only: X'X of the standardised data (both triangles), leading dimension and
    // allocated columns round_up(p, 128), zero padded; the seconds of Gram work left after the last byte arrived.
    DevBuf<T> gram;
    long long ldgram = 0;
    double t_gram_tail = 0;
    // Gram-form data (cross-validation folds formed as down-dates, cv.hip): X'y of the standardised data, round_up(p, 128)
    // entries, zero padded.  When set the tall solver takes X'y and the Gram from here and X / Y may be empty.
    DevBuf<T> xy;
};

// Convert the caller's double column-major x (n x p, ld n) and y to T on the device and apply
// DataStd::standardize (DataStd.h:89-155) there.  mem: ADMM_MEM_HOST / ADMM_MEM_DEVICE.
template <typename T>
void upload_standardize(DeviceData<T>& d, const double* x, const double* y, int n, int p, int mem,
                        bool standardize, bool intercept, hipStream_t st, long long n_total = 0);

// Host-input variant for the tall fp32 path: the same conversion and standardisation, column chunk by column chunk,
// with the Gram block row of every chunk (it only needs the chunks that have already arrived) enqueued behind it, so
// that standardisation and X'X run under the PCIe transfer instead of after it.  Results are bit-identical to
// upload_standardize + the one-shot matrix-core Gram.
void upload_standardize_gram_f32(DeviceData<float>& d, const double* x, const double* y, int n, int p,
                                 bool standardize, bool intercept, hipStream_t st);
// Another response of the same x (admm_hip_lasso_multi): X / statistics / Gram copied from `base`, y standardised alone.
void clone_with_response_f32(DeviceData<float>& d, const DeviceData<float>& base, const float* gram, long long ldgram,
                             const double* y_dev, hipStream_t st);

// The tall Gram on the 16-bit matrix cores (gram_bf16x3.hip): Z = X' as two fp16 planes (x = 2^e (h + l), three kept cross products;
// default) or three bf16 planes (x = h + m + l, six kept cross products).
struct GramSplit3 {
    int npl = 2;                        // planes: 2 = fp16 split, 3 = bf16 split
    DevBuf<float> rs;                   // fp16 split: [2][ldz] 2^e_i and 2^-e_i per output index
    DevBuf<unsigned short> planes;      // [npl][nkg][ldz][8] 16-bit values
    long long ldz = 0;                  // entries (of 8 bf16) per k group: order rounded up to 256
    int nkg = 0, M = 0;                 // k groups of 8 (K rounded up to 16); order
    void alloc(int order, int kdepth, hipStream_t st);
    void split_cols(const float* X, long long ldx, int rows, int c0, int nc, hipStream_t st);      // columns [c0, c0 + nc) of X (rows x nc at X)
    void gram_lower(float* C, long long ldc, const int* tilemap, int ntiles, hipStream_t st) const;
    // 256 x 256 macro-tiles (npl == 2), bit-identical; tail / ntail: 128 x 128 tiles (bi << 16 | bj) dispatched after the macro-tiles of every launch (the round that would straggle)
    void gram_lower256(float* C, long long ldc, const int* tilemap, int ntiles, hipStream_t st, const int* tail = nullptr, int ntail = 0) const;
    void gram_rows(int r0, int nr, float* C, long long ldc, hipStream_t st) const;
};
int gram_split_mode();                  // 2 (default) / 3 / 0 = the exact-fp32 matrix-core kernel (syrk_mfma.hip) as before: ADMM_HIP_GRAM_SPLIT=f16x2 | bf16x3 | 0

// Cross-validation folds formed as down-dates of the full-data Gram (cv.hip): what is formed once per call.
struct CvBase {
    DeviceData<float> full;         // the full data standardised with ITS statistics: Z (n x p), its response, m, s
    DevBuf<float> Gall;             // Z'Z, ld = ldp = round_up(p, 128), zero padded
    long long ldp = 0;
    std::vector<double> s1, s2;     // column sums of Z and of Z.^2
    double t_prepare = 0;
};
void cv_downdate_prepare(CvBase& b, const double* xd, const double* yd, int n, int p, bool standardize, bool intercept, hipStream_t st);
void cv_downdate_full(DeviceData<float>& d, const CvBase& b, hipStream_t st);
void cv_downdate_fold(DeviceData<float>& d, const CvBase& b, const double* yd, const int* d_train, int ntr, const int* d_test, int nte, hipStream_t st);

// y alone (device doubles, n entries): narrowed to float into Yout (ld entries, zero padded) and standardised by the kernels of
// upload_standardize (flag as there); the statistics come back in *meanY / *scaleY.
void standardize_response_f32(const double* y_dev, int n, int flag, long long n_total, float* Yout, long long ld,
                              float* meanY, float* scaleY, hipStream_t st);

// DataStd::recover (DataStd.h:157-207) on a host coefficient vector (length p) in precision T.
template <typename T>
void recover_coef(const DeviceData<T>& d, const T* coef, T* beta0, T* out);
// the same for a column given as its non-zeros (ascending indices): writes only those entries of `out` (the caller cleared it).  Same
// arithmetic in the same order as recover_coef -- the entries skipped there add exact zeros to the intercept's sum.
template <typename T>
void recover_coef_sparse(const DeviceData<T>& d, const int* idx, const T* val, long long cnt, T* beta0, T* out);

// C (k x k, ldc) = A' A for A (m x k, lda) [trans=true] or A A' for A (k x m) [trans=false], both
// triangles filled.  First cut: rocBLAS SYRK + symmetrise kernel.
template <typename T>
void gram_full(const T* A, long long lda, int rows, int cols, bool atA, T* C, long long ldc, hipStream_t st);

template <typename T>
void add_diag(T* A, long long lda, int n, T v, hipStream_t st);
template <typename T>
void symmetrize_from_lower(T* A, long long lda, int n, hipStream_t st);

// In place: A (SPD, lower triangle valid) -> full symmetric inverse. Throws ADMM_ERR_NOT_SPD.
template <typename T>
void spd_inverse_full(T* A, long long lda, int n, hipStream_t st);
// fp32, hand-written matrix-core path (syrk_mfma.hip).  Contract: lda >= round_up(n, 128) and the buffer holds
// round_up(n, 128) columns, padding zero.  ADMM_HIP_FACTOR=rocsolver forces the library path.
void spd_inverse_mfma_f32(float* A, long long lda, int n, hipStream_t st);
// fp32 SPD inverse of a matrix stored in whole 128-blocks (lda >= round_up(n, 128), that many zero-padded columns
// allocated): hand-written matrix-core kernels for n >= 256, potrf + two trsm below (ADMM_HIP_FACTOR=rocsolver forces the latter).
void spd_inverse_f32(float* A, long long lda, int n, hipStream_t st);
// spd_inverse_mfma_f32 with the factorisation's block columns dealt out to the ranks of the attached communicator; of the inverse
// only the lower 128 x 128 tiles in `need` (bi << 16 | bj) are formed (syrk_mfma.hip).  *flops: what this rank computed.
void spd_inverse_mfma_f32_dist(float* A, long long lda, int n, const std::vector<int>& need, double* flops, hipStream_t st);
// (A + diag I)^-1 of a float matrix, factorised and inverted in double and rounded to float once (same storage).
void spd_inverse_f32_via_f64(float* A, long long lda, int n, double diag, hipStream_t st);
// fp64 counterparts (gemm_f64_mfma.hip); same storage requirements.
void spd_inverse_mfma_f64(double* A, long long lda, int n, hipStream_t st);
void spd_inverse_f64(double* A, long long lda, int n, hipStream_t st);
// A -> Cholesky factor L in place (lower); returns U = L^-T (lda x round_up(n, 128), upper triangular, zero padded).
DevBuf<double> cholesky_linvt_mfma_f64(double* A, long long lda, int n, hipStream_t st);
// C (M x N) = A B' for operands with the output index contiguous (rows readable up to the next multiple of 128, K % 8 == 0).
// b_lower: B is lower triangular (B[j, k] = 0 for k > j, zero padded) -- the K loop of a tile ends at its last column: same bits, half the flops
void gemm_nt_f64(const double* A, long long lda, const double* B, long long ldb, double* C, long long ldc, int M, int N, int K, hipStream_t st, bool b_lower = false);
// In place Cholesky (lower). Throws ADMM_ERR_NOT_SPD.
template <typename T>
void cholesky_lower(T* A, long long lda, int n, hipStream_t st);
// B <- L^-1 B (left, lower, no-trans), B is n x m.
template <typename T>
void trsm_left_lower(const T* L, long long ldl, int n, T* B, long long ldb, int m, hipStream_t st);
// B <- B L^-T (right, lower, trans), B is m x n.
template <typename T>
void trsm_right_lower_t(const T* L, long long ldl, int n, T* B, long long ldb, int m, hipStream_t st);
// out (cols x rows, ldo) = in' for in (rows x cols, ldi)
template <typename T>
void transpose(const T* in, long long ldi, int rows, int cols, T* out, long long ldo, hipStream_t st);

// y = A v for a full symmetric device matrix (column-major, lda), host vectors; used by Lanczos.
template <typename T>
struct SymMatVec {
    const T* A; long long lda; int n; hipStream_t st;
    DevBuf<T> dv, dw, part;
    GemvTPlan pl; long long stride = 0;
    SymMatVec(const T* A_, long long lda_, int n_, hipStream_t st_);
    void operator()(const T* v_host, T* w_host);
};

// Gram-free operator of the wide solver's spectral-radius estimate (the idea of the reference's unused
// ADMMMatOp.h:31-40): w = X (X' v) for X n x p column-major, two passes over X per product instead of one n^2 p Gram up
// front -- at BASELINE configs[2] (n = 2000, p = 2 * 10^5) the 3-5 products of the ncv = 3 Lanczos run read 16 GB against the
// 8e11 flop of X X'.  Same value as the Gram-based call to float rounding (the sums associate differently).
struct GramFreeWideOp {
    const float* X; long long ldx; int n, p; hipStream_t st;
    DevBuf<float> dv, ds, dw, part;
    int cols_per_wg = 512; long long ldpart = 0; int nchunk = 0;
    GramFreeWideOp(const float* X_, long long ldx_, int n_, int p_, hipStream_t st_);
    void operator()(const float* v_host, float* w_host);
};

// The reference's Spectra call: SymEigsSolver<T, LARGEST_ALGE>(op, 1, 3); init(); compute(10, 0.1)
// (ADMMLassoTall.h:196-201, ADMMLassoWide.h:202-207).  Returns the Ritz value; throws
// ADMM_ERR_EIGS if it never passes the loose convergence test.
float lanczos_largest_f32(const std::function<void(const float*, float*)>& op, int n, int* nmatop);
double lanczos_largest_f64(const std::function<void(const double*, double*)>& op, int n, int* nmatop);

// max_j |v_j| of a device vector.
template <typename T>
T device_absmax(const T* v, int n, hipStream_t st);

// y = A' v (A m x k column-major, v length m) -> host-free device result; convenience wrapper
// over gemv_t with an internal partial buffer.
template <typename T>
void gemv_t_simple(const T* A, long long lda, int m, int k, const T* v, T* y, hipStream_t st);

}  // namespace admm
