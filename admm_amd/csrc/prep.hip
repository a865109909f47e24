// One-time preparation kernels and library wrappers (see prep.h).
#include "prep.h"
#include <thread>
#include <algorithm>
#include "gemv_kernels.h"
#include "comm.h"

#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>
#include <dlfcn.h>
#include <initializer_list>
#include <mutex>

namespace admm {

// ------------------------------------------------------------------ process-wide bits
static thread_local std::string g_last_error;
void set_last_error(const std::string& m) { g_last_error = m; }
const std::string& last_error_ref() { return g_last_error; }

void require_device() {
    int cnt = 0;
    hipError_t e = hipGetDeviceCount(&cnt);
    if (e != hipSuccess || cnt <= 0)
        throw Error(ADMM_ERR_NO_DEVICE, "no usable HIP device (libadmm_hip has no CPU fallback)");
}

namespace {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        // rocprofv3 --marker-trace listens to the rocprofiler-sdk's roctx library; the roctracer one is the fall-back for older tools
        for (const char* n : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            if (void* h = dlopen(n, RTLD_NOW | RTLD_GLOBAL)) {
                push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
                pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
                if (push && pop) return;
                push = nullptr; pop = nullptr;
            }
        }
    }
};
const Roctx& roctx() { static const Roctx* r = new Roctx(); return *r; }
}  // namespace
TraceRange::TraceRange(const char* name) : on(roctx().push != nullptr) { if (on) (void)roctx().push(name); }
TraceRange::~TraceRange() { if (on) (void)roctx().pop(); }

long long resident_workgroups(int occupancy_per_cu) {
    if (const char* e = option("TEST_RESIDENT_WGS")) { const long long v = std::atoll(e); if (v >= 0) return v; }
    return (long long)occupancy_per_cu * device_info().num_cu;
}

const DeviceInfo& device_info() {
    static std::mutex mu;
    static std::vector<DeviceInfo> cache;
    static std::vector<char> have;
    int dev = 0;
    ADMM_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    if ((int)cache.size() <= dev) { cache.resize(dev + 1); have.resize(dev + 1, 0); }
    if (!have[dev]) {
        hipDeviceProp_t prop;
        ADMM_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        cache[dev].num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        cache[dev].lds_per_block = prop.sharedMemPerBlock;
        int optin = 0;
        if (hipDeviceGetAttribute(&optin, hipDeviceAttributeSharedMemPerBlockOptin, dev) == hipSuccess && optin > 0)
            cache[dev].lds_optin = std::max((size_t)optin, cache[dev].lds_per_block);
        else
            cache[dev].lds_optin = cache[dev].lds_per_block;
        have[dev] = 1;
    }
    return cache[dev];
}

// rocBLAS / rocSOLVER are only used under the A/B knobs (ADMM_HIP_GRAM=rocblas, ADMM_HIP_FACTOR=rocsolver): they are
// loaded on first use instead of being link-time dependencies (loading librocblas cold costs seconds, creating a
// handle another 0.1-0.2 s; a default run needs neither).  The headers are used for types only.
struct BlasApi {
    void* hb = nullptr; void* hs = nullptr;
    decltype(&rocblas_create_handle) create_handle = nullptr;
    decltype(&rocblas_destroy_handle) destroy_handle = nullptr;
    decltype(&rocblas_set_stream) set_stream = nullptr;
    decltype(&rocblas_ssyrk) ssyrk = nullptr;
    decltype(&rocblas_dsyrk) dsyrk = nullptr;
    decltype(&rocblas_strsm) strsm = nullptr;
    decltype(&rocblas_dtrsm) dtrsm = nullptr;
    decltype(&rocsolver_spotrf) spotrf = nullptr;
    decltype(&rocsolver_dpotrf) dpotrf = nullptr;
    static void* open_any(std::initializer_list<const char*> names) {
        for (const char* n : names) if (void* h = dlopen(n, RTLD_NOW | RTLD_GLOBAL)) return h;
        return nullptr;
    }
    template <typename F> void sym(void* h, const char* name, F& f) {
        f = reinterpret_cast<F>(dlsym(h, name));
        if (!f) throw Error(ADMM_ERR_BLAS, std::string("symbol not found: ") + name);
    }
    BlasApi() {
        hb = open_any({"librocblas.so", "librocblas.so.5", "librocblas.so.4"});
        hs = open_any({"librocsolver.so", "librocsolver.so.0"});
        if (!hb || !hs) throw Error(ADMM_ERR_BLAS, "ADMM_HIP_GRAM=rocblas / ADMM_HIP_FACTOR=rocsolver need librocblas.so and librocsolver.so on the loader path");
        sym(hb, "rocblas_create_handle", create_handle); sym(hb, "rocblas_destroy_handle", destroy_handle);
        sym(hb, "rocblas_set_stream", set_stream);
        sym(hb, "rocblas_ssyrk", ssyrk); sym(hb, "rocblas_dsyrk", dsyrk);
        sym(hb, "rocblas_strsm", strsm); sym(hb, "rocblas_dtrsm", dtrsm);
        sym(hs, "rocsolver_spotrf", spotrf); sym(hs, "rocsolver_dpotrf", dpotrf);
    }
};
static BlasApi& blas_api() {
    static BlasApi api;
    return api;
}
struct BlasHandle {
    rocblas_handle h = nullptr;
    BlasHandle() {
        if (blas_api().create_handle(&h) != rocblas_status_success) throw Error(ADMM_ERR_BLAS, "rocblas_create_handle failed");
    }
    ~BlasHandle() { if (h) blas_api().destroy_handle(h); }
};
static rocblas_handle blas(hipStream_t st) {
    static thread_local BlasHandle bh;
    if (blas_api().set_stream(bh.h, st) != rocblas_status_success) throw Error(ADMM_ERR_BLAS, "rocblas_set_stream failed");
    return bh.h;
}
#define ADMM_BLAS_CHECK(expr)                                                                   \
    do {                                                                                        \
        rocblas_status _s = (expr);                                                             \
        if (_s != rocblas_status_success)                                                       \
            throw ::admm::Error(ADMM_ERR_BLAS, std::string(#expr) + " failed with rocblas_status " + std::to_string((int)_s)); \
    } while (0)

// ------------------------------------------------------------------ standardise (DataStd.h:89-155)
// The data are narrowed to T first (Lasso.cpp:49 copies double->float before standardising), then
// three column passes on the device copy: sums -> means, centred sums of squares -> population sd,
// apply.  Statistics accumulate in double; in a multi-process run the two statistic vectors are
// all-reduced so that every rank standardises with the GLOBAL moments (the reference standardises
// before it splits rows, ParLasso.cpp:68-72).  Index p of the statistic vectors is y.
template <typename T>
__global__ void __launch_bounds__(256)
convert_cols_kernel(const double* __restrict__ x, long long ldin, int n, T* __restrict__ X, long long ldx) {
    const int j = blockIdx.x;
    const double* src = x + (size_t)j * ldin;
    T* dst = X + (size_t)j * ldx;
    for (int i = blockIdx.y * 256 + threadIdx.x; i < n; i += gridDim.y * 256) dst[i] = (T)src[i];
}

template <typename T, int PASS>      // PASS 0: sum ; PASS 1: sum of (v - mean)^2
__global__ void __launch_bounds__(256)
colstat_kernel(const T* __restrict__ X, long long ldx, const T* __restrict__ Y, int n, int p,
               const T* __restrict__ mean, double* __restrict__ out) {
    __shared__ double scratch[4];
    const int j = blockIdx.x;
    const T* col = j < p ? X + (size_t)j * ldx : Y;
    const T m = PASS == 1 ? mean[j] : T(0);
    double s[1] = {0.0};
    for (int i = threadIdx.x; i < n; i += 256) {
        if (PASS == 0) s[0] += (double)col[i];
        else { const T c = col[i] - m; s[0] += (double)c * (double)c; }
    }
    block_sum<double, 1>(s, scratch);
    if (threadIdx.x == 0) out[j] = s[0];
}

template <typename T>
__global__ void finish_mean_kernel(const double* sum, double n_total, int count, T* mean) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < count) mean[j] = (T)(sum[j] / n_total);
}
template <typename T>
__global__ void finish_scale_kernel(const double* ss, double n_total, int count, T* scale, T* inv) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < count) {
        const T n_invsqrt = (T)(1.0 / sqrt((double)(T)n_total));
        const T sc = (T)((T)sqrt(ss[j]) * n_invsqrt);            // ||v - mean|| / sqrt(n): population sd
        scale[j] = sc;
        inv[j] = (T)(1.0 / (double)sc);
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
apply_std_kernel(T* __restrict__ X, long long ldx, T* __restrict__ Y, int n, int p, int flag,
                 const T* __restrict__ mean, const T* __restrict__ scale, const T* __restrict__ inv) {
    const int j = blockIdx.x;
    const bool center = (flag & 2) != 0;
    if (j < p) {
        T* col = X + (size_t)j * ldx;
        const T m = center ? mean[j] : T(0);
        const T iv = (flag & 1) ? inv[j] : T(1);
        for (int i = blockIdx.y * 256 + threadIdx.x; i < n; i += gridDim.y * 256) {
            T v = col[i];
            if (center) v = v - m;
            if (flag & 1) v = v * iv;                              // X.col(i) *= 1 / scaleX[i]
            col[i] = v;
        }
    } else {                                                       // y: flag 1 scales by sd without centring (DataStd.h:96-99)
        const T m = center ? mean[p] : T(0);
        const T sc = scale[p];
        for (int i = blockIdx.y * 256 + threadIdx.x; i < n; i += gridDim.y * 256) {
            T v = Y[i];
            if (center) v = v - m;
            Y[i] = v / sc;                                         // Y.array() /= scaleY
        }
    }
}

// The four steps above for one column in ONE launch (round 6; single process): the block narrows its column, keeps the sum, forms the
// mean, the centred sum of squares, the scale, and applies them -- every thread re-reads only what it wrote itself (same indices in
// every sweep), from the cache.  The statistics are the separate kernels' to the bit (same elements per thread in the same order, the
// same block sum), so the data are bit-identical to the unfused path's (PREP_FUSED=0); HBM sees the input once and the output once
// instead of seven sweeps.  Column p is y.  Rows [n, ldx) are zeroed here (no memset of the whole matrix).
template <typename T>
__global__ void __launch_bounds__(256)
convert_standardize_kernel(const double* __restrict__ x, const double* __restrict__ y, long long ldin, int n, int p, int j0,
                           T* __restrict__ X, T* __restrict__ Y, long long ldx, int flag, double n_total,
                           T* __restrict__ mean_out, T* __restrict__ scale_out) {
    __shared__ double scratch[4];
    const int jl = blockIdx.x;                      // column of this launch's input block; j0 + jl: its number
    const bool isy = y != nullptr;
    const double* src = isy ? y : x + (size_t)jl * ldin;
    T* dst = isy ? Y : X + (size_t)(j0 + jl) * ldx;
    const int j = isy ? p : j0 + jl;
    double s[1] = {0.0};
    {   // eight requests in flight per thread; the additions keep their order
        int i = threadIdx.x;
        for (; i + 7 * 256 < n; i += 8 * 256) {
            double v8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v8[u] = __builtin_nontemporal_load(src + i + u * 256);
#pragma unroll
            for (int u = 0; u < 8; ++u) { const T v = (T)v8[u]; dst[i + u * 256] = v; s[0] += (double)v; }
        }
        for (; i < n; i += 256) { const T v = (T)src[i]; dst[i] = v; s[0] += (double)v; }
    }
    for (long long i = n + threadIdx.x; i < ldx; i += 256) dst[i] = T(0);
    if (flag == 0) return;
    block_sum<double, 1>(s, scratch);
    const T m = (T)(s[0] / n_total);
    double ss[1] = {0.0};
    {
        int i = threadIdx.x;
        for (; i + 7 * 256 < n; i += 8 * 256) {
            T v8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v8[u] = dst[i + u * 256];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const T c = v8[u] - m; ss[0] += (double)c * (double)c; }
        }
        for (; i < n; i += 256) { const T c = dst[i] - m; ss[0] += (double)c * (double)c; }
    }
    __syncthreads();                                // (scratch is reused)
    block_sum<double, 1>(ss, scratch);
    const T n_invsqrt = (T)(1.0 / sqrt((double)(T)n_total));
    const T sc = (T)((T)sqrt(ss[0]) * n_invsqrt);
    const T iv = (T)(1.0 / (double)sc);
    const bool center = (flag & 2) != 0;
    if (!isy) {
        const T ivv = (flag & 1) ? iv : T(1);
        for (int i = threadIdx.x; i < n; i += 256) {
            T v = dst[i];
            if (center) v = v - m;
            if (flag & 1) v = v * ivv;
            dst[i] = v;
        }
    } else {
        for (int i = threadIdx.x; i < n; i += 256) {
            T v = dst[i];
            if (center) v = v - m;
            dst[i] = v / sc;
        }
    }
    if (threadIdx.x == 0) { mean_out[j] = m; scale_out[j] = sc; }
}

// ---- write_device (admm_internal.h): pinned staging ring + worker threads for the host-side copy
namespace {
struct H2DRing {
    static constexpr int kSlots = 3;
    static constexpr size_t kSlot = size_t(32) << 20;
    void* slot[kSlots] = {nullptr, nullptr, nullptr};
    hipEvent_t ev[kSlots] = {nullptr, nullptr, nullptr};
    hipStream_t st = nullptr;
    int dev = -1;
    bool ok() const { return st != nullptr; }
    void init() {
        ADMM_HIP_CHECK(hipGetDevice(&dev));
        for (int i = 0; i < kSlots; ++i) {
            ADMM_HIP_CHECK(hipHostMalloc(&slot[i], kSlot, hipHostMallocDefault));
            ADMM_HIP_CHECK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        }
        ADMM_HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    }
};
// One ring per host thread and device, owned by the thread: the holder's destructor releases the pinned slots, the events and the
// stream when the thread exits (a bare thread_local vector of pointers leaked 96 MB of pinned memory per short-lived
// thread that passed a host input).  A ring exists only in threads that pass LARGE host inputs (write_device: small transfers are
// plain copies).
struct H2DRingHolder {
    std::vector<H2DRing*> rings;
    ~H2DRingHolder() {
        for (H2DRing* r : rings) {
            if (r->st) { (void)hipStreamSynchronize(r->st); (void)hipStreamDestroy(r->st); }
            for (int i = 0; i < H2DRing::kSlots; ++i) {
                if (r->ev[i]) (void)hipEventDestroy(r->ev[i]);
                if (r->slot[i]) (void)hipHostFree(r->slot[i]);
            }
            delete r;
        }
    }
};
H2DRing& h2d_ring() {
    static thread_local H2DRingHolder holder;
    int dev = 0;
    ADMM_HIP_CHECK(hipGetDevice(&dev));
    for (H2DRing* r : holder.rings) if (r->dev == dev) return *r;
    H2DRing* r = new H2DRing();
    holder.rings.push_back(r);                           // owned from here on: a failed init() is cleaned up by the holder
    r->init();
    return *r;
}
int h2d_threads() {
    if (const char* e = option("H2D_THREADS")) { const int v = std::atoi(e); if (v >= 1) return std::min(v, 32); }
    static const int n = []() {
        const unsigned hw = std::thread::hardware_concurrency();
        return (int)std::max(1u, std::min(16u, hw ? hw / 2 : 4u));      // C2 host input on the 128-thread box: 8 threads 50 GB/s, 16 threads 56 GB/s = the pageable hipMemcpy's rate
    }();
    return n;
}
void parallel_memcpy(char* dst, const char* src, size_t bytes, int nthreads) {
    const size_t per = ((bytes + nthreads - 1) / nthreads + 4095) / 4096 * 4096;
    if (nthreads <= 1 || bytes < (size_t(4) << 20)) { std::memcpy(dst, src, bytes); return; }
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; ++t) {
        const size_t off = per * t;
        if (off >= bytes) break;
        th.emplace_back([=]() { std::memcpy(dst + off, src + off, std::min(per, bytes - off)); });
    }
    std::memcpy(dst, src, std::min(per, bytes));
    for (std::thread& t : th) t.join();
}
}  // namespace

void write_device(void* dst, const void* src, size_t bytes) {
    if (!bytes) return;
    const char* h2d = option("H2D");
    const bool pageable = h2d && std::string(h2d) == "pageable";
    if (pageable) { ADMM_HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); return; }
    if (bytes < (size_t(4) << 20)) { ADMM_HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); return; }      // (a y vector, a lambda grid: no ring for those)
    H2DRing& r = h2d_ring();
    const int nt = h2d_threads();
    int i = 0;
    bool used[H2DRing::kSlots] = {false, false, false};
    try {
        for (size_t off = 0; off < bytes; off += H2DRing::kSlot, i = (i + 1) % H2DRing::kSlots) {
            const size_t n = std::min(H2DRing::kSlot, bytes - off);
            if (used[i]) ADMM_HIP_CHECK(hipEventSynchronize(r.ev[i]));          // the DMA that read this slot has finished
            parallel_memcpy(static_cast<char*>(r.slot[i]), static_cast<const char*>(src) + off, n, nt);
            ADMM_HIP_CHECK(hipMemcpyAsync(static_cast<char*>(dst) + off, r.slot[i], n, hipMemcpyHostToDevice, r.st));
            ADMM_HIP_CHECK(hipEventRecord(r.ev[i], r.st));
            used[i] = true;
        }
        ADMM_HIP_CHECK(hipStreamSynchronize(r.st));
    } catch (...) {
        (void)hipStreamSynchronize(r.st);                   // no DMA may still be reading a slot when the next call starts with used[] = false
        throw;
    }
}

template <typename T>
void upload_standardize(DeviceData<T>& d, const double* x, const double* y, int n, int p, int mem,
                        bool standardize, bool intercept, hipStream_t st, long long n_total) {
    const TraceRange trace_range("admm:convert+standardize");
    const bool dist = n_total > 0;          // only the multi-process entry points pass n_total: all ranks are in this call
    if (n_total <= 0) n_total = n;
    d.n = n; d.p = p; d.n_total = n_total;
    d.flag = int(standardize) + 2 * int(intercept);
    d.ldx = round_up(n, 32);
    d.X.alloc((size_t)d.ldx * p);
    d.Y.alloc((size_t)d.ldx);
    // one launch per block of columns does everything (convert_standardize_kernel) unless the moments are global (several processes)
    bool fused = !dist;
    if (const char* e = option("PREP_FUSED")) fused = fused && std::string(e) != "0";
    DevBuf<T> fmean, fscale;
    if (fused) { fmean.alloc(p + 1); fscale.alloc(p + 1); }
    if (!fused) { d.X.zero(st); d.Y.zero(st); }
    const int ny = std::max(1, std::min(64, (n + 255) / 256));
    double t0 = now_s();
    double th = 0;
    auto convert_block = [&](const double* src, int c0, int nc) {       // columns [c0, c0 + nc) from a device block of nc columns
        if (fused) hipLaunchKernelGGL((convert_standardize_kernel<T>), dim3(nc), dim3(256), 0, st, src, (const double*)nullptr, (long long)n, n, p, c0,
                                      d.X.get(), d.Y.get(), d.ldx, d.flag, (double)n_total, fmean.get(), fscale.get());
        else hipLaunchKernelGGL((convert_cols_kernel<T>), dim3(nc, ny), dim3(256), 0, st, src, (long long)n, n, d.X.get() + (size_t)c0 * d.ldx, d.ldx);
    };
    auto convert_y = [&](const double* src) {
        if (fused) hipLaunchKernelGGL((convert_standardize_kernel<T>), dim3(1), dim3(256), 0, st, (const double*)nullptr, src, (long long)n, n, p, 0,
                                      d.X.get(), d.Y.get(), d.ldx, d.flag, (double)n_total, fmean.get(), fscale.get());
        else hipLaunchKernelGGL((convert_cols_kernel<T>), dim3(1, ny), dim3(256), 0, st, src, (long long)n, n, d.Y.get(), d.ldx);
    };
    // ---- pass 0: narrow to T on the device
    if (mem == ADMM_MEM_DEVICE) {
        convert_block(x, 0, p);
        convert_y(y);
    } else {
        // Host input (what R hands over): stream column chunks through two device staging buffers.
        const size_t chunk_bytes = (size_t)256 << 20;
        int cols_per_chunk = (int)std::max<size_t>(1, chunk_bytes / ((size_t)n * sizeof(double)));
        cols_per_chunk = std::min(cols_per_chunk, p);
        DevBuf<double> stage[2];
        stage[0].alloc((size_t)cols_per_chunk * n);
        stage[1].alloc((size_t)cols_per_chunk * n);
        Event ev[2];
        bool used[2] = {false, false};
        int b = 0;
        for (int c0 = 0; c0 < p; c0 += cols_per_chunk, b ^= 1) {
            const int nc = std::min(cols_per_chunk, p - c0);
            if (used[b]) ADMM_HIP_CHECK(hipEventSynchronize(ev[b].e));
            double t1 = now_s();
            write_device(stage[b].get(), x + (size_t)c0 * n, (size_t)nc * n * sizeof(double));
            th += now_s() - t1;
            convert_block(stage[b].get(), c0, nc);
            ADMM_HIP_CHECK(hipEventRecord(ev[b].e, st));
            used[b] = true;
        }
        DevBuf<double> ystage(n);
        double t1 = now_s();
        write_device(ystage.get(), y, (size_t)n * sizeof(double));
        th += now_s() - t1;
        convert_y(ystage.get());
        comm_stream_sync(st);
    }
    d.meanX.assign(p, T(0));
    d.scaleX.assign(p, T(1));
    d.meanY = T(0); d.scaleY = T(1);
    if (d.flag != 0 && fused) {
        const int cnt = p + 1;
        std::vector<T> hm(cnt), hs(cnt);
        ADMM_HIP_CHECK(hipMemcpyAsync(hm.data(), fmean.get(), cnt * sizeof(T), hipMemcpyDeviceToHost, st));
        ADMM_HIP_CHECK(hipMemcpyAsync(hs.data(), fscale.get(), cnt * sizeof(T), hipMemcpyDeviceToHost, st));
        comm_stream_sync(st);
        if (d.flag & 2) { for (int j = 0; j < p; ++j) d.meanX[j] = hm[j]; d.meanY = hm[p]; }
        if (d.flag & 1) for (int j = 0; j < p; ++j) d.scaleX[j] = hs[j];
        d.scaleY = hs[p];
    } else if (d.flag != 0) {
        const int cnt = p + 1;
        DevBuf<double> stat(cnt);
        DevBuf<T> mean(cnt), scale(cnt), inv(cnt);
        hipLaunchKernelGGL((colstat_kernel<T, 0>), dim3(cnt), dim3(256), 0, st, d.X.get(), d.ldx, d.Y.get(), n, p, mean.get(), stat.get());
        if (dist) allreduce_sum_f64(stat.get(), cnt, st);
        hipLaunchKernelGGL((finish_mean_kernel<T>), dim3((cnt + 255) / 256), dim3(256), 0, st, stat.get(), (double)n_total, cnt, mean.get());
        hipLaunchKernelGGL((colstat_kernel<T, 1>), dim3(cnt), dim3(256), 0, st, d.X.get(), d.ldx, d.Y.get(), n, p, mean.get(), stat.get());
        if (dist) allreduce_sum_f64(stat.get(), cnt, st);
        hipLaunchKernelGGL((finish_scale_kernel<T>), dim3((cnt + 255) / 256), dim3(256), 0, st, stat.get(), (double)n_total, cnt, scale.get(), inv.get());
        hipLaunchKernelGGL((apply_std_kernel<T>), dim3(cnt, ny), dim3(256), 0, st, d.X.get(), d.ldx, d.Y.get(), n, p, d.flag,
                           mean.get(), scale.get(), inv.get());
        std::vector<T> hm(cnt), hs(cnt);
        ADMM_HIP_CHECK(hipMemcpyAsync(hm.data(), mean.get(), cnt * sizeof(T), hipMemcpyDeviceToHost, st));
        ADMM_HIP_CHECK(hipMemcpyAsync(hs.data(), scale.get(), cnt * sizeof(T), hipMemcpyDeviceToHost, st));
        comm_stream_sync(st);
        if (d.flag & 2) { for (int j = 0; j < p; ++j) d.meanX[j] = hm[j]; d.meanY = hm[p]; }
        if (d.flag & 1) for (int j = 0; j < p; ++j) d.scaleX[j] = hs[j];
        d.scaleY = hs[p];
    }
    comm_stream_sync(st);
    d.t_h2d = th;
    d.t_std = now_s() - t0 - th;
}
template void upload_standardize<float>(DeviceData<float>&, const double*, const double*, int, int, int, bool, bool, hipStream_t, long long);
// Multi-response fits (admm_hip_lasso_multi): a DeviceData for another response y of the SAME x.  X, its column statistics
// and (if present) the Gram matrix are copied device to device from `base`; y (device doubles) is converted and
// standardised by the kernels upload_standardize runs on column p with the same flag -- column by column they do not depend
// on one another, so the result is bit-identical to upload_standardize(x, y).
void standardize_response_f32(const double* y_dev, int n, int flag, long long n_total, float* Yout, long long ld,
                              float* meanY, float* scaleY, hipStream_t st) {
    typedef float T;
    ADMM_HIP_CHECK(hipMemsetAsync(Yout, 0, (size_t)ld * sizeof(T), st));
    const int ny = std::max(1, std::min(64, (n + 255) / 256));
    hipLaunchKernelGGL((convert_cols_kernel<T>), dim3(1, ny), dim3(256), 0, st, y_dev, (long long)n, n, Yout, ld);
    *meanY = T(0); *scaleY = T(1);
    if (flag != 0) {
        DevBuf<double> stat(1);
        DevBuf<T> mean(1), scale(1), inv(1);
        // p = 0: the only "column" of these launches is y
        hipLaunchKernelGGL((colstat_kernel<T, 0>), dim3(1), dim3(256), 0, st, Yout, ld, Yout, n, 0, mean.get(), stat.get());
        hipLaunchKernelGGL((finish_mean_kernel<T>), dim3(1), dim3(256), 0, st, stat.get(), (double)n_total, 1, mean.get());
        hipLaunchKernelGGL((colstat_kernel<T, 1>), dim3(1), dim3(256), 0, st, Yout, ld, Yout, n, 0, mean.get(), stat.get());
        hipLaunchKernelGGL((finish_scale_kernel<T>), dim3(1), dim3(256), 0, st, stat.get(), (double)n_total, 1, scale.get(), inv.get());
        hipLaunchKernelGGL((apply_std_kernel<T>), dim3(1, ny), dim3(256), 0, st, Yout, ld, Yout, n, 0, flag, mean.get(), scale.get(), inv.get());
        T hm = 0, hs = 1;
        ADMM_HIP_CHECK(hipMemcpyAsync(&hm, mean.get(), sizeof(T), hipMemcpyDeviceToHost, st));
        ADMM_HIP_CHECK(hipMemcpyAsync(&hs, scale.get(), sizeof(T), hipMemcpyDeviceToHost, st));
        comm_stream_sync(st);
        if (flag & 2) *meanY = hm;
        *scaleY = hs;
    }
    ADMM_HIP_CHECK(hipGetLastError());
}

void clone_with_response_f32(DeviceData<float>& d, const DeviceData<float>& base, const float* gram, long long ldgram,
                             const double* y_dev, hipStream_t st) {
    typedef float T;
    const int n = base.n, p = base.p;
    d.n = n; d.p = p; d.n_total = base.n_total; d.ldx = base.ldx; d.flag = base.flag;
    d.meanX = base.meanX; d.scaleX = base.scaleX;
    d.X.alloc((size_t)d.ldx * p);
    ADMM_HIP_CHECK(hipMemcpyAsync(d.X.get(), base.X.get(), (size_t)d.ldx * p * sizeof(T), hipMemcpyDeviceToDevice, st));
    d.Y.alloc((size_t)d.ldx);
    standardize_response_f32(y_dev, n, d.flag, d.n_total, d.Y.get(), d.ldx, &d.meanY, &d.scaleY, st);
    if (gram != nullptr) {
        d.gram.alloc((size_t)ldgram * ldgram);
        ADMM_HIP_CHECK(hipMemcpyAsync(d.gram.get(), gram, (size_t)ldgram * ldgram * sizeof(T), hipMemcpyDeviceToDevice, st));
        d.ldgram = ldgram;
        d.t_gram_tail = 0;
    }
    ADMM_HIP_CHECK(hipGetLastError());
    comm_stream_sync(st);
}

template void upload_standardize<double>(DeviceData<double>&, const double*, const double*, int, int, int, bool, bool, hipStream_t, long long);

void gram_rows_mfma_f32(const float* Z, long long ldz, int r0, int nr, int K, float* C, long long ldc, hipStream_t st);   // syrk_mfma.hip

void upload_standardize_gram_f32(DeviceData<float>& d, const double* x, const double* y, int n, int p,
                                 bool standardize, bool intercept, hipStream_t st) {
    const TraceRange trace_range("admm:upload+standardize+gram (pipelined)");
    using T = float;
    d.n = n; d.p = p; d.n_total = n;
    d.flag = int(standardize) + 2 * int(intercept);
    d.ldx = round_up(n, 32);
    d.X.alloc((size_t)d.ldx * p);
    d.Y.alloc((size_t)d.ldx);
    d.X.zero(st);
    d.Y.zero(st);
    const long long ldz = round_up(p, 128);
    const int nk = (int)round_up(n, 16);
    // the operand of the matrix-core Gram: X' as three bf16 planes (gram_bf16x3.hip; p >= 4096 here, the same kernel the one-shot
    // Gram takes at this size), or as floats with the output index contiguous (ADMM_HIP_GRAM_SPLIT=0)
    const bool b3 = gram_split_mode() != 0;
    GramSplit3 z3;
    DevBuf<float> Z;
    if (b3) z3.alloc(p, n, st);
    else { Z.alloc((size_t)ldz * nk); Z.zero(st); }
    d.ldgram = ldz;
    d.gram.alloc((size_t)ldz * ldz);
    d.gram.zero(st);
    const int ny = std::max(1, std::min(64, (n + 255) / 256));
    const int cnt = p + 1;
    DevBuf<double> stat(cnt);
    DevBuf<T> mean(cnt), scale(cnt), inv(cnt);
    const double t0 = now_s();
    double th = 0;
    auto standardise_cols = [&](int c0, int nc, bool is_y, hipStream_t st) {       // columns [c0, c0 + nc) of X, or y (statistic index p)
        if (d.flag == 0) return;
        T* Xc = d.X.get() + (size_t)c0 * d.ldx;
        const int idx = is_y ? p : c0, np_ = is_y ? 0 : nc, nb = is_y ? 1 : nc;
        hipLaunchKernelGGL((colstat_kernel<T, 0>), dim3(nb), dim3(256), 0, st, Xc, d.ldx, d.Y.get(), n, np_, mean.get() + idx, stat.get() + idx);
        hipLaunchKernelGGL((finish_mean_kernel<T>), dim3((nb + 255) / 256), dim3(256), 0, st, stat.get() + idx, (double)n, nb, mean.get() + idx);
        hipLaunchKernelGGL((colstat_kernel<T, 1>), dim3(nb), dim3(256), 0, st, Xc, d.ldx, d.Y.get(), n, np_, mean.get() + idx, stat.get() + idx);
        hipLaunchKernelGGL((finish_scale_kernel<T>), dim3((nb + 255) / 256), dim3(256), 0, st, stat.get() + idx, (double)n, nb, scale.get() + idx, inv.get() + idx);
        hipLaunchKernelGGL((apply_std_kernel<T>), dim3(nb, ny), dim3(256), 0, st, Xc, d.ldx, d.Y.get(), n, np_, d.flag,
                           mean.get() + idx, scale.get() + idx, inv.get() + idx);
    };
    // y first
    {
        DevBuf<double> ystage(n);
        double t1 = now_s();
        write_device(ystage.get(), y, (size_t)n * sizeof(double));
        th += now_s() - t1;
        hipLaunchKernelGGL((convert_cols_kernel<T>), dim3(1, ny), dim3(256), 0, st, ystage.get(), (long long)n, n, d.Y.get(), d.ldx);
        standardise_cols(0, 0, true, st);
        comm_stream_sync(st);
    }
    // x: chunks of whole 128-column blocks (about 400 MB) through two staging buffers.  Chunks alternate between two
    // streams so that the block rows of consecutive chunks (each too few tiles to fill the chip) overlap when the GPU
    // lags behind the copies: the work that arrives with a chunk grows linearly with its position.
    int cols_per_chunk = (int)(((size_t)400 << 20) / ((size_t)n * sizeof(double)));
    cols_per_chunk = std::max(128, cols_per_chunk / 128 * 128);
    DevBuf<double> stage[2];
    stage[0].alloc((size_t)cols_per_chunk * n);
    stage[1].alloc((size_t)cols_per_chunk * n);
    Stream aux;
    const hipStream_t sq[2] = {st, aux.s};
    Event ev[2], evT[2], ev0;
    bool used[2] = {false, false};
    ADMM_HIP_CHECK(hipEventRecord(ev0.e, st));                     // zero fills / y above
    ADMM_HIP_CHECK(hipStreamWaitEvent(aux.s, ev0.e, 0));
    int b = 0;
    for (int c0 = 0; c0 < p; c0 += cols_per_chunk, b ^= 1) {
        const int nc = std::min(cols_per_chunk, p - c0);
        const hipStream_t s = sq[b];
        if (used[b]) ADMM_HIP_CHECK(hipEventSynchronize(ev[b].e));
        double t1 = now_s();
        write_device(stage[b].get(), x + (size_t)c0 * n, (size_t)nc * n * sizeof(double));
        th += now_s() - t1;
        hipLaunchKernelGGL((convert_cols_kernel<T>), dim3(nc, ny), dim3(256), 0, s, stage[b].get(), (long long)n, n,
                           d.X.get() + (size_t)c0 * d.ldx, d.ldx);
        ADMM_HIP_CHECK(hipEventRecord(ev[b].e, s));               // the staging buffer is free once the conversion has read it
        used[b] = true;
        standardise_cols(c0, nc, false, s);
        if (b3) z3.split_cols(d.X.get() + (size_t)c0 * d.ldx, d.ldx, n, c0, nc, s);
        else transpose<float>(d.X.get() + (size_t)c0 * d.ldx, d.ldx, n, nc, Z.get() + c0, ldz, s);
        ADMM_HIP_CHECK(hipEventRecord(evT[b].e, s));
        // rows of Z of every earlier chunk: the even ones are ordered by this stream, the odd ones by the other's event
        if (c0 > 0) ADMM_HIP_CHECK(hipStreamWaitEvent(s, evT[b ^ 1].e, 0));
        if (b3) z3.gram_rows(c0, nc, d.gram.get(), ldz, s);
        else gram_rows_mfma_f32(Z.get(), ldz, c0, nc, nk, d.gram.get(), ldz, s);
    }
    {
        Event done;
        ADMM_HIP_CHECK(hipEventRecord(done.e, aux.s));
        ADMM_HIP_CHECK(hipStreamWaitEvent(st, done.e, 0));
    }
    const double t_last = now_s();
    symmetrize_from_lower<float>(d.gram.get(), ldz, p, st);
    d.meanX.assign(p, T(0));
    d.scaleX.assign(p, T(1));
    d.meanY = T(0); d.scaleY = T(1);
    if (d.flag != 0) {
        std::vector<T> hm(cnt), hs(cnt);
        ADMM_HIP_CHECK(hipMemcpyAsync(hm.data(), mean.get(), cnt * sizeof(T), hipMemcpyDeviceToHost, st));
        ADMM_HIP_CHECK(hipMemcpyAsync(hs.data(), scale.get(), cnt * sizeof(T), hipMemcpyDeviceToHost, st));
        comm_stream_sync(st);
        if (d.flag & 2) { for (int j = 0; j < p; ++j) d.meanX[j] = hm[j]; d.meanY = hm[p]; }
        if (d.flag & 1) for (int j = 0; j < p; ++j) d.scaleX[j] = hs[j];
        d.scaleY = hs[p];
    }
    comm_stream_sync(st);
    d.t_gram_tail = now_s() - t_last;
    d.t_h2d = th;
    d.t_std = t_last - t0 - th;                    // host time between copies (launch overhead; the kernels overlap the copies)
}

template <typename T>
void recover_coef(const DeviceData<T>& d, const T* coef, T* beta0, T* out) {
    const int p = d.p;
    T b0 = T(0);
    switch (d.flag) {
        case 0:
            for (int j = 0; j < p; ++j) out[j] = coef[j];
            break;
        case 1:
            for (int j = 0; j < p; ++j) out[j] = (coef[j] / d.scaleX[j]) * d.scaleY;
            break;
        case 2: {
            T acc = T(0);
            for (int j = 0; j < p; ++j) { out[j] = coef[j] * d.scaleY; acc += out[j] * d.meanX[j]; }
            b0 = d.meanY - acc;
            break;
        }
        default: {
            T acc = T(0);
            for (int j = 0; j < p; ++j) { out[j] = (coef[j] / d.scaleX[j]) * d.scaleY; acc += out[j] * d.meanX[j]; }
            b0 = d.meanY - acc;
        }
    }
    *beta0 = b0;
}
template <typename T>
void recover_coef_sparse(const DeviceData<T>& d, const int* idx, const T* val, long long cnt, T* beta0, T* out) {
    T b0 = T(0), acc = T(0);
    for (long long k = 0; k < cnt; ++k) {
        const int j = idx[k];
        T o;
        switch (d.flag) {
            case 0: o = val[k]; break;
            case 1: o = (val[k] / d.scaleX[j]) * d.scaleY; break;
            case 2: o = val[k] * d.scaleY; acc += o * d.meanX[j]; break;
            default: o = (val[k] / d.scaleX[j]) * d.scaleY; acc += o * d.meanX[j];
        }
        out[j] = o;
    }
    if (d.flag >= 2) b0 = d.meanY - acc;
    *beta0 = b0;
}
template void recover_coef_sparse<float>(const DeviceData<float>&, const int*, const float*, long long, float*, float*);
template void recover_coef<float>(const DeviceData<float>&, const float*, float*, float*);
template void recover_coef<double>(const DeviceData<double>&, const double*, double*, double*);

// ------------------------------------------------------------------ small dense helpers
template <typename T>
__global__ void add_diag_kernel(T* A, long long lda, int n, T v) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) A[(size_t)i * lda + i] += v;
}
template <typename T>
void add_diag(T* A, long long lda, int n, T v, hipStream_t st) {
    hipLaunchKernelGGL((add_diag_kernel<T>), dim3((n + 255) / 256), dim3(256), 0, st, A, lda, n, v);
}
template void add_diag<float>(float*, long long, int, float, hipStream_t);
template void add_diag<double>(double*, long long, int, double, hipStream_t);

// upper(i<j) <- lower: A[i + j*lda] = A[j + i*lda]; 32x32 tiles through LDS so both sides coalesce.
template <typename T>
__global__ void __launch_bounds__(256)
symmetrize_kernel(T* A, long long lda, int n) {
    __shared__ T tile[32][33];
    const int bi = blockIdx.y, bj = blockIdx.x;      // tile (bi,bj) of the LOWER part: bi >= bj
    if (bi < bj) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    // read lower tile rows bi*32.., cols bj*32..
    for (int c = ty; c < 32; c += 8) {
        int r = bi * 32 + tx, cc = bj * 32 + c;
        tile[c][tx] = (r < n && cc < n) ? A[(size_t)cc * lda + r] : T(0);
    }
    __syncthreads();
    // write transposed into tile (bj,bi): element (r', c') = lower(c', r')
    for (int c = ty; c < 32; c += 8) {
        int r = bj * 32 + tx, cc = bi * 32 + c;     // target row r (in bj block), col cc (in bi block)
        if (r < n && cc < n && r < cc) A[(size_t)cc * lda + r] = tile[tx][c];
    }
}
template <typename T>
void symmetrize_from_lower(T* A, long long lda, int n, hipStream_t st) {
    int nb = (n + 31) / 32;
    hipLaunchKernelGGL((symmetrize_kernel<T>), dim3(nb, nb), dim3(256), 0, st, A, lda, n);
}
template void symmetrize_from_lower<float>(float*, long long, int, hipStream_t);
template void symmetrize_from_lower<double>(double*, long long, int, hipStream_t);

template <typename T>
__global__ void __launch_bounds__(256)
transpose_kernel(const T* __restrict__ in, long long ldi, int rows, int cols, T* __restrict__ out, long long ldo) {
    __shared__ T tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    for (int c = ty; c < 32; c += 8) {
        int r = r0 + tx, cc = c0 + c;
        tile[c][tx] = (r < rows && cc < cols) ? in[(size_t)cc * ldi + r] : T(0);
    }
    __syncthreads();
    for (int c = ty; c < 32; c += 8) {
        int orow = c0 + tx, ocol = r0 + c;     // out(orow, ocol) = in(ocol, orow)
        if (orow < cols && ocol < rows) out[(size_t)ocol * ldo + orow] = tile[tx][c];
    }
}
template <typename T>
void transpose(const T* in, long long ldi, int rows, int cols, T* out, long long ldo, hipStream_t st) {
    hipLaunchKernelGGL((transpose_kernel<T>), dim3((rows + 31) / 32, (cols + 31) / 32), dim3(256), 0, st,
                       in, ldi, rows, cols, out, ldo);
}
template void transpose<float>(const float*, long long, int, int, float*, long long, hipStream_t);
template void transpose<double>(const double*, long long, int, int, double*, long long, hipStream_t);

// ------------------------------------------------------------------ Gram / factorisation (library first cut)
void gram_mfma_f32(const float* A, long long lda, int rows, int cols, bool atA, float* C, long long ldc, hipStream_t st);   // syrk_mfma.hip
void gram_mfma_f64(const double* A, long long lda, int rows, int cols, bool atA, double* C, long long ldc, hipStream_t st);   // gemm_f64_mfma.hip

template <typename T>
void gram_full(const T* A, long long lda, int rows, int cols, bool atA, T* C, long long ldc, hipStream_t st) {
    const TraceRange trace_range("admm:gram");
    // hand-written matrix-core kernels (fp32: split-K when the triangle has few tiles; fp64) for every size, so that no
    // BLAS handle is ever created in a default run (rocBLAS handle creation alone costs 0.1-0.2 s per process);
    // ADMM_HIP_GRAM=rocblas forces the library path (A/B tests)
    {
        const char* e = option("GRAM");
        const bool force_lib = e && std::string(e) == "rocblas";
        if (!force_lib) {
            if constexpr (std::is_same<T, float>::value) gram_mfma_f32(A, lda, rows, cols, atA, C, ldc, st);
            else gram_mfma_f64(A, lda, rows, cols, atA, C, ldc, st);
            return;
        }
    }
    rocblas_handle h = blas(st);
    const T one = T(1), zero = T(0);
    const int nC = atA ? cols : rows;
    const int kk = atA ? rows : cols;
    const rocblas_operation op = atA ? rocblas_operation_transpose : rocblas_operation_none;
    if constexpr (std::is_same<T, float>::value) {
        ADMM_BLAS_CHECK(blas_api().ssyrk(h, rocblas_fill_lower, op, nC, kk, &one, A, (rocblas_int)lda, &zero, C, (rocblas_int)ldc));
    } else {
        ADMM_BLAS_CHECK(blas_api().dsyrk(h, rocblas_fill_lower, op, nC, kk, &one, A, (rocblas_int)lda, &zero, C, (rocblas_int)ldc));
    }
    symmetrize_from_lower<T>(C, ldc, nC, st);
}
template void gram_full<float>(const float*, long long, int, int, bool, float*, long long, hipStream_t);
template void gram_full<double>(const double*, long long, int, int, bool, double*, long long, hipStream_t);

static void check_info(const DevBuf<rocblas_int>& info, hipStream_t st, const char* what) {
    rocblas_int h = 0;
    ADMM_HIP_CHECK(hipMemcpyAsync(&h, info.get(), sizeof(h), hipMemcpyDeviceToHost, st));
    comm_stream_sync(st);
    if (h != 0) throw Error(ADMM_ERR_NOT_SPD, std::string(what) + ": matrix is not positive definite (info=" + std::to_string(h) + ")");
}

template <typename T>
void cholesky_lower(T* A, long long lda, int n, hipStream_t st) {
    rocblas_handle h = blas(st);
    DevBuf<rocblas_int> info(1);
    if constexpr (std::is_same<T, float>::value) {
        ADMM_BLAS_CHECK(blas_api().spotrf(h, rocblas_fill_lower, n, A, (rocblas_int)lda, info.get()));
    } else {
        ADMM_BLAS_CHECK(blas_api().dpotrf(h, rocblas_fill_lower, n, A, (rocblas_int)lda, info.get()));
    }
    check_info(info, st, "Cholesky");
}
template void cholesky_lower<float>(float*, long long, int, hipStream_t);
template void cholesky_lower<double>(double*, long long, int, hipStream_t);

template <typename T>
__global__ void set_diag_one_kernel(T* X, long long ldx, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) X[(size_t)i * ldx + i] = T(1);
}

// A(i, j) = A(j, i) = X(i, j) for i >= j
template <typename T>
__global__ void __launch_bounds__(256) mirror_lower_kernel(const T* __restrict__ X, long long ldx, T* __restrict__ A, long long lda, int n) {
    const int j = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && i >= j) {
        const T v = X[(size_t)j * ldx + i];
        A[(size_t)j * lda + i] = v;
        A[(size_t)i * lda + j] = v;
    }
}

// Symmetric inverse of an SPD matrix: potrf, then X = L^-T (L^-1 I) by two triangular solves.
// (rocSOLVER's potri is not used: on this ROCm its small-size path returned a NaN in the last diagonal
//  element when the handle's workspace had been used by an fp64 call before -- tests/tools/fuzz_parity.py.)
template <typename T>
void spd_inverse_full(T* A, long long lda, int n, hipStream_t st) {
    cholesky_lower<T>(A, lda, n, st);
    rocblas_handle h = blas(st);
    DevBuf<T> X((size_t)n * n);
    X.zero(st);
    hipLaunchKernelGGL((set_diag_one_kernel<T>), dim3((n + 255) / 256), dim3(256), 0, st, X.get(), (long long)n, n);
    const T one = T(1);
    if constexpr (std::is_same<T, float>::value) {
        ADMM_BLAS_CHECK(blas_api().strsm(h, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit,
                                      n, n, &one, A, (rocblas_int)lda, X.get(), n));
        ADMM_BLAS_CHECK(blas_api().strsm(h, rocblas_side_left, rocblas_fill_lower, rocblas_operation_transpose, rocblas_diagonal_non_unit,
                                      n, n, &one, A, (rocblas_int)lda, X.get(), n));
    } else {
        ADMM_BLAS_CHECK(blas_api().dtrsm(h, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit,
                                      n, n, &one, A, (rocblas_int)lda, X.get(), n));
        ADMM_BLAS_CHECK(blas_api().dtrsm(h, rocblas_side_left, rocblas_fill_lower, rocblas_operation_transpose, rocblas_diagonal_non_unit,
                                      n, n, &one, A, (rocblas_int)lda, X.get(), n));
    }
    hipLaunchKernelGGL((mirror_lower_kernel<T>), dim3((n + 255) / 256, n), dim3(256), 0, st, X.get(), (long long)n, A, lda, n);
    ADMM_HIP_CHECK(hipGetLastError());
    comm_stream_sync(st);        // X is released on return
}
template void spd_inverse_full<float>(float*, long long, int, hipStream_t);

// dst (double) = src (float) with `diag` added on the diagonal (in float, like the reference); and the way back (one rounding per entry).
__global__ void __launch_bounds__(256) widen_add_diag_kernel(const float* __restrict__ src, double* __restrict__ dst, long long ld, int n, double diag) {
    const int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (i < ld) {
        double v = 0.0;
        if (i < n && j < n) {
            const float a = src[(size_t)j * ld + i];
            v = (i == j) ? (double)__fadd_rn(a, (float)diag) : (double)a;      // XX.diagonal().array() += rho is a float addition (ADMMLassoTall.h:204)
        }
        dst[(size_t)j * ld + i] = v;
    }
}
__global__ void __launch_bounds__(256) narrow_kernel(const double* __restrict__ src, float* __restrict__ dst, long long ld, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (i < ld) dst[(size_t)j * ld + i] = (i < n && j < n) ? (float)src[(size_t)j * ld + i] : 0.f;
}

// A (float, SPD once `diag` is added, both triangles valid, whole 128-blocks as for spd_inverse_f32) -> (A + diag I)^-1:
// the Cholesky factorisation and the inverse are done in DOUBLE on the fp64 matrix cores and the result is rounded to
// float ONCE, so that each entry of the cached inverse carries half an ulp of error instead of the accumulated
// rounding of an fp32 factorisation (lasso_tall.hip: fewer stopping-rule / restart flips against a Cholesky solve).
void spd_inverse_f32_via_f64(float* A, long long lda, int n, double diag, hipStream_t st) {
    const TraceRange trace_range("admm:factor+inverse (f64)");
    const int pp = round_up(n, 128);
    ADMM_REQUIRE(lda >= pp, "spd_inverse_f32_via_f64: leading dimension must cover whole 128-row blocks");
    DevBuf<double> D((size_t)lda * pp);
    hipLaunchKernelGGL(widen_add_diag_kernel, dim3((unsigned)((lda + 255) / 256), pp), dim3(256), 0, st, A, D.get(), lda, n, diag);
    spd_inverse_f64(D.get(), lda, n, st);
    hipLaunchKernelGGL(narrow_kernel, dim3((unsigned)((lda + 255) / 256), pp), dim3(256), 0, st, D.get(), A, lda, n);
    ADMM_HIP_CHECK(hipGetLastError());
    comm_stream_sync(st);
}

void spd_inverse_f32(float* A, long long lda, int n, hipStream_t st) {
    const TraceRange trace_range("admm:factor+inverse (f32)");
    const char* e = option("FACTOR");
    if ((e && std::string(e) == "rocsolver") || lda < round_up(n, 128)) spd_inverse_full<float>(A, lda, n, st);
    else spd_inverse_mfma_f32(A, lda, n, st);
}

void spd_inverse_f64(double* A, long long lda, int n, hipStream_t st) {
    const TraceRange trace_range("admm:factor+inverse (f64)");
    const char* e = option("FACTOR");
    if ((e && std::string(e) == "rocsolver") || lda < round_up(n, 128)) spd_inverse_full<double>(A, lda, n, st);
    else spd_inverse_mfma_f64(A, lda, n, st);
}
template void spd_inverse_full<double>(double*, long long, int, hipStream_t);

template <typename T>
void trsm_left_lower(const T* L, long long ldl, int n, T* B, long long ldb, int m, hipStream_t st) {
    rocblas_handle h = blas(st);
    const T one = T(1);
    if constexpr (std::is_same<T, float>::value) {
        ADMM_BLAS_CHECK(blas_api().strsm(h, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit,
                                      n, m, &one, L, (rocblas_int)ldl, B, (rocblas_int)ldb));
    } else {
        ADMM_BLAS_CHECK(blas_api().dtrsm(h, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit,
                                      n, m, &one, L, (rocblas_int)ldl, B, (rocblas_int)ldb));
    }
}
template void trsm_left_lower<float>(const float*, long long, int, float*, long long, int, hipStream_t);
template void trsm_left_lower<double>(const double*, long long, int, double*, long long, int, hipStream_t);

template <typename T>
void trsm_right_lower_t(const T* L, long long ldl, int n, T* B, long long ldb, int m, hipStream_t st) {
    rocblas_handle h = blas(st);
    const T one = T(1);
    if constexpr (std::is_same<T, float>::value) {
        ADMM_BLAS_CHECK(blas_api().strsm(h, rocblas_side_right, rocblas_fill_lower, rocblas_operation_transpose, rocblas_diagonal_non_unit,
                                      m, n, &one, L, (rocblas_int)ldl, B, (rocblas_int)ldb));
    } else {
        ADMM_BLAS_CHECK(blas_api().dtrsm(h, rocblas_side_right, rocblas_fill_lower, rocblas_operation_transpose, rocblas_diagonal_non_unit,
                                      m, n, &one, L, (rocblas_int)ldl, B, (rocblas_int)ldb));
    }
}
template void trsm_right_lower_t<float>(const float*, long long, int, float*, long long, int, hipStream_t);
template void trsm_right_lower_t<double>(const double*, long long, int, double*, long long, int, hipStream_t);

// ------------------------------------------------------------------ reductions / simple gemv
template <typename T>
__global__ void __launch_bounds__(1024)
absmax_kernel(const T* v, int n, T* out) {
    __shared__ T sm[16];
    T m = T(0);
    for (int i = threadIdx.x; i < n; i += 1024) { T a = fabs(v[i]); m = a > m ? a : m; }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        T r = sm[0];
        for (int w = 1; w < 16; ++w) r = sm[w] > r ? sm[w] : r;
        out[0] = r;
    }
}
template <typename T>
T device_absmax(const T* v, int n, hipStream_t st) {
    DevBuf<T> out(1);
    hipLaunchKernelGGL((absmax_kernel<T>), dim3(1), dim3(1024), 0, st, v, n, out.get());
    T h;
    ADMM_HIP_CHECK(hipMemcpyAsync(&h, out.get(), sizeof(T), hipMemcpyDeviceToHost, st));
    comm_stream_sync(st);
    return h;
}
template float device_absmax<float>(const float*, int, hipStream_t);
template double device_absmax<double>(const double*, int, hipStream_t);

template <typename T>
void gemv_t_simple(const T* A, long long lda, int m, int k, const T* v, T* y, hipStream_t st) {
    GemvTPlan pl = plan_gemv_t<T>(m, k, 1, 4);
    const long long stride = round_up(k, 32);
    DevBuf<T> part((size_t)pl.nseg * stride);
    launch_gemv_t<T, 1, 4>(pl, A, lda, m, k, v, nullptr, part.get(), nullptr, stride, nullptr, st);
    hipLaunchKernelGGL((reduce_partials_kernel<T>), dim3((k + 255) / 256), dim3(256), 0, st, part.get(), stride, pl.nseg, k, y);
    comm_stream_sync(st);   // part is freed on return
}
template void gemv_t_simple<float>(const float*, long long, int, int, const float*, float*, hipStream_t);
template void gemv_t_simple<double>(const double*, long long, int, int, const double*, double*, hipStream_t);

template <typename T>
SymMatVec<T>::SymMatVec(const T* A_, long long lda_, int n_, hipStream_t st_) : A(A_), lda(lda_), n(n_), st(st_) {
    dv.alloc(round_up(n, 32));
    dw.alloc(round_up(n, 32));
    dv.zero(st);
    pl = plan_gemv_t<T>(n, n, 1, 4);
    stride = round_up(n, 32);
    part.alloc((size_t)pl.nseg * stride);               // allocated once: no hipMalloc per mat-vec
}
template <typename T>
void SymMatVec<T>::operator()(const T* v_host, T* w_host) {
    ADMM_HIP_CHECK(hipMemcpyAsync(dv.get(), v_host, (size_t)n * sizeof(T), hipMemcpyHostToDevice, st));
    launch_gemv_t<T, 1, 4>(pl, A, lda, n, n, dv.get(), nullptr, part.get(), nullptr, stride, nullptr, st);
    hipLaunchKernelGGL((reduce_partials_kernel<T>), dim3((n + 255) / 256), dim3(256), 0, st, part.get(), stride, pl.nseg, n, dw.get(), (const int*)nullptr);
    read_back(w_host, dw.get(), (size_t)n * sizeof(T), st);
}
template struct SymMatVec<float>;
template struct SymMatVec<double>;

// ---- w = X s: every workgroup takes `cols` consecutive columns and a tile of 1024 rows (one float4 per lane and wave),
// streams its columns once (1 KiB per wave instruction) and writes one partial row; the partial rows are summed in
// workgroup order afterwards (deterministic, no atomics).
__global__ void __launch_bounds__(256)
gemv_n_partial_kernel(const float* __restrict__ X, long long ldx, int n, int p, const float* __restrict__ s, int cols, float* __restrict__ part, long long ldpart) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int row = blockIdx.y * 1024 + wid * 256 + lane * 4;
    const int j0 = blockIdx.x * cols, j1 = min(p, j0 + cols);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < n) {                                        // ldx is padded to 32 rows and the padding is zero: the float4 stays in bounds
        const float* col = X + (size_t)j0 * ldx + row;
        int j = j0;
        for (; j + 4 <= j1; j += 4) {
            const float4 a0 = *reinterpret_cast<const float4*>(col), a1 = *reinterpret_cast<const float4*>(col + ldx);
            const float4 a2 = *reinterpret_cast<const float4*>(col + 2 * ldx), a3 = *reinterpret_cast<const float4*>(col + 3 * ldx);
            const float s0 = s[j], s1 = s[j + 1], s2 = s[j + 2], s3 = s[j + 3];
            acc.x = fmaf(s0, a0.x, acc.x); acc.y = fmaf(s0, a0.y, acc.y); acc.z = fmaf(s0, a0.z, acc.z); acc.w = fmaf(s0, a0.w, acc.w);
            acc.x = fmaf(s1, a1.x, acc.x); acc.y = fmaf(s1, a1.y, acc.y); acc.z = fmaf(s1, a1.z, acc.z); acc.w = fmaf(s1, a1.w, acc.w);
            acc.x = fmaf(s2, a2.x, acc.x); acc.y = fmaf(s2, a2.y, acc.y); acc.z = fmaf(s2, a2.z, acc.z); acc.w = fmaf(s2, a2.w, acc.w);
            acc.x = fmaf(s3, a3.x, acc.x); acc.y = fmaf(s3, a3.y, acc.y); acc.z = fmaf(s3, a3.z, acc.z); acc.w = fmaf(s3, a3.w, acc.w);
            col += 4 * ldx;
        }
        for (; j < j1; ++j) {
            const float4 a0 = *reinterpret_cast<const float4*>(col);
            const float s0 = s[j];
            acc.x = fmaf(s0, a0.x, acc.x); acc.y = fmaf(s0, a0.y, acc.y); acc.z = fmaf(s0, a0.z, acc.z); acc.w = fmaf(s0, a0.w, acc.w);
            col += ldx;
        }
        *reinterpret_cast<float4*>(part + (size_t)blockIdx.x * ldpart + row) = acc;
    }
}

GramFreeWideOp::GramFreeWideOp(const float* X_, long long ldx_, int n_, int p_, hipStream_t st_) : X(X_), ldx(ldx_), n(n_), p(p_), st(st_) {
    dv.alloc(round_up(n, 32)); dw.alloc(round_up(n, 32)); ds.alloc(round_up(p, 32));
    dv.zero(st); ds.zero(st);
    nchunk = (p + cols_per_wg - 1) / cols_per_wg;
    ldpart = round_up(n, 1024);
    part.alloc((size_t)nchunk * ldpart); part.zero(st);
}
void GramFreeWideOp::operator()(const float* v_host, float* w_host) {
    ADMM_HIP_CHECK(hipMemcpyAsync(dv.get(), v_host, (size_t)n * sizeof(float), hipMemcpyHostToDevice, st));
    gemv_t_simple<float>(X, ldx, n, p, dv.get(), ds.get(), st);                                             // s = X' v
    hipLaunchKernelGGL(gemv_n_partial_kernel, dim3(nchunk, (n + 1023) / 1024), dim3(256), 0, st, X, ldx, n, p, ds.get(), cols_per_wg, part.get(), ldpart);
    hipLaunchKernelGGL((reduce_partials_kernel<float>), dim3((n + 255) / 256), dim3(256), 0, st, part.get(), ldpart, nchunk, n, dw.get(), (const int*)nullptr);
    read_back(w_host, dw.get(), (size_t)n * sizeof(float), st);
}

}  // namespace admm
