// Internal helpers shared by the translation units of libadmm_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <string>
#include <utility>
#include <vector>
#include <stdexcept>
#include <chrono>

#include "../../include/admm_hip.h"

namespace admm {

// Error carried through the library as an exception and converted to a code at the C boundary.
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

void set_last_error(const std::string& m);

#define ADMM_HIP_CHECK(expr)                                                                       \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            throw ::admm::Error(ADMM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e) +  \
                                                  " (" __FILE__ ":" + std::to_string(__LINE__) + ")"); \
    } while (0)

#define ADMM_REQUIRE(cond, msg)                                             \
    do {                                                                    \
        if (!(cond)) throw ::admm::Error(ADMM_ERR_INVALID_ARG, (msg));      \
    } while (0)

inline double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Variant selectors / tuning values of the library (api.hip).  option("GRAM_SPLIT") returns the value in force for the CALLING THREAD, or
// nullptr for "library default": first what the thread set through the C ABI (admm_hip_options_set / admm_hip_option_set -- what
// admm_amd/api.py's options() and an R caller use), then the process-wide overlay of ADMM_HIP_<NAME> environment variables, which is
// captured ONCE when the library is first used (a debugging aid: nothing reads the environment per call, and two threads can run two
// different variants at the same time).  The pointer stays valid until the thread changes its options.
const char* option(const char* name);
int option_int(const char* name, int dflt);
void option_set_thread(const char* name, const char* value);      // value nullptr: back to the default / overlay
void options_reset_thread();

// roctx range around a phase of a call (setup: convert + standardise, Gram, Lanczos, factorisation + inverse; the ADMM loop; the
// read-back): shows as a named span in rocprofv3 --marker-trace / omnitrace next to the kernels.  libroctx64 is
// dlopen'ed on first use; without it the ranges cost one branch.
struct TraceRange {
    explicit TraceRange(const char* name);
    ~TraceRange();
    TraceRange(const TraceRange&) = delete;
    TraceRange& operator=(const TraceRange&) = delete;
    bool on;
};

// Cache of large device blocks (api.hip).  hipMalloc / hipFree of multi-GB buffers are synchronous page-table operations whose cost
// varies by box and by what the process freed before (measured on C2, second plan creation of a process: 0.07 s of kernels inside
// 0.07 .. 0.31 s of wall, the difference all in hipFree / hipMalloc of the 4 GB operands) -- a resident server or an R session that
// calls $fit() repeatedly should not pay it per call.  Blocks of at least 32 MB that a DevBuf releases are kept (per device, at most
// ADMM_HIP_POOL_MB megabytes in all, default 24576; 0 switches the cache off) and handed to the next allocation they fit (size <=
// block <= 1.25 size).  A failed hipMalloc empties the cache and tries again; pool_cached_bytes() is what a caller adds to
// hipMemGetInfo's free figure; admm_hip_trim_memory() returns everything to the driver.  Thread-safe (one mutex).
void* pool_alloc(size_t bytes, size_t* granted);
void pool_free(void* p, size_t granted);
size_t pool_cached_bytes();
void pool_trim();

// Owning device allocation.
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    size_t granted = 0;          // bytes of the block behind p (>= n * sizeof(T) when it came from the cache)
    DevBuf() = default;
    explicit DevBuf(size_t count) { alloc(count); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n), granted(o.granted) { o.p = nullptr; o.n = 0; o.granted = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; granted = o.granted; o.p = nullptr; o.n = 0; o.granted = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void alloc(size_t count) {
        release();
        n = count;
        if (count) p = static_cast<T*>(pool_alloc(count * sizeof(T), &granted));
    }
    void release() {
        if (p) pool_free(p, granted);
        p = nullptr; n = 0; granted = 0;
    }
    void zero(hipStream_t s) { if (n) ADMM_HIP_CHECK(hipMemsetAsync(p, 0, n * sizeof(T), s)); }
    T* get() const { return p; }
};

// A non-blocking stream borrowed from a per-thread, per-device pool: creating a HIP stream costs 4-20 ms and
// destroying one 3 ms on this runtime (measured), more than a whole small solve, so streams are created once per
// thread and device and handed back idle (the borrower synchronises before it returns one).
struct Stream {
    hipStream_t s = nullptr;
    int dev = 0;
    static std::vector<std::pair<int, hipStream_t>>& pool() {
        static thread_local std::vector<std::pair<int, hipStream_t>> p;      // never destroyed: the runtime may be gone at thread exit
        return p;
    }
    Stream() {
        ADMM_HIP_CHECK(hipGetDevice(&dev));
        auto& p = pool();
        for (size_t i = 0; i < p.size(); ++i)
            if (p[i].first == dev) { s = p[i].second; p.erase(p.begin() + (long)i); return; }
        ADMM_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    }
    ~Stream() {
        if (!s) return;
        if (hipStreamSynchronize(s) == hipSuccess) pool().push_back({dev, s});
        else (void)hipStreamDestroy(s);                  // a stream that saw an error is not reused
    }
    Stream(const Stream&) = delete;
    Stream& operator=(const Stream&) = delete;
    void sync() const { ADMM_HIP_CHECK(hipStreamSynchronize(s)); }
};

struct Event {
    hipEvent_t e = nullptr;
    Event() { ADMM_HIP_CHECK(hipEventCreate(&e)); }
    ~Event() { if (e) (void)hipEventDestroy(e); }
    Event(const Event&) = delete;
    Event& operator=(const Event&) = delete;
};

// Device-to-host read-back of results, traces and iterate dumps through a pinned bounce buffer of our own (per thread,
// never freed -- like the stream pool).  Why not hipMemcpy into the caller's pageable memory: under heavy host
// oversubscription (16 solver processes + their CPU checkers on one box) the runtime's pageable path was caught returning a
// few stale 64-byte lines at a fixed offset (~128 KB) of a large copy while the device buffer was intact -- a second read of
// the same device memory gave the right bytes (profiles/r04_transient_stale_lines.md).  A DMA into pinned memory followed by
// a host memcpy does not go through that staging path.  Copies below kDirectBytes stay on plain hipMemcpy.
// `st` must be the stream the producing kernels ran on (or nullptr after a device-wide sync); the call returns with the
// bytes in dst.
struct PinnedBounce {
    static constexpr size_t kChunk = size_t(4) << 20;
    static constexpr size_t kDirectBytes = size_t(32) << 10;
    static void* buffer() {
        static thread_local void* buf = nullptr;
        if (!buf) ADMM_HIP_CHECK(hipHostMalloc(&buf, kChunk, hipHostMallocDefault));
        return buf;
    }
};
inline void read_back(void* dst, const void* src, size_t bytes, hipStream_t st) {
    if (!bytes) return;
    if (bytes <= PinnedBounce::kDirectBytes) {
        if (st) ADMM_HIP_CHECK(hipStreamSynchronize(st));
        ADMM_HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
        return;
    }
    char* bounce = static_cast<char*>(PinnedBounce::buffer());
    for (size_t off = 0; off < bytes; off += PinnedBounce::kChunk) {
        const size_t n = bytes - off < PinnedBounce::kChunk ? bytes - off : PinnedBounce::kChunk;
        ADMM_HIP_CHECK(hipMemcpyAsync(bounce, static_cast<const char*>(src) + off, n, hipMemcpyDeviceToHost, st));
        ADMM_HIP_CHECK(hipStreamSynchronize(st));
        std::memcpy(static_cast<char*>(dst) + off, bounce, n);
    }
}

// Host-to-device transfer of the CALLER's data (x, y: pageable memory borrowed from R) through pinned staging of our own, the
// mirror image of read_back(): worker threads copy a piece into a pinned slot, the calling thread DMAs it from there while the
// next piece is being copied (prep.hip).  The runtime's pageable path -- a staging copy inside hipMemcpy -- is what returned stale
// lines in the D2H direction under host oversubscription (profiles/r04_transient_stale_lines.md); a corrupted INPUT line would be
// silent, so large inputs do not go through it at all.  Returns with the bytes in device memory.
// ADMM_HIP_H2D=pageable: plain hipMemcpy (A/B).
void write_device(void* dst, const void* src, size_t bytes);

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
inline size_t round_up_sz(size_t v, size_t m) { return (v + m - 1) / m * m; }

struct DeviceInfo {
    int num_cu = 256;
    size_t lds_per_block = 64 * 1024;    // dynamic LDS a launch may request without opting in
    size_t lds_optin = 64 * 1024;        // ... after hipFuncSetAttribute(hipFuncAttributeMaxDynamicSharedMemorySize) (160 KB on gfx950)
};
const DeviceInfo& device_info();   // queries the current device once per device id
// Workgroups of a kernel the device holds resident at once, given the occupancy query's answer per CU.  Launches whose
// workgroups WAIT for one another (the single-launch PEER exchange of the sharded solvers) are only chosen when their whole
// grid fits into half of this.  ADMM_HIP_TEST_RESIDENT_WGS overrides it (test hook: pretends the device holds fewer, so that
// the condition is violated on purpose and the fallback -- separate producer and consumer launches -- must be taken).
long long resident_workgroups(int occupancy_per_cu);
void require_device();             // throws ADMM_ERR_NO_DEVICE when no usable HIP device

}  // namespace admm
