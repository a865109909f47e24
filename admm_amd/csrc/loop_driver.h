// Host side of the device-resident ADMM loops: enqueue iterations in batches without waiting for
// the device, poll a sticky `done` word through pinned memory, always keeping one batch in flight.
#pragma once
#include "admm_internal.h"
#include "comm.h"

namespace admm {

struct LoopTimes {
    double wall_s = 0;
    double events_ms = 0;
    long long launched = 0;
};

// One int in pinned host memory that the deciding workgroup of a solver sets (one store over PCIe, once per solve) when
// the solve has finished: the host then needs no device-to-host copy per poll, only the batch's event.
struct PinnedFlag {
    int* p = nullptr;
    PinnedFlag() { ADMM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&p), sizeof(int), hipHostMallocDefault)); *p = 0; }
    ~PinnedFlag() { if (p) (void)hipHostFree(p); }
    PinnedFlag(const PinnedFlag&) = delete;
    PinnedFlag& operator=(const PinnedFlag&) = delete;
};

// enqueue(g): enqueue every kernel of iteration g (g = 0, 1, 2, ...) on `st`.
// d_done: device int that the iteration kernels set to non-zero once the solve is finished; all
// kernels must be no-ops afterwards.  `batch` iterations are enqueued between two polls.
// h_flag (optional, SINGLE-RANK solves only: when a flag store becomes visible to the host is not a function of the stream
// position, and ranks that exchange data per iteration must all stop after the same number of enqueued iterations):
// a PinnedFlag word the kernels set together with d_done -- then no copy is enqueued per poll (a
// 4-byte device-to-host copy is a blit launch plus a system-scope release: ~14 us of stream time per batch, measured on
// the wide path), and the batch grows from `batch` to 4 x `batch` as the solve gets long.
template <typename F>
inline LoopTimes run_until_done(hipStream_t st, const int* d_done, int batch, long long max_iters, F&& enqueue,
                                const volatile int* h_flag = nullptr, long long g_start = 0) {
    const TraceRange trace_range("admm:loop");
    int* h_done = nullptr;
    ADMM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h_done), 2 * sizeof(int), hipHostMallocDefault));
    struct HostFree { void* p; ~HostFree() { (void)hipHostFree(p); } } hf{h_done};
    h_done[0] = h_done[1] = 0;
    const CommLockstep lockstep;                       // exchanges enqueued from here are per-iteration: short wait bound
    const int batch0 = batch;
    int npoll = 0;
    Event ev0, ev1, poll[2];
    LoopTimes t;
    ADMM_HIP_CHECK(hipStreamSynchronize(st));
    const double t0 = now_s();
    ADMM_HIP_CHECK(hipEventRecord(ev0.e, st));
    long long g = g_start;                             // (a solver that resumes a halted loop continues its own count)
    auto enqueue_batch = [&](int slot) {
        for (int k = 0; k < batch; ++k, ++g) enqueue(g);
        if (h_flag == nullptr) ADMM_HIP_CHECK(hipMemcpyAsync(&h_done[slot], d_done, sizeof(int), hipMemcpyDeviceToHost, st));
        ADMM_HIP_CHECK(hipEventRecord(poll[slot].e, st));
    };
    int slot = 0;
    enqueue_batch(slot);
    ADMM_HIP_CHECK(hipGetLastError());                 // a failed launch (e.g. too much LDS requested) surfaces here, not as a hang
    bool done = false;
    while (!done) {
        enqueue_batch(slot ^ 1);
        comm_event_sync(poll[slot].e);
        comm_check();                                  // a timed-out exchange ends the solve with ADMM_ERR_COMM
        done = h_flag ? (*h_flag != 0) : (h_done[slot] != 0);
        slot ^= 1;
        if (h_flag && (++npoll & 3) == 0 && batch < 4 * batch0) batch *= 2;      // stays even: the parity pattern of g is kept
        if (!done && g - g_start > max_iters + 2 * batch)
            throw Error(ADMM_ERR_INTERNAL, "ADMM loop: iteration bound exceeded without completion");
    }
    ADMM_HIP_CHECK(hipEventRecord(ev1.e, st));
    comm_stream_sync(st);
    t.wall_s = now_s() - t0;
    float ms = 0.f;
    ADMM_HIP_CHECK(hipEventElapsedTime(&ms, ev0.e, ev1.e));
    t.events_ms = ms;
    t.launched = g - g_start;
    return t;
}

}  // namespace admm
