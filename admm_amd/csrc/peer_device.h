// Device side of the PEER exchange (comm.hip): a kernel of a solver can be the producer (write its values straight into
// every rank's slot and raise the flags) or the consumer (wait for the K flags, read the K slots) of one exchange, so
// that the exchange costs no launches of its own.  Protocol and buffer layout: comm.hip, "PEER backend".
//
// Memory ordering without cache write-backs: payload stores are RELAXED SYSTEM-scope atomic stores (written through to
// the destination's memory, never parked in this device's L2); a workgroup-scope release fence then only waits until
// they are acknowledged (s_waitcnt), and the flag store that follows is a relaxed system-scope store as well.  (A
// system-scope release FENCE instead writes back every dirty L2 line of the XCD -- the symmetric mat-vec's partial
// arrays -- and cost 5 us per exchange.)  The consumer's flag load is an ACQUIRE at system scope, executed by every wave.
#pragma once
#include <hip/hip_runtime.h>

namespace admm {

struct PeerExchange {
    unsigned char* const* remote;       // every rank's exchange buffer as mapped here (device array of nranks pointers)
    unsigned char* local;               // this rank's buffer
    size_t slot, flags_off;
    int nranks, rank;
    unsigned long long seq;             // number of this exchange (1, 2, ...): parity selects the slot set
    unsigned int* count;                // [nranks + 1] arrival counters of the producing launch (device memory)
    int* err;                           // device word: non-zero after a timed-out wait
    long long timeout_ticks;            // wall_clock64 ticks (100 MHz)
};

__device__ __forceinline__ unsigned long long* peer_flag(unsigned char* buf, size_t flags_off, unsigned long long seq, int nranks, int src) {
    return reinterpret_cast<unsigned long long*>(buf + flags_off + ((size_t)(seq & 1) * nranks + src) * 64);
}
// this rank's slot in rank q's buffer (producer) / rank r's slot in this rank's buffer (consumer)
__device__ __forceinline__ unsigned char* peer_dst_slot(const PeerExchange& e, int q) {
    return e.remote[q] + ((size_t)(e.seq & 1) * e.nranks + e.rank) * e.slot;
}
__device__ __forceinline__ const unsigned char* peer_src_slot(const PeerExchange& e, int r) {
    return e.local + ((size_t)(e.seq & 1) * e.nranks + r) * e.slot;
}
__device__ __forceinline__ void peer_store_f32(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned int*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void peer_store_u64(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Producer epilogue, called by EVERY thread of EVERY workgroup of the producing launch after its payload stores
// (peer_store_*): the last workgroup to arrive raises this rank's flag in every rank's buffer.  `nwg` = gridDim of the
// producing launch (all of whose workgroups call this exactly once).
__device__ __forceinline__ void peer_publish(const PeerExchange& e, unsigned int nwg) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");       // this wave's write-through stores are acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        // relaxed: the payload is already at its destination (write-through stores, acknowledged above); an acq_rel here
        // would write back the XCD's L2 once per workgroup
        const unsigned int prev = __hip_atomic_fetch_add(&e.count[e.nranks], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == nwg - 1) {
            __hip_atomic_store(&e.count[e.nranks], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int q = 0; q < e.nranks; ++q)
                peer_store_u64(peer_flag(e.remote[q], e.flags_off, e.seq, e.nranks, e.rank), e.seq);
        }
    }
}

// Cache-free variant for consumers with many workgroups: the flags are polled with RELAXED system-scope loads (no cache
// invalidation per wave -- an acquire in each of 1250 waves cost the tall tail 17 us) and the slot data is then read
// with peer_load_* (system-scope loads that bypass this device's caches, so no stale line of an earlier exchange can be
// returned).  The loads are issued after the polling loop has exited (in-order issue per wave, no speculation).
__device__ __forceinline__ bool peer_wait_relaxed(const PeerExchange& e) {
    const int lane = threadIdx.x & 63;
    bool ok = true;
    for (int r = lane; r < e.nranks; r += 64) {
        unsigned long long* f = peer_flag(e.local, e.flags_off, e.seq, e.nranks, r);
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < e.seq) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > e.timeout_ticks) {
                __hip_atomic_store(e.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = false;
                break;
            }
        }
    }
    return __all(ok) != 0;
}
__device__ __forceinline__ float2 peer_load_f32x2(const float* p) {
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return make_float2(__uint_as_float((unsigned int)v), __uint_as_float((unsigned int)(v >> 32)));
}
__device__ __forceinline__ void peer_store_f32x2(float* p, float x, float y) {
    peer_store_u64(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__float_as_uint(x) | ((unsigned long long)__float_as_uint(y) << 32));
}

// ---- AUX region: exchanges INSIDE one launch (the persistent active-set stretch of the column-sharded wide solver).
// The slots above are numbered by the host when it enqueues an exchange; a kernel that exchanges once per ADMM iteration for as many
// iterations as the decisions allow cannot be numbered that way.  A second, small region of every rank's exchange buffer therefore
// carries its own sequence: a DEVICE word (PeerAux::seq, replicated -- every rank performs the same exchanges because every rank
// takes the same decisions) read by the launch when it starts and advanced by its leader when it ends.  Exchange e uses parity e & 1:
//     data [2][nranks][kAuxRows floats | one 64-byte line for the agreement word]     flags [2][nranks][kAuxGroups + 1] lines
// The payload is flagged per GROUP of 256 floats (a row group of the stretch proceeds as soon as ITS rows have arrived from every
// rank).  Reuse of a parity is safe for the same reason as above, per group: rank A writes group r of e + 2 only after it consumed
// group r of e + 1, which rank B published only after all of ITS readers of group r of e had moved on (lasso_wide.hip).
constexpr int kAuxRows = 8192;
constexpr int kAuxGroups = 32;
constexpr int kAuxEntry = kAuxGroups;      // flag index of the once-per-launch agreement word
constexpr size_t kAuxSlotBytes = (size_t)kAuxRows * sizeof(float) + 64;
struct PeerAux {
    unsigned char* const* remote; unsigned char* local;
    size_t data_off, flags_off;
    int nranks, rank;
    unsigned long long* seq;            // device word: exchanges made through the region so far
    int* err; long long timeout_ticks;
};
__device__ __forceinline__ float* aux_slot(unsigned char* buf, const PeerAux& a, unsigned long long e, int src) {
    return reinterpret_cast<float*>(buf + a.data_off + ((size_t)(e & 1) * a.nranks + src) * kAuxSlotBytes);
}
__device__ __forceinline__ unsigned long long* aux_flag(unsigned char* buf, const PeerAux& a, unsigned long long e, int src, int idx) {
    return reinterpret_cast<unsigned long long*>(buf + a.flags_off + (((size_t)(e & 1) * a.nranks + src) * (kAuxGroups + 1) + idx) * 64);
}
// every wave that needs group `idx` of exchange e calls this: lanes < nranks poll one rank's flag each (bounded)
__device__ __forceinline__ bool aux_wait(const PeerAux& a, unsigned long long e, int idx) {
    const int lane = threadIdx.x & 63;
    bool ok = true;
    for (int r = lane; r < a.nranks; r += 64) {
        unsigned long long* f = aux_flag(a.local, a, e, r, idx);
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < e) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > a.timeout_ticks) { __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = false; break; }
        }
    }
    return __all(ok) != 0;
}
__device__ __forceinline__ float peer_load_f32(const float* p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned int*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
}
__device__ __forceinline__ unsigned long long peer_load_u64(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Consumer prologue, called by every thread of the consuming launch before it reads the slots: lanes < nranks of every
// wave spin (bounded) until the flag of their rank reads seq.  Returns false when the exchange failed (time limit).
__device__ __forceinline__ bool peer_wait(const PeerExchange& e) {
    const int lane = threadIdx.x & 63;
    if (*reinterpret_cast<volatile int*>(e.err)) return false;
    for (int r = lane; r < e.nranks; r += 64) {
        unsigned long long* f = peer_flag(e.local, e.flags_off, e.seq, e.nranks, r);
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < e.seq) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > e.timeout_ticks || *reinterpret_cast<volatile int*>(e.err)) {
                *reinterpret_cast<volatile int*>(e.err) = 1;
                break;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return *reinterpret_cast<volatile int*>(e.err) == 0;
}

}  // namespace admm
