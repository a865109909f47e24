// Exchange layer: RCCL, one-shot peer-mapped all-reduce (hipIpc), host shared memory.  See comm.h.
//
// Replaces the shared-memory "collectives" of the reference's OpenMP master/worker loop:
//   sum_i (x_i + y_i / rho)           /root/reference/src/PADMMLasso.h:99-108 (add_xu_to :65-68)
//   sum_i ||x_i||^2, ||y_i||^2, ||r_i||^2   /root/reference/src/PADMMBase.h:117-138,200-214
// The reference reads the workers' vectors directly (threads of one process); here each rank is a process on its own GPU.
//
// Bootstrap is caller-driven (the library never opens sockets): RCCL needs rank 0's 128-byte unique id on every rank,
// PEER needs every rank's 64-byte hipIpc handle on every rank, SHM needs a name all ranks agree on.  The Python harness
// moves those bytes with torch.distributed or a file; an R / MPI caller would use its own channel.
#include "comm.h"
#include "peer_device.h"
#include <rccl/rccl.h>
#include <atomic>
#include <cerrno>
#include <mutex>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace admm {
namespace {

constexpr int kMaxRanks = 64;
constexpr size_t kDefaultSlotBytes = 4u << 20;        // payload capacity of one exchange (longer messages are chunked)
// Every wait (host spin and device spin) is bounded: a missing rank yields ADMM_ERR_COMM, never a hung GPU.  Two bounds:
//   lock-step  the per-iteration exchanges of a solve (all ranks enqueue the same iterations at the same pace): a rank that
//              is 20 s late is dead.  ADMM_HIP_COMM_TIMEOUT_S.
//   patient    everything else -- setup reductions (ranks reach them after uploads / Grams of different sizes) and the
//              one join of the replica modes (cross-validation folds, several responses: 10 folds on 4 ranks is 3/3/2/2 whole
//              fits of imbalance BY DESIGN): RCCL would wait for ever there, so SHM / PEER wait long.
//              ADMM_HIP_COMM_PATIENT_TIMEOUT_S (default one hour).
// An exchange takes the bound in force when it is ENQUEUED (CommLockstep, comm.h).
double env_seconds(const char* name, double dflt) {
    if (const char* e = option(name)) { const double v = std::atof(e); if (v > 0.0) return v; }
    return dflt;
}
double wait_seconds_lockstep() { return env_seconds("COMM_TIMEOUT_S", 20.0); }
double wait_seconds_patient() { return env_seconds("COMM_PATIENT_TIMEOUT_S", 3600.0); }
std::atomic<int> g_lockstep{0};
double wait_seconds_now() { return g_lockstep.load(std::memory_order_relaxed) > 0 ? wait_seconds_lockstep() : wait_seconds_patient(); }

std::mutex g_mu;
CommInfo g_info;
ncclComm_t g_comm = nullptr;
size_t g_slot = kDefaultSlotBytes;
uint64_t g_seq = 0;                                   // exchanges enqueued so far (all ranks enqueue the same sequence)
int* g_err = nullptr;                                 // pinned host word: set by a timed-out wait of the SHM host function
int* g_derr = nullptr;                                // device word: set by a timed-out wait of a PEER kernel (a pinned host word
                                                      // would cost every workgroup a PCIe round trip: 30 us per exchange, measured)

#define ADMM_NCCL_CHECK(expr)                                                                     \
    do {                                                                                          \
        ncclResult_t _r = (expr);                                                                 \
        if (_r != ncclSuccess)                                                                    \
            throw ::admm::Error(ADMM_ERR_COMM, std::string(#expr) + ": " + ncclGetErrorString(_r)); \
    } while (0)

size_t slot_bytes_from_env() {
    if (const char* e = option("COMM_SLOT_BYTES")) {
        const long long v = std::atoll(e);
        if (v >= 4096) return (size_t)v / 16 * 16;
    }
    return kDefaultSlotBytes;
}

void alloc_err_word() {
    if (!g_err) {
        ADMM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&g_err), sizeof(int), hipHostMallocMapped));
        *g_err = 0;
    }
}

// ================================================================================================ SHM backend
// Segment: header | data[2][nranks][slot].  Exchange s: every rank copies its payload into data[s & 1][rank], publishes
// posted[rank] = s, waits until every posted[r] >= s, sums the nranks slots in rank order.  Two parities suffice: a rank
// can only post s + 2 after it completed s + 1, which needs every peer's post of s + 1, which a peer makes only after it
// has finished reading s.
struct ShmHeader {
    std::atomic<uint64_t> posted[kMaxRanks];
    std::atomic<uint32_t> attached;
    std::atomic<uint32_t> failed;
    uint64_t slot;
    uint32_t nranks;
    std::atomic<uint64_t> token;       // the job's generation token, stored LAST by rank 0 (release): a segment of the same
                                       // name left by a crashed run, or the live one of a concurrent job, never carries it
};
struct ShmOp { uint64_t seq; size_t nf, nd; double wait_s; };

struct ShmState {
    std::string name;
    int fd = -1;
    void* map = nullptr; size_t map_bytes = 0;
    ShmHeader* hdr = nullptr; unsigned char* data = nullptr;
    unsigned char* stage_in = nullptr; unsigned char* stage_out = nullptr;     // pinned
    std::vector<ShmOp> ring; size_t ring_pos = 0;
    bool owner = false;
} g_shm;

unsigned char* shm_slot(uint64_t seq, int rank) {
    return g_shm.data + ((size_t)(seq & 1) * g_info.nranks + (size_t)rank) * g_slot;
}

// Runs on a runtime thread when the stream reaches it: the payload is in stage_in, the result goes to stage_out.
void shm_host_fn(void* user) {
    const ShmOp op = *static_cast<ShmOp*>(user);
    const size_t off_d = round_up_sz(op.nf * sizeof(float), 16);
    const size_t bytes = off_d + op.nd * sizeof(double);
    ShmHeader* h = g_shm.hdr;
    if (h->failed.load(std::memory_order_acquire) || (g_err && *g_err)) { std::memset(g_shm.stage_out, 0, bytes); return; }
    std::memcpy(shm_slot(op.seq, g_info.rank), g_shm.stage_in, bytes);
    h->posted[g_info.rank].store(op.seq, std::memory_order_release);
    const double t0 = now_s();
    for (int r = 0; r < g_info.nranks; ++r) {
        int spins = 0;
        while (h->posted[r].load(std::memory_order_acquire) < op.seq) {
            if ((++spins & 1023) == 0) {
                sched_yield();
                if (h->failed.load(std::memory_order_acquire) || now_s() - t0 > op.wait_s) {
                    h->failed.store(1, std::memory_order_release);
                    if (g_err) *g_err = 1;
                    std::memset(g_shm.stage_out, 0, bytes);
                    return;
                }
            }
        }
    }
    float* of = reinterpret_cast<float*>(g_shm.stage_out);
    double* od = reinterpret_cast<double*>(g_shm.stage_out + off_d);
    for (int r = 0; r < g_info.nranks; ++r) {             // fixed rank order: every rank computes the identical sum
        const float* sf = reinterpret_cast<const float*>(shm_slot(op.seq, r));
        const double* sd = reinterpret_cast<const double*>(shm_slot(op.seq, r) + off_d);
        if (r == 0) {
            for (size_t i = 0; i < op.nf; ++i) of[i] = sf[i];
            for (size_t i = 0; i < op.nd; ++i) od[i] = sd[i];
        } else {
            for (size_t i = 0; i < op.nf; ++i) of[i] += sf[i];
            for (size_t i = 0; i < op.nd; ++i) od[i] += sd[i];
        }
    }
}

void shm_exchange(float* fbuf, size_t nf, double* dbuf, size_t nd, hipStream_t st) {
    const size_t off_d = round_up_sz(nf * sizeof(float), 16);
    ADMM_REQUIRE(off_d + round_up_sz(nd * sizeof(double), 16) <= g_slot, "exchange payload exceeds the slot size");
    ShmOp* op = &g_shm.ring[g_shm.ring_pos++ % g_shm.ring.size()];
    op->seq = ++g_seq; op->nf = nf; op->nd = nd; op->wait_s = wait_seconds_now();
    if (nf) ADMM_HIP_CHECK(hipMemcpyAsync(g_shm.stage_in, fbuf, nf * sizeof(float), hipMemcpyDeviceToHost, st));
    if (nd) ADMM_HIP_CHECK(hipMemcpyAsync(g_shm.stage_in + off_d, dbuf, nd * sizeof(double), hipMemcpyDeviceToHost, st));
    ADMM_HIP_CHECK(hipLaunchHostFunc(st, shm_host_fn, op));
    if (nf) ADMM_HIP_CHECK(hipMemcpyAsync(fbuf, g_shm.stage_out, nf * sizeof(float), hipMemcpyHostToDevice, st));
    if (nd) ADMM_HIP_CHECK(hipMemcpyAsync(dbuf, g_shm.stage_out + off_d, nd * sizeof(double), hipMemcpyHostToDevice, st));
}

void shm_close() {
    if (g_shm.map) munmap(g_shm.map, g_shm.map_bytes);
    if (g_shm.fd >= 0) close(g_shm.fd);
    if (g_shm.owner && !g_shm.name.empty()) shm_unlink(g_shm.name.c_str());
    if (g_shm.stage_in) (void)hipHostFree(g_shm.stage_in);
    if (g_shm.stage_out) (void)hipHostFree(g_shm.stage_out);
    g_shm = ShmState();
}

// ================================================================================================ PEER backend
// Every rank owns one exchange buffer in ITS device memory, mapped into every peer with hipIpc:
//     data[2][nranks][slot]  |  flags[2][nranks] (one 64-byte line each)
// Exchange s on rank `me`:
//   push   for every rank q (its own buffer included): copy the payload into q.data[s & 1][me], fence, and once all
//          workgroups of that copy are through, store s into q.flags[s & 1][me] (release, system scope);
//   sum    every workgroup first waits (lane r spins, acquire, system scope, bounded) until its own flags[s & 1][r] reads s,
//          then out[i] = sum over r = 0 .. nranks-1 of data[s & 1][r][i], in rank order, written back in place.
// The buffer is fine-grained device memory (peer_alloc_local), so a peer's stores are never shadowed by a stale L2 line here.
struct PeerState {
    unsigned char* local = nullptr;                      // this rank's buffer
    unsigned char* remote[kMaxRanks] = {};               // every rank's buffer as mapped here (remote[rank] == local)
    bool opened[kMaxRanks] = {};
    size_t bytes = 0, flags_off = 0;
    size_t aux_data_off = 0, aux_flags_off = 0;          // the AUX region (peer_device.h): exchanges numbered on the device
    unsigned long long* aux_seq = nullptr;               // device word
    unsigned int* count = nullptr;                       // [nranks] workgroup arrival counters of the push (device)
    unsigned char** d_remote = nullptr;                  // device copy of remote[]
} g_peer;

constexpr int kPushGroups = 4;                           // workgroups per destination rank
constexpr int kSumGroups = 128;                          // most workgroups of a consuming launch (they all wait for the flags)

struct PeerArgs : PeerExchange {
    const float* fsrc; size_t nf; const double* dsrc; size_t nd; size_t off_d;
    float* fdst; double* ddst;
    int bcast_root;                                      // >= 0: a broadcast -- only this rank's payload travels, the others only raise their flags
    int groups;                                          // workgroups per destination rank of the push
    size_t rs_stride;                                    // > 0: a reduce-scatter -- destination q receives the floats fsrc[q * rs_stride ...]
};

// All payload traffic is in 16-byte units per lane, 1 KiB per wave instruction (with an uncached buffer 4-byte accesses
// ran at ~1 GB/s: 90 us for an 80 KB sum).
// The payload [nf floats | pad to 16 | nd doubles] is copied as raw 16-byte units; both source buffers are padded
// allocations of the solvers (>= a multiple of 16 bytes readable), the tail unit of each segment is assembled lane-wise.
__device__ __forceinline__ uint4 peer_load_unit(const PeerArgs& a, size_t u) {
    // unit u of the payload: units [0, uf) come from the float source, the rest from the double source
    const size_t uf = a.off_d / 16;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (u < uf) {
        const size_t i = u * 4;
        if (i + 4 <= a.nf) v = *reinterpret_cast<const uint4*>(a.fsrc + i);
        else {
            unsigned int t[4] = {0u, 0u, 0u, 0u};
            for (int k = 0; k < 4; ++k) if (i + k < a.nf) t[k] = __float_as_uint(a.fsrc[i + k]);
            v = make_uint4(t[0], t[1], t[2], t[3]);
        }
    } else {
        const size_t i = (u - uf) * 2;
        if (i + 2 <= a.nd) v = *reinterpret_cast<const uint4*>(a.dsrc + i);
        else if (i < a.nd) { const unsigned long long b = (unsigned long long)__double_as_longlong(a.dsrc[i]); v = make_uint4((unsigned)b, (unsigned)(b >> 32), 0u, 0u); }
    }
    return v;
}

__global__ void __launch_bounds__(256) peer_push_kernel(PeerArgs a) {
    if (*reinterpret_cast<volatile int*>(a.err)) return;
    const int q = blockIdx.x / a.groups, g = blockIdx.x % a.groups;
    // a broadcast uses the whole slot set of this parity as ONE slot (only the root writes: nranks x slot bytes per exchange)
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.bcast_root >= 0 ? a.remote[q] + (size_t)(a.seq & 1) * a.nranks * a.slot : peer_dst_slot(a, q));
    const size_t units = (a.bcast_root >= 0 && a.rank != a.bcast_root) ? 0 : a.off_d / 16 + (a.nd + 1) / 2;
    PeerArgs src = a;
    src.fsrc = a.fsrc + (size_t)q * a.rs_stride;         // reduce-scatter: rank q gets its own chunk of the send buffer
    for (size_t u = (size_t)g * 256 + threadIdx.x; u < units; u += (size_t)a.groups * 256) {
        const uint4 v = peer_load_unit(src, u);
        peer_store_u64(dst + 2 * u, (unsigned long long)v.x | ((unsigned long long)v.y << 32));       // write-through, no fence needed
        peer_store_u64(dst + 2 * u + 1, (unsigned long long)v.z | ((unsigned long long)v.w << 32));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");       // this wave's stores are acknowledged ...
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int prev = __hip_atomic_fetch_add(&a.count[q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == (unsigned)a.groups - 1) {            // ... before the last workgroup raises the flag at the destination
            __hip_atomic_store(&a.count[q], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            peer_store_u64(peer_flag(a.remote[q], a.flags_off, a.seq, a.nranks, a.rank), a.seq);
        }
    }
}

// Wait for the K flags (the first nranks lanes of every workgroup spin, bounded), then sum the K slots in rank order,
// 16 bytes per lane.
__global__ void __launch_bounds__(256) peer_sum_kernel(PeerArgs a) {
    if (!peer_wait(a)) return;
    const size_t uf = a.off_d / 16, units = uf + (a.nd + 1) / 2;
    // grid-stride: the launch is capped at kSumGroups workgroups (peer_exchange).  Every workgroup of this kernel spins until the
    // flags are up, and with one workgroup per 4 KB of a 4 MB chunk the waiting waves of three rank processes on ONE GPU (the
    // test set-up) filled the device -- the fourth rank's push kernel found no slot and the exchange timed out (round 4: 5 of 8
    // four-rank runs).
    for (size_t u = (size_t)blockIdx.x * 256 + threadIdx.x; u < units; u += (size_t)gridDim.x * 256) {
        if (a.bcast_root >= 0) {                         // broadcast: the root's slot, copied out (the root keeps its own buffer)
            if (a.rank == a.bcast_root || u >= uf) return;
            const float4 s = reinterpret_cast<const float4*>(a.local + (size_t)(a.seq & 1) * a.nranks * a.slot)[u];
            const size_t i = u * 4;
            if (i + 4 <= a.nf) *reinterpret_cast<float4*>(a.fdst + i) = s;
            else {
                const float t[4] = {s.x, s.y, s.z, s.w};
                for (int k = 0; k < 4; ++k) if (i + k < a.nf) a.fdst[i + k] = t[k];
            }
            continue;
        }
        const unsigned char* base = peer_src_slot(a, 0);
        if (u < uf) {
            float4 s = reinterpret_cast<const float4*>(base)[u];
            for (int r = 1; r < a.nranks; ++r) {
                const float4 v = reinterpret_cast<const float4*>(base + (size_t)r * a.slot)[u];
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            const size_t i = u * 4;
            if (i + 4 <= a.nf) *reinterpret_cast<float4*>(a.fdst + i) = s;
            else {
                const float t[4] = {s.x, s.y, s.z, s.w};
                for (int k = 0; k < 4; ++k) if (i + k < a.nf) a.fdst[i + k] = t[k];
            }
        } else {
            double2 s = reinterpret_cast<const double2*>(base)[u];
            for (int r = 1; r < a.nranks; ++r) {
                const double2 v = reinterpret_cast<const double2*>(base + (size_t)r * a.slot)[u];
                s.x += v.x; s.y += v.y;
            }
            const size_t i = (u - uf) * 2;
            a.ddst[i] = s.x;
            if (i + 1 < a.nd) a.ddst[i + 1] = s.y;
        }
    }
}

void peer_fill(PeerExchange& a) {
    a.remote = g_peer.d_remote; a.local = g_peer.local; a.slot = g_slot; a.flags_off = g_peer.flags_off;
    a.nranks = g_info.nranks; a.rank = g_info.rank; a.seq = ++g_seq;
    a.count = g_peer.count; a.err = g_derr;
    a.timeout_ticks = (long long)(wait_seconds_now() * 100e6);          // wall_clock64: constant 100 MHz
}

// bcast_root >= 0: the payload of that rank only (floats).  Every rank still takes part in every exchange -- the non-roots raise
// their flags without payload and the root waits for them in its consuming launch -- so the slot parity protocol holds
// unchanged: the root cannot overwrite slot set s & 1 with chunk s + 2 before every rank has copied chunk s out (it waits for
// their flags of s + 1, raised in stream order after that copy).
void peer_exchange(float* fbuf, size_t nf, double* dbuf, size_t nd, hipStream_t st, int bcast_root = -1, const float* rs_send = nullptr, size_t rs_stride = 0) {
    PeerArgs a;
    peer_fill(a);
    a.bcast_root = bcast_root;
    a.rs_stride = rs_stride;
    a.groups = bcast_root >= 0 ? 32 : kPushGroups;        // bulk payload: more workgroups per destination
    a.off_d = round_up_sz(nf * sizeof(float), 16);
    ADMM_REQUIRE(a.off_d + round_up_sz(nd * sizeof(double), 16) <= (bcast_root >= 0 ? g_slot * (size_t)g_info.nranks : g_slot), "exchange payload exceeds the slot size");
    ADMM_REQUIRE((reinterpret_cast<uintptr_t>(fbuf) & 15) == 0 && (reinterpret_cast<uintptr_t>(dbuf) & 15) == 0, "exchange buffers must be 16-byte aligned");
    a.fsrc = rs_send ? rs_send : fbuf; a.nf = nf; a.dsrc = dbuf; a.nd = nd; a.fdst = fbuf; a.ddst = dbuf;
    ADMM_REQUIRE((reinterpret_cast<uintptr_t>(a.fsrc) & 15) == 0 && (rs_stride % 4) == 0, "reduce-scatter chunks must be 16-byte aligned");
    const size_t units = a.off_d / 16 + (nd + 1) / 2;
    hipLaunchKernelGGL(peer_push_kernel, dim3(g_info.nranks * a.groups), dim3(256), 0, st, a);
    hipLaunchKernelGGL(peer_sum_kernel, dim3((unsigned)std::min<size_t>((units + 255) / 256, (size_t)kSumGroups)), dim3(256), 0, st, a);
}

void peer_close() {
    for (int r = 0; r < kMaxRanks; ++r)
        if (g_peer.opened[r] && g_peer.remote[r]) (void)hipIpcCloseMemHandle(g_peer.remote[r]);
    if (g_peer.local) (void)hipFree(g_peer.local);
    if (g_peer.count) (void)hipFree(g_peer.count);
    if (g_peer.aux_seq) (void)hipFree(g_peer.aux_seq);
    if (g_peer.d_remote) (void)hipFree(g_peer.d_remote);
    g_peer = PeerState();
}

void peer_alloc_local(int nranks) {
    g_slot = slot_bytes_from_env();
    g_peer.flags_off = (size_t)2 * nranks * g_slot;
    g_peer.aux_data_off = g_peer.flags_off + (size_t)2 * nranks * 64;
    g_peer.aux_flags_off = g_peer.aux_data_off + (size_t)2 * nranks * kAuxSlotBytes;
    g_peer.bytes = g_peer.aux_flags_off + (size_t)2 * nranks * (kAuxGroups + 1) * 64;
    void* ptr = nullptr;
    // Fine-grained device memory: the HIP memory model makes system-scope release / acquire pairs (the flag protocol
    // below) order and publish plain accesses to it across agents, so stores arriving from a peer are never shadowed by
    // a stale line of this device's L2.  ADMM_HIP_PEER_MEM=uncached selects MTYPE UC instead (every access a memory
    // transaction of its own: measured 3-10x slower for the 80 KB payload; kept as the conservative fallback).
    const char* pm = option("PEER_MEM");
    const bool uncached = pm && std::string(pm) == "uncached";
    if (hipExtMallocWithFlags(&ptr, g_peer.bytes, uncached ? hipDeviceMallocUncached : hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        pool_trim();                                          // the cache of released device blocks must never starve this allocation
        ADMM_HIP_CHECK(hipExtMallocWithFlags(&ptr, g_peer.bytes, uncached ? hipDeviceMallocFinegrained : hipDeviceMallocUncached));
    }
    g_peer.local = static_cast<unsigned char*>(ptr);
    ADMM_HIP_CHECK(hipMemset(g_peer.local, 0, g_peer.bytes));
    ADMM_HIP_CHECK(hipDeviceSynchronize());
}

// ================================================================================================ dispatch
void exchange(float* fbuf, size_t nf, double* dbuf, size_t nd, hipStream_t st) {
    switch (g_info.backend) {
        case COMM_RCCL:
            if (nf && nd) ADMM_NCCL_CHECK(ncclGroupStart());
            if (nf) ADMM_NCCL_CHECK(ncclAllReduce(fbuf, fbuf, nf, ncclFloat, ncclSum, g_comm, st));
            if (nd) ADMM_NCCL_CHECK(ncclAllReduce(dbuf, dbuf, nd, ncclDouble, ncclSum, g_comm, st));
            if (nf && nd) ADMM_NCCL_CHECK(ncclGroupEnd());
            break;
        case COMM_SHM: shm_exchange(fbuf, nf, dbuf, nd, st); break;
        case COMM_PEER: peer_exchange(fbuf, nf, dbuf, nd, st); break;
        default: break;
    }
}

void require_free() {
    if (g_info.active) throw Error(ADMM_ERR_COMM, "communicator already initialised");
}
void check_ranks(int nranks, int rank) {
    ADMM_REQUIRE(nranks >= 1 && nranks <= kMaxRanks && rank >= 0 && rank < nranks, "bad rank / nranks");
}

}  // namespace

// The next exchange of the PEER backend for a solver whose own kernels produce / consume it (peer_device.h).
PeerExchange comm_peer_begin(size_t payload_bytes) {
    if (g_info.backend != COMM_PEER) throw Error(ADMM_ERR_COMM, "the PEER exchange is not attached");
    ADMM_REQUIRE(payload_bytes <= g_slot, "exchange payload exceeds the slot size");
    PeerExchange e;
    peer_fill(e);
    return e;
}

PeerAux comm_peer_aux() {
    if (g_info.backend != COMM_PEER) throw Error(ADMM_ERR_COMM, "the PEER exchange is not attached");
    PeerAux a;
    a.remote = g_peer.d_remote; a.local = g_peer.local; a.data_off = g_peer.aux_data_off; a.flags_off = g_peer.aux_flags_off;
    a.nranks = g_info.nranks; a.rank = g_info.rank; a.seq = g_peer.aux_seq; a.err = g_derr;
    a.timeout_ticks = (long long)(wait_seconds_now() * 100e6);
    return a;
}

CommInfo comm_info() {
    std::lock_guard<std::mutex> lk(g_mu);
    return g_info;
}

CommLockstep::CommLockstep() { g_lockstep.fetch_add(1, std::memory_order_relaxed); }
CommLockstep::~CommLockstep() { g_lockstep.fetch_sub(1, std::memory_order_relaxed); }

void comm_check() {
    if (g_info.backend == COMM_PEER && g_derr) {          // called at the solvers' polls (after an event sync) only
        int h = 0;
        if (hipMemcpy(&h, g_derr, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess && h && g_err) *g_err = 1;
    }
    if (g_err && *g_err) throw Error(ADMM_ERR_COMM, "exchange failed: a rank did not arrive within the time limit (or a peer reported failure)");
}

namespace {
// RCCL only: give up on the communicator (its in-flight kernels end) and report.
[[noreturn]] void rccl_fail(const std::string& why) {
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (g_comm) { (void)ncclCommAbort(g_comm); g_comm = nullptr; }
        g_info = CommInfo();
        g_seq = 0;
    }
    throw Error(ADMM_ERR_COMM, "RCCL exchange failed: " + why + " (communicator aborted; call admm_hip_comm_finalize and re-attach)");
}
template <typename Q>
void rccl_watched_wait(Q&& query) {
    const double t0 = now_s(), bound = wait_seconds_now();
    double next = t0 + 2e-3;
    for (;;) {
        const hipError_t q = query();
        if (q == hipSuccess) return;
        if (q != hipErrorNotReady) ADMM_HIP_CHECK(q);
        const double t = now_s();
        if (t >= next) {
            next = t + 5e-3;
            ncclResult_t ar = ncclSuccess;
            const ncclResult_t rc = ncclCommGetAsyncError(g_comm, &ar);
            if (rc != ncclSuccess) rccl_fail(std::string("ncclCommGetAsyncError: ") + ncclGetErrorString(rc));
            if (ar != ncclSuccess && ar != ncclInProgress) rccl_fail(std::string("asynchronous error: ") + ncclGetErrorString(ar));
            if (t - t0 > bound) rccl_fail("a rank did not arrive within " + std::to_string((int)bound) + " s");
        }
        if (t - t0 > 100e-6) sched_yield();              // the first 100 us spin: a batch of iterations is that short
    }
}
}  // namespace

void comm_stream_sync(hipStream_t st) {
    if (g_info.backend != COMM_RCCL || !g_comm) { ADMM_HIP_CHECK(hipStreamSynchronize(st)); return; }
    rccl_watched_wait([&] { return hipStreamQuery(st); });
}
void comm_event_sync(hipEvent_t ev) {
    if (g_info.backend != COMM_RCCL || !g_comm) { ADMM_HIP_CHECK(hipEventSynchronize(ev)); return; }
    rccl_watched_wait([&] { return hipEventQuery(ev); });
}
CommInfo comm_info_live() {
    std::lock_guard<std::mutex> lk(g_mu);
    CommInfo ci = g_info;
    if (ci.backend == COMM_RCCL && g_comm) {
        int n = 0, r = -1;
        ADMM_NCCL_CHECK(ncclCommCount(g_comm, &n));
        ADMM_NCCL_CHECK(ncclCommUserRank(g_comm, &r));
        ci.nranks = n; ci.rank = r;
    } else if (ci.backend == COMM_SHM && g_shm.hdr) {
        ci.nranks = (int)g_shm.hdr->attached.load(std::memory_order_acquire);
    } else if (ci.backend == COMM_PEER) {
        int n = 0;
        for (int r = 0; r < kMaxRanks; ++r) n += (g_peer.remote[r] != nullptr);
        ci.nranks = n;
    }
    return ci;
}

void allreduce_sum_f32(float* buf, size_t n, hipStream_t st) {
    if (!g_info.active || n == 0) return;
    if (g_info.backend == COMM_RCCL) { exchange(buf, n, nullptr, 0, st); return; }
    const size_t per = g_slot / sizeof(float);
    for (size_t o = 0; o < n; o += per) exchange(buf + o, std::min(per, n - o), nullptr, 0, st);
}
void allreduce_sum_f64(double* buf, size_t n, hipStream_t st) {
    if (!g_info.active || n == 0) return;
    if (g_info.backend == COMM_RCCL) { exchange(nullptr, 0, buf, n, st); return; }
    const size_t per = g_slot / sizeof(double);
    for (size_t o = 0; o < n; o += per) exchange(nullptr, 0, buf + o, std::min(per, n - o), st);
}
void broadcast_f32(float* buf, size_t n, int root, hipStream_t st) {
    if (!g_info.active || n == 0 || g_info.nranks == 1) return;
    if (g_info.backend == COMM_RCCL) { ADMM_NCCL_CHECK(ncclBroadcast(buf, buf, n, ncclFloat, root, g_comm, st)); return; }
    if (g_info.backend == COMM_PEER) {
        const size_t per = g_slot * (size_t)g_info.nranks / sizeof(float);
        for (size_t o = 0; o < n; o += per) peer_exchange(buf + o, std::min(per, n - o), nullptr, 0, st, root);
        return;
    }
    if (g_info.rank != root) ADMM_HIP_CHECK(hipMemsetAsync(buf, 0, n * sizeof(float), st));      // SHM (tests): x + 0 + ... + 0 is x bit for bit
    allreduce_sum_f32(buf, n, st);
}
void reduce_scatter_sum_f32(const float* send, float* recv, size_t count, hipStream_t st) {
    if (count == 0) return;
    if (!g_info.active || g_info.nranks == 1) {
        ADMM_HIP_CHECK(hipMemcpyAsync(recv, send, count * sizeof(float), hipMemcpyDeviceToDevice, st));
        return;
    }
    if (g_info.backend == COMM_RCCL) { ADMM_NCCL_CHECK(ncclReduceScatter(send, recv, count, ncclFloat, ncclSum, g_comm, st)); return; }
    if (g_info.backend == COMM_PEER) {
        // every rank pushes chunk q of its send buffer into rank q's slot and sums the nranks slots it received, in rank order:
        // (nranks - 1) / nranks of an all-reduce's traffic, 1 / nranks of its additions
        const size_t per = g_slot / sizeof(float);
        for (size_t o = 0; o < count; o += per) peer_exchange(recv + o, std::min(per, count - o), nullptr, 0, st, -1, send + o, count);
        return;
    }
    // SHM (tests): an all-reduce of a copy of the whole send buffer, then this rank's chunk -- the same sums in the same order
    DevBuf<float> tmp((size_t)g_info.nranks * count);
    ADMM_HIP_CHECK(hipMemcpyAsync(tmp.get(), send, (size_t)g_info.nranks * count * sizeof(float), hipMemcpyDeviceToDevice, st));
    allreduce_sum_f32(tmp.get(), (size_t)g_info.nranks * count, st);
    ADMM_HIP_CHECK(hipMemcpyAsync(recv, tmp.get() + (size_t)g_info.rank * count, count * sizeof(float), hipMemcpyDeviceToDevice, st));
    ADMM_HIP_CHECK(hipStreamSynchronize(st));             // tmp is freed on return
}
void allreduce_sum_f32_f64(float* fbuf, size_t nf, double* dbuf, size_t nd, hipStream_t st) {
    if (!g_info.active) return;
    if (g_info.backend != COMM_RCCL && round_up_sz(nf * sizeof(float), 16) + round_up_sz(nd * sizeof(double), 16) > g_slot) {
        allreduce_sum_f32(fbuf, nf, st);
        allreduce_sum_f64(dbuf, nd, st);
        return;
    }
    exchange(fbuf, nf, dbuf, nd, st);
}

// ---------------------------------------------------------------------------------------------- bootstrap (api.hip)
int comm_unique_id(void* out) {
    ncclUniqueId id;
    ADMM_NCCL_CHECK(ncclGetUniqueId(&id));
    std::memcpy(out, &id, sizeof(id));
    return 0;
}
void comm_init(int nranks, int rank, const void* idbytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    require_free();
    check_ranks(nranks, rank);
    ncclUniqueId id;
    std::memcpy(&id, idbytes, sizeof(id));
    ADMM_NCCL_CHECK(ncclCommInitRank(&g_comm, nranks, id, rank));
    g_info.nranks = nranks; g_info.rank = rank; g_info.active = true; g_info.backend = COMM_RCCL;
    g_seq = 0;
}

// `token`: a job-unique non-zero number every rank received over the caller's channel (like RCCL's unique id).  Rank 0
// creates the segment afresh (a stale name is unlinked first: whoever still maps the old inode keeps it, nobody of THIS
// job can mistake it for the new one) and stores the token into the header last; the other ranks map whatever carries the
// name, and keep re-opening it until the header shows the token.  Without this a rank that raced rank 0's unlink + create
// attached to the OLD segment, found `attached` and `posted[]` already high, skipped every wait and summed stale slots.
void comm_init_shm(int nranks, int rank, const char* name, unsigned long long token) {
    std::lock_guard<std::mutex> lk(g_mu);
    require_free();
    check_ranks(nranks, rank);
    ADMM_REQUIRE(name != nullptr && name[0] == '/' && std::strlen(name) < 200, "shm name must start with '/'");
    ADMM_REQUIRE(token != 0, "the shared-memory token must be non-zero (a job-unique number all ranks agree on)");
    alloc_err_word();
    g_slot = slot_bytes_from_env();
    const size_t hdr_bytes = round_up_sz(sizeof(ShmHeader), 4096);
    const size_t total = hdr_bytes + (size_t)2 * nranks * g_slot;
    g_shm.name = name;
    const double t0 = now_s();
    const double patience = std::min(wait_seconds_patient(), 60.0);
    auto map_segment = [&]() {
        g_shm.map = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, g_shm.fd, 0);
        if (g_shm.map == MAP_FAILED) { g_shm.map = nullptr; shm_close(); throw Error(ADMM_ERR_COMM, "mmap of the shared-memory segment failed"); }
        g_shm.map_bytes = total;
        g_shm.hdr = static_cast<ShmHeader*>(g_shm.map);
        g_shm.data = static_cast<unsigned char*>(g_shm.map) + hdr_bytes;
    };
    if (rank == 0) {
        for (int attempt = 0; ; ++attempt) {
            g_shm.fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
            if (g_shm.fd >= 0 || errno != EEXIST || attempt == 1) break;
            shm_unlink(name);                                               // a stale segment of a crashed run
        }
        if (g_shm.fd < 0 || ftruncate(g_shm.fd, (off_t)total) != 0) { shm_close(); throw Error(ADMM_ERR_COMM, "cannot create the shared-memory segment"); }
        g_shm.owner = true;
        map_segment();                                                      // fresh pages read zero: posted[], attached, failed
        g_shm.hdr->slot = g_slot; g_shm.hdr->nranks = (uint32_t)nranks;
        g_shm.hdr->token.store(token, std::memory_order_release);
    } else {
        for (;;) {                                                          // until the segment of THIS job is there
            g_shm.fd = shm_open(name, O_RDWR, 0600);
            struct stat sb;
            if (g_shm.fd >= 0 && fstat(g_shm.fd, &sb) == 0 && (size_t)sb.st_size >= total) {
                map_segment();
                if (g_shm.hdr->token.load(std::memory_order_acquire) == token) break;
                munmap(g_shm.map, g_shm.map_bytes);                         // another generation's segment (or not initialised yet)
                g_shm.map = nullptr; g_shm.hdr = nullptr; g_shm.data = nullptr; g_shm.map_bytes = 0;
            }
            if (g_shm.fd >= 0) { close(g_shm.fd); g_shm.fd = -1; }
            if (now_s() - t0 > patience) { shm_close(); throw Error(ADMM_ERR_COMM, "the shared-memory segment of this job did not appear (is rank 0 running? do all ranks pass the same token?)"); }
            usleep(2000);
        }
    }
    ADMM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&g_shm.stage_in), g_slot, hipHostMallocDefault));
    ADMM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&g_shm.stage_out), g_slot, hipHostMallocDefault));
    g_shm.ring.assign(8192, ShmOp());
    g_shm.hdr->attached.fetch_add(1, std::memory_order_acq_rel);
    while (g_shm.hdr->attached.load(std::memory_order_acquire) < (uint32_t)nranks) {       // everybody is mapped before anyone posts
        if (now_s() - t0 > patience) { shm_close(); throw Error(ADMM_ERR_COMM, "not every rank attached to the shared-memory segment"); }
        usleep(1000);
    }
    if (g_shm.hdr->attached.load(std::memory_order_acquire) > (uint32_t)nranks || g_shm.hdr->slot != g_slot || g_shm.hdr->nranks != (uint32_t)nranks) {
        shm_close();
        throw Error(ADMM_ERR_COMM, "ranks disagree on slot size / rank count (or more ranks attached than the job has)");
    }
    if (rank == 0) { shm_unlink(name); g_shm.owner = false; }              // everybody holds a mapping: the name is not needed any more and cannot leak
    g_info.nranks = nranks; g_info.rank = rank; g_info.active = true; g_info.backend = COMM_SHM;
    g_seq = 0;
}

// PEER, step 1: allocate this rank's exchange buffer and export it.
void comm_peer_prepare(int nranks, void* handle_out) {
    std::lock_guard<std::mutex> lk(g_mu);
    require_free();
    ADMM_REQUIRE(nranks >= 1 && nranks <= kMaxRanks, "bad nranks");
    if (g_peer.local) peer_close();
    peer_alloc_local(nranks);
    hipIpcMemHandle_t h;
    ADMM_HIP_CHECK(hipIpcGetMemHandle(&h, g_peer.local));
    static_assert(sizeof(hipIpcMemHandle_t) == ADMM_HIP_PEER_HANDLE_BYTES, "hipIpcMemHandle_t size");
    std::memcpy(handle_out, &h, sizeof(h));
}
// PEER, step 2: `handles` = the nranks handles in rank order (gathered by the caller).
void comm_init_peer(int nranks, int rank, const void* handles) {
    std::lock_guard<std::mutex> lk(g_mu);
    require_free();
    check_ranks(nranks, rank);
    if (!g_peer.local) throw Error(ADMM_ERR_COMM, "call admm_hip_comm_peer_prepare first");
    ADMM_REQUIRE(g_peer.flags_off == (size_t)2 * nranks * g_slot, "nranks differs from the prepared buffer");
    alloc_err_word();
    for (int r = 0; r < nranks; ++r) {
        if (r == rank) { g_peer.remote[r] = g_peer.local; continue; }
        hipIpcMemHandle_t h;
        std::memcpy(&h, static_cast<const unsigned char*>(handles) + (size_t)r * sizeof(h), sizeof(h));
        void* ptr = nullptr;
        ADMM_HIP_CHECK(hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess));
        g_peer.remote[r] = static_cast<unsigned char*>(ptr);
        g_peer.opened[r] = true;
    }
    if (!g_derr) ADMM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&g_derr), sizeof(int)));
    ADMM_HIP_CHECK(hipMemset(g_derr, 0, sizeof(int)));
    ADMM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&g_peer.count), (nranks + 1) * sizeof(unsigned int)));
    ADMM_HIP_CHECK(hipMemset(g_peer.count, 0, (nranks + 1) * sizeof(unsigned int)));
    ADMM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&g_peer.aux_seq), sizeof(unsigned long long)));
    ADMM_HIP_CHECK(hipMemset(g_peer.aux_seq, 0, sizeof(unsigned long long)));
    ADMM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&g_peer.d_remote), nranks * sizeof(unsigned char*)));
    ADMM_HIP_CHECK(hipMemcpy(g_peer.d_remote, g_peer.remote, nranks * sizeof(unsigned char*), hipMemcpyHostToDevice));
    ADMM_HIP_CHECK(hipDeviceSynchronize());
    g_info.nranks = nranks; g_info.rank = rank; g_info.active = true; g_info.backend = COMM_PEER;
    g_seq = 0;
}

void comm_finalize() {
    std::lock_guard<std::mutex> lk(g_mu);
    (void)hipDeviceSynchronize();
    if (g_comm) { (void)ncclCommDestroy(g_comm); g_comm = nullptr; }
    if (g_info.backend == COMM_SHM) shm_close();
    if (g_peer.local) peer_close();
    if (g_err) *g_err = 0;
    if (g_derr) (void)hipMemset(g_derr, 0, sizeof(int));
    g_info = CommInfo();
    g_seq = 0;
}

}  // namespace admm
