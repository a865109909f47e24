// Process-wide exchange layer (one process per GPU).  Only the solvers with a real exchange step use it: the row-block
// consensus solver (PADMMLasso: one all-reduce of p floats + 3 doubles per ADMM iteration), the row-sharded tall
// x-update (2p floats per iteration) and the global-moment standardisation / split-K Gram of their setup.
//
// Three interchangeable backends behind the same in-place sum all-reduce, all stream-ordered (the host never waits):
//   RCCL   ncclAllReduce over xGMI -- the default between GPUs and the correctness baseline.
//   PEER   one-shot all-reduce over peer-mapped device memory (hipIpc): every rank pushes its payload into a slot of
//          EVERY rank's exchange buffer and raises a flag there; each rank then waits for its K flags and sums the K
//          slots locally in rank order.  xGMI is point-to-point (7 links per GPU), the payload is <= 0.4 MB, so the
//          K-1 pushes travel over K-1 different links at once and the whole exchange is one link latency instead of a
//          ring's 2(K-1) hops.  Identical summation order on every rank => bit-identical results => identical decisions.
//   SHM    through POSIX shared memory on the host (D2H, a host function in stream order, H2D).  Slow; exists so that
//          the multi-rank code paths run as separate processes on ONE GPU / without RCCL (tests), bit-identical to PEER.
// Every call is a no-op when no communicator is attached.
#pragma once
#include "admm_internal.h"

namespace admm {

enum { COMM_NONE = 0, COMM_RCCL = 1, COMM_SHM = 2, COMM_PEER = 3 };
struct CommInfo { int nranks = 1, rank = 0; bool active = false; int backend = COMM_NONE; };
CommInfo comm_info();
// In-place sum all-reduce on device buffers, enqueued on `st` (any length: long messages go in slot-sized chunks).
void allreduce_sum_f32(float* buf, size_t n, hipStream_t st);
void allreduce_sum_f64(double* buf, size_t n, hipStream_t st);
// Broadcast of a device buffer from `root` to every rank, enqueued on `st` (setup only: the panel broadcasts of the distributed
// factorisation).  RCCL: ncclBroadcast.  PEER: the root pushes slot-sized chunks into every rank's exchange buffer, the others
// only raise their flags (same parity protocol as the all-reduce).  SHM (tests): the other ranks clear their buffer and the ranks
// sum -- x + 0 + ... + 0 is x bit for bit.
void broadcast_f32(float* buf, size_t n, int root, hipStream_t st);
// Sum reduce-scatter: `send` holds nranks chunks of `count` floats (chunk q is what rank q is to receive); `recv` (count floats) gets
// the sum over the ranks of chunk [this rank].  RCCL: ncclReduceScatter.  PEER: chunk q goes into rank q's slots only.  SHM (tests):
// all-reduce of a copy + own chunk.  No communicator: a copy.  (Setup only: the split-K Gram of the row-sharded tall solver.)
void reduce_scatter_sum_f32(const float* send, float* recv, size_t count, hipStream_t st);
// Two buffers in one exchange (the consensus payload: p floats + the norm doubles).
void allreduce_sum_f32_f64(float* fbuf, size_t nf, double* dbuf, size_t nd, hipStream_t st);
// While one of these is alive the exchanges this process enqueues take the short LOCK-STEP bound on their waits
// (ADMM_HIP_COMM_TIMEOUT_S, 20 s): the per-iteration exchanges of a solve.  Everything else -- setup reductions, the join
// of the replica modes -- takes the PATIENT bound (ADMM_HIP_COMM_PATIENT_TIMEOUT_S, one hour), like RCCL would wait.
struct CommLockstep { CommLockstep(); ~CommLockstep(); CommLockstep(const CommLockstep&) = delete; CommLockstep& operator=(const CommLockstep&) = delete; };
// Throws ADMM_ERR_COMM if an exchange of the SHM / PEER backends timed out or a peer reported failure (checked by the
// loop drivers at every poll; the kernels of a failed exchange return immediately instead of spinning).
void comm_check();
// Host waits of the solvers.  SHM / PEER: plain synchronisation (their device / host-function waits are bounded themselves).
// RCCL: an RCCL kernel waiting for a rank that died never returns, so the host polls instead -- hipStreamQuery / hipEventQuery,
// ncclCommGetAsyncError every few milliseconds, and the bound in force (lock-step 20 s inside a solve's loop, patient otherwise):
// on an asynchronous error or a timed-out wait the communicator is aborted (ncclCommAbort ends its kernels) and the call returns
// ADMM_ERR_COMM instead of hanging ("return an error code if a rank fails").
void comm_stream_sync(hipStream_t st);
void comm_event_sync(hipEvent_t ev);
// The ranks the attached communicator REALLY holds (RCCL: ncclCommCount / ncclCommUserRank of the live communicator, not what the
// caller asked for).
CommInfo comm_info_live();
// PEER backend only: start the next exchange for a solver whose own kernels write the slots / wait and read them
// (peer_device.h): no launches of the exchange layer itself.
struct PeerExchange;
PeerExchange comm_peer_begin(size_t payload_bytes);
// PEER backend only: the AUX region (peer_device.h) for a kernel that exchanges many times inside ONE launch.
struct PeerAux;
PeerAux comm_peer_aux();

}  // namespace admm
