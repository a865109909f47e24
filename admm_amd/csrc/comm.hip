// RCCL glue: communicator bootstrap through a caller-provided unique id (the Python harness
// broadcasts it with torch.distributed; an R / MPI caller would use its own channel) and the few
// collectives the consensus path needs.  xGMI is point-to-point and the payload is small (p floats
// + 3 doubles per iteration), so this is latency-bound: one grouped all-reduce per ADMM iteration.
#include "comm.h"
#include <rccl/rccl.h>
#include <mutex>

namespace admm {
namespace {
std::mutex g_mu;
ncclComm_t g_comm = nullptr;
CommInfo g_info;

#define ADMM_NCCL_CHECK(expr)                                                                     \
    do {                                                                                          \
        ncclResult_t _r = (expr);                                                                 \
        if (_r != ncclSuccess)                                                                    \
            throw ::admm::Error(ADMM_ERR_COMM, std::string(#expr) + ": " + ncclGetErrorString(_r)); \
    } while (0)
}  // namespace

CommInfo comm_info() {
    std::lock_guard<std::mutex> lk(g_mu);
    return g_info;
}

void allreduce_sum_f32(float* buf, size_t n, hipStream_t st) {
    if (!g_comm || n == 0) return;
    ADMM_NCCL_CHECK(ncclAllReduce(buf, buf, n, ncclFloat, ncclSum, g_comm, st));
}
void allreduce_sum_f64(double* buf, size_t n, hipStream_t st) {
    if (!g_comm || n == 0) return;
    ADMM_NCCL_CHECK(ncclAllReduce(buf, buf, n, ncclDouble, ncclSum, g_comm, st));
}
void allreduce_sum_f32_f64(float* fbuf, size_t nf, double* dbuf, size_t nd, hipStream_t st) {
    if (!g_comm) return;
    ADMM_NCCL_CHECK(ncclGroupStart());
    ADMM_NCCL_CHECK(ncclAllReduce(fbuf, fbuf, nf, ncclFloat, ncclSum, g_comm, st));
    ADMM_NCCL_CHECK(ncclAllReduce(dbuf, dbuf, nd, ncclDouble, ncclSum, g_comm, st));
    ADMM_NCCL_CHECK(ncclGroupEnd());
}

int comm_unique_id(void* out) {
    ncclUniqueId id;
    ADMM_NCCL_CHECK(ncclGetUniqueId(&id));
    std::memcpy(out, &id, sizeof(id));
    return 0;
}
void comm_init(int nranks, int rank, const void* idbytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_comm) throw Error(ADMM_ERR_COMM, "communicator already initialised");
    ADMM_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank / nranks");
    ncclUniqueId id;
    std::memcpy(&id, idbytes, sizeof(id));
    ADMM_NCCL_CHECK(ncclCommInitRank(&g_comm, nranks, id, rank));
    g_info.nranks = nranks; g_info.rank = rank; g_info.active = true;
}
void comm_finalize() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_comm) { (void)ncclCommDestroy(g_comm); g_comm = nullptr; }
    g_info = CommInfo();
}

}  // namespace admm
