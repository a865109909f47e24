// Process-wide RCCL communicator (one process per GPU).  Only the consensus solver and the global
// standardisation use it; every call is a no-op when no communicator is attached.
#pragma once
#include "admm_internal.h"

namespace admm {

struct CommInfo { int nranks = 1, rank = 0; bool active = false; };
CommInfo comm_info();
// In-place sum all-reduce on device buffers, enqueued on `st`.  No-ops without a communicator.
void allreduce_sum_f32(float* buf, size_t n, hipStream_t st);
void allreduce_sum_f64(double* buf, size_t n, hipStream_t st);
// Two buffers in one grouped RCCL launch (the consensus payload: p floats + the norm doubles).
void allreduce_sum_f32_f64(float* fbuf, size_t nf, double* dbuf, size_t nd, hipStream_t st);

}  // namespace admm
