// Launch geometry of gemv_t (host-only struct, shared by prep.h and gemv_kernels.h).
#pragma once
#include <cstddef>

namespace admm {

// Launch geometry for gemv_t: picks the row segmentation and the column blocking so that the
// grid is about `wg_per_cu` workgroups per CU, each with whole multiples of 4 column groups.
struct GemvTPlan {
    int seg_len = 0, seg_alloc = 0, nseg = 0, groups_per_wg = 0, num_cb = 0, grid = 0;
    size_t lds_bytes = 0;
    bool nt = false;              // stream the matrix with non-temporal loads (set by plan_gemv_t from its size)
};

// Operands above this size are streamed with non-temporal loads: the users of gemv_t re-read their matrices once per
// ADMM iteration, and whatever is larger than this cannot stay cache-resident next to the other operands of the
// iteration anyway (C4: 8 x 500 MB, C5: 2 x 2 GB); small cached inverses (the tall path below p = 2048: <= 16 MB) keep
// plain loads.
constexpr size_t kGemvNtBytes = (size_t)32 << 20;

}  // namespace admm
