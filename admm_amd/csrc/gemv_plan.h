// Launch geometry of gemv_t (host-only struct, shared by prep.h and gemv_kernels.h).
#pragma once
#include <cstddef>

namespace admm {

// Launch geometry for gemv_t: picks the row segmentation and the column blocking so that the
// grid is about `wg_per_cu` workgroups per CU, each with whole multiples of 4 column groups.
struct GemvTPlan {
    int seg_len = 0, seg_alloc = 0, nseg = 0, groups_per_wg = 0, num_cb = 0, grid = 0;
    size_t lds_bytes = 0;
};

}  // namespace admm
