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

// Non-temporal streaming policy.  A solver's products re-read their matrices once per ADMM iteration; when the matrices of
// one iteration together exceed what the 256 MB Infinity Cache can hold between two passes they are streamed with
// non-temporal loads (C4: 8 x 500 MB, C5: 2 x 2 GB: +10-13 % it/s), otherwise with plain loads so that they stay resident
// (the tall path's cached inverse: -8 % with nt).  Solvers that know their per-iteration working set say so with
// gemv_stream_nt(working_set_bytes); a product planned on its own decides from its matrix alone.
constexpr size_t kGemvNtWorkingSet = (size_t)220 << 20;      // working set of one iteration above which nothing is reused from the cache
constexpr size_t kGemvNtBytes = (size_t)128 << 20;           // a single matrix above this cannot share the cache with a second operand
inline bool gemv_stream_nt(size_t working_set_bytes) { return working_set_bytes > kGemvNtWorkingSet; }

}  // namespace admm
