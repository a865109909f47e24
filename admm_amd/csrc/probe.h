// Dev build only (-DADMM_HIP_PROBE through ADMM_HIP_EXTRA_CXXFLAGS): in-kernel wall-clock (100 MHz) timestamps of a few
// workgroups, one record per decision -- [4096 records][4 observers][8 stamps] -- dumped by the plan's run() when
// ADMM_HIP_PROBE_OUT names a file (scripts/wide_probe.py, scripts/tall_probe.py read it).  Compiled out of the product.
#pragma once
#ifdef ADMM_HIP_PROBE
#define WIDE_PROBE_DECL long long pt_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define WIDE_PROBE(k) do { pt_[k] = wall_clock64(); } while (0)
#define WIDE_PROBE_FLUSH(obs, total) do { if (q.probe && threadIdx.x == 0 && (obs) >= 0) {                                  \
        long long* d_ = q.probe + ((size_t)((total) & 4095) * 4 + (obs)) * 8;                                                \
        _Pragma("unroll") for (int k_ = 0; k_ < 8; ++k_) d_[k_] = pt_[k_]; } } while (0)
#else
#define WIDE_PROBE_DECL
#define WIDE_PROBE(k) do {} while (0)
#define WIDE_PROBE_FLUSH(obs, total) do {} while (0)
#endif
