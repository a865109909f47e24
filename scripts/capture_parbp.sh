#!/bin/bash
# Kernel trace + PMC bytes + plain bench line of the sharing-BP config alone (run from the repo root through gpurun; outputs in
# gpurun_out/r05q/: copy c5parbp_kernel_stats.md / c5parbp_pmc_hbm_bytes.md to profiles/r05_* and merge pmc_traffic.json's
# configs.c5parbp into profiles/pmc_traffic.json).  The other configs: scripts/capture_profiles.sh, scripts/capture_pmc_configs.sh.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05q
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace_c5parbp -o k -- python scripts/bench_configs.py c5parbp > $OUT/trace_c5parbp.log 2>&1
python scripts/rocpd_summary.py $OUT/trace_c5parbp/k_results.db $OUT/c5parbp_kernel_stats.md 24
rm -rf $OUT/trace_c5parbp
bash scripts/capture_pmc_configs.sh r05q c5parbp
python scripts/bench_configs.py c5parbp > $OUT/bench_c5parbp_plain.jsonl 2>/dev/null
ls $OUT
