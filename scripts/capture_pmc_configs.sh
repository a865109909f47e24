#!/bin/bash
# PMC HBM bytes of the dominant kernels of C4 (batched consensus products) and of the sharing basis pursuit (separate passes per
# counter, kernel-trace only).  Run from the repo root through gpurun; summaries go to gpurun_out/<tag>/.
set -u
TAG=${1:-r03p}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
for c in c4 c5parbp; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/f_$c -o k -- python scripts/bench_configs.py $c > $OUT/f_$c.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/w_$c -o k -- python scripts/bench_configs.py $c > $OUT/w_$c.log 2>&1
  python scripts/rocpd_pmc.py $OUT/f_$c/k_results.db $OUT/w_$c/k_results.db $OUT/${c}_pmc_hbm_bytes.md > /dev/null
  rm -rf $OUT/f_$c $OUT/w_$c
done
head -8 $OUT/c4_pmc_hbm_bytes.md $OUT/c5parbp_pmc_hbm_bytes.md
