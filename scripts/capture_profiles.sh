#!/bin/bash
# Capture the judged evidence for the headline bench on the GPU box (run from the repo root through gpurun):
#   kernel-trace stats of `bench.py`, PMC HBM bytes (separate passes per counter), the bench line itself.
# Outputs go to gpurun_out/; copy the summaries into profiles/ afterwards.
set -u
TAG=${1:-r01_final}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o c2 -- python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --consensus-seconds 0 > $OUT/trace_bench.log 2>&1
python scripts/rocpd_summary.py $OUT/trace/c2_results.db $OUT/kernel_stats.md 14
grep '^{"metric"' $OUT/trace_bench.log | tail -1 > $OUT/kernel_stats_benchline.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o c2 -- python bench.py --steps 1 --warmup 0 --cpu-seconds 0 --consensus-seconds 0 --nlambda 10 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o c2 -- python bench.py --steps 1 --warmup 0 --cpu-seconds 0 --consensus-seconds 0 --nlambda 10 > $OUT/pmc_write.log 2>&1
python scripts/rocpd_pmc.py $OUT/pmc_fetch/c2_results.db $OUT/pmc_write/c2_results.db $OUT/pmc_hbm_bytes.md
python bench.py > $OUT/bench_default.log 2>&1
grep '^{"metric"' $OUT/bench_default.log | tail -1 > $OUT/bench_default.json
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write      # the databases are large; the summaries are what is kept
ls -la $OUT
