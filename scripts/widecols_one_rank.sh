#!/bin/bash
# The column-sharded wide solver as ONE rank over the PEER exchange at the full BASELINE configs[2] shape (n = 2000, p = 200 000, 20-lambda
# path): microseconds per iteration with and without the persistent active-set stretch.  Output: gpurun_out/widecols_one_rank.txt
mkdir -p gpurun_out
for v in 1 0; do
  ADMM_HIP_WIDE_PERSIST_COLS=$v python bench.py --child widecols:peer:/tmp/wc_$v.json --seed 123 > /dev/null 2>gpurun_out/wc_$v.err
  echo "ADMM_HIP_WIDE_PERSIST_COLS=$v: $(cat /tmp/wc_$v.json)"
done | tee gpurun_out/widecols_one_rank.txt
