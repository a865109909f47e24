#!/bin/bash
# Dev tool (GPU box): kernel time of the C2 Gram launch with and without the quarter-tile tail (syrk_mfma.hip, gram_work_lists).
# Usage (from the repo root through gpurun): bash scripts/gram_variants.sh ; the table goes to gpurun_out/gram_variants.md
export TMPDIR=/tmp
OUT=gpurun_out/gram_variants
mkdir -p $OUT
echo "| tail (ADMM_HIP_GRAM_TAIL) | Gram kernel (us) | work lists | t_gram / t_factor of the plan |" > gpurun_out/gram_variants.md
echo "|---|---|---|---|" >> gpurun_out/gram_variants.md
for tail in 0 0.55 0 0.55; do
  rm -rf $OUT/t
  ADMM_HIP_GRAM_TAIL=$tail ADMM_HIP_GRAM_DEBUG=1 rocprofv3 --kernel-trace --stats -d $OUT/t -o k -- python scripts/gram_ab.py square > $OUT/log_${tail}.txt 2>&1
  python scripts/rocpd_summary.py $OUT/t/k_results.db $OUT/stats_${tail}.md 10 > /dev/null 2>&1
  line=$(grep "gemm_nt_mfma_kernel<1" $OUT/stats_${tail}.md | head -1)
  echo "| $tail | $(echo "$line" | awk -F'|' '{print $7}') | $(grep '\[gram\]' $OUT/log_${tail}.txt | head -1) | $(grep t_gram $OUT/log_${tail}.txt | head -1) |" >> gpurun_out/gram_variants.md
done
rm -rf $OUT/t
cat gpurun_out/gram_variants.md
