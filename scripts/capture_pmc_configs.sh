#!/bin/bash
# PMC HBM bytes per ADMM iteration of every non-headline BASELINE config (separate passes per counter, kernel-trace only, as
# MI355X_MICROARCH.md prescribes; gpurun refuses --pmc together with the sys/hip trace domains).  Run from the repo root through
# gpurun; summaries go to gpurun_out/<tag>/ and profiles/pmc_traffic.json (configs.<name>) is updated in gpurun_out/<tag>/ as a copy.
# Usage: scripts/capture_pmc_configs.sh <tag> [configs...]
set -u
TAG=${1:-r05p}; shift || true
CONFIGS=${@:-c3 c4 c5lad c5bp c5parbp dantzig}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
for c in $CONFIGS; do
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/f_$c -o k -- python scripts/bench_configs.py $c > $OUT/f_$c.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/w_$c -o k -- python scripts/bench_configs.py $c > $OUT/w_$c.log 2>&1
  fdb=$(find $OUT/f_$c -name '*_results.db' | head -1); wdb=$(find $OUT/w_$c -name '*_results.db' | head -1)
  grep '^{' $OUT/f_$c.log > $OUT/bench_$c.jsonl
  python scripts/pmc_config_traffic.py $c $fdb $wdb $OUT/bench_$c.jsonl $OUT/${c}_pmc_hbm_bytes.md --update $OUT/pmc_traffic.json
  rm -rf $OUT/f_$c $OUT/w_$c
done
