#!/bin/bash
# Dev tool (GPU box): long randomised soak -- many seeds of the small sweep, a few of the medium and dense ones.
# Prints only the suspect lines and the per-seed totals.   bash tests/tools/soak.sh <first_seed> <nseeds>
first=${1:-201}; n=${2:-10}
for ((s=first; s<first+n; s++)); do
  timeout 900 python tests/tools/fuzz_parity.py 150 $s 2>&1 | grep -E "SUSPECT|suspects|EXCEPTION|library error|Traceback" | sed "s/^/seed $s: /"
done
for ((s=first; s<first+3; s++)); do
  timeout 1200 python tests/tools/fuzz_medium.py 12 $s 2>&1 | grep -E "SUSPECT|Traceback" | sed "s/^/medium $s: /"
  timeout 600 python tests/tools/fuzz_dense.py 12 $s 2>&1 | grep -E "SUSPECT|Traceback" | sed "s/^/dense $s: /"
done
echo soak done
