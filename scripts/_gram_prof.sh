#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r05g
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/kt -o k -- python scripts/gram_b3_sweep.py 512 > $OUT/kt.log 2>&1
python scripts/rocpd_summary.py $(find $OUT/kt -name '*_results.db' | head -1) 2>/dev/null | head -30 > $OUT/stats.md
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/f -o k -- python scripts/gram_b3_sweep.py 512 > $OUT/f.log 2>&1
python - <<'PY' > gpurun_out/r05g/fetch.md
import sqlite3, glob
db = sqlite3.connect(glob.glob("gpurun_out/r05g/f/**/*_results.db", recursive=True)[0])
for r in db.execute("select kernel_name, count(*), avg(value), max(value) from counters_collection where counter_name='FETCH_SIZE' group by kernel_name order by sum(value) desc limit 12"):
    print(r[0][:70], r[1], "avg MB (2x):", round(2*r[2]*1024/1e6,1), "max:", round(2*r[3]*1024/1e6,1))
PY
rm -rf $OUT/kt $OUT/f
cat $OUT/stats.md $OUT/fetch.md
