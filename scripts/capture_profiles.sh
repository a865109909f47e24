#!/bin/bash
# Capture the judged evidence on the GPU box (run from the repo root through gpurun):
#   headline bench: kernel-trace stats, PMC HBM bytes (separate passes per counter, no trace domains mixed in), the bench line;
#   the other BASELINE configs (C3 wide, C4 consensus K=8 on one GPU, C5 LAD / BP): bench lines + kernel-trace stats each.
# Outputs go to gpurun_out/<tag>/; copy the summaries into profiles/ afterwards.
set -u
TAG=${1:-r04}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o c2 -- python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --consensus-seconds 0 --config-seconds 0 > $OUT/trace_bench.log 2>&1
python scripts/rocpd_summary.py $OUT/trace/c2_results.db $OUT/tall_c2_kernel_stats.md 14
grep '^{"metric"' $OUT/trace_bench.log | tail -1 > $OUT/tall_c2_kernel_stats_benchline.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o c2 -- python bench.py --steps 1 --warmup 0 --cpu-seconds 0 --consensus-seconds 0 --config-seconds 0 --nlambda 10 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o c2 -- python bench.py --steps 1 --warmup 0 --cpu-seconds 0 --consensus-seconds 0 --config-seconds 0 --nlambda 10 > $OUT/pmc_write.log 2>&1
python scripts/rocpd_pmc.py $OUT/pmc_fetch/c2_results.db $OUT/pmc_write/c2_results.db $OUT/tall_c2_pmc_hbm_bytes.md
python bench.py > $OUT/bench_default.log 2>&1
grep '^{"metric"' $OUT/bench_default.log | tail -1 > $OUT/bench_all_configs.json
python scripts/bench_configs.py c3 c4 c5lad c5bp c5parbp dantzig > $OUT/bench_configs.jsonl 2> $OUT/bench_configs.err
for c in c3 c4 c5lad c5bp c5parbp dantzig; do
  rocprofv3 --kernel-trace --stats -d $OUT/trace_$c -o k -- python scripts/bench_configs.py $c > $OUT/trace_$c.log 2>&1
  python scripts/rocpd_summary.py $OUT/trace_$c/k_results.db $OUT/${c}_kernel_stats.md 12
done
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/trace_c3 $OUT/trace_c4 $OUT/trace_c5lad $OUT/trace_c5bp $OUT/trace_c5parbp $OUT/trace_dantzig      # the databases are large; the summaries are what is kept
ls -la $OUT
# round 6: the column-sharded wide solver as one rank over PEER (persistent stretch with the exchange inside the launch), and the roctx ranges
rocprofv3 --kernel-trace --stats -d $OUT/trace_wc -o k -- python bench.py --child widecols:peer:$OUT/widecols_one_rank.json --seed 123 > $OUT/trace_wc.log 2>&1
python scripts/rocpd_summary.py $OUT/trace_wc/k_results.db $OUT/widecols_one_rank_kernel_stats.md 10
rocprofv3 --marker-trace --kernel-trace --output-format csv -d $OUT/markers -o m -- python __graft_entry__.py smoke > $OUT/markers.log 2>&1
f=$(find $OUT/markers -name '*marker*trace*.csv' | head -1)
if [ -n "$f" ]; then python - "$f" > $OUT/roctx_ranges.md <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    name = r.get("Function") or r.get("Name") or r.get("Message") or str(r)
    try:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    except Exception:
        d = 0.0
    a = agg.setdefault(name, [0, 0.0]); a[0] += 1; a[1] += d
print("| roctx range | count | total us |\n|---|---|---|")
for k, (c, t) in agg.items():
    print(f"| `{k}` | {c} | {t:.1f} |")
PY
fi
rm -rf $OUT/trace_wc $OUT/markers
