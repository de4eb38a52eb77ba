#!/bin/bash
# rocprofv3 kernel-trace stats of the headline bench command on the closing tree (same command as r03_rocprofv3_kernel_stats_final.csv)
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r3prof2; mkdir -p $O
export TMPDIR=/tmp
rm -rf /tmp/prof && mkdir -p /tmp/prof
R=$(pwd)
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o run -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batch --no-pmc --no-tp-projection > $R/$O/bench_under_rocprofv3.json 2> $R/$O/prof.err); echo "rocprof rc=$?"
for f in $(find /tmp/prof -name "*kernel_stats.csv"); do cp $f $O/kernel_stats.csv; done
python - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/r3prof2/kernel_stats.csv')))
for r in rows[:14]:
    print(f"{r['Name'][:100]:100s} calls={r['Calls']} avg_us={float(r['AverageNs'])/1e3:9.2f} pct={r['Percentage']}")
PY
