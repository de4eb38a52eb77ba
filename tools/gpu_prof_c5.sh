#!/bin/bash
# rocprofv3 kernel-trace stats of the training-step bench (config 5, 16 x 2048, one warm-up + one timed step); summary -> gpurun_out/prof_c5_kernel_stats.csv
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/prof5 && mkdir -p /tmp/prof5
R=$(pwd)
(cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof5 -o run -- python $R/bench.py --workload config5 --train-batch ${C5_BATCH:-16} --train-seq 2048 --steps 1 --warmup 1 > $R/gpurun_out/prof_c5_bench.json 2> $R/gpurun_out/prof_c5.err); echo "rocprof rc=$?"
for f in $(find /tmp/prof5 -name "*kernel_stats.csv"); do cp $f gpurun_out/prof_c5_kernel_stats.csv; done
python - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/prof_c5_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:32]:
    print(f"{r['Name'][:100]:100s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.1f} total_ms={float(r['TotalDurationNs'])/1e6:9.2f} pct={float(r['TotalDurationNs'])/tot*100:5.2f}")
print('total ms', tot / 1e6)
PY
tail -2 gpurun_out/prof_c5.err
