#!/bin/bash
# rocprofv3 kernel-trace stats of the headline bench command (short run), summaries copied to gpurun_out/prof/
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
rm -rf /tmp/prof && mkdir -p /tmp/prof
R=$(pwd)
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o run -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-tp-projection --no-batch > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err); echo "rocprof rc=$?"
find /tmp/prof -type f | head -20
for f in $(find /tmp/prof -name "*stats*.csv"); do cp $f gpurun_out/prof/; done
python - <<'PY'
import csv, glob
for f in glob.glob('/tmp/prof/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r.get('TotalDurationNs', r.get('TotalDuration(ns)', 0)) or 0))
    print(f, len(rows))
    for r in rows[:25]:
        name = r.get('Name', r.get('KernelName', ''))[:110]
        print(f"{name:110s} calls={r.get('Calls')} avg_us={float(r.get('AverageNs', 0))/1e3:9.2f} total_ms={float(r.get('TotalDurationNs', 0))/1e6:9.2f} pct={r.get('Percentage')}")
PY
tail -3 gpurun_out/prof.err
