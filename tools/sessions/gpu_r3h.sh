#!/bin/bash
# Round-3 GPU session H: evidence for profiles/: rocprofv3 kernel stats of the bench command, then the PMC passes (tools/gpu_pmc_r3.sh).
set -u
cd "$(dirname "$0")/../.."
R=$(pwd); O=$R/gpurun_out/r3h; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_r3
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r3 -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batch --no-pmc --no-tp-projection > $O/bench_under_rocprofv3.json 2> $O/bench_under_rocprofv3.err
echo "rocprof bench rc=$?"
f=$(find /tmp/prof_r3 -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then python - "$f" $O/r03_rocprofv3_kernel_stats.csv <<'PY'
import csv, sys
csv.field_size_limit(1 << 30)
rows = list(csv.reader(open(sys.argv[1])))
with open(sys.argv[2], "w", newline="") as fo:
    w = csv.writer(fo)
    for r in rows:
        r[0] = r[0][:160]
        w.writerow(r)
print("".join(",".join(r)[:200] + "\n" for r in rows[:14]))
PY
else echo "no kernel_stats.csv"; find /tmp/prof_r3 | head; tail -5 $O/bench_under_rocprofv3.err; fi
cd $R
bash tools/gpu_pmc_r3.sh 2>&1 | tail -60
cp gpurun_out/r3pmc/r03_pmc*.json gpurun_out/r3pmc/r03_pmc_*.csv $O/ 2>/dev/null
ls $O | head -40
