#!/bin/bash
# Round-3 GPU session U: where the config-5 step (4 x 2048) spends its time: rocprofv3 kernel stats.
set -u
cd "$(dirname "$0")/../.."
R=$(pwd); O=$R/gpurun_out/r3u; mkdir -p $O
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/prof_c5
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -o p -- python $R/bench.py --workload config5 --train-batch 4 --train-seq 2048 --steps 1 --warmup 1 > $O/bench.json 2> $O/bench.err
echo "rc=$?"
f=$(find /tmp/prof_c5 -name "*kernel_stats.csv" | head -1)
python - "$f" $O/r03_config5_kernel_stats.csv <<'PY'
import csv, sys
csv.field_size_limit(1 << 30)
rows = list(csv.reader(open(sys.argv[1])))
with open(sys.argv[2], "w", newline="") as fo:
    w = csv.writer(fo)
    for r in rows:
        r[0] = r[0][:140]
        w.writerow(r)
for r in rows[:22]:
    print(r[0][:90], r[1], r[3][:9] if len(r) > 3 else "", r[4] if len(r) > 4 else "")
PY
