#!/bin/bash
# Round-2 GPU session O: rocprofv3 kernel stats of the config-5 training step.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r2o; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o c5 -- python $GRAFT_REPO_ROOT/bench.py --workload config5 --steps 2 --warmup 1 > $GRAFT_REPO_ROOT/$O/bench.json 2> $GRAFT_REPO_ROOT/$O/bench.err
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print("%-90s calls %6s  total %8.2f ms  avg %8.1f us  %5.1f%%" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
