#!/bin/bash
# rocprofv3 evidence for the decode-batch path: (1) kernel-trace stats of tools/mb_batch.py (B = 8 and 32, 7B, ctx 1087),
# (2) PMC FETCH_SIZE of the skinny linear at M = 8 on the four 7B layer shapes (HBM-side bytes vs algorithmic).
set -u
cd "$(dirname "$0")/.."
R=$(pwd); mkdir -p gpurun_out/prof_batch; export TMPDIR=/tmp
rm -rf /tmp/pb && mkdir -p /tmp/pb
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o run -- python $R/tools/mb_batch.py llava15_7b 512 8,32 > $R/gpurun_out/prof_batch/mb_batch.log 2>&1); echo "rocprof rc=$?"
for f in $(find /tmp/pb -name "*kernel_stats.csv"); do cp $f gpurun_out/prof_batch/kernel_stats.csv; done
python - <<'PY'
import csv, glob
for f in glob.glob('/tmp/pb/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r.get('TotalDurationNs', 0) or 0))
    for r in rows[:14]:
        print(f"{r.get('Name','')[:100]:100s} calls={r.get('Calls')} avg_us={float(r.get('AverageNs', 0))/1e3:9.2f} total_ms={float(r.get('TotalDurationNs', 0))/1e6:9.2f} pct={r.get('Percentage')}")
PY
grep -h '"path"\|profile_B' gpurun_out/prof_batch/mb_batch.log
cd /tmp
rm -rf /tmp/pmcb
MS=8 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmcb -o p -- python $R/tools/mb_skinny.py > /tmp/pmcb.log 2>&1
f=$(find /tmp/pmcb -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then
python - "$f" <<'PY' | tee $R/gpurun_out/prof_batch/pmc_fetch_skinny.txt
import csv, sys, collections
acc = collections.defaultdict(float); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != "FETCH_SIZE": continue
    k = r["Kernel_Name"].split("(")[0]
    if "skinny" not in k: continue
    key = (k[-70:], r.get("Grid_Size", ""))
    acc[key] += float(r["Counter_Value"]); cnt[key] += 1
for key in sorted(acc):
    print(f"PMC FETCH_SIZE {key[0]:70s} grid={key[1]:>8s} n={cnt[key]:4d} avg={acc[key] / cnt[key]:14.1f}")
PY
else echo "no pmc csv"; tail -3 /tmp/pmcb.log; fi
