#!/bin/bash
# PMC passes over one GEMM variant (counters in their own runs, kernel-trace only)
set -u
cd "$(dirname "$0")/.."
R=$(pwd); mkdir -p gpurun_out/pmc; export TMPDIR=/tmp
V=${V:-7}; M=${M:-1087}; N=${N:-12288}; K=${K:-4096}
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|TCC|TCP|GRBM|TA)_[A-Z0-9_]+\b" | sort -u > $R/gpurun_out/pmc/counters.txt
wc -l $R/gpurun_out/pmc/counters.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE" \
           "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_REQ_sum GRBM_COUNT"; do
  i=$((i+1)); rm -rf /tmp/pmc$i
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc$i -o p -- python $R/tools/mb_gemm_one.py $V $M $N $K 4 > /tmp/pmc$i.log 2>&1
  f=$(find /tmp/pmc$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r.get('Kernel_Name', '')
    if 'gemm' not in k: continue
    acc[k[:60]][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k[:60], r['Counter_Name'])] += 1
for k, d in acc.items():
    print(k)
    for c, v in d.items(): print(f"   {c:36s} {v / max(cnt[(k, c)], 1):16.1f} per dispatch")
PY
  else echo "no counter csv for set $i"; tail -5 /tmp/pmc$i.log; fi
done
