#!/bin/bash
# HBM-side traffic per launch of the dominant kernels from rocprofv3 PMC (one counter per pass, kernel-trace only).
# Drivers are the single-kernel microbenchmarks (rocprofv3 --pmc segfaults on the full bench.py process in this image).
set -u
cd "$(dirname "$0")/.."
R=$(pwd); mkdir -p gpurun_out/pmc; export TMPDIR=/tmp
cd /tmp
i=0
for c in FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum; do
  for drv in "tools/mb_gemv_cold.py" "tools/mb_gemm_one.py 9 1087 12288 4096 4" "tools/mb_gemm_one.py 7 1087 4096 11008 4"; do
    i=$((i+1)); rm -rf /tmp/pmc$i
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc$i -o p -- python $R/$drv > /tmp/pmc$i.log 2>&1
    rc=$?
    f=$(find /tmp/pmc$i -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then
      python - "$f" "$c" "$drv" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); cnt = collections.Counter(); grid = {}
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != sys.argv[2]: continue
    k = r["Kernel_Name"].split("(")[0]
    if "lmx::" not in k: continue
    key = (k, r.get("Grid_Size", ""))
    acc[key] += float(r["Counter_Value"]); cnt[key] += 1
for key in acc:
    print(f"PMC {sys.argv[2]:20s} drv={sys.argv[3][:28]:28s} {key[0][-60:]:60s} grid={key[1]:>8s} n={cnt[key]:4d} avg={acc[key] / cnt[key]:14.1f}")
PY
    else echo "$c / $drv: rc=$rc no csv"; tail -2 /tmp/pmc$i.log; fi
  done
done
