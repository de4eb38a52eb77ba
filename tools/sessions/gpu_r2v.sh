#!/bin/bash
cd "$(dirname "$0")/../.."
for rp in "auto 2" "auto 4" "auto 6" "1 2" "1 4" "1 6" "2 4" "2 6" "4 4"; do
  set -- $rp
  if [ "$1" = auto ]; then R=""; else R="LMX_GEMV_R=$1"; fi
  env $R LMX_GEMV_P=$2 timeout 200 python tools/mb_gemv_cold.py 2>&1 | grep gemv_cold | python -c "
import sys, json
rows=[json.loads(l) for l in sys.stdin]
print('R=$1 P=$2', {r['name']: r['us'] for r in rows})"
done
