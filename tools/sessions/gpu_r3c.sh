#!/bin/bash
# Round-3 GPU session C: decode ms/token, one PROCESS per configuration (for switches that are latched per process, e.g. LMX_GEMV_R / LMX_GEMV_P).
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/${OUT:-r3c}; mkdir -p $O
export TMPDIR=/tmp
: > $O/mb_decode.jsonl
for conf in "$@"; do
  timeout 300 python tools/mb_decode.py "$conf" --rounds 2 >> $O/mb_decode.jsonl 2>> $O/mb_decode.err
done
python - $O/mb_decode.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    try: r = json.loads(l)
    except Exception: continue
    u = r.get("us_per_launch", {})
    print(r["config"], r["best_us"], {k.replace("decode.", ""): v for k, v in u.items() if "gemv" in k or "attn" in k or "flow" in k}, "ids", r.get("ids_hash"))
PY
tail -3 $O/mb_decode.err
