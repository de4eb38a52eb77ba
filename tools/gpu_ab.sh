#!/bin/bash
# A/B of env-var switches on the headline bench (short runs)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for cfg in "$@"; do
  env $cfg timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print('$cfg', {k: round(r[k], 3) for k in ('value','prefill_ms','decode_tokens_per_s','decode_ms_per_token')}, r['greedy_ids_identical_across_steps'])
"
done
