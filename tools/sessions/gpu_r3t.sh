#!/bin/bash
# Round-3 GPU session T: training parity at the reference's length (7B layer x 2048, attention backward 32 x 128 x 2048) + config 5 at 4 x 2048.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r3t; mkdir -p $O
export TMPDIR=/tmp
SECONDS=0
timeout 900 python -m pytest tests/test_train_slices_gpu.py -x -q -m gpu -k "attn_bwd or composes" > $O/pytest.txt 2>&1; echo "pytest rc=$? (${SECONDS}s)"; tail -5 $O/pytest.txt
SECONDS=0
timeout 900 python bench.py --workload config5 --train-batch 4 --train-seq 2048 --steps 2 --warmup 1 > $O/bench_config5_4x2048.json 2> $O/bench_config5.err; echo "config5 rc=$? (${SECONDS}s)"; tail -3 $O/bench_config5.err
python - $O/bench_config5_4x2048.json <<'PY'
import json, sys
for l in open(sys.argv[1]):
    try: r = json.loads(l)
    except Exception: continue
    print({k: r.get(k) for k in ('metric','value','ms_per_step','config')})
    print({k: v for k, v in r.items() if k not in ('metric','value','ms_per_step','config','kernel_breakdown_ms_per_step')})
PY
