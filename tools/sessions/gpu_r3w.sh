#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r3w; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_train_slices_gpu.py tests/test_train_step_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?"; grep -E "^E  |passed|failed|Error" $O/pytest.txt | tail -8
timeout 600 python bench.py --workload config5 --train-batch 4 --train-seq 2048 --steps 2 --warmup 1 > $O/bench_config5_4x2048.json 2> $O/bench_config5.err; echo "config5 rc=$?"
python - $O/bench_config5_4x2048.json <<'PY'
import json, sys
for l in open(sys.argv[1]):
    try: r = json.loads(l)
    except Exception: continue
    print({k: r.get(k) for k in ('value','ms_per_step','forward_ms','forward_backward_ms','optimizer_ms','linear_tflops_in_fwd_bwd','loss_first_last')})
PY
