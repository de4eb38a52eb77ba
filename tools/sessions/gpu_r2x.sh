#!/bin/bash
# Round-2 GPU session X: tests touched since the closing session + the N=2 code path of bench.py with both ranks on this GPU (dry run).
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r2x; mkdir -p $O
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_tp_gpu.py tests/test_decode_persist_gpu.py tests/test_beam_gpu.py tests/test_ops_gpu.py tests/test_model_gpu.py -q -m gpu 2>&1 | tail -5
LMX_BENCH_SHARE_GPU=1 LMX_TP_P2P_ALL=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-batch --no-pmc > $O/bench_tp2_shared.json 2> $O/bench_tp2_shared.err
python - $O/bench_tp2_shared.json <<'PY'
import json, sys
ok = False
for l in open(sys.argv[1]):
    try: r = json.loads(l)
    except Exception: continue
    ok = True
    print({k: r[k] for k in ('value', 'n_gpus', 'scaling', 'prefill_ms')}, r['config'], 'ids_same', r.get('greedy_ids_identical_across_steps'), 'replicas', r.get('replicas'))
print("json line found:", ok)
PY
tail -3 $O/bench_tp2_shared.err | cut -c1-300
