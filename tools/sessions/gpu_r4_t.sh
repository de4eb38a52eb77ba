#!/bin/bash
# Round 4, session t: the RMSNorm-in-the-linear policy inside the engine's batched step on TP = 8 / 4 shards; N = 2 dry run of the bench on the final tree.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
for w in 8 4; do for x in 0 policy 1; do
  if [ "$x" = policy ]; then timeout 200 python tools/mb_tp_batch_step.py $w 8 2>/dev/null | tee -a gpurun_out/r04_tp_batch_step_xnorm.jsonl
  else LMX_SKINNY_XNORM=$x timeout 200 python tools/mb_tp_batch_step.py $w 8 2>/dev/null | tee -a gpurun_out/r04_tp_batch_step_xnorm.jsonl; fi
done; done
( time LMX_BENCH_SHARE_GPU=1 LMX_TP_P2P_ALL=1 LMX_TP_P2P_BIG=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r04_bench_tp2_shared_gpu_dry_run.json 2> gpurun_out/r04_t_dry.err ) 2>&1 | tail -3
python - <<'PY'
import json
for l in open("gpurun_out/r04_bench_tp2_shared_gpu_dry_run.json"):
    try: r = json.loads(l)
    except Exception: continue
    print({k: r.get(k) for k in ("value", "scaling", "n_gpus", "ms_per_step")}); print("weak", r.get("weak_job")); print("strong", r.get("strong_single_request")); print("replicas", r.get("replicas"))
PY
