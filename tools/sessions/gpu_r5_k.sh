#!/bin/bash
# Round 5, session K: N = 2 dry run of the bench (weak job + single request + replicas + the same job on one GPU) with both ranks on ONE GPU — proves the N > 1 code path
# of this round's bench line (schema 5); its speeds are meaningless.  LMX_TP_P2P_BIG=0: see tools/sessions/gpu_r4_p.sh.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( time LMX_BENCH_SHARE_GPU=1 LMX_TP_P2P_ALL=1 LMX_TP_P2P_BIG=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r05_bench_tp2_shared_gpu_dry_run.json 2> gpurun_out/r05_k_dry.err ) 2>&1 | tail -3; grep -v "^W0\|^\[W\|amdgpu.ids\|^$\|^\*\*\*\|OMP_NUM" gpurun_out/r05_k_dry.err | tail -12
python - <<'PY'
import json
for l in open("gpurun_out/r05_bench_tp2_shared_gpu_dry_run.json"):
    try: r = json.loads(l)
    except Exception: continue
    print({k: r.get(k) for k in ("value", "scaling", "n_gpus", "ms_per_step", "schema", "value_definition", "strong_single_request_value", "same_job_on_one_gpu_value", "value_vs_same_job_on_one_gpu")})
    print("weak", r.get("weak_job")); print("strong", r.get("strong_single_request")); print("replicas", r.get("replicas")); print("one_gpu", r.get("same_job_on_one_gpu"))
PY
