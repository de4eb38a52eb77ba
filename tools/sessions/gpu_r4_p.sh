#!/bin/bash
# Round 4, session p: two-shot all-reduce at the real width between two processes; N = 2 dry run of the bench (weak job + single request + replicas) on one GPU.
# The dry run keeps prefill-sized sums on 32-row launches (LMX_TP_P2P_BIG=0): with both ranks on ONE GPU the 544 waiting workgroups of a rank's two-shot launch sit on
# every CU and the other rank's prefill GEMM (one workgroup = a whole CU's registers) can never start — session o's 30 s time-out; between GPUs each rank has its own CUs.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_tp_p2p_gpu.py -q -x -s -k real_width -p no:cacheprovider 2>&1 | tail -8 ) 2>&1
( time LMX_BENCH_SHARE_GPU=1 LMX_TP_P2P_ALL=1 LMX_TP_P2P_BIG=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r04_bench_tp2_shared_gpu_dry_run.json 2> gpurun_out/r04_p_dry.err ) 2>&1 | tail -3; grep -v "^W0\|^\[W\|amdgpu.ids\|^$\|^\*\*\*\|OMP_NUM" gpurun_out/r04_p_dry.err | tail -12
python - <<'PY'
import json
for l in open("gpurun_out/r04_bench_tp2_shared_gpu_dry_run.json"):
    try: r = json.loads(l)
    except Exception: continue
    print({k: r.get(k) for k in ("value", "scaling", "n_gpus", "ms_per_step", "unit")}); print(r["config"]); print("weak", r.get("weak_job")); print("strong", r.get("strong_single_request")); print("replicas", r.get("replicas"))
PY
