#!/bin/bash
# Round 4, session o: tower data parallel under tensor parallelism (tests), weak-scaling projection in the bench line, N = 2 dry run of the new bench path on one GPU.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_tp_gpu.py tests/test_tp_serving_gpu.py tests/test_batching_gpu.py tests/test_reuse_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -8 ) 2>&1
( time timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > gpurun_out/r04_bench_weak_projection.json 2> gpurun_out/r04_o_bench.err ) 2>&1 | tail -3; tail -3 gpurun_out/r04_o_bench.err
python tools/bench_brief.py gpurun_out/r04_bench_weak_projection.json "weak projection"
python - <<'PY'
import json
for l in open("gpurun_out/r04_bench_weak_projection.json"):
    try: r = json.loads(l)
    except Exception: continue
    tp = r.get("tp_projection") or {}
    print("measured_tp1", json.dumps(tp.get("measured_tp1")))
    for k, v in (tp.get("by_world") or {}).items():
        print("W", k, "prefill", round(v["rank_compute_prefill_ms"], 2), "dec", round(v["rank_compute_decode_ms_per_token"], 3), "proj x", round(v.get("projected_speedup_vs_tp1", 0), 2), "weak", json.dumps(v.get("weak"))[:900])
    if "error" in tp: print(tp)
PY
( time LMX_BENCH_SHARE_GPU=1 LMX_TP_P2P_ALL=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r04_bench_tp2_shared_gpu_dry_run.json 2> gpurun_out/r04_o_dry.err ) 2>&1 | tail -3; tail -5 gpurun_out/r04_o_dry.err
python - <<'PY'
import json
for l in open("gpurun_out/r04_bench_tp2_shared_gpu_dry_run.json"):
    try: r = json.loads(l)
    except Exception: continue
    print({k: r.get(k) for k in ("value", "scaling", "n_gpus", "ms_per_step")}); print(r["config"]["workload"]); print("weak", r.get("weak_job")); print("strong", r.get("strong_single_request")); print("replicas", r.get("replicas"))
PY
