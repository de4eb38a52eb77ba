#!/bin/bash
# Round 4, last session: whole GPU suite on the final tree (skinny stream on), then a short bench line with the serving batch and the projection.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 420 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -4 ) > gpurun_out/r04_pytest_gpu_summary.txt 2>&1; cat gpurun_out/r04_pytest_gpu_summary.txt
timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > gpurun_out/r04_bench_final_tree_short.json 2> gpurun_out/r04_f4.err
python tools/bench_brief.py gpurun_out/r04_bench_final_tree_short.json "final tree (short)"
python - <<'PY'
import json
for l in open("gpurun_out/r04_bench_final_tree_short.json"):
    try: r = json.loads(l)
    except Exception: continue
    print("serving", {k: (round(v["ms_per_step"], 3), round(v["decode_tokens_per_s"])) for k, v in r["serving_batch"]["by_batch"].items()})
    tp = r.get("tp_projection") or {}
    print("same jobs on one GPU", json.dumps((tp.get("measured_tp1") or {}).get("same_jobs_on_one_gpu")))
    for k, v in (tp.get("by_world") or {}).items():
        w = v.get("weak", {})
        print("W", k, "single x", round(v.get("projected_speedup_vs_tp1", 0), 2), "b32", round(v["serving_batch"]["rank_compute_ms_per_step"], 3), "x", round(v.get("projected_batch32_speedup_vs_tp1", 0), 2), "| weak job", round(w.get("rank_compute_job_ms", 0), 1), "+", round(w.get("p2p", {}).get("modelled_comm_ms", 0), 1), "->", round(w.get("projected_value_tokens_per_s", 0)),
              "x", round(w.get("projected_speedup_vs_one_gpu_one_request", 0), 2), "vs replicas", round(w.get("projected_vs_replicas", 0), 3), "vs one GPU", round(w.get("projected_vs_same_job_on_one_gpu", 0), 2))
PY
