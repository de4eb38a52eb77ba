#!/bin/bash
# Round 4, closing session of the re-entered build, part 2: whole GPU suite, smoke, the driver's bench command on the final tree (projection timed with a C no-op all-reduce stand-in).
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -6 ) > gpurun_out/r04_pytest_gpu_summary.txt 2>&1; cat gpurun_out/r04_pytest_gpu_summary.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3
( time timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_driver_flags.json 2> gpurun_out/r04_bench.err ) 2>&1 | tail -3
python tools/bench_brief.py gpurun_out/r04_bench_driver_flags.json "driver flags"
python - <<'PY'
import json
for l in open("gpurun_out/r04_bench_driver_flags.json"):
    try: r = json.loads(l)
    except Exception: continue
    print("roofline", round(r["roofline"]["frac"], 4), "prefill frac", round(r["roofline_prefill"]["frac"], 4), "gemm family", round(r["roofline_prefill"]["gemm_family_frac"], 4), "cpu", (r.get("cpu_baseline") or {}).get("kind"), (r.get("cpu_baseline") or {}).get("value"))
    tp = r.get("tp_projection") or {}
    print("same jobs on one GPU", json.dumps((tp.get("measured_tp1") or {}).get("same_jobs_on_one_gpu")))
    for k, v in (tp.get("by_world") or {}).items():
        w = v.get("weak", {})
        print("W", k, "prefill", round(v["rank_compute_prefill_ms"], 2), "+", round(v["modelled_comm_prefill_ms"], 2), "dec", round(v["rank_compute_decode_ms_per_token"], 3), "+", round(v["modelled_comm_decode_ms_per_token"], 3), "single x", round(v.get("projected_speedup_vs_tp1", 0), 2),
              "b32", round(v["serving_batch"]["rank_compute_ms_per_step"], 3), "x", round(v.get("projected_batch32_speedup_vs_tp1", 0), 2), "| weak job", round(w.get("rank_compute_job_ms", 0), 1), "+", round(w.get("p2p", {}).get("modelled_comm_ms", 0), 1), "ms ->", round(w.get("projected_value_tokens_per_s", 0)),
              "tok/s x", round(w.get("projected_speedup_vs_one_gpu_one_request", 0), 2), "vs replicas", round(w.get("projected_vs_replicas", 0), 3), "vs one GPU", round(w.get("projected_vs_same_job_on_one_gpu", 0), 2), "ring", round(w.get("ring", {}).get("projected_value_tokens_per_s", 0)))
    if "error" in tp: print(tp)
PY
