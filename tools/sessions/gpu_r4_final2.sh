#!/bin/bash
# Round 4, closing session of the re-entered build, part 1: the driver's bench command, rocprofv3 kernel stats of the same command (short form).
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$(pwd)
( time timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_driver_flags.json 2> gpurun_out/r04_bench.err ) 2>&1 | tail -3; grep -v amdgpu.ids gpurun_out/r04_bench.err | tail -3
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
        print("W", k, "prefill", round(v["rank_compute_prefill_ms"], 2), "+", round(v["modelled_comm_prefill_ms"], 2), "dec", round(v["rank_compute_decode_ms_per_token"], 3), "single x", round(v.get("projected_speedup_vs_tp1", 0), 2),
              "b32 x", round(v.get("projected_batch32_speedup_vs_tp1", 0), 2), "| weak job", round(w.get("rank_compute_job_ms", 0), 1), "+", round(w.get("p2p", {}).get("modelled_comm_ms", 0), 1), "ms ->", round(w.get("projected_value_tokens_per_s", 0)),
              "tok/s x", round(w.get("projected_speedup_vs_one_gpu_one_request", 0), 2), "vs replicas", round(w.get("projected_vs_replicas", 0), 3), "vs one GPU", round(w.get("projected_vs_same_job_on_one_gpu", 0), 2), "ring", round(w.get("ring", {}).get("projected_value_tokens_per_s", 0)))
    if "error" in tp: print(tp)
PY
rm -rf /tmp/prof && mkdir -p /tmp/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o run -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-tp-projection --no-batch > $R/gpurun_out/r04_bench_under_rocprofv3.json 2> $R/gpurun_out/r04_prof.err); echo "rocprof rc=$?"
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r04_rocprofv3_kernel_stats.csv && head -12 "$f" | cut -c1-150
