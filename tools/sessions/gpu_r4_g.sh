#!/bin/bash
# Round 4, session G: two-shot P2P all-reduce (2 processes on one GPU), TP regression, full bench line (reference-kind CPU baseline, new tp_projection).
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_tp_p2p_gpu.py tests/test_tp_gpu.py tests/test_tp_serving_gpu.py -m gpu -q -x -p no:cacheprovider -s 2>&1 | tail -30 > gpurun_out/r04_g_tp.log; tail -12 gpurun_out/r04_g_tp.log
( time timeout 1500 python bench.py --steps 5 --warmup 2 > gpurun_out/r04_g_bench.json 2> gpurun_out/r04_g_bench.err ) 2>&1 | tail -3; tail -3 gpurun_out/r04_g_bench.err
python tools/bench_brief.py gpurun_out/r04_g_bench.json "full" | head -2
python - <<'P'
import json
for l in open("gpurun_out/r04_g_bench.json"):
    try: r = json.loads(l)
    except Exception: continue
    if "value" not in r: continue
    c = r["cpu_baseline"]; print("cpu:", {k: c.get(k) for k in ("kind", "value", "cores", "prefill_ms", "decode_tokens_per_s", "decode_steps_timed", "model_build_s", "sourceless", "error", "reference_error", "port_vs_reference")})
    if c.get("port"): print("port:", {k: c["port"].get(k) for k in ("value", "prefill_ms", "decode_tokens_per_s", "decode_steps_timed")})
    t = r["tp_projection"]
    if "by_world" in t:
        for w, v in t["by_world"].items():
            print("TP", w, {k: (round(v[k], 3) if isinstance(v[k], float) else v[k]) for k in ("rank_compute_prefill_ms", "of_which_replicated_tower_and_splice_ms", "rank_compute_decode_ms_per_token", "allreduce_calls_prefill", "projected_prefill_ms", "projected_decode_ms_per_token", "projected_value_tokens_per_s", "projected_speedup_vs_tp1", "projected_batch32_speedup_vs_tp1")}, v["serving_batch"], {k: round(x, 3) for k, x in v["modelled_comm"]["p2p"].items()}, "ring", {k: round(x, 1) for k, x in v["ring"].items()})
    else: print(t)
    print("roofline", r["roofline"]["frac"], "prefill", r["roofline_prefill"]["frac"], r["roofline_prefill"].get("prefill_end_to_end_frac"))
P
